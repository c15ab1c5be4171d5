// soc_brick.hip -- "brick sweep" execution of SimRAM_PB on Cartesian grids: absorption
// tallies live in LDS, not on the fabric.
//
// Why: the direct kernel issues one scattered global float atomic per cell step.  On
// MI355X those execute at the memory side at ~2e10 64-B requests/s chip-wide whatever the
// footprint (MI355X_MICROARCH.md "Global float atomics"; measured here: TCC_EA0_ATOMIC ==
// tally events, kernel pinned at 1.94e10 steps/s while the same walk without tallies runs
// at 1.2e11 steps/s).  No scope or cache policy moves them into L2.
//
// How: the grid is cut into bricks of B^3 root cells.  In-flight packets (one per logical
// work item -- a work item's packets are sequential in its RNG stream, so one is in
// flight at a time) are kept sorted by the brick they are in.  One pass =
//   soc_brick_step    a workgroup takes a chunk of one brick's queue, keeps the brick's
//                     tally in LDS (ds_add_f32), walks each packet until it leaves the brick
//                     (or ends: the work item's next packet is created on the spot), writes
//                     the packet back with its destination brick, then flushes the brick
//                     tally with row-contiguous atomics (>= 16x fewer fabric requests);
//   soc_brick_scan    exclusive scan of the arrival histogram -> next queue offsets and
//                     workgroup descriptors (one workgroup, exact sizes, no capacity guess);
//   soc_brick_scatter counting-sort placement of the packet ids into the next queues.
// A kernel boundary separates the phases, so no in-launch inter-workgroup hand-off exists.
//
// What does not change: the logical work items, their MWC64X streams, every fp32 operation
// of a packet's life (same code as soc_walk.h, same operand order).  Trajectories are
// identical to the direct kernel and to the oracle; only the order of fp32 tally additions
// differs (as it already does between any two runs of the atomic version).
#include "soc_walk.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define SOC_BRICK_T 256          // threads per workgroup (scatter kernel; step kernel uses A.T)
#define SOC_BRICK_PMAX 4096      // upper bound of packets per workgroup chunk

enum { SOC_BM_STEP = 0, SOC_BM_CREATE = 1, SOC_BM_SCATTER = 2, SOC_BM_SWAP = 3, SOC_BM_IDLE = 4 };

struct __align__(16) SocPacket {
    float px, py, pz, ux;
    float uy, uz, photons, free_path;
    float tau, dens;
    uint32_t rx, rc;
    int   ind, III;
    uint32_t misc;               // scat | mode << 8
    int   lid;                   // tally slot of the current cell inside its brick
};

struct SocDesc { int brick, start, count, pad; };

struct SocBrickArgs {
    int LB, NBX, NBY, NBZ, NB;   // brick edge = 1 << LB root cells; NB bricks
    int T, P, KCAP, FTH;         // step-kernel threads, packets per chunk, max steps per packet per pass, fetch threshold
    SocPacket *pk;
    const uint32_t *idq;         // current queue (ids sorted by brick)
    uint32_t *idq_next;
    uint32_t *keyq;              // destination brick of every entry of the current queue
    int *hist;                   // [NB+1] arrivals per brick in the next pass; [NB] = finished
    int *off;                    // [NB+1] offsets of the next queues (scan output)
    int *cursor;                 // [NB]
    const SocDesc *desc;         // descriptors of the current pass
    const int *ndesc;
    SocDesc *desc_next;
    int *ndesc_next;
    int *total;                  // packets still in flight after this pass
    long long *dbg;              // diagnostic build only: 8 timestamps per workgroup (NULL otherwise)
};

// ---------------------------------------------------------------------------------------

__device__ __forceinline__ void soc_cell_brick(const SocBrickArgs &A, float px, float py, float pz, int &brick, int &lid)
{
    const int ix = (int)soc_floorf(px), iy = (int)soc_floorf(py), iz = (int)soc_floorf(pz);
    const int M = (1 << A.LB) - 1;
    brick = ((iz >> A.LB) * A.NBY + (iy >> A.LB)) * A.NBX + (ix >> A.LB);
    lid   = ((iz & M) << (2 * A.LB)) | ((iy & M) << A.LB) | (ix & M);
}

// minimal walker interface for soc_pb_create()
struct SocBrickLane {
    float px, py, pz, ux, uy, uz;
    float photons, free_path, tau, dens;
    int   level, ind, scat;
    soc_rng_t rng;

    __device__ __forceinline__ void begin()
    {
        if (soc_fabsf(ux) < SOC_DEPS) ux = SOC_DEPS;
        if (soc_fabsf(uy) < SOC_DEPS) uy = SOC_DEPS;
        if (soc_fabsf(uz) < SOC_DEPS) uz = SOC_DEPS;
        soc_normalize(ux, uy, uz);
        scat = 0;
        tau  = 0.0f;
        free_path = -soc_logf(soc_rand(&rng));
    }
};

__global__ void soc_brick_init(const SocGrid G, const SocSim S, SocBrickArgs A, uint32_t count, uint32_t *idq0,
                               SocDesc *desc0, int *ndesc0, int *hist)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) {
        SocPacket p;
        soc_rng_t r = soc_seed_stream(S.seed_mul, S.seed_tab, S.gid0 + t);
        p.px = p.py = p.pz = p.ux = p.uy = p.uz = 0.0f;
        p.photons = p.free_path = p.tau = p.dens = 0.0f;
        p.rx = r.x;  p.rc = r.c;
        p.ind = -1;  p.III = 0;  p.lid = 0;
        p.misc = (uint32_t)SOC_BM_CREATE << 8;
        A.pk[t] = p;
        idq0[t] = t;
    }
    const uint32_t nd = (count + A.P - 1) / A.P;
    if (t < nd) {
        SocDesc d;
        d.brick = -1;
        d.start = (int)(t * A.P);
        d.count = (int)min((uint32_t)A.P, count - t * A.P);
        d.pad = 0;
        desc0[t] = d;
    }
    if (t == 0) *ndesc0 = (int)nd;
    if (t <= (uint32_t)A.NB) hist[t] = 0;
}

template <bool ABU, bool WINT, bool SD>
__global__ __launch_bounds__(1024) void soc_brick_step(const SocGrid G, const SocSim S, const SocBrickArgs A)
{
    if ((int)blockIdx.x >= *A.ndesc) return;
    const SocDesc D = A.desc[blockIdx.x];
    const int BV = 1 << (3 * A.LB);
    const int nthr = (int)blockDim.x;
    long long t0 = 0, t1 = 0, t2 = 0;
    if (A.dbg) t0 = wall_clock64();

    extern __shared__ float lds[];
    float *sT   = lds;                                   // [BV] TABS of this brick
    float *sD   = sT + BV;                               // [BV] densities of this brick (SD)
    float *sI   = sD + (SD ? BV : 0);                    // [BV] INT (WINT)
    int   *sH   = (int *)(sI + (WINT ? BV : 0));         // [NB+1] arrivals per brick, next pass
    int   *sCtl = sH + A.NB + 1;                         // [0] next packet, [1..3] stats, [4] = 0 (OFF[0])
    for (int i = threadIdx.x; i < BV; i += nthr) { sT[i] = 0.0f; if (WINT) sI[i] = 0.0f; }
    if (SD) {
        // stage the brick's densities: no global access is left inside the step loop
        const int Bq = 1 << A.LB, Mq = Bq - 1;
        const int bx = D.brick % A.NBX, by = (D.brick / A.NBX) % A.NBY, bz = D.brick / (A.NBX * A.NBY);
        for (int i = threadIdx.x; i < BV; i += nthr) {
            float d = 0.0f;
            if (D.brick >= 0) {
                const int ix = bx * Bq + (i & Mq), iy = by * Bq + ((i >> A.LB) & Mq), iz = bz * Bq + (i >> (2 * A.LB));
                if (ix < G.NX && iy < G.NY && iz < G.NZ) d = G.DENS[(iz * G.NY + iy) * G.NX + ix];
            }
            sD[i] = d;
        }
    }
    for (int i = threadIdx.x; i <= A.NB; i += nthr) sH[i] = 0;
    if (threadIdx.x < 5) sCtl[threadIdx.x] = 0;
    __syncthreads();
    if (A.dbg) t1 = wall_clock64();

    const int   NX = G.NX, NY = G.NY, NZ = G.NZ;
    const float fNX = (float)NX, fNY = (float)NY, fNZ = (float)NZ;
    const int   mybrick = D.brick, LB = A.LB, M = (1 << A.LB) - 1;
    SocBrickLane w;
    w.level = 0;  w.ind = -1;  w.scat = 0;
    w.px = w.py = w.pz = w.ux = w.uy = w.uz = 0.0f;
    w.photons = w.free_path = w.tau = w.dens = 0.0f;
    w.rng.x = w.rng.c = 0u;
    int   mode = SOC_BM_SWAP, next_mode = 0, slot = 0, III = 0, lid = 0, nvisit = 0, key = 0;
    bool  have = false;
    uint32_t wid = 0;
    unsigned int n_tally = 0, n_scat = 0, n_pkt = 0;
    const int *sOFF0 = sCtl + 4;                          // OFF[0] == 0: Cartesian grids only

    while (true) {
        bool nobody_steps = (__ballot(mode == SOC_BM_STEP) == 0ull);
        // ---- swap: write back packets that are through with this brick, take the next ones ----
        {
            const unsigned long long m = __ballot(mode == SOC_BM_SWAP);
            if (m != 0ull && (nobody_steps || __popcll(m) >= A.FTH)) {
                if (mode == SOC_BM_SWAP) {
                    if (have) {
                        SocPacket p;
                        p.px = w.px;  p.py = w.py;  p.pz = w.pz;  p.ux = w.ux;  p.uy = w.uy;  p.uz = w.uz;
                        p.photons = w.photons;  p.free_path = w.free_path;  p.tau = w.tau;  p.dens = w.dens;
                        p.rx = w.rng.x;  p.rc = w.rng.c;
                        p.ind = w.ind;  p.III = III;  p.lid = lid;
                        p.misc = (uint32_t)(w.scat & 0xff) | ((uint32_t)next_mode << 8);
                        A.pk[wid] = p;
                        A.keyq[D.start + slot] = (uint32_t)key;
                        atomicAdd(&sH[key], 1);
                    }
                    slot = atomicAdd(&sCtl[0], 1);
                    have = slot < D.count;
                    if (!have) {
                        mode = SOC_BM_IDLE;
                    } else {
                        wid = A.idq[D.start + slot];
                        const SocPacket p = A.pk[wid];
                        w.px = p.px;  w.py = p.py;  w.pz = p.pz;  w.ux = p.ux;  w.uy = p.uy;  w.uz = p.uz;
                        w.photons = p.photons;  w.free_path = p.free_path;  w.tau = p.tau;
                        w.rng.x = p.rx;  w.rng.c = p.rc;
                        w.ind = p.ind;  III = p.III;  lid = p.lid;
                        w.scat = (int)(p.misc & 0xffu);
                        mode = (int)((p.misc >> 8) & 0xffu);
                        w.dens = SD ? ((mybrick >= 0) ? sD[lid] : 0.0f) : p.dens;
                        nvisit = 0;
                    }
                }
                nobody_steps = (__ballot(mode == SOC_BM_STEP) == 0ull);
            }
        }
        // ---- create the work item's next packet ----
        if (soc_service_now(mode == SOC_BM_CREATE, nobody_steps)) {
            if (mode == SOC_BM_CREATE) {
                if (III >= S.BATCH) {
                    mode = SOC_BM_SWAP;  key = A.NB;  next_mode = SOC_BM_CREATE;     // work item finished
                } else {
                    const int id = (int)(S.gid0 + wid);
                    const SocSurfElem E = soc_surface_element(G, S, id);
                    soc_pb_create<false>(G, S, sOFF0, E, III, w);
                    III++;
                    n_pkt++;
                    w.begin();
                    if (w.ind >= 0) {
                        int b;
                        soc_cell_brick(A, w.px, w.py, w.pz, b, lid);
                        mode = SOC_BM_STEP;
                        if (b != mybrick) { mode = SOC_BM_SWAP;  key = b;  next_mode = SOC_BM_STEP; }
                    }                                                        // else: missed the cloud, create again
                }
            }
        }
        // ---- scattering block ----
        if (soc_service_now(mode == SOC_BM_SCATTER, nobody_steps)) {
            if (mode == SOC_BM_SCATTER) {
                const int oind = w.ind;
                float kabs, ksca;
                if (ABU) { float2 o = S.OPT[oind];  kabs = o.x;  ksca = o.y; }
                else     { kabs = S.ABS;  ksca = S.SCA; }
                w.scat++;
                float dt = w.free_path - w.tau;
                float dx = dt / (ksca * w.dens);
                float tauA = dx * w.dens * kabs;
                float e = soc_expf(-tauA);
                float delta = (tauA > SOC_TAULIM) ? (w.photons * (1.0f - e)) : (w.photons * tauA * (1.0f - 0.5f * tauA));
                atomicAdd(&sT[lid], delta * S.TW);
                if (WINT) atomicAdd(&sI[lid], delta);
                n_tally++;
                n_scat++;
                dx = soc_scale_up(dx, 0);
                dx = __builtin_fmaxf(0.0f, dx - 2.0f * SOC_PEPS);
                w.px = w.px + dx * w.ux;
                w.py = w.py + dx * w.uy;
                w.pz = w.pz + dx * w.uz;
                w.photons *= e;
                w.free_path = -soc_logf(soc_rand(&w.rng));
                soc_scatter(w.ux, w.uy, w.uz, S.CSC, S.BINS, &w.rng);
                w.tau = 0.0f;
                mode = (w.scat > 20) ? SOC_BM_CREATE : SOC_BM_STEP;          // dropped after 20 scatterings
                if (w.scat > 20) w.ind = -1;
            }
        }
        if (__ballot(mode != SOC_BM_IDLE) == 0ull) break;
        // ---- one cell step (kernel_ASOC.c:565-683, LEVELS == 1) ----
        if (mode == SOC_BM_STEP) {
            const int   oind = w.ind, lid0 = lid;
            const float p0x = w.px, p0y = w.py, p0z = w.pz, d0 = w.dens;
            float kabs, ksca;
            if (ABU) { float2 o = S.OPT[oind];  kabs = o.x;  ksca = o.y; }
            else     { kabs = S.ABS;  ksca = S.SCA; }
            float ax = (w.ux > 0.0f) ? (((1.0f + SOC_PEPS) - soc_fmod1f(w.px)) / w.ux) : ((-SOC_PEPS - soc_fmod1f(w.px)) / w.ux);
            float ay = (w.uy > 0.0f) ? (((1.0f + SOC_PEPS) - soc_fmod1f(w.py)) / w.uy) : ((-SOC_PEPS - soc_fmod1f(w.py)) / w.uy);
            float az = (w.uz > 0.0f) ? (((1.0f + SOC_PEPS) - soc_fmod1f(w.pz)) / w.uz) : ((-SOC_PEPS - soc_fmod1f(w.pz)) / w.uz);
            float ds = __builtin_fminf(ax, __builtin_fminf(ay, az));
            w.px += ds * w.ux;
            w.py += ds * w.uy;
            w.pz += ds * w.uz;
            ds = soc_scale_down(ds, 0);
            // new cell, without branches: inside <=> 0 < p < N on every axis (same outcome as the
            // reference's "<= 0 || >= N" exit test for every finite position)
            const bool inside = (w.px > 0.0f) & (w.px < fNX) & (w.py > 0.0f) & (w.py < fNY) & (w.pz > 0.0f) & (w.pz < fNZ);
            const int ix = inside ? (int)soc_floorf(w.px) : 0;
            const int iy = inside ? (int)soc_floorf(w.py) : 0;
            const int iz = inside ? (int)soc_floorf(w.pz) : 0;
            const int nind = iz * NX * NY + iy * NX + ix;
            const int nb   = ((iz >> LB) * A.NBY + (iy >> LB)) * A.NBX + (ix >> LB);
            const int nlid = ((iz & M) << (2 * LB)) | ((iy & M) << LB) | (ix & M);
            const bool stay = inside & (nb == mybrick);
            float ndens;
            if (SD) ndens = sD[stay ? nlid : 0];
            else    ndens = G.DENS[nind];
            const float tauA = ds * d0 * kabs;
            const float dtau = ds * d0 * ksca;
            if (w.free_path < (w.tau + dtau)) {
                w.px = p0x;  w.py = p0y;  w.pz = p0z;                        // back to the start of the step
                mode = SOC_BM_SCATTER;
            } else {
                const float e = soc_expf(-tauA);
                const float delta = (tauA > SOC_TAULIM) ? (w.photons * (1.0f - e)) : (w.photons * tauA * (1.0f - 0.5f * tauA));
                atomicAdd(&sT[lid0], delta * S.TW);
                if (WINT) atomicAdd(&sI[lid0], delta);
                n_tally++;
                w.photons *= e;
                w.tau += dtau;
                w.ind = inside ? nind : -1;
                w.dens = ndens;
                lid = nlid;
                const bool failed = (w.ind == oind);                         // failed step: nudge
                w.px += failed ? (SOC_PEPS * w.ux) : 0.0f;
                w.py += failed ? (SOC_PEPS * w.uy) : 0.0f;
                w.pz += failed ? (SOC_PEPS * w.uz) : 0.0f;
                nvisit++;
                // out of the cloud -> next packet; out of the brick, or this pass's step budget used
                // up -> back to the queue of the brick it is in (bounds the pass length)
                if (!inside)                              mode = SOC_BM_CREATE;
                else if (!stay || nvisit >= A.KCAP)     { mode = SOC_BM_SWAP;  key = nb;  next_mode = SOC_BM_STEP; }
            }
        }
    }

    if (A.dbg) t2 = wall_clock64();
    // ---- flush: brick tally -> global (rows of 1<<LB contiguous cells), histogram, stats ----
    atomicAdd(&sCtl[1], (int)n_tally);
    atomicAdd(&sCtl[2], (int)n_pkt);
    atomicAdd(&sCtl[3], (int)n_scat);
    __syncthreads();
    if (mybrick >= 0) {
        const int B = 1 << A.LB;
        const int bx = mybrick % A.NBX, by = (mybrick / A.NBX) % A.NBY, bz = mybrick / (A.NBX * A.NBY);
        for (int i = threadIdx.x; i < BV; i += nthr) {
            const float v = sT[i];
            const float vi = WINT ? sI[i] : 0.0f;
            if (v != 0.0f || vi != 0.0f) {
                const int ix = bx * B + (i & M), iy = by * B + ((i >> A.LB) & M), iz = bz * B + (i >> (2 * A.LB));
                const int cell = iz * NX * NY + iy * NX + ix;
                soc_tally(S.TABS, cell, v);
                if (WINT) soc_tally(S.INT, cell, vi);
            }
        }
    }
    for (int i = threadIdx.x; i <= A.NB; i += nthr) {
        const int c = sH[i];
        if (c) atomicAdd(&A.hist[i], c);
    }
    if (A.dbg && (threadIdx.x & 63) == 0) {
        // per wave: loop end; per WG (wave 0): start, init end, end, brick, count, n_tally
        long long *d = A.dbg + (size_t)blockIdx.x * 24;
        const int wv = threadIdx.x >> 6;
        d[8 + wv] = t2;
        if (wv == 0) { d[0] = t0; d[1] = t1; d[2] = wall_clock64(); d[3] = D.brick; d[4] = D.count; d[5] = sCtl[1]; d[6] = blockDim.x >> 6;
                       unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); d[7] = xcc; }
    }
    if (threadIdx.x == 0 && S.stats) {
        atomicAdd(S.stats + 0, (unsigned long long)(unsigned int)sCtl[1]);
        atomicAdd(S.stats + 1, (unsigned long long)(unsigned int)sCtl[2]);
        atomicAdd(S.stats + 2, (unsigned long long)(unsigned int)sCtl[3]);
    }
}

// one workgroup: hist -> offsets of the next queues + descriptors of the next pass
__global__ __launch_bounds__(1024) void soc_brick_scan(SocBrickArgs A)
{
    __shared__ int sSum[1024], sSumD[1024];
    const int NB = A.NB, tid = threadIdx.x;
    const int per = (NB + 1023) / 1024;
    const int b0 = tid * per, b1 = min(NB, b0 + per);
    int s = 0, sd = 0;
    for (int b = b0; b < b1; b++) {
        const int c = A.hist[b];
        s += c;
        sd += (c + A.P - 1) / A.P;
    }
    sSum[tid] = s;
    sSumD[tid] = sd;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan
        int v = 0, vd = 0;
        if (tid >= d) { v = sSum[tid - d];  vd = sSumD[tid - d]; }
        __syncthreads();
        sSum[tid] += v;
        sSumD[tid] += vd;
        __syncthreads();
    }
    int off = sSum[tid] - s, offd = sSumD[tid] - sd;
    for (int b = b0; b < b1; b++) {
        const int c = A.hist[b];
        A.off[b] = off;
        A.cursor[b] = 0;
        for (int k = 0; k * A.P < c; k++) {
            SocDesc d;
            d.brick = b;
            d.start = off + k * A.P;
            d.count = min(A.P, c - k * A.P);
            d.pad = 0;
            A.desc_next[offd++] = d;
        }
        off += c;
        A.hist[b] = 0;
    }
    if (tid == 1023) {
        A.off[NB] = sSum[1023];
        *A.total = sSum[1023];
        *A.ndesc_next = sSumD[1023];
        A.hist[NB] = 0;
    }
}

// counting-sort placement: ids of the current queue -> next queues, by destination brick
__global__ __launch_bounds__(SOC_BRICK_T) void soc_brick_scatter(SocBrickArgs A)
{
    if ((int)blockIdx.x >= *A.ndesc) return;
    const SocDesc D = A.desc[blockIdx.x];
    extern __shared__ int sB[];                          // [NB]
    for (int i = threadIdx.x; i < A.NB; i += SOC_BRICK_T) sB[i] = 0;
    __syncthreads();
    uint32_t key[SOC_BRICK_PMAX / SOC_BRICK_T];
    int      rank[SOC_BRICK_PMAX / SOC_BRICK_T];
#pragma unroll
    for (int k = 0; k < SOC_BRICK_PMAX / SOC_BRICK_T; k++) {
        const int j = k * SOC_BRICK_T + threadIdx.x;
        key[k] = (uint32_t)A.NB;
        rank[k] = 0;
        if (j < D.count) {
            key[k] = A.keyq[D.start + j];
            if (key[k] < (uint32_t)A.NB) rank[k] = atomicAdd(&sB[key[k]], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A.NB; i += SOC_BRICK_T) {
        const int c = sB[i];
        if (c) sB[i] = A.off[i] + atomicAdd(&A.cursor[i], c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SOC_BRICK_PMAX / SOC_BRICK_T; k++) {
        const int j = k * SOC_BRICK_T + threadIdx.x;
        if (j < D.count && key[k] < (uint32_t)A.NB) A.idq_next[sB[key[k]] + rank[k]] = A.idq[D.start + j];
    }
}

// ---------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------

struct SocBrickBuffers {
    size_t cap_items = 0;
    int    cap_nb = 0, cap_desc = 0;
    SocPacket *pk = nullptr;
    uint32_t *idq[2] = { nullptr, nullptr }, *keyq = nullptr;
    int *hist = nullptr, *off = nullptr, *cursor = nullptr, *ndesc = nullptr, *total = nullptr;
    SocDesc *desc[2] = { nullptr, nullptr };
};

static SocBrickBuffers g_bb[16];                          // one set per device ordinal

#define BCHK(call)                          \
    do {                                    \
        hipError_t e_ = (call);             \
        if (e_ != hipSuccess) return e_;    \
    } while (0)

template <typename T>
static hipError_t brick_alloc(T **p, size_t n)
{
    if (*p) { (void)hipFree(*p);  *p = nullptr; }
    return hipMalloc((void **)p, (n ? n : 1) * sizeof(T));
}

void soc_brick_release(int device)
{
    if (device < 0 || device >= 16) return;
    SocBrickBuffers &b = g_bb[device];
    void *ptrs[] = { b.pk, b.idq[0], b.idq[1], b.keyq, b.hist, b.off, b.cursor, b.ndesc, b.total, b.desc[0], b.desc[1] };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    b = SocBrickBuffers();
}

// LB: log2 of the brick edge.  Returns hipErrorNotSupported when the launch cannot use bricks.
hipError_t soc_brick_run_pb(int device, const SocGrid &G, const SocSim &Sin, const SocVariant &V, int LB, hipStream_t st,
                            int *passes_out)
{
    if (V.octree || device < 0 || device >= 16) return hipErrorNotSupported;
    SocSim S = Sin;
    const int B = 1 << LB;
    SocBrickArgs A{};
    A.LB = LB;
    A.NBX = (G.NX + B - 1) / B;  A.NBY = (G.NY + B - 1) / B;  A.NBZ = (G.NZ + B - 1) / B;
    A.NB = A.NBX * A.NBY * A.NBZ;
    if (A.NB > 8192) return hipErrorNotSupported;
    // workgroup shape; overridable for experiments
    // measured on C2 (128^3, 786k packets in flight): T=512, P=2048, KCAP=32 is the best of the sweep
    A.T = 512;
    A.P = 4 * A.T;
    A.KCAP = 32;
    if (const char *e = getenv("SOC_BRICK_T")) A.T = atoi(e);
    if (const char *e = getenv("SOC_BRICK_P")) A.P = atoi(e);
    if (const char *e = getenv("SOC_BRICK_KCAP")) A.KCAP = atoi(e);
    A.FTH = 8;
    if (const char *e = getenv("SOC_BRICK_FTH")) A.FTH = atoi(e);
    if (A.T < 64 || A.T > 1024 || (A.T & 63) || A.P < 1 || A.P > SOC_BRICK_PMAX || A.KCAP < 1) return hipErrorInvalidValue;
    uint32_t count = S.gid_count;
    if (S.SOURCE == 1) {
        const long long lim = 8LL * 2 * ((long long)G.NX * G.NY + (long long)G.NY * G.NZ + (long long)G.NZ * G.NX);
        if ((long long)S.gid0 >= lim) return hipSuccess;
        if ((long long)S.gid0 + count > lim) count = (uint32_t)(lim - S.gid0);
    }
    if (count == 0 || S.BATCH <= 0) return hipSuccess;
    const int maxdesc = (int)((count + A.P - 1) / A.P) + A.NB + 1;

    SocBrickBuffers &bb = g_bb[device];
    if (bb.cap_items < count) {
        BCHK(hipStreamSynchronize(st));
        BCHK(brick_alloc(&bb.pk, count));
        BCHK(brick_alloc(&bb.idq[0], count));
        BCHK(brick_alloc(&bb.idq[1], count));
        BCHK(brick_alloc(&bb.keyq, count));
        bb.cap_items = count;
    }
    if (bb.cap_nb < A.NB + 1) {
        BCHK(hipStreamSynchronize(st));
        BCHK(brick_alloc(&bb.hist, A.NB + 1));
        BCHK(brick_alloc(&bb.off, A.NB + 1));
        BCHK(brick_alloc(&bb.cursor, A.NB + 1));
        bb.cap_nb = A.NB + 1;
    }
    if (bb.cap_desc < maxdesc) {
        BCHK(hipStreamSynchronize(st));
        BCHK(brick_alloc(&bb.desc[0], maxdesc));
        BCHK(brick_alloc(&bb.desc[1], maxdesc));
        bb.cap_desc = maxdesc;
    }
    if (!bb.ndesc) { BCHK(brick_alloc(&bb.ndesc, 2));  BCHK(brick_alloc(&bb.total, 1)); }

    A.pk = bb.pk;  A.keyq = bb.keyq;  A.hist = bb.hist;  A.off = bb.off;  A.cursor = bb.cursor;  A.total = bb.total;
    const int BV = 1 << (3 * LB);
    bool use_sd = false;      // LDS copy of the brick densities: measured no gain (268 vs 274 ms at C2), costs LDS
    if (const char *e = getenv("SOC_BRICK_SD")) use_sd = atoi(e) != 0;
    const size_t lds_step = (size_t)(BV * (1 + (use_sd ? 1 : 0) + (V.wint ? 1 : 0)) + A.NB + 1 + 8) * 4;
    const size_t lds_scat = (size_t)A.NB * 4;

    soc_brick_init<<<(max(count, (uint32_t)A.NB + 1) + 255) / 256, 256, 0, st>>>(G, S, A, count, bb.idq[0], bb.desc[0], bb.ndesc, bb.hist);
    BCHK(hipGetLastError());
    int cur = 0, passes = 0, total = 1;
    int dbg_pass = -1;
    long long *dbg_buf = nullptr;
    if (const char *e = getenv("SOC_BRICK_DBG")) {
        dbg_pass = atoi(e);
        BCHK(hipMalloc((void **)&dbg_buf, (size_t)maxdesc * 24 * sizeof(long long)));
        BCHK(hipMemset(dbg_buf, 0, (size_t)maxdesc * 24 * sizeof(long long)));
    }
    while (total > 0) {
        for (int k = 0; k < 64; k++, passes++) {
            A.dbg = (passes == dbg_pass) ? dbg_buf : nullptr;
            A.idq = bb.idq[cur];  A.idq_next = bb.idq[1 - cur];
            A.desc = bb.desc[cur];  A.ndesc = bb.ndesc + cur;
            A.desc_next = bb.desc[1 - cur];  A.ndesc_next = bb.ndesc + (1 - cur);
            const int key = (V.abu ? 4 : 0) | (V.wint ? 2 : 0) | (use_sd ? 1 : 0);
            switch (key) {
            case 0: soc_brick_step<false, false, false><<<maxdesc, A.T, lds_step, st>>>(G, S, A); break;
            case 1: soc_brick_step<false, false, true><<<maxdesc, A.T, lds_step, st>>>(G, S, A); break;
            case 2: soc_brick_step<false, true, false><<<maxdesc, A.T, lds_step, st>>>(G, S, A); break;
            case 3: soc_brick_step<false, true, true><<<maxdesc, A.T, lds_step, st>>>(G, S, A); break;
            case 4: soc_brick_step<true, false, false><<<maxdesc, A.T, lds_step, st>>>(G, S, A); break;
            case 5: soc_brick_step<true, false, true><<<maxdesc, A.T, lds_step, st>>>(G, S, A); break;
            case 6: soc_brick_step<true, true, false><<<maxdesc, A.T, lds_step, st>>>(G, S, A); break;
            default: soc_brick_step<true, true, true><<<maxdesc, A.T, lds_step, st>>>(G, S, A); break;
            }
            soc_brick_scan<<<1, 1024, 0, st>>>(A);
            soc_brick_scatter<<<maxdesc, SOC_BRICK_T, lds_scat, st>>>(A);
            cur = 1 - cur;
        }
        BCHK(hipGetLastError());
        BCHK(hipMemcpyAsync(&total, bb.total, sizeof(int), hipMemcpyDeviceToHost, st));
        BCHK(hipStreamSynchronize(st));
        if (passes > 4000000) return hipErrorUnknown;     // cannot happen: every pass retires work
    }
    if (dbg_buf) {
        std::vector<long long> h((size_t)maxdesc * 24);
        BCHK(hipMemcpy(h.data(), dbg_buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        if (FILE *fp = fopen("gpurun_out/brick_dbg.bin", "wb")) { fwrite(h.data(), sizeof(long long), h.size(), fp); fclose(fp); }
        (void)hipFree(dbg_buf);
    }
    if (passes_out) *passes_out = passes;
    return hipSuccess;
}
