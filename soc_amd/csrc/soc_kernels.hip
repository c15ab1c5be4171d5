// soc_kernels.hip -- photon-packet kernels for gfx950 (MI355X).
//
// What is computed is what the reference's SimRAM_PB / SimRAM_CL compute
// (kernel_ASOC.c:15-824, 1223-1689 with helpers kernel_ASOC_aux.c:131-561): packets are
// created, walked cell by cell through the (optionally hierarchical) grid, deposit
// absorbed photons in per-cell tallies and scatter off a tabulated cumulative phase
// function, with one MWC64X stream per logical work item.  How it is computed is new:
//
//  * one lane = one logical work item, and a lane never waits at a packet boundary:
//    packet creation, the cell step and the scattering event are arms of ONE flat
//    per-lane state machine, so the 64 lanes of a wave stay busy until each has
//    finished its whole BATCH (the reference nests three loops per work item);
//  * geometry is a run-time argument (no -D NX=... rebuilds); the feature switches of
//    the reference (#if LEVELS / NX>DIMLIM / WITH_ABU / NOABSORBED) are template
//    parameters instantiated ahead of time;
//  * the cumulative scattering function and the level offsets are staged in LDS once
//    per workgroup; the density of the cell a packet has just entered is fetched when
//    the cell is found, one step before it is needed;
//  * tallies use the hardware's no-return global_atomic_add_f32 (no CAS loop);
//  * RNG streams are seeded with 4 table reads + 4 modular multiplications (soc_rng.h);
//  * all fp32 arithmetic follows the reference's operand order with contraction off and
//    transcendental functions from soc_math.h, so a host build of the same header
//    (the test oracle) reproduces every trajectory bit for bit.
#include "soc_walk.h"

// ------------------------------------------------------------------------------------
// SimRAM_PB: point sources (SOURCE 0) and isotropic background (SOURCE 1)
// ------------------------------------------------------------------------------------

template <bool OCT, bool DBL, bool ABU, bool WINT>
__global__ __launch_bounds__(256) void soc_sim_pb_kernel(const SocGrid G, const SocSim S)
{
    extern __shared__ float lds[];
    float *sCSC = lds;
    int   *sOFF = (int *)(lds + S.BINS);
    int   *sLC  = sOFF + SOC_MAXL;
    soc_stage_lds(G, S, sCSC, sOFF, sLC);

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.gid_count) return;
    const int id = (int)(S.gid0 + t);                      // logical get_global_id(0)
    const int AREA = 2 * (G.NX * G.NY + G.NY * G.NZ + G.NZ * G.NX);
    if ((S.SOURCE == 1) && (id >= 8 * AREA)) return;
    if ((S.SOURCE == 3) && (!S.ROI || !S.ROI->load || (id >= 100 * S.ROI->NELEM))) return;     // kernel_ASOC.c:97-105

    SocWalker<OCT, DBL, ABU, WINT> w;
    w.rng = soc_seed_stream(S.seed_mul, S.seed_tab, (uint32_t)id);
    w.ind = -1;  w.level = 0;  w.n_tally = 0;  w.n_scat = 0;
    w.ux = w.uy = w.uz = 0.0f;  w.px = w.py = w.pz = 0.0f;
    w.dens = 0.0f;  w.photons = 0.0f;  w.tau = 0.0f;  w.free_path = 0.0f;  w.scat = 0;
    w.roi_on = (S.ROI != nullptr) && (S.ROI->save != 0);

    const SocSurfElem E = soc_surface_element(G, S, id);

    // SOURCE == 3 (kernel_ASOC.c:141-179): 100 work items per surface element of the loaded record, each sends
    // BATCH packets, the Healpix pixels of the element in turn; patch centre (RDX, RDY) on side pair `rside`
    int   relem = 0, rside = 0;
    float RDX = 0.0f, RDY = 0.0f, rd = 1.0f, RX0 = 0.0f;
    if (S.SOURCE == 3) {
        const SocRoi &R = *S.ROI;
        relem = id % R.NELEM;
        int iside = relem;
        rd = (float)G.NX / ((float)R.DIM[0]);
        if (iside < (R.DIM[1] * R.DIM[2])) {
            RDX = ((float)(iside % R.DIM[1]) + 0.5f) * rd;  RDY = ((float)(iside / R.DIM[1]) + 0.5f) * rd;  rside = 0;
        } else {
            iside -= R.DIM[1] * R.DIM[2];
            if (iside < (R.DIM[0] * R.DIM[2])) {
                RDX = ((float)(iside % R.DIM[0]) + 0.5f) * rd;  RDY = ((float)(iside / R.DIM[0]) + 0.5f) * rd;  rside = 1;
            } else {
                iside -= R.DIM[0] * R.DIM[2];
                rside = 3;
                if (iside < (R.DIM[0] * R.DIM[1])) {
                    RDX = ((float)(iside % R.DIM[0]) + 0.5f) * rd;  RDY = ((float)(iside / R.DIM[0]) + 0.5f) * rd;  rside = 2;
                }
            }
        }
        RX0 = (float)((double)(R.NSIDE * R.NSIDE) * 12.0 / (100.0 * (double)S.BATCH));
    }

    int III = 0;
    int mode = SOC_M_CREATE;
    while (true) {
        const bool nobody_steps = (__ballot(mode == SOC_M_STEP) == 0ull);
        if (soc_service_now(mode == SOC_M_CREATE, nobody_steps)) {
            if (mode == SOC_M_CREATE) {
                if (S.SOURCE == 3) {
                    // kernel_ASOC.c:469-501; an empty pixel is skipped without a draw
                    const SocRoi &R = *S.ROI;
                    const int npix = 12 * R.NSIDE * R.NSIDE;
                    mode = SOC_M_DONE;
                    while (III < S.BATCH) {
                        const int pix = III % npix;
                        III++;
                        w.photons = RX0 * R.LOAD[(size_t)relem * npix + pix];
                        if (w.photons <= 0.0f) continue;
                        float v1, v2, s1, c1, s2, c2;
                        soc_pixel2angles_ring(R.NSIDE, pix, v1, v2);
                        v1 += (soc_rand(&w.rng) - 0.5f) * 0.05f;
                        v2 += (soc_rand(&w.rng) - 0.5f) * 0.05f;
                        soc_sincosf(v1, &s1, &c1);
                        soc_sincosf(v2, &s2, &c2);
                        w.ux = s2 * c1;  w.uy = s2 * s1;  w.uz = c2;
                        if (rside == 0) {
                            w.py = RDX + (-0.49f + 0.98f * soc_rand(&w.rng)) * rd;  w.pz = RDY + (-0.49f + 0.98f * soc_rand(&w.rng)) * rd;
                            w.px = (w.ux > 0.0f) ? SOC_PEPS : ((float)G.NX - SOC_PEPS);
                        }
                        if (rside == 1) {
                            w.px = RDX + (-0.49f + 0.98f * soc_rand(&w.rng)) * rd;  w.pz = RDY + (-0.49f + 0.98f * soc_rand(&w.rng)) * rd;
                            w.py = (w.uy > 0.0f) ? SOC_PEPS : ((float)G.NY - SOC_PEPS);
                        }
                        if (rside == 2) {
                            w.px = RDX + (-0.49f + 0.98f * soc_rand(&w.rng)) * rd;  w.py = RDY + (-0.49f + 0.98f * soc_rand(&w.rng)) * rd;
                            w.pz = (w.uz > 0.0f) ? SOC_PEPS : ((float)G.NZ - SOC_PEPS);
                        }
                        soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                        w.begin(S);
                        if (w.roi_on) w.roi = soc_inroi(G, sOFF, *S.ROI, w.level, w.ind);
                        mode = (w.ind >= 0) ? SOC_M_STEP : SOC_M_CREATE;
                        break;
                    }
                } else if (III >= S.BATCH) {
                    mode = SOC_M_DONE;
                } else {
                    soc_pb_create<OCT>(G, S, sOFF, E, III, w);
                    III++;
                    w.begin(S);
                    if (w.roi_on) w.roi = soc_inroi(G, sOFF, *S.ROI, w.level, w.ind);       // kernel_ASOC.c:550
                    mode = (w.ind >= 0) ? SOC_M_STEP : SOC_M_CREATE;
                }
            }
        }
        if (soc_service_now(mode == SOC_M_SCATTER, nobody_steps)) {
            if (mode == SOC_M_SCATTER) {
                w.template scatter<false>(S, sCSC, sOFF);
                mode = (w.ind >= 0) ? SOC_M_STEP : SOC_M_CREATE;
            }
        }
        if (__ballot(mode != SOC_M_DONE) == 0ull) break;
        if (mode == SOC_M_STEP) {
            if (w.template step<false>(G, S, sOFF)) mode = SOC_M_SCATTER;
            else if (w.ind < 0) mode = SOC_M_CREATE;
        }
    }
    soc_flush_stats(S, w.n_tally, (unsigned int)III, w.n_scat);
}

// ------------------------------------------------------------------------------------
// SimRAM_HP: background from a Healpix sky map (kernel_ASOC.c:826-1207); the walk is SimRAM_PB's
// ------------------------------------------------------------------------------------

template <bool OCT, bool DBL, bool ABU, bool WINT>
__global__ __launch_bounds__(256) void soc_sim_hp_kernel(const SocGrid G, const SocSim S)
{
    extern __shared__ float lds[];
    float *sCSC = lds;
    int   *sOFF = (int *)(lds + S.BINS);
    int   *sLC  = sOFF + SOC_MAXL;
    soc_stage_lds(G, S, sCSC, sOFF, sLC);

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.gid_count) return;
    const int id = (int)(S.gid0 + t);                      // logical get_global_id(0)
    const int AREA = 2 * (G.NX * G.NY + G.NY * G.NZ + G.NZ * G.NX);
    if (id >= 8 * AREA) return;                            // kernel_ASOC.c:874

    SocWalker<OCT, DBL, ABU, WINT> w;
    w.rng = soc_seed_stream(S.seed_mul, S.seed_tab, (uint32_t)id);
    w.ind = -1;  w.level = 0;  w.n_tally = 0;  w.n_scat = 0;
    w.ux = w.uy = w.uz = 0.0f;  w.px = w.py = w.pz = 0.0f;
    w.dens = 0.0f;  w.photons = 0.0f;  w.tau = 0.0f;  w.free_path = 0.0f;  w.scat = 0;


    int III = 0;
    int mode = SOC_M_CREATE;
    while (true) {
        const bool nobody_steps = (__ballot(mode == SOC_M_STEP) == 0ull);
        if (soc_service_now(mode == SOC_M_CREATE, nobody_steps)) {
            if (mode == SOC_M_CREATE) {
                if (III >= S.BATCH) {
                    mode = SOC_M_DONE;
                } else {
                    soc_hp_create<OCT>(G, S, sOFF, w);
                    III++;
                    w.begin_conditioned(S);
                    mode = (w.ind >= 0) ? SOC_M_STEP : SOC_M_CREATE;
                }
            }
        }
        if (soc_service_now(mode == SOC_M_SCATTER, nobody_steps)) {
            if (mode == SOC_M_SCATTER) {
                w.template scatter<false>(S, sCSC, sOFF);
                mode = (w.ind >= 0) ? SOC_M_STEP : SOC_M_CREATE;
            }
        }
        if (__ballot(mode != SOC_M_DONE) == 0ull) break;
        if (mode == SOC_M_STEP) {
            if (w.template step<false>(G, S, sOFF)) mode = SOC_M_SCATTER;
            else if (w.ind < 0) mode = SOC_M_CREATE;
        }
    }
    soc_flush_stats(S, w.n_tally, (unsigned int)III, w.n_scat);
}

// ------------------------------------------------------------------------------------
// SimRAM_CL: emission from the cells themselves (diffuse field / dust re-emission)
// ------------------------------------------------------------------------------------

template <bool OCT, bool DBL, bool ABU, bool WINT>
__global__ __launch_bounds__(256) void soc_sim_cl_kernel(const SocGrid G, const SocSim S)
{
    extern __shared__ float lds[];
    float *sCSC = lds;
    int   *sOFF = (int *)(lds + S.BINS);
    int   *sLC  = sOFF + SOC_MAXL;
    soc_stage_lds(G, S, sCSC, sOFF, sLC);

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.gid_count) return;
    const int id = (int)(S.gid0 + t);
    if (id >= G.CELLS) return;
    const int NX = G.NX, NY = G.NY;

    SocWalker<OCT, DBL, ABU, WINT> w;
    w.rng = soc_seed_stream(S.seed_mul, S.seed_tab, (uint32_t)id);
    w.ind = -1;  w.level = 0;  w.n_tally = 0;  w.n_scat = 0;
    w.ux = w.uy = w.uz = 0.0f;  w.px = w.py = w.pz = 0.0f;
    w.dens = 0.0f;  w.photons = 0.0f;  w.tau = 0.0f;  w.free_path = 0.0f;  w.scat = 0;
    w.roi_on = (S.ROI != nullptr) && (S.ROI->save != 0);

    long long ICELL = (long long)id - S.GLOBAL;            // kernel_ASOC.c:1283-1290
    long long IND = (long long)id - S.GLOBAL;              // USE_EMWEIGHT == 2: position in EMINDEX (:1759)
    int   IRAY = 0, batch = -1;
    float PWEI = 1.0f;
    unsigned int n_pkt = 0;
    int   mode = SOC_M_CREATE;

    while (true) {
        const bool nobody_steps = (__ballot(mode == SOC_M_STEP) == 0ull);
        if (soc_service_now(mode == SOC_M_CREATE, nobody_steps)) {
            if (mode == SOC_M_CREATE) {
                if ((S.USE_EMWEIGHT == 2) && (IRAY >= batch)) {
                    // the host's list of emitting cells, 100 packets from each (kernel_ASOC.c:1775-1790);
                    // a listed cell 0 is skipped by the reference's "> 0" test, -1 ends the list
                    IRAY = 0;
                    batch = 100;
                    while (true) {
                        IND += S.GLOBAL;
                        if (IND >= G.CELLS) { mode = SOC_M_DONE; break; }
                        const int e = S.EMINDEX[IND];
                        if (e < 0) { mode = SOC_M_DONE; break; }
                        if (e > 0) { ICELL = e;  PWEI = S.EMWEI[e];  break; }
                    }
                } else if (IRAY >= batch) {                // next emitting cell (kernel_ASOC.c:1318-1355)
                    IRAY = 0;
                    PWEI = 1.0f;
                    while (true) {
                        ICELL += S.GLOBAL;
                        if (ICELL >= G.CELLS) { mode = SOC_M_DONE; break; }
                        if (S.USE_EMWEIGHT > 0) {
                            PWEI = S.EMWEI[ICELL];
                            if ((PWEI < 1e-10f) || (G.DENS[ICELL] <= 0.0f)) continue;
                            batch = (int)soc_floorf(PWEI);
                            if (batch < 1) {
                                batch = 1;
                                PWEI  = (float)(1.0 / (double)(PWEI + 1.0e-30f));
                            } else {
                                PWEI = (float)(1.0 / (double)(batch + 1.0e-9f));
                            }
                        } else {
                            batch = S.BATCH;
                            PWEI  = 1.0f / (batch + 1.0e-9f);
                        }
                        break;
                    }
                }
                if (mode != SOC_M_DONE) {
                    int ind = (int)ICELL;
                    IRAY += 1;
                    int level;
                    for (level = 0; level < G.LEVELS - 1; level++) {
                        ind -= sLC[level];
                        if (ind < 0) {
                            ind += sLC[level];
                            break;
                        }
                    }
                    float X0, Y0, Z0;
                    if (level == 0) {
                        X0 = (ind % NX);
                        Y0 = ((ind / NX) % NY);
                        Z0 = (ind / (NX * NY));
                    } else {
                        int sid = ind % 8;
                        X0 = (sid % 2);
                        Y0 = ((sid % 4) > 1) ? 1.0f : 0.0f;
                        Z0 = (sid / 4);
                    }
                    w.level   = level;
                    w.ind     = ind;
                    w.e_index = sOFF[level] + ind;
                    w.dens    = G.DENS[sOFF[level] + ind];
                    w.photons = S.EMIT[sOFF[level] + ind] * PWEI;
                    w.px = X0 + soc_rand(&w.rng);
                    w.py = Y0 + soc_rand(&w.rng);
                    w.pz = Z0 + soc_rand(&w.rng);
                    float phi       = SOC_TWOPI * soc_rand(&w.rng);
                    float cos_theta = 0.999997f - 1.999995f * soc_rand(&w.rng);
                    float sin_theta = soc_sqrtf(1.0f - cos_theta * cos_theta);
                    float sp, cp;
                    soc_sincosf(phi, &sp, &cp);
                    w.ux = sin_theta * cp;
                    w.uy = sin_theta * sp;
                    w.uz = cos_theta;
                    n_pkt++;
                    w.begin(S);
                    if (w.roi_on) w.roi = soc_inroi(G, sOFF, *S.ROI, w.level, w.ind);       // kernel_ASOC.c:1439
                    mode = SOC_M_STEP;
                }
            }
        }
        if (soc_service_now(mode == SOC_M_SCATTER, nobody_steps)) {
            if (mode == SOC_M_SCATTER) {
                w.template scatter<true>(S, sCSC, sOFF);
                mode = (w.ind >= 0) ? SOC_M_STEP : SOC_M_CREATE;
            }
        }
        if (__ballot(mode != SOC_M_DONE) == 0ull) break;
        if (mode == SOC_M_STEP) {
            if (w.template step<true>(G, S, sOFF)) mode = SOC_M_SCATTER;
            else if (w.ind < 0) mode = SOC_M_CREATE;
        }
    }
    soc_flush_stats(S, w.n_tally, n_pkt, w.n_scat);
}

// ------------------------------------------------------------------------------------
// Parents (kernel_ASOC_aux.c:688-718): child -> parent link table
// ------------------------------------------------------------------------------------
__global__ void soc_parents_kernel(const SocGrid G, int *PAR)
{
    const int stride = gridDim.x * blockDim.x;
    for (int level = 0; level < G.LEVELS - 1; level++) {
        const int nchild = G.LCELLS[level + 1];
        for (int ipar = blockIdx.x * blockDim.x + threadIdx.x; ipar < G.LCELLS[level]; ipar += stride) {
            float link = G.DENS[G.OFF[level] + ipar];
            if (link < 1.0e-10f) {
                int first = soc_link_index(link);
                if ((first >= 0) && (first + 8 <= nchild)) {
                    for (int i = 0; i < 8; i++) PAR[G.OFF[level + 1] - G.NXYZ + first + i] = ipar;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// probes used by the parity tests (RNG streams, math header, single-ray traces)
// ------------------------------------------------------------------------------------
__global__ void soc_seed_probe_kernel(uint64_t seed_mul, const uint64_t *tab, uint32_t gid0, uint32_t n,
                                      int ndraw, uint32_t *out_state, uint32_t *out_draws)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    soc_rng_t s = soc_seed_stream(seed_mul, tab, gid0 + t);
    out_state[2 * t]     = s.x;
    out_state[2 * t + 1] = s.c;
    for (int i = 0; i < ndraw; i++) out_draws[(size_t)t * ndraw + i] = soc_next_uint(&s);
}

__global__ void soc_math_probe_kernel(int fn, const float *x, float *y, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i], r = 0.0f;
    switch (fn) {
    case 0: r = soc_expf(v); break;
    case 1: r = soc_logf(v); break;
    case 2: r = soc_sinf(v); break;
    case 3: r = soc_cosf(v); break;
    case 4: r = soc_acosf(v); break;
    case 5: r = soc_sqrtf(v); break;
    case 6: r = soc_fmod1f(v); break;
    case 7: r = 1.0f / v; break;
    case 8: r = soc_expm1f(v); break;
    case 9: r = soc_pow15f(v); break;
    case 10: r = (float)soc_logd((double)v); break;
    default: break;
    }
    y[i] = r;
}

template <bool OCT, bool DBL>
__global__ void soc_trace_kernel(const SocGrid G, const float *pos, const float *dir, int maxsteps,
                                 int *levels, int *inds, float *dss, float *endpos, int *nsteps)
{
    __shared__ int sOFF[SOC_MAXL];
    if (threadIdx.x < SOC_MAXL) sOFF[threadIdx.x] = G.OFF[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    float px = pos[0], py = pos[1], pz = pos[2];
    float ux = dir[0], uy = dir[1], uz = dir[2];
    int   level = 0, ind = -1, n = 0;
    float dens = 0.0f;
    soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens);
    while ((ind >= 0) && (n < maxsteps)) {
        levels[n] = level;
        inds[n]   = ind;
        dss[n]    = soc_getstep<OCT, DBL>(G, sOFF, px, py, pz, ux, uy, uz, level, ind, dens);
        n++;
    }
    endpos[0] = px;  endpos[1] = py;  endpos[2] = pz;
    *nsteps = n;
}

// ------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------

static inline size_t soc_lds_bytes(const SocSim &S) { return (size_t)S.BINS * 4 + 2 * SOC_MAXL * 4; }

static inline void soc_launch_shape(uint32_t count, dim3 &grid, dim3 &block)
{
    // 64-lane workgroups while the launch is too small to give every CU a 256-thread group
    int bs = (count >= 256u * 1024u) ? 256 : 64;
    block = dim3(bs);
    grid  = dim3((count + bs - 1) / bs);
}

#define SOC_DISPATCH(KERNEL)                                                                      \
    do {                                                                                          \
        const int key = (V.octree ? 8 : 0) | (V.dbl ? 4 : 0) | (V.abu ? 2 : 0) | (V.wint ? 1 : 0); \
        switch (key) {                                                                            \
        case 0:  KERNEL<false, false, false, false><<<grid, block, lds, st>>>(G, S); break;       \
        case 1:  KERNEL<false, false, false, true><<<grid, block, lds, st>>>(G, S); break;        \
        case 2:  KERNEL<false, false, true, false><<<grid, block, lds, st>>>(G, S); break;        \
        case 3:  KERNEL<false, false, true, true><<<grid, block, lds, st>>>(G, S); break;         \
        case 8:  KERNEL<true, false, false, false><<<grid, block, lds, st>>>(G, S); break;        \
        case 9:  KERNEL<true, false, false, true><<<grid, block, lds, st>>>(G, S); break;         \
        case 10: KERNEL<true, false, true, false><<<grid, block, lds, st>>>(G, S); break;         \
        case 11: KERNEL<true, false, true, true><<<grid, block, lds, st>>>(G, S); break;          \
        case 12: KERNEL<true, true, false, false><<<grid, block, lds, st>>>(G, S); break;         \
        case 13: KERNEL<true, true, false, true><<<grid, block, lds, st>>>(G, S); break;          \
        case 14: KERNEL<true, true, true, false><<<grid, block, lds, st>>>(G, S); break;          \
        case 15: KERNEL<true, true, true, true><<<grid, block, lds, st>>>(G, S); break;           \
        default: return hipErrorInvalidValue;   /* dbl without octree never differs from float */ \
        }                                                                                         \
    } while (0)

hipError_t soc_launch_sim_pb(const SocGrid &G, const SocSim &S, const SocVariant &Vin, hipStream_t st)
{
    if (S.gid_count == 0) return hipSuccess;
    SocVariant V = Vin;
    if (!V.octree) V.dbl = 0;        // Cartesian: Index() touches no double arithmetic at level 0
    dim3 grid, block;
    soc_launch_shape(S.gid_count, grid, block);
    const size_t lds = soc_lds_bytes(S);
    SOC_DISPATCH(soc_sim_pb_kernel);
    return hipGetLastError();
}

hipError_t soc_launch_sim_hp(const SocGrid &G, const SocSim &S, const SocVariant &Vin, hipStream_t st)
{
    if (S.gid_count == 0) return hipSuccess;
    SocVariant V = Vin;
    if (!V.octree) V.dbl = 0;
    dim3 grid, block;
    soc_launch_shape(S.gid_count, grid, block);
    const size_t lds = soc_lds_bytes(S);
    SOC_DISPATCH(soc_sim_hp_kernel);
    return hipGetLastError();
}

hipError_t soc_launch_sim_cl(const SocGrid &G, const SocSim &S, const SocVariant &Vin, hipStream_t st)
{
    if (S.gid_count == 0) return hipSuccess;
    SocVariant V = Vin;
    if (!V.octree) V.dbl = 0;
    dim3 grid, block;
    soc_launch_shape(S.gid_count, grid, block);
    const size_t lds = soc_lds_bytes(S);
    SOC_DISPATCH(soc_sim_cl_kernel);
    return hipGetLastError();
}

hipError_t soc_launch_parents(const SocGrid &G, int *PAR, hipStream_t st)
{
    if (G.LEVELS < 2) return hipSuccess;
    soc_parents_kernel<<<1024, 256, 0, st>>>(G, PAR);
    return hipGetLastError();
}

hipError_t soc_launch_seed_probe(uint64_t seed_mul, const uint64_t *tab, uint32_t gid0, uint32_t n,
                                 int ndraw, uint32_t *out_state, uint32_t *out_draws, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    soc_seed_probe_kernel<<<(n + 255) / 256, 256, 0, st>>>(seed_mul, tab, gid0, n, ndraw, out_state, out_draws);
    return hipGetLastError();
}

hipError_t soc_launch_math_probe(int fn, const float *x, float *y, long n, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    soc_math_probe_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(fn, x, y, n);
    return hipGetLastError();
}

hipError_t soc_launch_trace(const SocGrid &G, const SocVariant &V, const float *pos, const float *dir,
                            int maxsteps, int *levels, int *inds, float *dss, float *endpos, int *nsteps,
                            hipStream_t st)
{
    if (!V.octree)
        soc_trace_kernel<false, false><<<1, 64, 0, st>>>(G, pos, dir, maxsteps, levels, inds, dss, endpos, nsteps);
    else if (!V.dbl)
        soc_trace_kernel<true, false><<<1, 64, 0, st>>>(G, pos, dir, maxsteps, levels, inds, dss, endpos, nsteps);
    else
        soc_trace_kernel<true, true><<<1, 64, 0, st>>>(G, pos, dir, maxsteps, levels, inds, dss, endpos, nsteps);
    return hipGetLastError();
}
