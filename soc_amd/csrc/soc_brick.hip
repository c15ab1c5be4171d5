// soc_brick.hip -- "brick sweep" execution of SimRAM_PB / SimRAM_HP / SimRAM_CL, on Cartesian grids and on
// hierarchies: absorption tallies live in LDS, not on the fabric.
//
// Why: the direct kernel issues one scattered global float atomic per cell step.  On
// MI355X those execute at the memory side at ~2e10 64-B requests/s chip-wide whatever the
// footprint (MI355X_MICROARCH.md "Global float atomics"; measured here: TCC_EA0_ATOMIC ==
// tally events, kernel pinned at 1.94e10 steps/s while the same walk without tallies runs
// at 1.2e11 steps/s).  No scope or cache policy moves them into L2.
//
// How: the grid is cut into bricks (B^3 root cells; on a hierarchy sets of <= CAP neighbouring cells, see
// soc_oct_build).  In-flight packets (one per logical
// work item -- a work item's packets are sequential in its RNG stream, so one is in
// flight at a time) are kept sorted by the queue they wait in: one queue per brick, plus
// one for work items whose packet has left the cloud and one for packets whose free path
// has ended.  One pass =
//   soc_brick_walk    a workgroup takes a chunk of one brick's queue, keeps the brick's
//                     tally in LDS (ds_add_f32), walks each packet until it leaves the brick,
//                     leaves the cloud or must scatter, writes it back with its destination
//                     queue, then flushes the brick tally with row-contiguous atomics;
//   soc_brick_events  one lane per packet of the two event queues: scattering block, or
//                     creation of the work item's next packet (all RNG use is here);
//   soc_brick_scan    exclusive scan of the arrival histogram -> next queue offsets and
//                     workgroup descriptors (one workgroup, exact sizes, no capacity guess);
//                     and admission of waiting work items (population control);
//   soc_brick_scatter permutation of the packet ids into the next queues (every packet's place was settled
//                     in the pass: its rank in the workgroup's LDS count + what the histogram add returned).
// A kernel boundary separates the phases, so no in-launch inter-workgroup hand-off exists.
// Several launches (frequencies of a run, steps of the benchmark) can share one sweep: their
// work items are one population, each launch with its own pair of event queues.
//
// What does not change: the logical work items, their MWC64X streams, every fp32 operation
// of a packet's life (same code as soc_walk.h, same operand order).  Trajectories are
// identical to the direct kernel and to the oracle; only the order of fp32 tally additions
// differs (as it already does between any two runs of the atomic version).
#include "soc_walk.h"
#include "soc_ltree.h"

#include <cstdio>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "soc_lbricks.h"

#define SOC_BRICK_T 256          // threads per workgroup (scatter kernel; step kernel uses A.T)

// wave-level counters of the walk loop for experiments (-DSOC_BRICK_PROF): iterations, lanes stepping, arm executions
#if defined(SOC_BRICK_PROF)
__device__ unsigned long long g_soc_prof[24];
#define SOC_PROF_DECL unsigned int prof_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, dprof_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };  unsigned long long tprof_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlast_ = __builtin_readcyclecounter()
#define SOC_PROF_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter();  tprof_[i] += t_ - tlast_;  tlast_ = t_; } while (0)
#define SOC_PROF_ENTRY const unsigned long long tentry_ = __builtin_readcyclecounter()
#define SOC_PROF_SINCE_ENTRY(i) do { tprof_[i] += tlast_ - tentry_; } while (0)
#define SOC_PROF(i, n) do { prof_[i] += (unsigned int)(n); } while (0)           /* n is wave-uniform */
#define SOC_DPROF(i, n) do { dprof_[i] += (unsigned int)(n); } while (0)         /* diagnostics of the lane-bound walk: [16..23] */
#define SOC_PROF_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_soc_prof[16 + i_], (unsigned long long)dprof_[i_]); \
                            if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_soc_prof[i_], (unsigned long long)prof_[i_]); \
                            if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_soc_prof[8 + i_], tprof_[i_]); } while (0)   /* [6] is counted per lane */
extern "C" __attribute__((visibility("default"))) void soc_prof_read(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_soc_prof), sizeof(unsigned long long) * 24);
    if (reset) { unsigned long long z[24] = { 0 };  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_soc_prof), z, sizeof(z)); }
}
#else
#define SOC_PROF_DECL
#define SOC_PROF_ENTRY
#define SOC_PROF_SINCE_ENTRY(i) do { } while (0)
#define SOC_PROF_T(i) do { } while (0)
#define SOC_PROF(i, n) do { } while (0)
#define SOC_DPROF(i, n) do { } while (0)
#define SOC_PROF_FLUSH do { } while (0)
#endif
#define SOC_BRICK_PMAX 4096      // upper bound of packets per workgroup chunk (walks that keep the chunk's ranks in LDS)
#define SOC_RANK_BITS 16         // a packet's rank among its workgroup's packets for one queue; the table entry sits above (<= 4096 queues or 1024 hash entries)
#define SOC_LBRICK_PMAX 32768    // ... of soc_lbrick_walk, whose chunk lives in global memory only

enum { SOC_BM_STEP = 0, SOC_BM_CLIMB = 1, SOC_BM_SWAP = 3, SOC_BM_IDLE = 4 };

struct __align__(16) SocPk2 { float4 A, B, C; uint4 D; };   // 64-B packet record, see soc_brick_walk

struct SocDesc { int brick, start, count, pad; };

// Packet records and queue entries are touched once per pass.  Streaming them past the caches (-DSOC_BRICK_NT:
// nontemporal) was measured: no change on a 256^3 hierarchy, -23 % on C2 (whose packets fit the last-level cache).
#if defined(SOC_BRICK_NT)
#define SOC_NT_LOAD(p)      __builtin_nontemporal_load(p)
#define SOC_NT_STORE(v, p)  __builtin_nontemporal_store((v), (p))
#else
#define SOC_NT_LOAD(p)      (*(p))
#define SOC_NT_STORE(v, p)  (*(p) = (v))
#endif
typedef float soc_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t soc_u2v __attribute__((ext_vector_type(2)));
// Pointers that reach a kernel inside a struct are generic: the compiler emits FLAT loads and stores for them, and a flat
// operation counts as an LDS operation too (lgkmcnt) -- every wait for an LDS read then waits for the global loads in flight
// as well, the prefetched packet records included.  The walk's loop casts its pointers to the global address space.
#define SOC_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float4 soc_ld4(const float4 *p) { const soc_f4v v = SOC_NT_LOAD((const soc_f4v *)p);  return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void soc_st4(float4 *p, float4 a) { soc_f4v v = { a.x, a.y, a.z, a.w };  SOC_NT_STORE(v, (soc_f4v *)p); }

struct SocBrickArgs {
    int LB, NBX, NBY, NBZ, NB;   // brick edge = 1 << LB root cells; NB bricks
    int NBQ;                     // brick queues: NB, or NB per launch (launches with the INT tally: a workgroup's LDS
                                 // tallies then belong to one launch; queue = launch * NB + brick)
    int T, P, KCAP, FTH;         // step-kernel threads, packets per chunk, max steps per packet per pass, fetch threshold
    int CTH;                     // lanes waiting for the deferred Index() before that arm is entered (hierarchies)
    int TAIL;                    // a wave with this many lanes out of work sends its last packets back to the queue (0: never)
    int PARK;                    // brick queues shorter than this and than the mean brick queue are not walked this pass: their packets wait for company (0: never)
    SocPk2 *pk;
    const uint32_t *idq;         // current queue (ids sorted by brick)
    uint32_t *idq_next;
    uint32_t *keyq;              // destination brick of every entry of the current queue
    uint32_t *posq;              // ... and its place in that queue
    int *hist;                   // [NB+1] arrivals per brick in the next pass; [NB] = finished
    int *off;                    // [NB+1] offsets of the next queues (scan output)
    const SocDesc *desc;         // descriptors of the current pass
    const int *ndesc;
    SocDesc *desc_next;
    int *ndesc_next;
    int *total;                  // packets still in flight after this pass
    int ev_brick;                // scan: first event queue (descriptors from here on belong to soc_brick_events)
    int HS;                      // arrivals per destination, per workgroup: 0 = LDS table indexed by queue, else hash table of HS entries
    long long opt_stride;        // per-cell opacities of launch l start at OPT + l * opt_stride (several launches with abundances)
    // population control: work items [0, target) start at once, the others are admitted (in order) as work items
    // finish, so that the sweep runs with `target` packets in flight until the last launch drains
    int target, nl;
    const uint32_t *first;       // [nl + 1] first work item of every launch (the SocSimPack's, in device memory)
    int *admit;                  // [0] next work item to admit; per launch l: [1+3l] first id, [2+3l] how many, [3+3l] where (this pass)
    // hierarchical grids: bricks are sets of <= CAP leaf cells (soc_oct_build)
    int CAP;
    const float2 *DS;            // [CELLS] density or link as in DENS | brick << 14 + tally slot of a leaf (bits)
    float sib_thr;               // coordinates from here up climb to the parent exactly in double (see the walk)
    const int *bcell;            // cells of every brick in slot order
    const int *bbase;            // [NB+1] first entry of a brick in bcell
    // brick-local hierarchies (soc_ltree.h): the brick's cells live in LDS, the packet carries integer cell coordinates
    int LT;                      // 1: this form of the walk
    int EQ;                      // event queues per launch: creation, scattering (+ slow steps with LT)
    int slow_every;              // test knob: every n-th step below the root grid takes the slow-step queue (0: only the degenerate ones)
    const SocLBrick *lbr;        // [NB] boxes
    const float *btree;          // slots of every brick: density or link to the octet's slots
    const int *rbrick;           // [NX*NY*NZ] brick of every root cell
    int kexp;                    // k - 30 with 2^k > max(NX, NY, NZ): the bounds of soc_lt_move
    int int_only;                // the INT tally alone in LDS, TABS = TW * INT at the flush (WINT 3 of soc_lbrick_walk)
    int ali;                     // -D WITH_ALI: the launches (SimRAM_CL) tally what the emitting cell absorbs of its own into XAB
    int roi_on;                  // -D WITH_ROI_SAVE: a packet that steps into the region of interest goes through a fourth event queue of its launch
    // scattered-light images on brick-local hierarchies (soc_sca_events): the view, and the parked packet of every work item
    SocPk2 *park;
    SocSca sca;
};

#include "soc_octbricks.h"     // SOC_SLOT_BITS, SocOctBuilder
#define SOC_LVL_SHIFT 16         // packet word C.z: tally slot | level << 16 | launch << 20
#define SOC_LCH_SHIFT 20

// Arrivals per destination queue, counted per workgroup in LDS and added to the global histogram once.
// Small models: a table with one entry per queue.  Large ones (thousands of bricks): a workgroup's packets go to
// its neighbours and the event queues only, so an open-addressing table of HS entries; a key that finds no
// place within 32 probes is counted in global memory directly.
__device__ __forceinline__ void soc_qh_init(int *sH, int HS, int NQ)
{
    if (HS == 0) { for (int i = threadIdx.x; i < NQ; i += blockDim.x) sH[i] = 0; }
    else         { for (int i = threadIdx.x; i < HS; i += blockDim.x) { sH[i] = -1;  sH[HS + i] = 0; } }
}
// returns the table entry (for soc_brick_scatter's ranks), -1 when counted globally
__device__ __forceinline__ int soc_qh_find(int *sH, int HS, int key)
{
    uint32_t h = (((uint32_t)key * 0x9E3779B1u) >> 12) & (uint32_t)(HS - 1);
    for (int t = 0; t < 32; t++) {
        const int old = atomicCAS(&sH[h], -1, key);
        if (old == -1 || old == key) return (int)h;
        h = (h + 1) & (uint32_t)(HS - 1);
    }
    return -1;
}
// The place of a packet in its destination queue is settled in the pass itself: the LDS count a packet bumps
// is its rank among the workgroup's packets for that queue (returned as entry << SOC_RANK_BITS | rank); at the end the
// workgroup adds its counts to the global histogram, and what the add returns is where its packets start in
// the queue.  The sort is then a plain permutation (soc_brick_scatter).  SOC_POS_FINAL: counted in global
// memory directly (hash table full around that key), the value is the place itself.
#define SOC_POS_FINAL 0x80000000u
__device__ __forceinline__ uint32_t soc_qh_rank(int *sH, int HS, int key, int *ghist)
{
    if (HS == 0) return ((uint32_t)key << SOC_RANK_BITS) | (uint32_t)atomicAdd(&sH[key], 1);
    const int h = soc_qh_find(sH, HS, key);
    if (h >= 0) return ((uint32_t)h << SOC_RANK_BITS) | (uint32_t)atomicAdd(&sH[HS + h], 1);
    return SOC_POS_FINAL | (uint32_t)atomicAdd(&ghist[key], 1);
}
// counts -> first places (all threads of the workgroup; barrier before and after by the caller)
__device__ __forceinline__ void soc_qh_bases(int *sH, int HS, int NQ, int *ghist)
{
    if (HS == 0) {
        for (int i = threadIdx.x; i < NQ; i += blockDim.x) { const int c = sH[i];  if (c) sH[i] = atomicAdd(&ghist[i], c); }
    } else {
        for (int i = threadIdx.x; i < HS; i += blockDim.x) { const int k = sH[i], c = sH[HS + i];  if (k >= 0 && c) sH[HS + i] = atomicAdd(&ghist[k], c); }
    }
}
__device__ __forceinline__ uint32_t soc_qh_place(const int *sH, int HS, uint32_t pack)
{
    if (pack & SOC_POS_FINAL) return pack & ~SOC_POS_FINAL;
    return (uint32_t)sH[(HS ? HS : 0) + (pack >> SOC_RANK_BITS)] + (pack & ((1u << SOC_RANK_BITS) - 1u));
}

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

    __device__ __forceinline__ void begin_conditioned(const SocSim &S)    // SimRAM_HP: the direction is conditioned at creation
    {
        scat = 0;
        tau  = 0.0f;
        free_path = soc_draw_free_path(S, &rng, photons);
    }
    __device__ __forceinline__ void begin(const SocSim &S)
    {
        if (soc_fabsf(ux) < SOC_DEPS) ux = SOC_DEPS;
        if (soc_fabsf(uy) < SOC_DEPS) uy = SOC_DEPS;
        if (soc_fabsf(uz) < SOC_DEPS) uz = SOC_DEPS;
        soc_normalize(ux, uy, uz);
        scat = 0;
        tau  = 0.0f;
        free_path = soc_draw_free_path(S, &rng, photons);
    }
};


// ---------------------------------------------------------------------------------------
// Stepping and events in separate kernels.
//
// Measured on the first form of this file (one kernel with creation and scattering as arms of
// the stepping loop; rocprofv3 SQ counters, C2): 43 % of the lanes of an average VALU
// instruction active, waves 44 % of their cycles in s_waitcnt, 88 VGPRs.  The lanes idled
// because creation and scattering -- a few hundred instructions each, needed by ~1 lane in 130
// per iteration -- were arms of the same loop, and the kernel carried their registers.
// Here the two rare events are simply two more destinations of the sort the sweep does anyway:
//   queue NB   : work items whose packet has left the cloud  -> soc_brick_events creates the next
//   queue NB+1 : packets whose free path ends in their cell  -> soc_brick_events scatters them
//   queue NB+2 : finished work items (never scheduled)
// soc_brick_events runs one lane per queued packet at full utilisation and sends every packet
// to the queue of the brick it is in.  soc_brick_walk is left with one arm (the cell step) plus
// the swap, no RNG, no tables, and half the registers.
// Packet record (64 B): A = position, photons | B = direction, free_path | C = tau, density of
// the current cell, tally slot, cell index | D = RNG state, III | scat << 24, brick.
// ---------------------------------------------------------------------------------------
__global__ void soc_brick2_init(const SocSimPack *Kp, SocBrickArgs A, uint32_t count, uint32_t *idq0, SocDesc *desc0, int *ndesc0, int *hist)
{
    const SocSimPack &K = *Kp;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    SocPk2 *pk = A.pk;
    if (t < count) {
        int l = 0, hi = K.n;                                // the launch of work item t: first[l] <= t < first[l + 1]
        while (hi - l > 1) { const int m = (l + hi) >> 1;  if (t >= K.first[m]) l = m; else hi = m; }
        const soc_rng_t r = soc_seed_stream(K.S[l].seed_mul, K.S[l].seed_tab, K.S[l].gid0 + (t - K.first[l]));
        SocPk2 p;
        p.A = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        p.B = p.A;
        p.C = make_float4(0.0f, 0.0f, __int_as_float(l << SOC_LCH_SHIFT), __int_as_float(-1));
        // SimRAM_CL: the work item's cell, one GLOBAL before its first one (kernel_ASOC.c:1283)
        const uint32_t cell0 = (K.S[l].SOURCE == SOC_SOURCE_CL) ? (uint32_t)((int)(K.S[l].gid0 + (t - K.first[l])) - K.S[l].GLOBAL) : 0u;
        p.D = make_uint4(r.x, r.c, 0u, cell0);
        pk[t] = p;
        idq0[t] = t;
        if (A.park) { p.C = p.A;  p.D = make_uint4(0u, 0u, 0u, 0u);  A.park[t] = p; }      // no ray yet (SOC_RM_NONE)
    }
    // every launch starts in its own creation queue NB + 2l; the first A.target work items at once
    const uint32_t active = min(count, (uint32_t)A.target);
    uint32_t dbase = 0;
    for (int l = 0; l < K.n; l++) {
        const uint32_t hi = min(K.first[l + 1], active);
        const uint32_t cnt = (hi > K.first[l]) ? (hi - K.first[l]) : 0u;
        const uint32_t nd = (cnt + A.P - 1) / A.P;
        if (t >= dbase && t < dbase + nd) {
            const uint32_t k = t - dbase;
            SocDesc d;
            d.brick = A.NBQ + A.EQ * l;
            d.start = (int)(K.first[l] + k * A.P);
            d.count = (int)min((uint32_t)A.P, cnt - k * A.P);
            d.pad = 0;
            desc0[t] = d;
        }
        dbase += nd;
    }
    if (t == 0) { ndesc0[0] = (int)dbase;  ndesc0[2] = 0;  A.admit[0] = (int)active;  *A.total = (int)active; }
    if (t <= (uint32_t)(A.NBQ + A.EQ * K.n)) hist[t] = 0;
}

template <bool OCT, bool DBL, bool ABU, bool WINT, int KIND>
__device__ __forceinline__ void soc_brick_walk(const SocGrid &G, const SocSimPack &K, const SocBrickArgs &A, const int bid)
{
    constexpr bool CL = (KIND == 2);                       // SimRAM_CL: no nudge after a failed step, D.w holds the emitting cell
    if (bid >= *A.ndesc) return;
    const SocDesc D = A.desc[bid];
    if (D.brick >= A.NBQ) return;                          // an event queue: soc_brick_events
    const int BV = OCT ? A.CAP : (1 << (3 * A.LB));        // tally slots in LDS
    const int nthr = (int)blockDim.x;
    SocPk2 *pk = A.pk;

    extern __shared__ float lds[];
    float *sT   = lds;                                     // [BV] TABS of this brick
    float *sI   = sT + BV;                                 // [BV] INT (WINT)
    const int NQ = A.NBQ + A.EQ * K.n + 1;                 // brick queues, (creation, scattering) per launch, finished
    int   *sH   = (int *)(sI + (WINT ? BV : 0));           // arrivals per queue, next pass
    int   *sCtl = sH + (A.HS ? 2 * A.HS : NQ);             // [0] next packet, [1] tally events
    float *sL   = (float *)(sCtl + 2);                     // [3 * MAXLAUNCH] ABS, SCA, TW of every launch
    int   *sOFF = (int *)(sL + 3 * SOC_MAXLAUNCH);         // [SOC_MAXL] first cell of every level
    uint32_t *sPos = (uint32_t *)(sOFF + SOC_MAXL);        // [P] table entry << 12 | rank of every packet of the chunk
    const SocSim &S = K.S[0];                              // what the launches share: tallies, stats, OPT (n == 1)
    if ((int)threadIdx.x < K.n) {
        sL[3 * threadIdx.x] = K.S[threadIdx.x].ABS;  sL[3 * threadIdx.x + 1] = K.S[threadIdx.x].SCA;  sL[3 * threadIdx.x + 2] = K.S[threadIdx.x].TW;
    }
    if (threadIdx.x < SOC_MAXL) sOFF[threadIdx.x] = G.OFF[threadIdx.x];
    const int brick0 = (A.NBQ > A.NB) ? (D.brick % A.NB) : D.brick;         // the brick of this queue
    const int nslot = OCT ? (A.bbase[brick0 + 1] - A.bbase[brick0]) : BV;
    for (int i = threadIdx.x; i < nslot; i += nthr) { sT[i] = 0.0f; if (WINT) sI[i] = 0.0f; }
    soc_qh_init(sH, A.HS, NQ);
    if (threadIdx.x < 2) sCtl[threadIdx.x] = 0;
    __syncthreads();

    const int   NX = G.NX, NY = G.NY;
    const float fNX = (float)G.NX, fNY = (float)G.NY, fNZ = (float)G.NZ;
    const int   mybrick = brick0, LB = A.LB, M = (1 << A.LB) - 1;
    const int   qbase = D.brick - mybrick;                 // first brick queue of this workgroup's launch (0 when the launches share queues)
    float px = 0.0f, py = 0.0f, pz = 0.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f;
    float photons = 0.0f, free_path = 0.0f, tau = 0.0f, dens = 0.0f;
    float rux = 1.0f, ruy = 1.0f, ruz = 1.0f;              // correctly rounded reciprocals of the direction
    float kabs = 0.0f, ksca = 0.0f, tw = 0.0f;             // of the packet's launch
    int   lsh = 0;                                         // launch index << SOC_LCH_SHIFT
    int   ind = -1, level = 0, lid = 0, nvisit = 0, key = 0, slot = 0;
    int   ind0 = -1, level0 = 0;                           // cell at the start of the step (hierarchies)
    int   mode = SOC_BM_SWAP;
    bool  have = false;
    uint32_t wid = 0;
    unsigned int n_tally = 0;
    SOC_PROF_DECL;

    while (true) {
        {
            SOC_PROF(0, 1);  SOC_PROF(1, __popcll(__ballot(mode == SOC_BM_STEP)));  SOC_PROF(7, __popcll(__ballot(mode == SOC_BM_IDLE)));
            // the chunk has run out and most of the wave idles behind its longest walks: those continue in the next
            // pass, from this brick's queue, among a full wave again (between steps the packet state is complete)
            if (A.TAIL > 0 && __popcll(__ballot(mode == SOC_BM_IDLE)) >= A.TAIL && mode == SOC_BM_STEP) { mode = SOC_BM_SWAP;  key = D.brick; }
            const unsigned long long m = __ballot(mode == SOC_BM_SWAP);
            const bool nobody_steps = (__ballot(mode == SOC_BM_STEP) == 0ull);
            if (m != 0ull && (nobody_steps || __popcll(m) >= A.FTH)) {
                SOC_PROF(4, 1);  SOC_PROF(5, __popcll(m));
                if (mode == SOC_BM_SWAP) {
                    if (have) {
                        SocPk2 *q = pk + wid;
                        soc_st4(&q->A, make_float4(px, py, pz, photons));
                        soc_st4(&q->C, make_float4(tau, dens, __int_as_float(lid | (level << SOC_LVL_SHIFT) | lsh), __int_as_float(ind)));
                        if (!CL && key >= A.NBQ) SOC_NT_STORE((uint32_t)D.brick, &q->D.w);   // scattering: the brick to come back to (SimRAM_CL keeps its cell there)
                        SOC_NT_STORE((uint32_t)key, &A.keyq[D.start + slot]);
                        sPos[slot] = soc_qh_rank(sH, A.HS, key, A.hist);
                    }
                    slot = atomicAdd(&sCtl[0], 1);
                    have = slot < D.count;
                    if (!have) {
                        mode = SOC_BM_IDLE;
                    } else {
                        wid = SOC_NT_LOAD(&A.idq[D.start + slot]);
                        const SocPk2 *q = pk + wid;
                        const float4 a = soc_ld4(&q->A), b = soc_ld4(&q->B), c = soc_ld4(&q->C);
                        px = a.x;  py = a.y;  pz = a.z;  photons = a.w;
                        ux = b.x;  uy = b.y;  uz = b.z;  free_path = b.w;
                        rux = 1.0f / ux;  ruy = 1.0f / uy;  ruz = 1.0f / uz;
                        tau = c.x;  dens = c.y;  ind = __float_as_int(c.w);
                        const int cz = __float_as_int(c.z);
                        lid = cz & 0xffff;  level = (cz >> SOC_LVL_SHIFT) & 15;  lsh = cz & ~((1 << SOC_LCH_SHIFT) - 1);
                        { const int l3 = 3 * (lsh >> SOC_LCH_SHIFT);  kabs = sL[l3];  ksca = sL[l3 + 1];  tw = sL[l3 + 2]; }
                        nvisit = 0;
                        mode = SOC_BM_STEP;
                    }
                }
            }
        }
        SOC_PROF_T(0);                                     // swap
        if (__ballot(mode != SOC_BM_IDLE) == 0ull) break;
        if (OCT) {
            // ---- one cell step on the hierarchy (kernel_ASOC.c:565-683) ----
            // GetStep as on Cartesian grids (local coordinates: the arithmetic does not depend on the level).
            // Index (kernel_ASOC_aux.c:198-278) by case, each with the reference's results:
            //  * from a root cell: the root-grid lookup, then the descent through refined cells -- positions are
            //    floats >= 0, for which 2*fmod(p,1) is exact in float and in double alike;
            //  * to a sibling of the same octet (DBL, all coordinates in [sib_thr, 2)): the reference climbs to the
            //    parent (P = 0.5*p + octant, exact in double above sib_thr), finds the position inside the parent cell
            //    and descends again to 2*fmod(P,1) == p: same position, cell = octet base + octant of p, no reads;
            //  * anything else: soc_index() itself, for several lanes at a time (mode SOC_BM_CLIMB).
            bool finish = false;
            uint32_t sl = 0;
            if (mode == SOC_BM_STEP) {
                ind0 = ind;  level0 = level;
                const int   lid0 = lid;
                const float p0x = px, p0y = py, p0z = pz, d0 = dens;
                if (ABU) { float2 o = S.OPT[(lsh >> SOC_LCH_SHIFT) * A.opt_stride + sOFF[level] + ind];  kabs = o.x;  ksca = o.y; }
                float fx, fy, fz;
                if (__ballot(__builtin_fminf(px, __builtin_fminf(py, pz)) < 0.0f) == 0ull) {
                    fx = __builtin_amdgcn_fractf(px);  fy = __builtin_amdgcn_fractf(py);  fz = __builtin_amdgcn_fractf(pz);
                } else {
                    fx = soc_fmod1f(px);  fy = soc_fmod1f(py);  fz = soc_fmod1f(pz);
                }
                const float ax = soc_div_by_rcp(((ux > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fx, ux, rux);
                const float ay = soc_div_by_rcp(((uy > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fy, uy, ruy);
                const float az = soc_div_by_rcp(((uz > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fz, uz, ruz);
                float ds = __builtin_fminf(ax, __builtin_fminf(ay, az));
                px += ds * ux;
                py += ds * uy;
                pz += ds * uz;
                ds = soc_scale_down(ds, level);
                const float tauA = ds * d0 * kabs;
                const float dtau = ds * d0 * ksca;
                if (free_path < (tau + dtau)) {
                    px = p0x;  py = p0y;  pz = p0z;                               // back to the start of the step
                    mode = SOC_BM_SWAP;  key = A.NBQ + A.EQ * (lsh >> SOC_LCH_SHIFT) + 1;
                } else {
                    const float e = (__ballot(!(tauA < 0.34f)) == 0ull) ? soc_expf_small(-tauA) : soc_expf(-tauA);
                    const float delta = (tauA > SOC_TAULIM) ? (photons * (1.0f - e)) : (photons * tauA * (1.0f - 0.5f * tauA));
                    atomicAdd(&sT[lid0], delta * tw);
                    if (WINT) atomicAdd(&sI[lid0], delta);
                    n_tally++;
                    photons *= e;
                    tau += dtau;
                    int cell = -1;                                               // global index of the cell to look at next
                    if (level == 0) {
                        const bool inside = (px > 0.0f) & (px < fNX) & (py > 0.0f) & (py < fNY) & (pz > 0.0f) & (pz < fNZ);
                        ind = inside ? ((int)pz * NX * NY + (int)py * NX + (int)px) : -1;
                        cell = ind;
                        finish = true;
                    } else if (DBL && (__builtin_fminf(px, __builtin_fminf(py, pz)) >= A.sib_thr)
                                   && (__builtin_fmaxf(px, __builtin_fmaxf(py, pz)) < 2.0f)) {
                        ind = (ind & ~7) + 4 * (int)pz + 2 * (int)py + (int)px;
                        cell = sOFF[level] + ind;
                        finish = true;
                    } else {
                        mode = SOC_BM_CLIMB;
                    }
                    if (cell >= 0) {
                        float2 rec = A.DS[cell];
                        while (!(rec.x > 0.0f)) {                                 // descend to the leaf
                            SOC_PROF(6, 1);                                      /* per lane */
                            px = 2.0f * __builtin_amdgcn_fractf(px);
                            py = 2.0f * __builtin_amdgcn_fractf(py);
                            pz = 2.0f * __builtin_amdgcn_fractf(pz);
                            level++;
                            ind = soc_link_index(rec.x) + 4 * (int)pz + 2 * (int)py + (int)px;
                            rec = A.DS[sOFF[level] + ind];
                        }
                        dens = rec.x;
                        sl = __float_as_uint(rec.y);
                    }
                }
            }
            SOC_PROF_T(1);                                 // step
            {
                const unsigned long long mc = __ballot(mode == SOC_BM_CLIMB);
                if (mc != 0ull && (__popcll(mc) >= A.CTH || __ballot(mode == SOC_BM_STEP) == 0ull)) {
                    SOC_PROF(2, 1);  SOC_PROF(3, __popcll(mc));
                    if (mode == SOC_BM_CLIMB) {
                        if (DBL) soc_index<true, double>(G, sOFF, px, py, pz, level, ind, dens);
                        else     soc_index<true, float>(G, sOFF, px, py, pz, level, ind, dens);
                        if (ind >= 0) sl = __float_as_uint(A.DS[sOFF[level] + ind].y);
                        mode = SOC_BM_STEP;
                        finish = true;
                    }
                }
            }
            SOC_PROF_T(2);                                 // climb
            if (finish) {
                if (!CL && (level == level0) && (ind == ind0)) {                  // failed step: nudge (SimRAM_PB / HP only)
                    px += SOC_PEPS * ux;  py += SOC_PEPS * uy;  pz += SOC_PEPS * uz;
                }
                nvisit++;
                if (ind < 0) {
                    mode = SOC_BM_SWAP;  key = A.NBQ + A.EQ * (lsh >> SOC_LCH_SHIFT);                          // -> creation queue
                } else {
                    const int nb = (int)(sl >> SOC_SLOT_BITS);
                    lid = (int)(sl & SOC_SLOT_MASK);
                    if (nb != mybrick || nvisit >= A.KCAP) { mode = SOC_BM_SWAP;  key = qbase + nb; }
                }
            }
        } else
        // ---- one cell step (kernel_ASOC.c:565-683, LEVELS == 1) ----
        if (mode == SOC_BM_STEP) {
            const int   oind = ind, lid0 = lid;
            const float p0x = px, p0y = py, p0z = pz, d0 = dens;
            if (ABU) { float2 o = S.OPT[(lsh >> SOC_LCH_SHIFT) * A.opt_stride + oind];  kabs = o.x;  ksca = o.y; }
            // GetStep (kernel_ASOC_aux.c:282-315) with the same results from fewer instructions:
            //  * fmod(p,1) of a positive p is v_fract (exact); a lane with a negative coordinate (possible
            //    only right after a failed-step nudge) sends the wave through the general form;
            //  * n/u from the cached correctly rounded 1/u (soc_div_by_rcp: bit-identical to the division);
            //  * floor of a positive coordinate is the float->int conversion.
            float fx, fy, fz;
            if (__ballot(__builtin_fminf(px, __builtin_fminf(py, pz)) < 0.0f) == 0ull) {
                fx = __builtin_amdgcn_fractf(px);  fy = __builtin_amdgcn_fractf(py);  fz = __builtin_amdgcn_fractf(pz);
            } else {
                fx = soc_fmod1f(px);  fy = soc_fmod1f(py);  fz = soc_fmod1f(pz);
            }
            const float ax = soc_div_by_rcp(((ux > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fx, ux, rux);
            const float ay = soc_div_by_rcp(((uy > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fy, uy, ruy);
            const float az = soc_div_by_rcp(((uz > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fz, uz, ruz);
            float ds = __builtin_fminf(ax, __builtin_fminf(ay, az));
            px += ds * ux;
            py += ds * uy;
            pz += ds * uz;
            ds = soc_scale_down(ds, 0);
            const bool inside = (px > 0.0f) & (px < fNX) & (py > 0.0f) & (py < fNY) & (pz > 0.0f) & (pz < fNZ);
            const int ix = inside ? (int)px : 0;
            const int iy = inside ? (int)py : 0;
            const int iz = inside ? (int)pz : 0;
            const int nind = iz * NX * NY + iy * NX + ix;
            const int nb   = ((iz >> LB) * A.NBY + (iy >> LB)) * A.NBX + (ix >> LB);
            const int nlid = ((iz & M) << (2 * LB)) | ((iy & M) << LB) | (ix & M);
            const bool stay = inside & (nb == mybrick);
            const float ndens = G.DENS[nind];
            const float tauA = ds * d0 * kabs;
            const float dtau = ds * d0 * ksca;
            if (free_path < (tau + dtau)) {
                px = p0x;  py = p0y;  pz = p0z;                               // back to the start of the step
                mode = SOC_BM_SWAP;  key = A.NBQ + A.EQ * (lsh >> SOC_LCH_SHIFT) + 1;   // -> scattering queue of its launch
            } else {
                // every lane of the wave in the interval where soc_expf_small == soc_expf (the common case)
                const float e = (__ballot(!(tauA < 0.34f)) == 0ull) ? soc_expf_small(-tauA) : soc_expf(-tauA);
                const float delta = (tauA > SOC_TAULIM) ? (photons * (1.0f - e)) : (photons * tauA * (1.0f - 0.5f * tauA));
                atomicAdd(&sT[lid0], delta * tw);
                if (WINT) atomicAdd(&sI[lid0], delta);
                n_tally++;
                photons *= e;
                tau += dtau;
                ind = inside ? nind : -1;
                dens = ndens;
                lid = nlid;
                const bool failed = !CL && (ind == oind);                     // failed step: nudge (SimRAM_PB / HP only)
                px += failed ? (SOC_PEPS * ux) : 0.0f;
                py += failed ? (SOC_PEPS * uy) : 0.0f;
                pz += failed ? (SOC_PEPS * uz) : 0.0f;
                nvisit++;
                if (!inside)                          { mode = SOC_BM_SWAP;  key = A.NBQ + A.EQ * (lsh >> SOC_LCH_SHIFT); }   // -> creation queue
                else if (!stay || nvisit >= A.KCAP)   { mode = SOC_BM_SWAP;  key = qbase + nb; }
            }
        }
    }

    SOC_PROF_FLUSH;
    atomicAdd(&sCtl[1], (int)n_tally);
    __syncthreads();
    if (OCT) {
        const int *cells = A.bcell + A.bbase[mybrick];                        // consecutive slots: row neighbours, octet siblings
        for (int i = threadIdx.x; i < nslot; i += nthr) {
            const float v = sT[i];
            const float vi = WINT ? sI[i] : 0.0f;
            if (v != 0.0f || vi != 0.0f) {
                const int cell = cells[i];
                soc_tally(S.TABS, cell, v);
                if (WINT) soc_tally(K.S[K.gfirst[qbase / A.NB]].INT, cell, vi);      // the launches this workgroup's queue belongs to
            }
        }
    } else {
        const int B = 1 << A.LB;
        const int bx = mybrick % A.NBX, by = (mybrick / A.NBX) % A.NBY, bz = mybrick / (A.NBX * A.NBY);
        for (int i = threadIdx.x; i < BV; i += nthr) {
            const float v = sT[i];
            const float vi = WINT ? sI[i] : 0.0f;
            if (v != 0.0f || vi != 0.0f) {
                const int ix = bx * B + (i & M), iy = by * B + ((i >> A.LB) & M), iz = bz * B + (i >> (2 * A.LB));
                const int cell = iz * NX * NY + iy * NX + ix;
                soc_tally(S.TABS, cell, v);
                if (WINT) soc_tally(K.S[K.gfirst[qbase / A.NB]].INT, cell, vi);      // the launches this workgroup's queue belongs to
            }
        }
    }
    soc_qh_bases(sH, A.HS, NQ, A.hist);
    __syncthreads();
    for (int j = threadIdx.x; j < D.count; j += nthr) A.posq[D.start + j] = soc_qh_place(sH, A.HS, sPos[j]);
    if (threadIdx.x == 0 && S.stats) atomicAdd(S.stats + 0, (unsigned long long)(unsigned int)sCtl[1]);
}

// ---------------------------------------------------------------------------------------
// The walk on brick-local hierarchies (soc_ltree.h): hierarchies whose Index() the reference evaluates in double.
// The workgroup copies its brick's cells into LDS (sD: density, or the link to the octet's slots), so a cell step
// reads and writes LDS only: GetStep's arithmetic, the tally (ds_add_f32), then the new cell from the packet's
// integer cell coordinates -- a sibling by slot arithmetic, anything else by a descent from the brick's root cells
// (soc_lt_aim / soc_lt_land) -- and the new local position from one fma.  Global memory is touched when a packet is taken from
// the queue or put back, and for the brick of the root cell a leaving packet goes to (rbrick).
// Packet record here: A = position, photons | B = direction, free path | C = tau, cell coordinates (cx, cy, cz on the
// cell's level) | D = RNG state, III | scatterings << 24 | level << 29, SimRAM_CL: emitting cell; bit 31: the step
// into the next brick is not finished (old cell + advanced position: the Index() part completes it there).
// ---------------------------------------------------------------------------------------
#define SOC_LT_ARRIVE 0x80000000u
#define SOC_LQ_SHIFT 24           // packet word C.w of this form: cz | launch << 24 (coordinates stay below 2^24, see soc_brick_run_pb)

// integer coordinates of cell (level, ind) on its level: octants on the way up through PAR, then the root cell
__device__ __forceinline__ void soc_cell_coords(const SocGrid &G, const int *sOFF, int level, int ind, int &cx, int &cy, int &cz)
{
    int x = 0, y = 0, z = 0;
    for (int j = 0; j < level; j++) {
        x |= (ind & 1) << j;  y |= ((ind >> 1) & 1) << j;  z |= ((ind >> 2) & 1) << j;
        ind = G.PAR[sOFF[level - j] + ind - G.NXYZ];
    }
    cx = x | ((ind % G.NX) << level);
    cy = y | (((ind / G.NX) % G.NY) << level);
    cz = z | ((ind / (G.NX * G.NY)) << level);
}

// the inverse: index within its level of the cell with coordinates (cx, cy, cz) on `level` (descent through DENS)
__device__ __forceinline__ int soc_cell_index(const SocGrid &G, int level, int cx, int cy, int cz, float &dens)
{
    int ind = ((cz >> level) * G.NY + (cy >> level)) * G.NX + (cx >> level);
    dens = G.DENS[ind];
    for (int l = 1; l <= level; l++) {
        const int sh = level - l;
        ind = soc_link_index(dens) + (((cx >> sh) & 1) | (((cy >> sh) & 1) << 1) | (((cz >> sh) & 1) << 2));
        dens = G.DENS[G.OFF[l] + ind];
    }
    return ind;
}

// RAY: the read-only rays of the scattered-light kernels (soc_sca_events below): no tallies (the LDS holds the cells only), the optical
// depth grows with the factor the record carries in place of the photons (kappa_sca for the look-ahead and the packet, kappa_abs + kappa_sca
// for a peel-off ray: kernel_ASOC_sca.c:895-897, :975-990, :1035-1040), no nudge after a failed step (GetStep alone moves these rays).
// WINT: 0 TABS only, 1 the INT tally beside it, 2 INT and the vector sums INTX, INTY, INTZ (-D SAVE_INTENSITY=2, kernel_ASOC.c:604-612),
// 3 the INT tally ALONE in LDS: the launches of the workgroup's queue share one weight TW (the source blocks of one frequency), so
//   TABS = TW * INT is formed when the tallies go to global memory -- one LDS float atomic per step instead of two (they are what the LDS
//   unit spends its time on), 8 B per cell instead of 12.
// ALI: -D WITH_ALI (kernel_ASOC.c:1394-1396, :1486-1494; SimRAM_CL only): what a packet deposits in the cell that emitted it goes to the XAB
// tally instead of TABS.  The LDS then holds the cell numbers of the brick's slots (sC) and an XAB tally (sX) as well: 16 B per cell.
template <int WINT, bool RAY = false, bool ALI = false>
__device__ __forceinline__ void soc_lbrick_walk(const SocGrid &G, const SocSimPack &K, const SocBrickArgs &A, const int bid)
{
    if (bid >= *A.ndesc) return;
    SocDesc D = A.desc[bid];
    D.brick = __builtin_amdgcn_readfirstlane(D.brick);  D.start = __builtin_amdgcn_readfirstlane(D.start);  D.count = __builtin_amdgcn_readfirstlane(D.count);
    if (D.brick >= A.NBQ) return;                          // an event queue: soc_brick_events
    const bool parked = __builtin_amdgcn_readfirstlane(D.pad) != 0;      // too few packets for a workgroup: they stay in the queue this pass (soc_brick_scan)
    SOC_PROF_ENTRY;
    const int BV = A.CAP;                                  // slots in LDS
    const int nthr = (int)blockDim.x;
    SOC_GLOBAL SocPk2 *pk = (SOC_GLOBAL SocPk2 *)A.pk;

    extern __shared__ float lds[];
    float *sT   = lds;                                     // [BV] TABS of this brick
    float *sI   = sT + ((RAY || WINT == 3) ? 0 : BV);      // [BV] INT (WINT)
    float *sV   = sI + (WINT ? BV : 0);                    // [3 BV] INTX | INTY | INTZ (WINT == 2)
    float *sX   = sV + ((WINT == 2) ? 3 * BV : 0);         // [BV] XAB (ALI)
    int   *sC   = (int *)(sX + (ALI ? BV : 0));            // [BV] global number of the cell in every slot (ALI)
    float *sD   = (float *)(sC + (ALI ? BV : 0));          // [BV] density | link of every cell of the brick (RAY: nothing else)
    const int NQ = A.NBQ + A.EQ * A.nl + 1;
    int   *sH   = (int *)(sD + BV);                        // arrivals per queue, next pass
    int   *sCtl = sH + (A.HS ? 2 * A.HS : ((NQ + 3) & ~3));   // [0] next packet, [1] tally events
    float *sL   = (float *)(sCtl + 4);                     // [4 * MAXLAUNCH] ABS, SCA, TW, flags (bit 0: SimRAM_CL) of every launch
    const SocSim &S = K.S[0];
    for (int l = threadIdx.x; l < K.n; l += nthr) {
        sL[4 * l] = K.S[l].ABS;  sL[4 * l + 1] = K.S[l].SCA;  sL[4 * l + 2] = K.S[l].TW;
        sL[4 * l + 3] = __int_as_float((K.S[l].SOURCE == SOC_SOURCE_CL) ? 1 : 0);
    }
    const int mybrick = (A.NBQ > A.NB) ? (D.brick % A.NB) : D.brick;
    const int qbase = D.brick - mybrick;                   // first brick queue of this workgroup's launch (0 when the launches share queues)
    SocLBrick KB = A.lbr[mybrick];
    // workgroup-uniform values that came from global memory go to scalar registers now: a use inside the loop would
    // otherwise wait for every load in flight (s_waitcnt vmcnt(0)), the prefetched packets included
    KB.x0 = __builtin_amdgcn_readfirstlane(KB.x0);  KB.y0 = __builtin_amdgcn_readfirstlane(KB.y0);  KB.z0 = __builtin_amdgcn_readfirstlane(KB.z0);
    KB.bx = __builtin_amdgcn_readfirstlane(KB.bx);  KB.by = __builtin_amdgcn_readfirstlane(KB.by);  KB.bz = __builtin_amdgcn_readfirstlane(KB.bz);
    KB.base = __builtin_amdgcn_readfirstlane(KB.base);  KB.nslot = __builtin_amdgcn_readfirstlane(KB.nslot);
    if (!parked) {
        const float *src = A.btree + KB.base;
        for (int i = threadIdx.x; i < KB.nslot; i += nthr) {
            sD[i] = src[i];  if (!RAY && WINT != 3) sT[i] = 0.0f;  if (WINT) sI[i] = 0.0f;
            if (WINT == 2) { sV[i] = 0.0f;  sV[BV + i] = 0.0f;  sV[2 * BV + i] = 0.0f; }
            if (ALI) { sX[i] = 0.0f;  sC[i] = A.bcell[KB.base + i]; }
        }
    }
    soc_qh_init(sH, A.HS, NQ);
    if (threadIdx.x < 2) sCtl[threadIdx.x] = 0;
    if (parked) for (int j = threadIdx.x; j < D.count; j += nthr) A.keyq[D.start + j] = (uint32_t)D.brick;
    __syncthreads();

    // Uniform values the loop needs now and then.  Left as plain kernel arguments the compiler, out of scalar registers,
    // re-reads them from the argument segment where they are used -- an s_load and a wait for it (and for every LDS
    // operation in flight) inside the loop.  Made opaque here they are values it has to keep: in scalar registers or in
    // lanes of a spill VGPR (v_readlane, no memory).
    int NX = G.NX, NY = G.NY, NZ = G.NZ, Lmax = G.LEVELS - 1, kexp = A.kexp;
    asm volatile("" : "+s"(NX), "+s"(NY), "+s"(NZ), "+s"(Lmax), "+s"(kexp));
    // the region of interest (root cells, inclusive limits): InRoi of kernel_ASOC_aux.c:1031-1048 on the packet's integer coordinates
    int rx0 = 1, rx1 = 0, ry0 = 1, ry1 = 0, rz0 = 1, rz1 = 0;
    if (!RAY && A.roi_on) {
        const SocRoi *R = S.ROI;
        rx0 = __builtin_amdgcn_readfirstlane(R->ROI[0]);  rx1 = __builtin_amdgcn_readfirstlane(R->ROI[1]);  ry0 = __builtin_amdgcn_readfirstlane(R->ROI[2]);
        ry1 = __builtin_amdgcn_readfirstlane(R->ROI[3]);  rz0 = __builtin_amdgcn_readfirstlane(R->ROI[4]);  rz1 = __builtin_amdgcn_readfirstlane(R->ROI[5]);
    }
    SOC_GLOBAL uint32_t *keyq_c = (SOC_GLOBAL uint32_t *)(A.keyq + D.start);          // this chunk's part of the queue arrays
    const SOC_GLOBAL uint32_t *idq_c = (const SOC_GLOBAL uint32_t *)(A.idq + D.start);
    asm volatile("" : "+s"(pk), "+s"(keyq_c), "+s"(idq_c));
    float px = 0.0f, py = 0.0f, pz = 0.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f;
    float photons = 0.0f, free_path = 0.0f, tau = 0.0f, dens = 0.0f;
    float rux = 1.0f, ruy = 1.0f, ruz = 1.0f;              // correctly rounded reciprocals of the direction
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;                 // GetStep's target inside the cell per axis: 1 + PEPS or -PEPS
    float kabs = 0.0f, ksca = 0.0f, tw = 0.0f;
    int   cx = 0, cy = 0, cz = 0, level = 0, slot = 0, obase = 0, nvisit = 0, key = 0, cslot = 0, lq = 0, evq = 0;      // evq: first event queue of the packet's launch      // obase: slot of the first cell of the packet's octet
    int   what = SOC_LTM_STEP;                              // what the Index() part has to do for the lane: finish a step, an arrival, or find the packet's cell
    bool  nonudge = false;                                 // SimRAM_CL: no nudge after a failed step (kernel_ASOC.c:1530-1540 has none)
    uint32_t dz = 0, dw = 0, wid = 0;
    int   mode = SOC_BM_SWAP;
    bool  have = false, nhave = false;                                       // the packet in hand, the prefetched one
    soc_f4v na = { 0.0f, 0.0f, 0.0f, 0.0f }, nb = na, nc = na;
    soc_u2v ndzw = { 0u, 0u };
    uint32_t nwid = 0, nnwid = 0;
    int   nslot = 0, nnslot = 0;
    bool  nnhave = false;                                                    // the packet after the prefetched one: its id is on the way
    unsigned int n_tally = 0;
    SOC_PROF_DECL;
    SOC_PROF_SINCE_ENTRY(3);                               // the workgroup's prologue (brick -> LDS, tables), per wave

    // The loop has two arms.  SWAP (entered when A.FTH lanes of the wave wait for it, or nobody can step): the lane's packet
    // goes back to memory with the queue it belongs to next, the prefetched one is taken up.  STEP (every iteration): GetStep's
    // arithmetic and the tally -- skipped by a lane whose packet has just come in from another brick or still needs its
    // cell looked up -- and Index() in the one-path form of soc_ltree.h (soc_lt_aim / soc_lt_land), which serves all of them.
    // Reads of LDS are issued ahead of the arithmetic that does not need them (the slot Index() starts from before the
    // exponential of the tally; the three reads of the swap arm together, before the stores), so a wave waits for LDS once
    // per arm, not once per read.
    while (!parked) {
        {
            SOC_PROF(0, 1);  SOC_PROF(1, __popcll(__ballot(mode == SOC_BM_STEP)));  SOC_PROF(7, __popcll(__ballot(mode == SOC_BM_IDLE)));
            if (D.count < 4 * nthr)  { SOC_DPROF(0, 1);  SOC_DPROF(1, __popcll(__ballot(mode == SOC_BM_IDLE))); }
            if (D.count < 16 * nthr) { SOC_DPROF(2, 1);  SOC_DPROF(3, __popcll(__ballot(mode == SOC_BM_IDLE))); }
            if (__ballot(nhave | nnhave) == 0ull) { SOC_DPROF(4, 1);  SOC_DPROF(5, __popcll(__ballot(mode == SOC_BM_IDLE))); }
            // (tuning tail_lanes: a long chunk is used up and that many lanes of the wave are out of work -- the others send their packets back
            //  to this brick's queue instead of finishing their visits before mostly idle lanes.  Only packets that have made a step in this
            //  visit, and only in chunks of at least 8 packets per lane: every pass makes progress, short queues run to their end.)
            if (A.TAIL > 0 && D.count >= 8 * nthr && __popcll(__ballot(mode == SOC_BM_IDLE)) >= A.TAIL
                && mode == SOC_BM_STEP && what == SOC_LTM_STEP && nvisit > 0) { mode = SOC_BM_SWAP;  key = D.brick; }
            const unsigned long long m = __ballot(mode == SOC_BM_SWAP);
            const bool nobody_steps = (__ballot(mode == SOC_BM_STEP) == 0ull);
            if (m != 0ull && (nobody_steps || __popcll(m) >= A.FTH)) {
                SOC_PROF(4, 1);  SOC_PROF(5, __popcll(m));
                if (mode == SOC_BM_SWAP) {
                    // A lane holds three packets: the one it walks, the next one, whose record was asked for when the
                    // current one was taken up, and the one after that, of which the id has been asked for -- each a
                    // visit ahead of its use, so nothing in this arm waits for global memory.  The chunk itself is never
                    // copied: ids, queues and places stay in global memory (idq, keyq, posq), so a chunk may be the whole
                    // queue of the brick.
                    // (1) the lane reserves the packet after next: one LDS atomic per wave, neighbouring lanes read neighbouring ids
                    const unsigned long long am = __ballot(true);
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                    int base = 0;
                    if (rank == 0) base = atomicAdd(&sCtl[0], __popcll(am));
                    asm volatile("" :: "v"(na), "v"(nb), "v"(nc), "v"(ndzw), "v"(nnwid));      // what is in flight has landed: no later use waits behind the stores below
                    // (2) what the prefetched packet needs from LDS -- its launch's constants and, for a root-level packet, the root cell
                    // it is in (or comes into: ARRIVE) -- is asked for now; addresses from a stale record (no packet) stay inside the arrays
                    const int  n_cz = __float_as_int(nc.w);
                    const int  n_lq = (int)(((uint32_t)n_cz >> SOC_LQ_SHIFT) & (SOC_MAXLAUNCH - 1));
                    const bool n_arrive = (ndzw.y & SOC_LT_ARRIVE) != 0u;
                    const int  n_ix = n_arrive ? (int)soc_floorf(na.x) : __float_as_int(nc.y), n_iy = n_arrive ? (int)soc_floorf(na.y) : __float_as_int(nc.z),
                               n_iz = n_arrive ? (int)soc_floorf(na.z) : (n_cz & ((1 << SOC_LQ_SHIFT) - 1));
                    int n_s2 = SOC_MAD24(SOC_MAD24(n_iz - KB.z0, KB.by, n_iy - KB.y0), KB.bx, n_ix - KB.x0);
                    n_s2 = ((unsigned)n_s2 < (unsigned)KB.nslot) ? n_s2 : 0;
                    const soc_f4v l4 = *(const soc_f4v *)&sL[4 * n_lq];
                    const float n_rec = sD[n_s2];
                    // (3) the packet in hand goes back to memory
                    if (have) {
                        SOC_GLOBAL SocPk2 *q = pk + wid;
                        // (the brick of the root cell a leaving packet goes to -- key < 0: -1 - root cell -- is looked up after
                        // the walk, for all packets of the chunk at once: loaded when the packet leaves it made EVERY iteration
                        // wait for all loads in flight, 19 % of the wave's cycles; loaded here, every entry of this arm one L2 latency)
                        { const soc_f4v v = { px, py, pz, photons };  SOC_NT_STORE(v, (SOC_GLOBAL soc_f4v *)&q->A); }
                        { const soc_f4v v = { tau, __int_as_float(cx), __int_as_float(cy), __int_as_float(cz | (lq << SOC_LQ_SHIFT)) };  SOC_NT_STORE(v, (SOC_GLOBAL soc_f4v *)&q->C); }
                        q->D.z = (dz & 0x1fffffffu) | ((uint32_t)level << 29);
                        q->D.w = dw;
                        SOC_NT_STORE((uint32_t)key, &keyq_c[cslot]);              // its rank in that queue is settled after the walk, for all packets at once
                    }
                    // (4) the prefetched packet becomes the current one
                    have = nhave;
                    wid = nwid;  cslot = nslot;
                    {   // (all lanes of the arm: what a lane without a prefetched packet decodes here is never used)
                        dz = ndzw.x;  dw = ndzw.y & ~SOC_LT_ARRIVE;
                        px = na.x;  py = na.y;  pz = na.z;  photons = na.w;
                        ux = nb.x;  uy = nb.y;  uz = nb.z;  free_path = nb.w;
                        rux = 1.0f / ux;  ruy = 1.0f / uy;  ruz = 1.0f / uz;
                        gx = (ux > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS;  gy = (uy > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS;  gz = (uz > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS;
                        tau = nc.x;  cx = __float_as_int(nc.y);  cy = __float_as_int(nc.z);  cz = n_cz & ((1 << SOC_LQ_SHIFT) - 1);
                        lq = n_lq;                                                // the launch of the work item
                        evq = A.NBQ + A.EQ * n_lq;
                        level = (int)(dz >> 29);
                        kabs = l4.x;  ksca = l4.y;  tw = l4.z;  nonudge = RAY || ((__float_as_int(l4.w) & 1) != 0);
                        nvisit = 0;
                        mode = have ? SOC_BM_STEP : mode;
                        slot = -1;  obase = -1;
                        // The first pass of the packet through the Index() part finds its cell in this brick: the second half of
                        // the step that brought it here (ARRIVE), or the cell its coordinates name.  A root-level packet in (or
                        // into) a root cell that is not refined -- the common case -- is settled here.
                        // (with the record of packets entering ROI an arrival is always finished in the step arm, where the test for it sits)
                        const bool settled = (level == 0) && (!n_arrive || ((n_rec > 0.0f) && !(!RAY && A.roi_on)));
                        what = settled ? SOC_LTM_STEP : (n_arrive ? SOC_LTM_ARRIVE : SOC_LTM_PLACE);
                        slot = settled ? n_s2 : slot;  dens = settled ? n_rec : dens;
                        cx = settled ? n_ix : cx;  cy = settled ? n_iy : cy;  cz = settled ? n_iz : cz;
                    }
                    // (5) the record of the packet after it (its id has arrived), and the id of the one after that
                    nhave = nnhave;
                    nwid = nnwid;  nslot = nnslot;
                    asm volatile("" : "+v"(nwid) :: "memory");                // (the copy is made HERE: nnwid is dead below and the load at the end of the arm lands
                                                                              //  in its register; left to itself the compiler loads into a fresh register, waits for
                                                                              //  EVERY load in flight -- the records just asked for -- and copies: 14 % of the cycles)
                    if (nhave) {
                        // (vector-typed loop variables: the loads land in the registers that carry the values around the loop)
                        const SOC_GLOBAL SocPk2 *q = pk + nwid;
                        na = SOC_NT_LOAD((const SOC_GLOBAL soc_f4v *)&q->A);  nb = SOC_NT_LOAD((const SOC_GLOBAL soc_f4v *)&q->B);  nc = SOC_NT_LOAD((const SOC_GLOBAL soc_f4v *)&q->C);
                        ndzw = *(const SOC_GLOBAL soc_u2v *)&q->D.z;
                    }
                    nnslot = __builtin_amdgcn_readfirstlane(base) + rank;
                    nnhave = nnslot < D.count;
                    if (nnhave) nnwid = SOC_NT_LOAD(&idq_c[nnslot]);
                    if (!have) mode = (nhave | nnhave) ? SOC_BM_SWAP : SOC_BM_IDLE;      // (the first two turns of a lane only reserve)
                }
            }
        }
        SOC_PROF_T(0);                                     // swap
        if (__ballot(mode != SOC_BM_IDLE) == 0ull) break;
        if (mode == SOC_BM_STEP) {
            bool move = true;
            const int slot0 = slot;
            float tauA = 0.0f, dtau = 0.0f;
            // ---- one cell step (kernel_ASOC.c:565-683): GetStep ----
            if (what == SOC_LTM_STEP) {
                const float p0x = px, p0y = py, p0z = pz;
                float fx, fy, fz;
                if (__ballot(__builtin_fminf(px, __builtin_fminf(py, pz)) < 0.0f) == 0ull) {
                    fx = __builtin_amdgcn_fractf(px);  fy = __builtin_amdgcn_fractf(py);  fz = __builtin_amdgcn_fractf(pz);
                } else {
                    fx = soc_fmod1f(px);  fy = soc_fmod1f(py);  fz = soc_fmod1f(pz);
                }
                const float ax = soc_div_by_rcp(gx - fx, ux, rux);
                const float ay = soc_div_by_rcp(gy - fy, uy, ruy);
                const float az = soc_div_by_rcp(gz - fz, uz, ruz);
                float ds = __builtin_fminf(ax, __builtin_fminf(ay, az));
                px += ds * ux;
                py += ds * uy;
                pz += ds * uz;
                ds = ds * soc_lt_pow2(-level);                                    // ldexp(ds, -level)
                tauA = ds * dens * kabs;
                dtau = ds * dens * (RAY ? photons : ksca);
                const bool scat = free_path < (tau + dtau);                       // the free path ends in this cell:
                px = scat ? p0x : px;  py = scat ? p0y : py;  pz = scat ? p0z : pz;   // back to the start of the step
                mode = scat ? SOC_BM_SWAP : mode;  key = scat ? (evq + 1) : key;  // -> scattering queue of its launch
                move = !scat;
            }
            // ---- Index (kernel_ASOC_aux.c:198-278), first half: where the point is, and the read of the slot the descent starts from;
            // the tally of the step while that read is in flight; the descent to the leaf; the outcome.  ONE divergent region, and the
            // outcome by selects: every `if` of a divergent wave costs scalar bookkeeping of the exec mask that the wave issues in order
            // with its vector instructions (the walk is bound by instruction issue) ----
            if (move) {
                SocLtAim AM;
                int   r;
                float rec = 1.0f;
                if ((A.slow_every > 0) && (what == SOC_LTM_STEP) && (level > 0) && (((n_tally + 1u + wid) % (unsigned)A.slow_every) == 0u)) {
                    r = SOC_LT_SLOW;                                              // test knob: this step goes through the slow-step queue
                    AM = SocLtAim{};
                } else {
                    soc_lt_aim(KB, NX, NY, NZ, Lmax, kexp, what, px, py, pz, level, cx, cy, cz, obase, AM);
                    r = AM.r;
                    rec = sD[AM.s];
                }
                // the tally of the step.  A lane that only places its packet (no GetStep: tauA = dtau = 0) adds zeros to slot 0
                // and multiplies its photons by exp(-0) = 1: cheaper than a branch of its own
                {
                    const bool stepped = (what == SOC_LTM_STEP);
                    n_tally += stepped ? 1u : 0u;
                    tau += dtau;
                    if (!RAY) {
                        const float e = (__ballot(!(tauA < 0.34f)) == 0ull) ? soc_expf_small(-tauA) : soc_expf(-tauA);
                        const float delta = (tauA > SOC_TAULIM) ? (photons * (1.0f - e)) : (photons * tauA * (1.0f - 0.5f * tauA));
                        const int st = stepped ? slot0 : 0;
                        if (ALI) atomicAdd((sC[st] == (int)dw) ? &sX[st] : &sT[st], delta * tw);      // (SimRAM_CL: the record's last word is the emitting cell)
                        else if (WINT != 3) atomicAdd(&sT[st], delta * tw);
                        if (WINT) atomicAdd(&sI[st], delta);
                        if (WINT == 2) { atomicAdd(&sV[st], delta * ux);  atomicAdd(&sV[BV + st], delta * uy);  atomicAdd(&sV[2 * BV + st], delta * uz); }
                        photons *= e;
                    }
                }
                SOC_PROF_T(1);                                 // GetStep + tally
                const int orx = cx >> level, ory = cy >> level, orz = cz >> level;      // (the root cell the step starts from: for the record of packets entering ROI)
                if (r == SOC_LT_INSIDE) r = soc_lt_land(sD, AM, Lmax, what, rec, px, py, pz, level, cx, cy, cz, slot, obase, dens);
                const bool inside = (r == SOC_LT_INSIDE);
                const bool moved  = inside & (what != SOC_LTM_PLACE);
                // failed step: nudge (SimRAM_PB / HP only); + 0 * u leaves the position as it is
                const float nz = (moved & !nonudge & (slot == slot0)) ? SOC_PEPS : 0.0f;
                px += nz * ux;  py += nz * uy;  pz += nz * uz;
                nvisit += moved ? 1 : 0;
                const bool leave = (r == SOC_LT_LEAVE), slowq = (r == SOC_LT_SLOW);
                // where the packet goes when it does not stay: its own queue again (step budget used), the brick of the root cell it
                // steps into (looked up after the walk: -1 - root cell; N < 4096), the creation queue of its launch (outside the model),
                // the slow-step queue (old cell, advanced position), or -- cannot happen, the sender looked the brick up -- retirement
                const int kleave = -1 - SOC_MAD24(SOC_MAD24(AM.Rz, NY, AM.Ry), NX, AM.Rx);
                const int kout = leave ? kleave : ((r == SOC_LT_EXIT) ? evq : (slowq ? (evq + 2) : (NQ - 1)));
                const bool out = !inside | (moved & (nvisit >= A.KCAP));
                key  = out ? (inside ? D.brick : kout) : key;
                mode = out ? SOC_BM_SWAP : mode;
                if (!RAY && A.roi_on) {
                    // WITH_ROI_SAVE (kernel_ASOC.c:615-642, :1510-1535), at the end of a full step: the packet was outside ROI and is inside
                    // now -> the event workgroups add it to the record (root position, direction, photons) and send it back here
                    const int nrx = cx >> level, nry = cy >> level, nrz = cz >> level;
                    const bool was = (orx >= rx0) & (orx <= rx1) & (ory >= ry0) & (ory <= ry1) & (orz >= rz0) & (orz <= rz1);
                    const bool is  = (nrx >= rx0) & (nrx <= rx1) & (nry >= ry0) & (nry <= ry1) & (nrz >= rz0) & (nrz <= rz1);
                    const bool entered = moved & is & !was;
                    key  = entered ? (evq + 3) : key;
                    mode = entered ? SOC_BM_SWAP : mode;
                }
                dw  |= (leave | slowq) ? SOC_LT_ARRIVE : 0u;
                what = SOC_LTM_STEP;
            }
            SOC_PROF_T(2);                                 // Index
        }
    }

    SOC_PROF_T(5);                                         // (what the last iteration left: next to nothing)
    atomicAdd(&sCtl[1], (int)n_tally);
    __syncthreads();
    SOC_PROF_T(6);                                         // the wait for the workgroup's other waves
    if (!parked && !RAY) {
        const int *cells = A.bcell + KB.base;
        for (int i = threadIdx.x; i < KB.nslot; i += nthr) {
            const float vi = WINT ? sI[i] : 0.0f;
            const float v = (WINT == 3) ? (vi * K.S[K.gfirst[qbase / A.NB]].TW) : sT[i];
            if (v != 0.0f || vi != 0.0f || (ALI && sX[i] != 0.0f)) {
                const int cell = cells[i];
                soc_tally(S.TABS, cell, v);
                if (WINT) soc_tally(K.S[K.gfirst[qbase / A.NB]].INT, cell, vi);      // the launches this workgroup's queue belongs to
                if (ALI && (sX[i] != 0.0f)) soc_tally(S.XAB, cell, sX[i]);
                if (WINT == 2) {
                    float *IV = K.S[K.gfirst[qbase / A.NB]].INTV;
                    const long C = K.S[0].CELLS;
                    soc_tally(IV, cell, sV[i]);  soc_tally(IV + C, cell, sV[BV + i]);  soc_tally(IV + 2 * C, cell, sV[2 * BV + i]);
                }
            }
        }
    }
    SOC_PROF_T(7);                                         // tallies -> global memory
    // every packet of the chunk: its queue -> its rank among the workgroup's packets for that queue (kept in posq
    // meanwhile; the barrier above made the workgroup's keyq stores visible to all its threads)
    for (int j = threadIdx.x; j < D.count; j += nthr) {
        int k = (int)A.keyq[D.start + j];
        if (k < 0) { k = qbase + A.rbrick[-1 - k];  A.keyq[D.start + j] = (uint32_t)k; }      // a packet that left: root cell -> brick queue
        A.posq[D.start + j] = soc_qh_rank(sH, A.HS, k, A.hist);
    }
    __syncthreads();
    soc_qh_bases(sH, A.HS, NQ, A.hist);
    __syncthreads();
    for (int j = threadIdx.x; j < D.count; j += nthr) A.posq[D.start + j] = soc_qh_place(sH, A.HS, A.posq[D.start + j]);
    if (!RAY && threadIdx.x == 0 && S.stats) atomicAdd(S.stats + 0, (unsigned long long)(unsigned int)sCtl[1]);
    if (RAY && threadIdx.x == 0 && S.stats) atomicAdd(S.stats + 3, (unsigned long long)(unsigned int)sCtl[1]);      // cell steps of all rays
    SOC_PROF_T(4);                                         // ranks and places of the chunk's packets in their next queues
    SOC_PROF_FLUSH;
}


// SOURCE == 3: the next packet of work item `id` from the loaded record; III counts the pixels tried.  false: none left.
template <bool OCT, typename W>
__device__ __forceinline__ bool soc_roi_create(const SocGrid &G, const SocSim &S, const int *sOFF, const int id, int &III, W &w)
{
    const SocRoi &R = *S.ROI;
    const int relem = id % R.NELEM;
    int   iside = relem, rside;
    float RDX, RDY;
    const float rd = (float)G.NX / ((float)R.DIM[0]);
    if (iside < (R.DIM[1] * R.DIM[2])) {
        RDX = ((float)(iside % R.DIM[1]) + 0.5f) * rd;  RDY = ((float)(iside / R.DIM[1]) + 0.5f) * rd;  rside = 0;
    } else {
        iside -= R.DIM[1] * R.DIM[2];
        if (iside < (R.DIM[0] * R.DIM[2])) {
            RDX = ((float)(iside % R.DIM[0]) + 0.5f) * rd;  RDY = ((float)(iside / R.DIM[0]) + 0.5f) * rd;  rside = 1;
        } else {
            iside -= R.DIM[0] * R.DIM[2];
            rside = 3;
            RDX = 0.0f;  RDY = 0.0f;
            if (iside < (R.DIM[0] * R.DIM[1])) {
                RDX = ((float)(iside % R.DIM[0]) + 0.5f) * rd;  RDY = ((float)(iside / R.DIM[0]) + 0.5f) * rd;  rside = 2;
            }
        }
    }
    const float RX0 = (float)((double)(R.NSIDE * R.NSIDE) * 12.0 / (100.0 * (double)S.BATCH));
    const int npix = 12 * R.NSIDE * R.NSIDE;
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
        return true;
    }
    return false;
}

// creation and scattering, one lane per queued packet
template <bool OCT, bool ABU, int WINT, int KIND, bool LT>
__device__ __forceinline__ void soc_brick_events(const SocGrid &G, const SocSimPack &K, const SocBrickArgs &A, const int ebid, const int slice)
{
    constexpr int  SRC = (KIND == 3) ? 1 : -1;             // KIND 3: SimRAM_PB with SOURCE == 1 (background) only
    // ebid counts from the first event descriptor (event queues sort last);
    // slice: blockDim.x packets of the chunk -- one packet per lane
    const int di = A.ndesc[2] + ebid;
    if (di >= *A.ndesc) return;
    SocDesc D = A.desc[di];
    if (D.brick < A.NBQ) return;
    {
        const int first = (int)(slice * blockDim.x);
        if (first >= D.count) return;
        D.start += first;
        D.count = min((int)blockDim.x, D.count - first);
    }
    SocPk2 *pk = A.pk;
    const int NQ = A.NBQ + A.EQ * K.n + 1;
    const int lq = (D.brick - A.NBQ) / A.EQ;                  // the launch this queue belongs to (workgroup-uniform)
    const SocSim &S = K.S[lq];
    // KIND 4: launches of several kinds share the sweep, the kind is the launch's (uniform in the workgroup: an event queue belongs to one launch)
    const int  skind = (KIND == 4) ? ((S.SOURCE == SOC_SOURCE_CL) ? 2 : (S.SOURCE == SOC_SOURCE_HP) ? 1 : 0) : KIND;
    const bool CL = (skind == 2), HP = (skind == 1);
    const int qbase = (A.NBQ > A.NB) ? K.grp[lq] * A.NB : 0;      // first brick queue of this launch's group
    extern __shared__ float lds[];
    int   *sH   = (int *)lds;                              // arrivals per queue
    int   *sCtl = sH + (A.HS ? 2 * A.HS : NQ);             // [0..2] stats
    int   *sOFF = sCtl + 4;                                // [SOC_MAXL]
    soc_qh_init(sH, A.HS, NQ);
    if (threadIdx.x < 4) sCtl[threadIdx.x] = 0;
    if (threadIdx.x < SOC_MAXL) sOFF[threadIdx.x] = G.OFF[threadIdx.x];
    __syncthreads();
    unsigned int n_tally = 0, n_pkt = 0, n_scat = 0;
    uint32_t mypack = 0;                                   // one packet per lane (D.count <= blockDim.x)

    for (int j = threadIdx.x; j < D.count; j += blockDim.x) {
        const uint32_t wid = A.idq[D.start + j];
        SocPk2 p = pk[wid];
        SocBrickLane w;
        w.px = p.A.x;  w.py = p.A.y;  w.pz = p.A.z;  w.photons = p.A.w;
        w.ux = p.B.x;  w.uy = p.B.y;  w.uz = p.B.z;  w.free_path = p.B.w;
        w.tau = p.C.x;  w.dens = p.C.y;
        int lid = __float_as_int(p.C.z) & 0xffff;
        w.level = OCT ? ((__float_as_int(p.C.z) >> SOC_LVL_SHIFT) & 15) : 0;
        w.ind = __float_as_int(p.C.w);
        w.rng.x = p.D.x;  w.rng.c = p.D.y;
        int III = (int)(p.D.z & 0xffffffu);
        w.scat = (int)(p.D.z >> 24);
        int key = (int)p.D.w;
        uint32_t cl_cell = p.D.w;                              // SimRAM_CL: the cell the work item emits from
        const int evk = (D.brick - A.NBQ) % A.EQ;              // 0 creation, 1 scattering, 2 slow step (brick-local hierarchies)
        bool create = (evk == 0);
        int ccx = 0, ccy = 0, ccz = 0;                         // brick-local hierarchies: cell coordinates on the cell's level
        if (LT) {
            // record of soc_lbrick_walk: C = tau, cell coordinates; D.z = III | scatterings << 24 | level << 29
            ccx = __float_as_int(p.C.y);  ccy = __float_as_int(p.C.z);  ccz = __float_as_int(p.C.w) & ((1 << SOC_LQ_SHIFT) - 1);      // (the launch above it: this queue's)
            w.level = (int)(p.D.z >> 29);
            w.scat  = (int)((p.D.z >> 24) & 31u);
            if (evk == 2) cl_cell = p.D.w & ~SOC_LT_ARRIVE;                      // (a work item that has not started holds a negative cell)
            w.ind = -1;
            if (!create) w.ind = soc_cell_index(G, w.level, ccx, ccy, ccz, w.dens);      // the cell the packet is in (or stepped from)
        }
        // Reflecting faces (the `mirror` key; Mirror, kernel_ASOC_aux.c:1054-1083, called where a step has taken the packet out of the
        // model: kernel_ASOC.c:686-688, :1064, :1540).  Brick-local hierarchies: such a packet arrives in its launch's creation
        // queue with its old cell and the advanced position; the root-grid position Index() leaves behind (:238-241) follows
        // from them, Mirror works on that.  A packet that comes back inside goes on in the brick of its new cell.
        bool mirrored = false;
        if (LT && create && (S.MIRROR > 0)) {
            const bool started = CL ? (((int)cl_cell >= 0) && (III > 0)) : (III > 0);      // (a work item that has sent no packet yet holds none)
            if (started) {
                if (w.level > 0) {
                    const float sc = soc_lt_pow2(-w.level);
                    w.px = SOC_FMA(w.px, sc, (float)(ccx & ~1) * sc);  w.py = SOC_FMA(w.py, sc, (float)(ccy & ~1) * sc);  w.pz = SOC_FMA(w.pz, sc, (float)(ccz & ~1) * sc);
                }
                soc_mirror<true>(G, sOFF, S.MIRROR, w.px, w.py, w.pz, w.ux, w.uy, w.uz, w.level, w.ind, w.dens);
                mirrored = (w.ind >= 0);
            }
        }
        if (LT && (evk == 3)) {
            // the walk saw the packet step into the region of interest: its place in the record (kernel_ASOC.c:615-642, :1510-1535), then on
            // with the walk in the brick of its cell
            soc_roi_save(G, sOFF, *S.ROI, w.px, w.py, w.pz, w.ux, w.uy, w.uz, w.level, w.ind, w.photons);
            key = qbase + A.rbrick[((ccz >> w.level) * G.NY + (ccy >> w.level)) * G.NX + (ccx >> w.level)];
        } else
        if (LT && (evk == 2)) {
            // slow step: Index() itself, in double, for a step that exact geometry does not decide (soc_ltree.h).  The
            // packet holds its old cell and the advanced position; tallies of the step are done.
            const int ind0 = w.ind, level0 = w.level;
            soc_index<true, double>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
            if (S.ROISAVE && (w.ind >= 0) && (soc_inroi(G, sOFF, *S.ROI, level0, ind0) < 0) && (soc_inroi(G, sOFF, *S.ROI, w.level, w.ind) >= 0))
                soc_roi_save(G, sOFF, *S.ROI, w.px, w.py, w.pz, w.ux, w.uy, w.uz, w.level, w.ind, w.photons);      // the step took it into ROI (kernel_ASOC.c:615-642)
            if (!CL && (w.ind >= 0) && (w.level == level0) && (w.ind == ind0)) {   // failed step: nudge (SimRAM_PB / HP only), before Mirror as in the step
                w.px += SOC_PEPS * w.ux;  w.py += SOC_PEPS * w.uy;  w.pz += SOC_PEPS * w.uz;
            }
            if ((w.ind < 0) && (S.MIRROR > 0)) soc_mirror<true>(G, sOFF, S.MIRROR, w.px, w.py, w.pz, w.ux, w.uy, w.uz, w.level, w.ind, w.dens);
            if (w.ind < 0) {
                create = true;                                                  // left the model: the work item's next packet
            } else {
                soc_cell_coords(G, sOFF, w.level, w.ind, ccx, ccy, ccz);
                key = qbase + A.rbrick[((ccz >> w.level) * G.NY + (ccy >> w.level)) * G.NX + (ccx >> w.level)];
            }
        } else
        if (mirrored) {
            create = false;
            soc_cell_coords(G, sOFF, w.level, w.ind, ccx, ccy, ccz);
            key = qbase + A.rbrick[((ccz >> w.level) * G.NY + (ccy >> w.level)) * G.NX + (ccx >> w.level)];
        } else
        if (!create) {
            // scattering block (kernel_ASOC.c:700-804); the packet is at the start of the step
            const int oind = (OCT ? sOFF[w.level] : 0) + w.ind;
            float kabs, ksca;
            if (ABU) { float2 o = S.OPT[oind];  kabs = o.x;  ksca = o.y; }
            else     { kabs = S.ABS;  ksca = S.SCA; }
            w.scat++;
            if (CL && (w.scat > 20)) {                                        // SimRAM_CL drops it before the deposit (kernel_ASOC.c:1552-1553)
                w.ind = -1;
                create = true;
            } else {
            float dt = w.free_path - w.tau;
            float dx = dt / (ksca * w.dens);
            float tauA = dx * w.dens * kabs;
            float e = soc_expf(-tauA);
            float delta = (tauA > SOC_TAULIM) ? (w.photons * (1.0f - e)) : (w.photons * tauA * (1.0f - 0.5f * tauA));
            if (CL && S.XAB && (oind == (int)cl_cell)) soc_tally(S.XAB, oind, delta * S.TW);      // WITH_ALI: absorbed in the emitting cell itself (kernel_ASOC.c:1486-1494)
            else                                      soc_tally(S.TABS, oind, delta * S.TW);
            if (WINT) soc_tally(S.INT, oind, delta);
            if (WINT == 2) {                                                  // -D SAVE_INTENSITY=2 (kernel_ASOC.c:724-732)
                soc_tally(S.INTV, oind, delta * w.ux);  soc_tally(S.INTV + S.CELLS, oind, delta * w.uy);  soc_tally(S.INTV + 2 * (long)S.CELLS, oind, delta * w.uz);
            }
            n_tally++;
            n_scat++;
            dx = soc_scale_up(dx, w.level);
            dx = __builtin_fmaxf(0.0f, dx - 2.0f * SOC_PEPS);
            w.px = w.px + dx * w.ux;
            w.py = w.py + dx * w.uy;
            w.pz = w.pz + dx * w.uz;
            w.photons *= e;
            w.free_path = soc_draw_free_path(S, &w.rng, w.photons);
            if (CL) soc_new_direction<true>(S, S.CSC, oind, w.ux, w.uy, w.uz, w.free_path, &w.rng);   // one table read per event: no staging
            else    soc_new_direction<false>(S, S.CSC, oind, w.ux, w.uy, w.uz, w.free_path, &w.rng);
            w.tau = 0.0f;
            if (!CL && (w.scat > 20)) { w.ind = -1;  create = true; }         // dropped after 20 scatterings
            if (LT) {                                                        // back to the brick of its cell
                key = qbase + A.rbrick[((ccz >> w.level) * G.NY + (ccy >> w.level)) * G.NX + (ccx >> w.level)];
            } else
            if (CL) {                                                        // back to the brick of its cell (D.w holds the emitting cell)
                if (OCT) {
                    const uint32_t si = __float_as_uint(A.DS[oind].y);
                    key = (int)(si >> SOC_SLOT_BITS);  lid = (int)(si & SOC_SLOT_MASK);
                } else {
                    const int M = (1 << A.LB) - 1, ix = w.ind % G.NX, iy = (w.ind / G.NX) % G.NY, iz = w.ind / (G.NX * G.NY);
                    key = ((iz >> A.LB) * A.NBY + (iy >> A.LB)) * A.NBX + (ix >> A.LB);
                    lid = ((iz & M) << (2 * A.LB)) | ((iy & M) << A.LB) | (ix & M);
                }
                key += qbase;
            }
            }
        }
        if (create && CL) {
            // SimRAM_CL (kernel_ASOC.c:1283-1432): the work item walks the cells id, id+GLOBAL, ... and sends `batch`
            // packets from each.  Its place is kept in the record: D.w = cell, III = packets sent from it; batch and
            // the packet weight follow from the cell (EMWEI) and are recomputed.
            int ICELL = (int)cl_cell, IRAY = III, batch = -1;
            float PWEI = 1.0f;
            if (ICELL >= 0) {
                if (S.USE_EMWEIGHT > 0) {
                    PWEI  = S.EMWEI[ICELL];
                    batch = (int)soc_floorf(PWEI);
                    if (batch < 1) { batch = 1;  PWEI = (float)(1.0 / (double)(PWEI + 1.0e-30f)); }
                    else           { PWEI = (float)(1.0 / (double)(batch + 1.0e-9f)); }
                } else {
                    batch = S.BATCH;
                    PWEI  = 1.0f / (batch + 1.0e-9f);
                }
            }
            key = -1;
            if (IRAY >= batch) {                                              // next emitting cell (:1318-1355)
                IRAY = 0;
                long long IC = ICELL;
                while (true) {
                    IC += S.GLOBAL;
                    if (IC >= G.CELLS) { key = NQ - 1;  break; }              // work item finished
                    if (S.USE_EMWEIGHT > 0) {
                        PWEI = S.EMWEI[IC];
                        if ((PWEI < 1e-10f) || (G.DENS[IC] <= 0.0f)) continue;
                        batch = (int)soc_floorf(PWEI);
                        if (batch < 1) { batch = 1;  PWEI = (float)(1.0 / (double)(PWEI + 1.0e-30f)); }
                        else           { PWEI = (float)(1.0 / (double)(batch + 1.0e-9f)); }
                    } else {
                        batch = S.BATCH;
                        PWEI  = 1.0f / (batch + 1.0e-9f);
                    }
                    break;
                }
                ICELL = (int)IC;
            }
            if (key < 0) {
                IRAY += 1;
                int level = 0, ind = ICELL;
                if (OCT) {
                    for (level = 0; level < G.LEVELS - 1; level++) {
                        if (ind < sOFF[level + 1] - sOFF[level]) break;
                        ind -= sOFF[level + 1] - sOFF[level];
                    }
                }
                float X0, Y0, Z0;
                if (level == 0) {
                    X0 = (float)(ind % G.NX);  Y0 = (float)((ind / G.NX) % G.NY);  Z0 = (float)(ind / (G.NX * G.NY));
                } else {
                    const int sid = ind % 8;
                    X0 = (float)(sid % 2);  Y0 = ((sid % 4) > 1) ? 1.0f : 0.0f;  Z0 = (float)(sid / 4);
                }
                const int oabs = (OCT ? sOFF[level] : 0) + ind;
                w.level = level;  w.ind = ind;
                w.dens    = G.DENS[oabs];
                w.photons = S.EMIT[oabs] * PWEI;
                w.px = X0 + soc_rand(&w.rng);
                w.py = Y0 + soc_rand(&w.rng);
                w.pz = Z0 + soc_rand(&w.rng);
                const float phi       = SOC_TWOPI * soc_rand(&w.rng);
                const float cos_theta = 0.999997f - 1.999995f * soc_rand(&w.rng);
                const float sin_theta = soc_sqrtf(1.0f - cos_theta * cos_theta);
                float sp, cp;
                soc_sincosf(phi, &sp, &cp);
                w.ux = sin_theta * cp;  w.uy = sin_theta * sp;  w.uz = cos_theta;
                n_pkt++;
                w.begin(S);
                if (LT) {
                    soc_cell_coords(G, sOFF, level, ind, ccx, ccy, ccz);
                    key = A.rbrick[((ccz >> level) * G.NY + (ccy >> level)) * G.NX + (ccx >> level)];
                } else
                if (OCT) {
                    const uint32_t si = __float_as_uint(A.DS[oabs].y);
                    key = (int)(si >> SOC_SLOT_BITS);  lid = (int)(si & SOC_SLOT_MASK);
                } else {
                    const int M = (1 << A.LB) - 1, ix = ind % G.NX, iy = (ind / G.NX) % G.NY, iz = ind / (G.NX * G.NY);
                    key = ((iz >> A.LB) * A.NBY + (iy >> A.LB)) * A.NBX + (ix >> A.LB);
                    lid = ((iz & M) << (2 * A.LB)) | ((iy & M) << A.LB) | (ix & M);
                }
                key += qbase;
            }
            III = IRAY;
            cl_cell = (uint32_t)ICELL;
        }
        if (create && !CL) {
            const int id = (int)(S.gid0 + (wid - K.first[lq]));
            const SocSurfElem E = soc_surface_element<SRC>(G, S, id);
            while (true) {
                if (III >= S.BATCH) { key = NQ - 1;  break; }                  // work item finished
                if ((SRC < 0) && (S.SOURCE == 3)) {
                    // packets of a loaded region-of-interest record (-D WITH_ROI_LOAD, kernel_ASOC.c:141-179, :469-501): 100 work items per
                    // surface element, the Healpix pixels of the element in turn; an empty pixel is skipped without a draw
                    const int before = III;
                    const bool made = soc_roi_create<OCT>(G, S, sOFF, id, III, w);
                    n_pkt += (unsigned int)(III - before);
                    if (!made) { key = NQ - 1;  break; }
                    w.begin(S);
                    III--;  n_pkt--;                                           // (counted again below)
                } else
                if (HP) {                                                      // SimRAM_HP (kernel_ASOC.c:878-955)
                    soc_hp_create<OCT>(G, S, sOFF, w);
                    w.begin_conditioned(S);
                } else {
                    soc_pb_create<OCT, SocBrickLane, SRC>(G, S, sOFF, E, III, w);
                    w.begin(S);
                }
                III++;
                n_pkt++;
                if (w.ind >= 0) {
                    if (LT) {
                        soc_cell_coords(G, sOFF, w.level, w.ind, ccx, ccy, ccz);
                        key = A.rbrick[((ccz >> w.level) * G.NY + (ccy >> w.level)) * G.NX + (ccx >> w.level)];
                    } else
                    if (OCT) {
                        const uint32_t si = __float_as_uint(A.DS[sOFF[w.level] + w.ind].y);
                        key = (int)(si >> SOC_SLOT_BITS);  lid = (int)(si & SOC_SLOT_MASK);
                    } else {
                        soc_cell_brick(A, w.px, w.py, w.pz, key, lid);
                    }
                    key += qbase;
                    break;
                }
            }
        }
        p.A = make_float4(w.px, w.py, w.pz, w.photons);
        p.B = make_float4(w.ux, w.uy, w.uz, w.free_path);
        if (LT) {
            p.C = make_float4(w.tau, __int_as_float(ccx), __int_as_float(ccy), __int_as_float(ccz | (lq << SOC_LQ_SHIFT)));
            p.D = make_uint4(w.rng.x, w.rng.c, (uint32_t)III | ((uint32_t)w.scat << 24) | ((uint32_t)w.level << 29), CL ? cl_cell : 0u);
        } else {
        p.C = make_float4(w.tau, w.dens, __int_as_float(lid | ((OCT ? w.level : 0) << SOC_LVL_SHIFT) | (lq << SOC_LCH_SHIFT)), __int_as_float(w.ind));
        p.D = make_uint4(w.rng.x, w.rng.c, (uint32_t)III | ((uint32_t)w.scat << 24), CL ? cl_cell : (uint32_t)key);
        }
        pk[wid] = p;
        A.keyq[D.start + j] = (uint32_t)key;
        mypack = soc_qh_rank(sH, A.HS, key, A.hist);
    }
    atomicAdd(&sCtl[0], (int)n_tally);
    atomicAdd(&sCtl[1], (int)n_pkt);
    atomicAdd(&sCtl[2], (int)n_scat);
    __syncthreads();
    soc_qh_bases(sH, A.HS, NQ, A.hist);
    __syncthreads();
    if ((int)threadIdx.x < D.count) A.posq[D.start + threadIdx.x] = soc_qh_place(sH, A.HS, mypack);
    if (threadIdx.x == 0 && S.stats) {
        atomicAdd(S.stats + 0, (unsigned long long)(unsigned int)sCtl[0]);
        atomicAdd(S.stats + 1, (unsigned long long)(unsigned int)sCtl[1]);
        atomicAdd(S.stats + 2, (unsigned long long)(unsigned int)sCtl[2]);
    }
}

// One launch per pass for both: blocks [0, nwalk) walk the descriptor of their index (and leave at
// once if it belongs to an event queue), the blocks after them are the event workgroups.  The short,
// latency-bound event work runs beside the walk instead of after it.
template <bool OCT, bool DBL, bool ABU, bool WINT, int KIND>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6))) void soc_brick_pass(const SocGrid G, const SocSimPack *Kp, const SocBrickArgs A, const int nwalk, const int slices)
{
    const SocSimPack &K = *Kp;
    const int b = (int)blockIdx.x;
    if (b < nwalk) {
        soc_brick_walk<OCT, DBL, ABU, WINT, KIND>(G, K, A, b);
    } else {
        const int e = b - nwalk;
        soc_brick_events<OCT, ABU, WINT, KIND, false>(G, K, A, e / slices, e % slices);
    }
}

// the same for brick-local hierarchies (soc_lbrick_walk).  The brick's cells take 8 bytes of LDS each, which is what
// bounds the waves per SIMD here, not the registers.
template <int WINT, int KIND>
__global__ __launch_bounds__(1024) void soc_lbrick_pass(const SocGrid G, const SocSimPack *Kp, const SocBrickArgs A, const int nwalk, const int slices)
{
    const SocSimPack &K = *Kp;
    const int b = (int)blockIdx.x;
    if (b < nwalk) {
        soc_lbrick_walk<WINT>(G, K, A, b);
    } else {
        const int e = b - nwalk;
        soc_brick_events<true, false, WINT, KIND, true>(G, K, A, e / slices, e % slices);
    }
}


// SimRAM_CL launches with the XAB tally of -D WITH_ALI
template <int WINT>
__global__ __launch_bounds__(1024) void soc_lbrick_pass_ali(const SocGrid G, const SocSimPack *Kp, const SocBrickArgs A, const int nwalk, const int slices)
{
    const SocSimPack &K = *Kp;
    const int b = (int)blockIdx.x;
    if (b < nwalk) {
        soc_lbrick_walk<WINT, false, true>(G, K, A, b);
    } else {
        const int e = b - nwalk;
        soc_brick_events<true, false, WINT, 2, true>(G, K, A, e / slices, e % slices);
    }
}

// ---------------------------------------------------------------------------------------
// Scattered-light images (kernel_ASOC_sca.c) on brick-local hierarchies.
//
// A work item of the reference alternates between read-only walks -- the look-ahead of forced first scattering (:888-906), the
// packet's free walk (:967-990), one walk to the surface per observer at every scattering (:1019-1047) -- and short blocks in
// between.  Here every such walk is a RAY: a record of the same 64 bytes as an absorption packet, queued per brick and stepped by
// soc_lbrick_walk<false, true> on the brick's cells in LDS (4 B per cell: no tallies).  What a ray does not need while it walks
// stays behind in a second record per work item that never moves (`park`): the packet's position and direction at the scattering
// (or at its creation), its photons, its cell, the observer the current ray looks at.  The blocks in between run here, one lane
// per ray that has ended: it left the model (queue 0 of its launch), the packet's free path ends in the cell (queue 1), or the
// step was one that exact geometry does not decide (queue 2, soc_ltree.h).  The chain of blocks a lane runs through is the one
// of soc_sca_kernel's service arm (soc_sca.hip), whose order of operations and of RNG draws is the reference's; it ends when the
// lane has a ray to walk again or the work item is finished.
//   ray record:  A = position, kappa factor | B = direction, free path (+inf for look-ahead and peel-off rays) | C = optical
//                depth so far, cell coordinates | D = RNG state, III | scatterings << 24 | level << 29, SimRAM_CL: emitting cell
//   park record: A = packet position, photons | B = packet direction | C = its cell coordinates, level | D = kind of the current
//                ray (SOC_RM_*), observer index
// Not on this path (the caller falls back to soc_sca_kernel): Healpix images, SimRAM_HP, per-cell opacities, WITH_MSF.
// ---------------------------------------------------------------------------------------
enum { SOC_RM_NONE = 0, SOC_RM_FFS = 1, SOC_RM_MAIN = 2, SOC_RM_PEEL = 3 };
enum { SOC_RE_SCAT = 4, SOC_RE_PEEL_END = 5, SOC_RE_FFS_END = 6, SOC_RE_CREATE = 7, SOC_RE_DONE = 8 };     // blocks between rays
#define SOC_SCA_MAX_SCATTERINGS 30                                    /* kernel_ASOC_sca.c:5 */

struct SocRayLane {                                                   // what soc_pb_create fills in
    float px, py, pz, ux, uy, uz, photons, dens;
    int   level, ind;
    soc_rng_t rng;
};

__device__ __forceinline__ void soc_sca_events(const SocGrid &G, const SocSimPack &K, const SocBrickArgs &A, const int ebid, const int slice)
{
    const SocSca &V = A.sca;
    const int di = A.ndesc[2] + ebid;
    if (di >= *A.ndesc) return;
    SocDesc D = A.desc[di];
    if (D.brick < A.NBQ) return;
    {
        const int first = (int)(slice * blockDim.x);
        if (first >= D.count) return;
        D.start += first;
        D.count = min((int)blockDim.x, D.count - first);
    }
    SocPk2 *pk = A.pk, *park = A.park;
    const int NQ = A.NBQ + A.EQ * K.n + 1;
    const int lq = (D.brick - A.NBQ) / A.EQ;
    const int evk = (D.brick - A.NBQ) % A.EQ;                 // 0 left the model (or not started), 1 scattering, 2 slow step
    const SocSim &S = K.S[lq];
    const int  KIND = S.SCAKIND - 1;                          // the kernel of this queue's launch (uniform in the workgroup)
    const bool CLW = (KIND == SOC_SCA_CL);
    const int qbase = 0;                                      // rays of all launches share the brick queues (no tallies in LDS)
    extern __shared__ float lds[];
    int   *sH   = (int *)lds;
    int   *sCtl = sH + (A.HS ? 2 * A.HS : NQ);
    int   *sOFF = sCtl + 4;
    soc_qh_init(sH, A.HS, NQ);
    if (threadIdx.x < 4) sCtl[threadIdx.x] = 0;
    if (threadIdx.x < SOC_MAXL) sOFF[threadIdx.x] = G.OFF[threadIdx.x];
    __syncthreads();
    unsigned int n_add = 0, n_pkt = 0, n_scat = 0;
    uint32_t mypack = 0;
    const float kabs = S.ABS, ksca = S.SCA;
    const int   NDIRS = V.NDIR;

    for (int j = threadIdx.x; j < D.count; j += blockDim.x) {
        const uint32_t wid = A.idq[D.start + j];
        const SocPk2 p = pk[wid];
        const SocPk2 q = park[wid];
        SocRayLane w;
        w.px = p.A.x;  w.py = p.A.y;  w.pz = p.A.z;
        w.ux = p.B.x;  w.uy = p.B.y;  w.uz = p.B.z;
        float free_path = p.B.w, tau = p.C.x;
        int   ccx = __float_as_int(p.C.y), ccy = __float_as_int(p.C.z), ccz = __float_as_int(p.C.w) & ((1 << SOC_LQ_SHIFT) - 1);
        w.level = (int)(p.D.z >> 29);
        int   scat = (int)((p.D.z >> 24) & 31u), III = (int)(p.D.z & 0xffffffu);
        w.rng.x = p.D.x;  w.rng.c = p.D.y;
        uint32_t cl_cell = p.D.w & ~SOC_LT_ARRIVE;
        if ((evk == 0) && (q.D.x == SOC_RM_NONE)) cl_cell = p.D.w;       // (a work item that has not started holds a negative cell)
        float mx = q.A.x, my = q.A.y, mz = q.A.z;
        w.photons = q.A.w;
        float dx_ = q.B.x, dy_ = q.B.y, dz_ = q.B.z;
        int   mcx = __float_as_int(q.C.x), mcy = __float_as_int(q.C.y), mcz = __float_as_int(q.C.z), mlevel = __float_as_int(q.C.w);
        int   rmode = (int)q.D.x, idir = (int)q.D.y;
        w.ind = -1;  w.dens = 0.0f;
        int   lvl_post = 0, mode;
        bool  cc_ok = true;                                   // (ccx, ccy, ccz) are the coordinates of cell (w.level, w.ind)

        if (rmode == SOC_RM_NONE) {
            mode = SOC_RE_CREATE;
        } else if (evk == 1) {
            // the packet's free path ends in this step (:1000-1015 PB, :1295-1310 CL, :1789-1805 PS); the record holds the start of the step.
            // The offset of the scattering is scaled with the level AFTER the step: GetStep once more, on the global hierarchy.
            w.ind = soc_cell_index(G, w.level, ccx, ccy, ccz, w.dens);
            float tx = w.px, ty = w.py, tz = w.pz, td = w.dens;
            int   tl = w.level, ti = w.ind;
            (void)soc_getstep_rcp<true, true>(G, sOFF, tx, ty, tz, w.ux, w.uy, w.uz, 1.0f / w.ux, 1.0f / w.uy, 1.0f / w.uz, tl, ti, td);
            lvl_post = tl;
            mode = SOC_RE_SCAT;
        } else {
            if (evk == 2) {
                // slow step: Index() itself, in double; the record holds the old cell and the advanced position, the optical depth of the step is added
                w.ind = soc_cell_index(G, w.level, ccx, ccy, ccz, w.dens);
                soc_index<true, double>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                cc_ok = false;
            } else {
                // the ray has left the model: the root-grid position Index() leaves behind (kernel_ASOC_aux.c:238-241)
                if (w.level > 0) {
                    const float sc = soc_lt_pow2(-w.level);
                    w.px = SOC_FMA(w.px, sc, (float)(ccx & ~1) * sc);  w.py = SOC_FMA(w.py, sc, (float)(ccy & ~1) * sc);  w.pz = SOC_FMA(w.pz, sc, (float)(ccz & ~1) * sc);
                }
                w.level = 0;  w.ind = -1;
            }
            if (w.ind >= 0) {
                mode = rmode;                                 // the slow step stayed inside: the ray goes on
            } else if (rmode == SOC_RM_FFS) {
                mode = SOC_RE_FFS_END;
            } else if (rmode == SOC_RM_PEEL) {
                mode = SOC_RE_PEEL_END;
            } else {
                if (S.MIRROR > 0) {                           // kernel_ASOC_sca.c:983, :1283, :1781
                    soc_mirror<true>(G, sOFF, S.MIRROR, w.px, w.py, w.pz, w.ux, w.uy, w.uz, w.level, w.ind, w.dens);
                    cc_ok = false;
                }
                mode = (w.ind >= 0) ? SOC_RM_MAIN : SOC_RE_CREATE;
            }
        }

        const int id = (int)(S.gid0 + (wid - K.first[lq]));
        SocSurfElem E;
        if (KIND != SOC_SCA_CL) E = soc_surface_element(G, S, id);
        while (mode > SOC_RM_PEEL && mode != SOC_RE_DONE) {
            // ---- start of a scattering event
            if (mode == SOC_RE_SCAT) {
                scat++;
                n_scat++;
                float dx = (free_path - tau) / (ksca * w.dens);
                dx = soc_scale_up(dx, lvl_post);
                w.px = w.px + dx * w.ux;
                w.py = w.py + dx * w.uy;
                w.pz = w.pz + dx * w.uz;
                w.photons *= soc_expf(-free_path * kabs / ksca);
                mx = w.px;  my = w.py;  mz = w.pz;  dx_ = w.ux;  dy_ = w.uy;  dz_ = w.uz;
                mlevel = w.level;  mcx = ccx;  mcy = ccy;  mcz = ccz;
                idir = 0;
                if (NDIRS > 0) {
                    const float4 o = V.ODIRS[0];
                    w.ux = o.x;  w.uy = o.y;  w.uz = o.z;
                    tau = 0.0f;
                    mode = SOC_RM_PEEL;
                } else {
                    mode = SOC_RE_PEEL_END;                   // no observers: straight to the deflection
                    idir = -1;
                }
            }
            // ---- a peel-off ray has reached the surface: image contribution, next observer or deflection
            if (mode == SOC_RE_PEEL_END) {
                if (idir >= 0) {
                    const float taup = tau;
                    const float CL = CLW ? 0.9999f : 0.999f;
                    const float cos_theta = soc_clampf(dx_ * w.ux + dy_ * w.uy + dz_ * w.uz, -CL, +CL);
                    float delta;
                    if (CLW) {
                        const float g = 0.65f;
                        const float fraction = (1.0f / (4.0f * SOC_PI)) * (1.0f - g * g) / soc_pow15f(1.0f + g * g - 2.0f * g * cos_theta);
                        delta = w.photons * fraction * ((taup > SOC_TAULIM) ? (1.0f - soc_expf(-taup)) : (taup * (1.0f - 0.5f * taup)));
                    } else {
                        int b = (int)(S.BINS * (1.0f + cos_theta) * 0.5f);
                        b = b < 0 ? 0 : (b > S.BINS - 1 ? S.BINS - 1 : b);
                        delta = w.photons * soc_expf(-taup) * S.DSC[b];
                    }
                    const float qx = w.px - V.CX, qy = w.py - V.CY, qz = w.pz - V.CZ;
                    const float4 ra = V.ORA[idir], de = V.ODE[idir];
                    int i = (int)((0.5f * V.NPIX_X - 0.00005f) + (qx * ra.x + qy * ra.y + qz * ra.z) / V.MAP_DX);
                    int jj = (int)((0.5f * V.NPIX_Y - 0.00005f) + (qx * de.x + qy * de.y + qz * de.z) / V.MAP_DX);
                    if ((i >= 0) && (jj >= 0) && (i < V.NPIX_X) && (jj < V.NPIX_Y)) {
                        i += idir * V.NPIX_X * V.NPIX_Y + jj * V.NPIX_X;
                        soc_tally(S.OUT, i, delta);
                        n_add++;
                    }
                    idir++;
                }
                // back to the packet at the scattering position
                w.px = mx;  w.py = my;  w.pz = mz;  w.level = mlevel;  ccx = mcx;  ccy = mcy;  ccz = mcz;  cc_ok = true;  w.ind = 0;
                tau = 0.0f;
                if ((idir >= 0) && (idir < NDIRS)) {
                    const float4 o = V.ODIRS[idir];
                    w.ux = o.x;  w.uy = o.y;  w.uz = o.z;
                    mode = SOC_RM_PEEL;
                } else {
                    // new direction, new free path
                    w.ux = dx_;  w.uy = dy_;  w.uz = dz_;
                    soc_scatter(w.ux, w.uy, w.uz, S.CSC, S.BINS, &w.rng);
                    free_path = -soc_logf(soc_rand(&w.rng));
                    mode = (scat == SOC_SCA_MAX_SCATTERINGS) ? SOC_RE_CREATE : SOC_RM_MAIN;
                }
            }
            // ---- the FFS look-ahead has left the cloud (:899-909 PB, :1249-1258 CL, :1733-1745 PS); tau = optical depth of scattering along the line of sight
            if (mode == SOC_RE_FFS_END) {
                w.px = mx;  w.py = my;  w.pz = mz;  w.level = mlevel;  ccx = mcx;  ccy = mcy;  ccz = mcz;  cc_ok = true;
                bool inside = (rmode != SOC_RM_NONE);         // (a packet created outside the cloud comes here without a look-ahead)
                bool alive = true;
                if (tau < 1.0e-22f) {
                    inside = false;
                    if (CLW) alive = false;                   // no random number drawn
                }
                if (alive) {
                    float W;
                    if (KIND == SOC_SCA_PS) {
                        W = -soc_expm1f(-tau);
                        free_path = -soc_logf(1.0f - W * soc_rand(&w.rng));
                    } else {
                        W = 1.0f - soc_expf(-tau);
                        free_path = -(float)soc_logd(1.0 - (double)(W * soc_rand(&w.rng)));
                    }
                    w.photons *= W;
                }
                tau  = 0.0f;
                scat = 0;
                w.ind = inside ? 0 : -1;
                mode = inside ? SOC_RM_MAIN : SOC_RE_CREATE;
            }
            // ---- next packet of this work item
            if (mode == SOC_RE_CREATE) {
                bool have = false;
                rmode = SOC_RM_NONE;
                if (KIND == SOC_SCA_CL) {
                    // :1158-1222; the work item's place is kept in the record: D.w = cell, III = packets sent from it
                    int ICELL = (int)cl_cell, IRAY = III, batch = -1;
                    float PWEI = 1.0f;
                    if (ICELL >= 0) {
                        if (S.USE_EMWEIGHT > 0) {
                            PWEI  = S.EMWEI[ICELL];
                            batch = (int)soc_floorf(PWEI);
                            if (batch < 1) { batch = 1;  PWEI = (float)(1.0 / (double)(PWEI + 1.0e-30f)); }
                            else           { PWEI = (float)(1.0 / (double)((float)batch + 1.0e-9f)); }
                        } else {
                            batch = S.BATCH;
                            PWEI  = 1.0f / ((float)batch + 1.0e-9f);
                        }
                    }
                    bool more = true;
                    if (IRAY >= batch) {
                        IRAY = 0;
                        PWEI = 1.0f;
                        long long IC = ICELL;
                        while (true) {
                            IC += S.GLOBAL;
                            if (IC >= G.CELLS) { more = false; break; }
                            if (S.USE_EMWEIGHT > 0) {
                                PWEI = S.EMWEI[IC];
                                if ((PWEI < 1e-10f) || (G.DENS[IC] <= 0.0f)) continue;
                                batch = (int)soc_floorf(PWEI);
                                if (batch < 1) { batch = 1;  PWEI = (float)(1.0 / (double)(PWEI + 1.0e-30f)); }
                                else           { PWEI = (float)(1.0 / (double)((float)batch + 1.0e-9f)); }
                            } else {
                                batch = S.BATCH;
                                PWEI  = 1.0f / ((float)batch + 1.0e-9f);
                            }
                            break;
                        }
                        ICELL = (int)IC;
                    }
                    if (!more) {
                        mode = SOC_RE_DONE;
                    } else {
                        int ind = ICELL, level;
                        IRAY += 1;
                        for (level = 0; level < G.LEVELS - 1; level++) {
                            if (ind < sOFF[level + 1] - sOFF[level]) break;
                            ind -= sOFF[level + 1] - sOFF[level];
                        }
                        float X0, Y0, Z0;
                        if (level == 0) {
                            X0 = (float)(ind % G.NX);  Y0 = (float)((ind / G.NX) % G.NY);  Z0 = (float)(ind / (G.NX * G.NY));
                        } else {
                            const int sid = ind % 8;
                            X0 = (float)(sid % 2);  Y0 = ((sid % 4) > 1) ? 1.0f : 0.0f;  Z0 = (float)(sid / 4);
                        }
                        w.level = level;  w.ind = ind;
                        w.dens = G.DENS[sOFF[level] + ind];
                        w.photons = S.EMIT[sOFF[level] + ind] * PWEI;
                        w.px = X0 + soc_rand(&w.rng);
                        w.py = Y0 + soc_rand(&w.rng);
                        w.pz = Z0 + soc_rand(&w.rng);
                        const float phi       = SOC_TWOPI * soc_rand(&w.rng);
                        const float cos_theta = 0.999997f - 1.999995f * soc_rand(&w.rng);
                        const float sin_theta = soc_sqrtf(1.0f - cos_theta * cos_theta);
                        float sp, cp;
                        soc_sincosf(phi, &sp, &cp);
                        w.ux = sin_theta * cp;
                        w.uy = sin_theta * sp;
                        w.uz = cos_theta;
                        have = true;
                    }
                    III = IRAY;
                    cl_cell = (uint32_t)ICELL;
                } else {
                    if (III >= S.BATCH) {
                        mode = SOC_RE_DONE;
                    } else {
                        soc_pb_create<true>(G, S, sOFF, E, III, w);
                        III++;
                        have = true;
                    }
                }
                if (have) {
                    n_pkt++;
                    if (soc_fabsf(w.ux) < SOC_DEPS) w.ux = SOC_DEPS;
                    if (soc_fabsf(w.uy) < SOC_DEPS) w.uy = SOC_DEPS;
                    if (soc_fabsf(w.uz) < SOC_DEPS) w.uz = SOC_DEPS;
                    soc_normalize(w.ux, w.uy, w.uz);
                    tau  = 0.0f;
                    scat = 0;
                    cc_ok = false;
                    if (w.ind >= 0) { soc_cell_coords(G, sOFF, w.level, w.ind, ccx, ccy, ccz);  cc_ok = true; }
                    if (V.FFS > 0) {
                        mx = w.px;  my = w.py;  mz = w.pz;  mlevel = w.level;  mcx = ccx;  mcy = ccy;  mcz = ccz;
                        if (w.ind >= 0) { mode = SOC_RM_FFS;  rmode = SOC_RM_FFS; }
                        else            { mode = SOC_RE_FFS_END; }                  // (rmode NONE: see there)
                    } else {
                        free_path = -soc_logf(soc_rand(&w.rng));
                        mode = (w.ind >= 0) ? SOC_RM_MAIN : SOC_RE_CREATE;
                    }
                }
            }
        }

        int key = NQ - 1;                                     // work item finished
        SocPk2 o, r;
        if (mode != SOC_RE_DONE) {
            if (!cc_ok) soc_cell_coords(G, sOFF, w.level, w.ind, ccx, ccy, ccz);
            key = qbase + A.rbrick[((ccz >> w.level) * G.NY + (ccy >> w.level)) * G.NX + (ccx >> w.level)];
            rmode = mode;
        }
        const float kk = (mode == SOC_RM_PEEL) ? (kabs + ksca) : ksca;
        const float fp = (mode == SOC_RM_MAIN) ? free_path : __builtin_inff();
        o.A = make_float4(w.px, w.py, w.pz, kk);
        o.B = make_float4(w.ux, w.uy, w.uz, fp);
        o.C = make_float4(tau, __int_as_float(ccx), __int_as_float(ccy), __int_as_float(ccz | (lq << SOC_LQ_SHIFT)));
        o.D = make_uint4(w.rng.x, w.rng.c, (uint32_t)III | ((uint32_t)scat << 24) | ((uint32_t)w.level << 29), CLW ? cl_cell : 0u);
        r.A = make_float4(mx, my, mz, w.photons);
        r.B = make_float4(dx_, dy_, dz_, 0.0f);
        r.C = make_float4(__int_as_float(mcx), __int_as_float(mcy), __int_as_float(mcz), __int_as_float(mlevel));
        r.D = make_uint4((uint32_t)rmode, (uint32_t)idir, 0u, 0u);
        pk[wid] = o;
        park[wid] = r;
        A.keyq[D.start + j] = (uint32_t)key;
        mypack = soc_qh_rank(sH, A.HS, key, A.hist);
    }
    atomicAdd(&sCtl[0], (int)n_add);
    atomicAdd(&sCtl[1], (int)n_pkt);
    atomicAdd(&sCtl[2], (int)n_scat);
    __syncthreads();
    soc_qh_bases(sH, A.HS, NQ, A.hist);
    __syncthreads();
    if ((int)threadIdx.x < D.count) A.posq[D.start + threadIdx.x] = soc_qh_place(sH, A.HS, mypack);
    if (threadIdx.x == 0 && S.stats) {
        atomicAdd(S.stats + 0, (unsigned long long)(unsigned int)sCtl[0]);
        atomicAdd(S.stats + 1, (unsigned long long)(unsigned int)sCtl[1]);
        atomicAdd(S.stats + 2, (unsigned long long)(unsigned int)sCtl[2]);
    }
}

__global__ __launch_bounds__(1024) void soc_lray_pass(const SocGrid G, const SocSimPack *Kp, const SocBrickArgs A, const int nwalk, const int slices)
{
    const SocSimPack &K = *Kp;
    const int b = (int)blockIdx.x;
    if (b < nwalk) {
        soc_lbrick_walk<false, true>(G, K, A, b);
    } else {
        const int e = b - nwalk;
        soc_sca_events(G, K, A, e / slices, e % slices);
    }
}

// one workgroup: hist -> offsets of the next queues + descriptors of the next pass
__global__ __launch_bounds__(1024) void soc_brick_scan(SocBrickArgs A)
{
    __shared__ int sSum[1024], sSumD[1024];
    __shared__ int sAdm[SOC_MAXLAUNCH];
    const int NB = A.NB, tid = threadIdx.x;
    // admission: the population of the last pass (*A.total) against the target; new work items go to the end
    // of their launch's creation queue (soc_brick_scatter writes their ids)
    if (tid < SOC_MAXLAUNCH) sAdm[tid] = 0;
    __syncthreads();
    if (tid == 0) {
        const int next = A.admit[0], count = (int)A.first[A.nl];
        int n = min(A.target - *A.total, count - next);
        if (n < 0) n = 0;
        for (int l = 0; l < A.nl; l++) {
            const int lo = max(next, (int)A.first[l]), hi = min(next + n, (int)A.first[l + 1]);
            const int m = (hi > lo) ? (hi - lo) : 0;
            sAdm[l] = m;
            A.admit[1 + 3 * l] = lo;
            A.admit[2 + 3 * l] = m;
            if (m) atomicAdd(&A.hist[A.ev_brick + A.EQ * l], m);            // at L2: the loads below must see it
        }
        A.admit[0] = next + n;
    }
    __syncthreads();
    // Descriptors of the next pass in three groups: brick queues with long chunks (>= P/2 packets) first, then the short
    // ones, then the event queues -- workgroups start in descriptor order, so the long walks are under way when the
    // short ones fill the end of the pass (the queues themselves keep their order in the id array).
    __shared__ int sSumL[1024], sSumE[1024];
    const int per = (NB + 1023) / 1024;
    const int b0 = tid * per, b1 = min(NB, b0 + per);
    const int longc = (A.P + 1) / 2;
    int s = 0, sd = 0, sl = 0, se = 0;                   // packets; descriptors of short chunks, of long chunks, of event queues
    for (int b = b0; b < b1; b++) {
        const int c = A.hist[b];
        const int nk = (c + A.P - 1) / A.P;
        s += c;
        if (b >= A.ev_brick) se += nk;
        else if (c >= longc) sl += nk;
        else                 sd += nk;
    }
    sSum[tid] = s;
    sSumD[tid] = sd;
    sSumL[tid] = sl;
    sSumE[tid] = se;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan
        int v = 0, vd = 0, vl = 0, ve = 0;
        if (tid >= d) { v = sSum[tid - d];  vd = sSumD[tid - d];  vl = sSumL[tid - d];  ve = sSumE[tid - d]; }
        __syncthreads();
        sSum[tid] += v;
        sSumD[tid] += vd;
        sSumL[tid] += vl;
        sSumE[tid] += ve;
        __syncthreads();
    }
    const int nlong = sSumL[1023], nbrickdesc = nlong + sSumD[1023];
    // parking: a brick queue shorter than min(PARK, mean length of the non-empty brick queues) is not walked in the next
    // pass -- its workgroup hands the packets back to the same queue, where they wait for company (a workgroup with a
    // packet or less per lane runs until its longest visit ends with most lanes idle).  Queues of at least the mean
    // length always exist, so every pass walks; the event queues never wait.
    int park = 0;
    if (A.PARK > 0) {
        __shared__ int sCnt[1024], sTot[1024];
        int n1 = 0, t1 = 0;
        for (int b = b0; b < b1; b++) if (b < A.ev_brick) { const int c = A.hist[b];  if (c > 0) { n1++;  t1 += c; } }
        sCnt[tid] = n1;  sTot[tid] = t1;
        __syncthreads();
        for (int d = 512; d > 0; d >>= 1) {
            if (tid < d) { sCnt[tid] += sCnt[tid + d];  sTot[tid] += sTot[tid + d]; }
            __syncthreads();
        }
        park = (sCnt[0] > 0) ? min(A.PARK, sTot[0] / sCnt[0]) : 0;
    }
    int off = sSum[tid] - s;
    int offl = sSumL[tid] - sl, offs = nlong + sSumD[tid] - sd, offe = nbrickdesc + sSumE[tid] - se;
    for (int b = b0; b < b1; b++) {
        const int c = A.hist[b];
        A.off[b] = off;
        if (b == A.ev_brick) A.ndesc_next[2] = nbrickdesc;
        int &offd = (b >= A.ev_brick) ? offe : ((c >= longc) ? offl : offs);
        if (b >= A.ev_brick && ((b - A.ev_brick) % A.EQ) == 0) {        // a creation queue: the admitted ids go to its end
            const int l = (b - A.ev_brick) / A.EQ;
            if (l < A.nl && sAdm[l]) A.admit[3 + 3 * l] = off + c - sAdm[l];
        }
        const int nk = (c + A.P - 1) / A.P;               // chunks of equal size (a queue of P+1 is not 2048 + 1)
        for (int k = 0, o = off; k < nk; k++) {
            SocDesc d;
            d.brick = b;
            d.start = o;
            d.count = c / nk + (k < c % nk ? 1 : 0);
            d.pad = ((b < A.ev_brick) && (c < park)) ? 1 : 0;
            o += d.count;
            A.desc_next[offd++] = d;
        }
        off += c;
        A.hist[b] = 0;
    }
    if (tid == 1023) {
        A.off[NB] = sSum[1023];
        *A.total = sSum[1023];
        *A.ndesc_next = nbrickdesc + sSumE[1023];
        A.hist[NB] = 0;
    }
}

// the sort: ids of the current queue -> next queues; every entry knows its queue (keyq) and its place in it (posq)
__global__ __launch_bounds__(SOC_BRICK_T) void soc_brick_scatter(SocBrickArgs A, const int nsort)
{
    if ((int)blockIdx.x >= nsort) {
        // admission (see soc_brick_scan): 16 workgroups per launch write the new ids to the end of its creation queue
        const int e = (int)blockIdx.x - nsort, l = e >> 4, part = e & 15;
        const int start = A.admit[1 + 3 * l], n = A.admit[2 + 3 * l];
        if (n > 0) {
            const int dst = A.admit[3 + 3 * l];
            for (int i = part * SOC_BRICK_T + threadIdx.x; i < n; i += 16 * SOC_BRICK_T) A.idq_next[dst + i] = (uint32_t)(start + i);
        }
        return;
    }
    if ((int)blockIdx.x >= *A.ndesc) return;
    const SocDesc D = A.desc[blockIdx.x];
    for (int j = threadIdx.x; j < D.count; j += SOC_BRICK_T) {
        const uint32_t key = A.keyq[D.start + j];
        if (key < (uint32_t)A.NB) A.idq_next[A.off[key] + A.posq[D.start + j]] = A.idq[D.start + j];
    }
}

// ---------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------

struct SocBrickBuffers {
    size_t cap_items = 0;
    int    cap_nq = 0, cap_desc = 0;
    size_t cap_park = 0;
    SocPk2 *pk = nullptr, *park = nullptr;
    uint32_t *idq[2] = { nullptr, nullptr }, *keyq = nullptr, *posq = nullptr;
    SocSimPack *pack = nullptr;
    int *hist = nullptr, *off = nullptr, *ndesc = nullptr, *total = nullptr, *admit = nullptr;
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

static void soc_oct_release(int device);

void soc_brick_release(int device)
{
    if (device < 0 || device >= 16) return;
    SocBrickBuffers &b = g_bb[device];
    void *ptrs[] = { b.pack, b.pk, b.park, b.idq[0], b.idq[1], b.keyq, b.posq, b.hist, b.off, b.ndesc, b.total, b.admit, b.desc[0], b.desc[1] };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    b = SocBrickBuffers();
    soc_oct_release(device);
}

// ---------------------------------------------------------------------------------------
// Bricks of a hierarchical grid.  A brick is a set of <= CAP cells that are close in space, each with a tally
// slot (leaves, and the refined cells above them: see place_subtree).  Built on the host once per grid:
//   1. leaves in every subtree, bottom-up over the levels;
//   2. the root grid is visited in cubes of 16^3 cells; a cube (and below that: an octant of it, a root cell's
//      subtree, a child's subtree) that holds more than CAP leaves is split into its eight parts; a part that
//      fits goes to the open brick, or starts a new one when the open brick has no room for it;
//   3. slots are given in visiting order (x fastest, octet siblings together), so that the end-of-pass flush
//      touches neighbouring cells with neighbouring lanes.
// The reference layout of DENS/OFF/PAR is untouched; the walk reads density and slot together from DS[cell].
// ---------------------------------------------------------------------------------------
struct SocOctBricks {
    bool valid = false;
    int  NB = 0, CAP = 0;
    size_t cells = 0, leaves = 0;
    const float *dens_key = nullptr;           // the hierarchy the bricks were built for
    float2 *DS = nullptr;
    int *bcell = nullptr, *bbase = nullptr;
};
static SocOctBricks g_ob[16];

// brick-local hierarchies (soc_lbricks.h) on the device, built once per grid and cap
struct SocLBricksDev {
    bool valid = false, failed = false;
    int  NB = 0, cap = 0, max_slots = 0;
    size_t cells = 0;
    const float *dens_key = nullptr;
    SocLBrick *lbr = nullptr;
    float *btree = nullptr;
    int *bcell = nullptr, *bbase = nullptr, *rbrick = nullptr;
};
static SocLBricksDev g_lb[16];

static void soc_lb_release(int device)
{
    SocLBricksDev &o = g_lb[device];
    void *ptrs[] = { o.lbr, o.btree, o.bcell, o.bbase, o.rbrick };
    for (void *q : ptrs) if (q) (void)hipFree(q);
    o = SocLBricksDev();
}

void soc_brick_invalidate(int device)
{
    if (device >= 0 && device < 16) { g_ob[device].valid = false;  g_lb[device].valid = false;  g_lb[device].failed = false; }
}

static void soc_oct_release(int device)
{
    soc_lb_release(device);
    SocOctBricks &o = g_ob[device];
    if (o.DS) (void)hipFree(o.DS);
    if (o.bcell) (void)hipFree(o.bcell);
    if (o.bbase) (void)hipFree(o.bbase);
    o = SocOctBricks();
}


static hipError_t soc_oct_build(int device, const SocGrid &G, int CAP, hipStream_t st, bool verbose)
{
    SocOctBricks &ob = g_ob[device];
    if (ob.valid && ob.CAP == CAP && ob.cells == (size_t)G.CELLS && ob.dens_key == G.DENS) return hipSuccess;
    std::vector<float> D((size_t)G.CELLS);
    BCHK(hipStreamSynchronize(st));
    BCHK(hipMemcpy(D.data(), G.DENS, (size_t)G.CELLS * 4, hipMemcpyDeviceToHost));
    SocOctBuilder B(G.NX, G.NY, G.NZ, G.LEVELS, G.CELLS, G.LCELLS, G.OFF, D.data(), CAP);
    B.build();
    const int NB = (int)B.bbase.size() - 1;
    if (NB < 1 || NB >= (1 << (32 - SOC_SLOT_BITS))) return hipErrorNotSupported;
    std::vector<float2> DS((size_t)G.CELLS);
    for (size_t i = 0; i < (size_t)G.CELLS; i++) { float y;  memcpy(&y, &B.slotmap[i], 4);  DS[i] = make_float2(D[i], y); }
    BCHK(brick_alloc(&ob.DS, (size_t)G.CELLS));
    BCHK(brick_alloc(&ob.bcell, B.bcell.size()));
    BCHK(brick_alloc(&ob.bbase, B.bbase.size()));
    BCHK(hipMemcpy(ob.DS, DS.data(), (size_t)G.CELLS * 8, hipMemcpyHostToDevice));
    BCHK(hipMemcpy(ob.bcell, B.bcell.data(), B.bcell.size() * 4, hipMemcpyHostToDevice));
    BCHK(hipMemcpy(ob.bbase, B.bbase.data(), B.bbase.size() * 4, hipMemcpyHostToDevice));
    ob.NB = NB;  ob.CAP = CAP;  ob.cells = (size_t)G.CELLS;  ob.leaves = B.bcell.size();  ob.dens_key = G.DENS;
    ob.valid = true;
    if (verbose)
        fprintf(stderr, "soc_brick: hierarchy of %d cells, %zu leaves -> %d bricks of <= %d leaves (mean %.0f)\n",
                G.CELLS, ob.leaves, NB, CAP, (double)ob.leaves / NB);
    return hipSuccess;
}

// Bricks for the walk on brick-local hierarchies.  hipErrorNotSupported: the hierarchy cannot be cut that way (a root
// cell with more than cap cells below it) -- the caller keeps the sweep that reads the hierarchy from global memory.
static hipError_t soc_lb_build(int device, const SocGrid &G, int cap, hipStream_t st, bool verbose)
{
    SocLBricksDev &lb = g_lb[device];
    if (lb.cap == cap && lb.cells == (size_t)G.CELLS && lb.dens_key == G.DENS) {
        if (lb.valid) return hipSuccess;
        if (lb.failed) return hipErrorNotSupported;
    }
    soc_lb_release(device);
    lb.cap = cap;  lb.cells = (size_t)G.CELLS;  lb.dens_key = G.DENS;
    std::vector<float> D((size_t)G.CELLS);
    BCHK(hipStreamSynchronize(st));
    BCHK(hipMemcpy(D.data(), G.DENS, (size_t)G.CELLS * 4, hipMemcpyDeviceToHost));
    SocLBricksHost H;
    if (!soc_lbricks_build(G.NX, G.NY, G.NZ, G.LEVELS, G.LCELLS, G.OFF, D.data(), cap, H) || H.bricks.size() >= (1u << 20)) {
        lb.failed = true;
        return hipErrorNotSupported;
    }
    const int NB = (int)H.bricks.size();
    std::vector<int> bbase((size_t)NB + 1);
    for (int b = 0; b < NB; b++) bbase[b] = H.bricks[b].base;
    bbase[NB] = (int)H.btree.size();
    BCHK(brick_alloc(&lb.lbr, (size_t)NB));
    BCHK(brick_alloc(&lb.btree, H.btree.size()));
    BCHK(brick_alloc(&lb.bcell, H.bcell.size()));
    BCHK(brick_alloc(&lb.bbase, bbase.size()));
    BCHK(brick_alloc(&lb.rbrick, H.rbrick.size()));
    BCHK(hipMemcpy(lb.lbr, H.bricks.data(), (size_t)NB * sizeof(SocLBrick), hipMemcpyHostToDevice));
    BCHK(hipMemcpy(lb.btree, H.btree.data(), H.btree.size() * 4, hipMemcpyHostToDevice));
    BCHK(hipMemcpy(lb.bcell, H.bcell.data(), H.bcell.size() * 4, hipMemcpyHostToDevice));
    BCHK(hipMemcpy(lb.bbase, bbase.data(), bbase.size() * 4, hipMemcpyHostToDevice));
    BCHK(hipMemcpy(lb.rbrick, H.rbrick.data(), H.rbrick.size() * 4, hipMemcpyHostToDevice));
    lb.NB = NB;  lb.max_slots = H.max_slots;
    lb.valid = true;
    if (verbose)
        fprintf(stderr, "soc_brick: hierarchy of %d cells -> %d brick-local hierarchies of <= %d cells (largest %d, mean %.0f)\n",
                G.CELLS, NB, cap, H.max_slots, (double)G.CELLS / NB);
    return hipSuccess;
}

template <int WINT, int KIND>
static void soc_lbrick_launch_one(int nblocks, int T, size_t lds, hipStream_t st, const SocGrid &G, const SocSimPack *K,
                                  const SocBrickArgs &A, int nwalk, int slices)
{
    if (lds > 64 * 1024)                           // more dynamic LDS than the default limit (per device and kernel; cheap)
        (void)hipFuncSetAttribute((const void *)soc_lbrick_pass<WINT, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    soc_lbrick_pass<WINT, KIND><<<nblocks, T, lds, st>>>(G, K, A, nwalk, slices);
}

static void soc_lbrick_launch_pass(int wint, int kind, int nblocks, int T, size_t lds, hipStream_t st, const SocGrid &G, const SocSimPack *K,
                                   const SocBrickArgs &A, int nwalk, int slices)
{
    if (A.ali) {                                             // SimRAM_CL with the XAB tally (soc_brick_run_pb has checked kind and wint)
#define SOC_LBA_CASE(W) do { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)soc_lbrick_pass_ali<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                             soc_lbrick_pass_ali<W><<<nblocks, T, lds, st>>>(G, K, A, nwalk, slices); } while (0)
        if (wint) SOC_LBA_CASE(1); else SOC_LBA_CASE(0);
#undef SOC_LBA_CASE
        return;
    }
#define SOC_LB_CASE(W, KD) soc_lbrick_launch_one<W, KD>(nblocks, T, lds, st, G, K, A, nwalk, slices)
    if (wint == 3) { if (kind == 4) SOC_LB_CASE(3, 4);  else if (kind == 3) SOC_LB_CASE(3, 3);  else if (kind == 2) SOC_LB_CASE(3, 2);  else if (kind == 1) SOC_LB_CASE(3, 1);  else SOC_LB_CASE(3, 0); }
    else if (wint == 2) { if (kind == 4) SOC_LB_CASE(2, 4);  else if (kind == 3) SOC_LB_CASE(2, 3);  else if (kind == 2) SOC_LB_CASE(2, 2);  else if (kind == 1) SOC_LB_CASE(2, 1);  else SOC_LB_CASE(2, 0); }
    else if (wint) { if (kind == 4) SOC_LB_CASE(1, 4);  else if (kind == 3) SOC_LB_CASE(1, 3);  else if (kind == 2) SOC_LB_CASE(1, 2);  else if (kind == 1) SOC_LB_CASE(1, 1);  else SOC_LB_CASE(1, 0); }
    else      { if (kind == 4) SOC_LB_CASE(false, 4); else if (kind == 3) SOC_LB_CASE(false, 3); else if (kind == 2) SOC_LB_CASE(false, 2); else if (kind == 1) SOC_LB_CASE(false, 1); else SOC_LB_CASE(false, 0); }
#undef SOC_LB_CASE
}

static hipError_t soc_lray_launch_pass(int nblocks, int T, size_t lds, hipStream_t st, const SocGrid &G, const SocSimPack *K,
                                       const SocBrickArgs &A, int nwalk, int slices)
{
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)soc_lray_pass, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    soc_lray_pass<<<nblocks, T, lds, st>>>(G, K, A, nwalk, slices);
    return hipSuccess;
}

// LB: log2 of the brick edge (Cartesian grids; hierarchies use bricks of <= CAP leaves).  nlaunch launches
// (same geometry, same tallies; no INT tally when nlaunch > 1) share one sweep: more
// packets in flight per pass, and the passes in which one launch's last work items finish are filled by the
// others.  Returns hipErrorNotSupported when the launches cannot use bricks.
template <bool OCT, bool DBL, bool ABU, bool WINT, int KIND>
static void soc_brick_launch_one(int nblocks, int T, size_t lds, hipStream_t st, const SocGrid &G, const SocSimPack *K,
                                 const SocBrickArgs &A, int nwalk, int slices)
{
    if (lds > 64 * 1024)                           // more dynamic LDS than the default limit; the attribute is per device, so it is set
        (void)hipFuncSetAttribute((const void *)soc_brick_pass<OCT, DBL, ABU, WINT, KIND>,     // on every launch (microseconds)
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    soc_brick_pass<OCT, DBL, ABU, WINT, KIND><<<nblocks, T, lds, st>>>(G, K, A, nwalk, slices);
}

// vkey: bit 0 INT tally, bit 1 per-cell opacities; kind 0 SimRAM_PB, 1 SimRAM_HP, 2 SimRAM_CL, 3 SimRAM_PB with
// background packets only (they differ in how the
// event workgroups create a packet; separate kernels so that none carries the registers of the others)
template <bool OCT, bool DBL, int KIND>
static void soc_brick_launch_kind(int vkey, int nblocks, int T, size_t lds, hipStream_t st, const SocGrid &G, const SocSimPack *K,
                                  const SocBrickArgs &A, int nwalk, int slices)
{
    switch (vkey) {
    case 0:  soc_brick_launch_one<OCT, DBL, false, false, KIND>(nblocks, T, lds, st, G, K, A, nwalk, slices); break;
    case 1:  soc_brick_launch_one<OCT, DBL, false, true, KIND>(nblocks, T, lds, st, G, K, A, nwalk, slices); break;
    case 2:  soc_brick_launch_one<OCT, DBL, true, false, KIND>(nblocks, T, lds, st, G, K, A, nwalk, slices); break;
    default: soc_brick_launch_one<OCT, DBL, true, true, KIND>(nblocks, T, lds, st, G, K, A, nwalk, slices); break;
    }
}

template <bool OCT, bool DBL>
static void soc_brick_launch_pass(int vkey, int kind, int nblocks, int T, size_t lds, hipStream_t st, const SocGrid &G, const SocSimPack *K,
                                  const SocBrickArgs &A, int nwalk, int slices)
{
    if (kind == 3)      soc_brick_launch_kind<OCT, DBL, 3>(vkey, nblocks, T, lds, st, G, K, A, nwalk, slices);
    else if (kind == 2) soc_brick_launch_kind<OCT, DBL, 2>(vkey, nblocks, T, lds, st, G, K, A, nwalk, slices);
    else if (kind == 1) soc_brick_launch_kind<OCT, DBL, 1>(vkey, nblocks, T, lds, st, G, K, A, nwalk, slices);
    else                soc_brick_launch_kind<OCT, DBL, 0>(vkey, nblocks, T, lds, st, G, K, A, nwalk, slices);
}

hipError_t soc_brick_run_pb(int device, const SocGrid &G, const SocSim *Sin, int nlaunch, const SocVariant &V, int LB,
                            int population, const SocBrickTune &tune, hipStream_t st, int *passes_out, int *form_out, const SocSca *sca)
{
    if (form_out) *form_out = 0;
    // rays of the scattered-light kernels (soc_sca_events): flat images of SimRAM_PB / PS / CL with scalar opacities and one scattering function
    if (sca) {
        if (sca->NDIR < 0 || V.abu || V.wint) return hipErrorNotSupported;
        for (int l = 0; l < nlaunch; l++) {
            const int k = Sin[l].SCAKIND - 1;
            if ((k != SOC_SCA_PB && k != SOC_SCA_PS && k != SOC_SCA_CL) || Sin[l].NDUST > 1 || Sin[l].BINS < 1 || !Sin[l].OUT) return hipErrorNotSupported;
            if ((k != SOC_SCA_CL) && !Sin[l].DSC) return hipErrorNotSupported;
            if ((k == SOC_SCA_CL) != (Sin[l].SOURCE == SOC_SOURCE_CL)) return hipErrorInvalidValue;
        }
    }
    if (device < 0 || device >= 16 || nlaunch < 1 || nlaunch > SOC_MAXLAUNCH) return hipErrorNotSupported;
    const int B = 1 << LB;
    SocBrickArgs A{};
    A.LB = LB;
    A.ev_brick = -1;
    // workgroup shape; soc_set_tuning overrides (measured on C2 and on the 256^3-root hierarchy)
    A.T = 512;
    A.P = (V.octree ? 8 : 4) * A.T;                  // hierarchies: one chunk per brick queue (measured)
    A.KCAP = V.octree ? 32 : 48;                     // measured on C2 / on the 256^3-root hierarchy
    A.FTH = V.octree ? 16 : 24;
    A.CAP = 6144;                                    // with P = 4096: 48 KB of LDS, three workgroups per CU (measured)
    A.TAIL = 0;
    if (tune.TAIL > 0) A.TAIL = tune.TAIL;
    if (tune.T > 0)    A.T = tune.T;
    if (tune.P > 0)    A.P = tune.P;
    if (tune.KCAP > 0) A.KCAP = tune.KCAP;
    if (tune.FTH > 0)  A.FTH = tune.FTH;
    A.CTH = A.FTH;
    if (tune.CTH > 0)  A.CTH = tune.CTH;
    if (tune.CAP > 0)  A.CAP = tune.CAP;
    if (A.T < 64 || A.T > 1024 || (A.T & 63) || A.P < 1 || A.P > SOC_LBRICK_PMAX || A.KCAP < 1) return hipErrorInvalidValue;
    A.EQ = 2;
    // hierarchies whose Index() is evaluated in double: the walk on brick-local hierarchies (soc_ltree.h), unless the
    // grid does not allow it (per-cell opacities, more than 8 levels, coordinates beyond 24 bits, a root cell whose
    // subtree exceeds the brick) or soc_set_tuning("global_tree", 1) asks for the older form
    if (V.octree && V.dbl && !V.abu && !tune.global_tree && G.LEVELS <= 8
        && ((long long)std::max(G.NX, std::max(G.NY, G.NZ)) << (G.LEVELS - 1)) < (1LL << 24)
        && std::max(G.NX, std::max(G.NY, G.NZ)) < 4096) {        // (root-cell numbers from 24-bit multiplies: SOC_MAD24)
        // cells per brick: what lets two workgroups share a CU's 160 KB of LDS (8 B per cell, 12 B with the INT tally, + 9 KB)
        // (rays: 4 B per cell, twice the cells in the same LDS)
        // (the vector sums of SAVE_INTENSITY 2: 24 B per cell)
        // WITH_ALI (every launch a SimRAM_CL one with the XAB tally): 8 B per cell more for XAB and the cell numbers
        // INT tally with one weight per group of launches (an absorbed-file sweep: the source blocks of ONE frequency): the INT-only form
        bool int_only = !sca && (V.wint == 1);
        for (int l = 0; l < nlaunch && int_only; l++)
            for (int m = 0; m < l; m++) if (Sin[m].INT == Sin[l].INT && Sin[m].TW != Sin[l].TW) { int_only = false;  break; }
        bool ali = !sca && (V.wint != 2);
        for (int l = 0; l < nlaunch; l++) ali = ali && (Sin[l].SOURCE == SOC_SOURCE_CL) && (Sin[l].XAB != nullptr) && (Sin[l].XAB == Sin[0].XAB);
        if (ali) int_only = false;
        const int capl = (tune.CAP > 0) ? tune.CAP : (sca ? 17408 : (ali ? (V.wint ? 3456 : 4352) : (V.wint == 2 ? 2944 : (V.wint && !int_only) ? 5888 : 8704)));
        if (capl < 8 || capl > 36864) return hipErrorInvalidValue;
        const hipError_t e = soc_lb_build(device, G, capl, st, tune.verbose != 0);
        if (e == hipSuccess) {
            const SocLBricksDev &lb = g_lb[device];
            A.LT = 1;  A.EQ = 3;
            A.ali = ali ? 1 : 0;
            A.int_only = int_only ? 1 : 0;
            for (int l = 0; l < nlaunch; l++) if (Sin[l].ROISAVE && Sin[l].ROI) A.roi_on = 1;
            if (A.roi_on) {
                for (int l = 0; l < nlaunch; l++) if (!(Sin[l].ROISAVE && Sin[l].ROI) || Sin[l].MIRROR) return hipErrorNotSupported;      // one record, no reflecting faces
                A.EQ = 4;                                                    // + the queue of packets that have just stepped into ROI
            }
            A.NBX = A.NBY = A.NBZ = 0;
            A.NB = lb.NB;
            A.CAP = (lb.max_slots + 63) & ~63;                           // slots in LDS
            A.lbr = lb.lbr;  A.btree = lb.btree;  A.bcell = lb.bcell;  A.bbase = lb.bbase;  A.rbrick = lb.rbrick;
            int k = 1;
            while ((1 << k) <= std::max(G.NX, std::max(G.NY, G.NZ))) k++;
            A.sib_thr = ldexpf(1.0f, k - 29);
            A.kexp = k - 30;
            A.slow_every = tune.slow_every;
            A.P = (tune.P > 0) ? tune.P : 16384;
            A.KCAP = (tune.KCAP > 0) ? tune.KCAP : 64;
            A.FTH = (tune.FTH > 0) ? tune.FTH : 16;
            A.CTH = (tune.CTH > 0) ? tune.CTH : 8;
            // the tail of a long chunk: 32 of 64 lanes out of work (measured: 24 ... 32 +2 %, 48 +1 %, 56 0; 65 = never)
            A.TAIL = (tune.TAIL > 0) ? tune.TAIL : 32;
            // short brick queues wait (soc_brick_scan): 4096 = 8 packets per lane, measured on config 3 (1024 ... 16384; +5 % point source,
            // +9 % diffuse emission against no parking); soc_set_tuning("park_below", 1) = never
            A.PARK = (tune.park > 0) ? tune.park : 4096;
        } else if (e != hipErrorNotSupported) {
            return e;
        }
    }
    if ((sca || V.wint == 2) && !A.LT) return hipErrorNotSupported;
    if (!A.LT) for (int l = 0; l < nlaunch; l++) if (Sin[l].ROISAVE) return hipErrorNotSupported;      // region-of-interest records: brick-local hierarchies only
    if (!A.LT && (A.T > 512 || A.P > SOC_BRICK_PMAX || A.CAP < 8 || A.CAP > (1 << SOC_SLOT_BITS))) return hipErrorInvalidValue;
    if (A.LT) {
        // set above
    } else
    if (V.octree) {
        if (G.LEVELS > 15) return hipErrorNotSupported;                  // the level shares a packet word with slot and launch
        BCHK(soc_oct_build(device, G, A.CAP, st, tune.verbose != 0));
        const SocOctBricks &ob = g_ob[device];
        A.NBX = A.NBY = A.NBZ = 0;
        A.NB = ob.NB;
        A.DS = ob.DS;  A.bcell = ob.bcell;  A.bbase = ob.bbase;
        int k = 1;
        while ((1 << k) <= std::max(G.NX, std::max(G.NY, G.NZ))) k++;
        A.sib_thr = ldexpf(1.0f, k - 29);
    } else {
        A.NBX = (G.NX + B - 1) / B;  A.NBY = (G.NY + B - 1) / B;  A.NBZ = (G.NZ + B - 1) / B;
        A.NB = A.NBX * A.NBY * A.NBZ;
        if (A.NB > (1 << 18)) return hipErrorNotSupported;
    }

    SocSimPack K{};
    uint32_t count = 0;
    for (int l = 0; l < nlaunch; l++) {
        const SocSim &S = Sin[l];
        if (S.BATCH >= (1 << 24)) return hipErrorNotSupported;       // III shares a word with the scattering count
        uint32_t c = S.gid_count;
        if ((S.SOURCE == 1 || S.SOURCE == SOC_SOURCE_HP) && !tune.oversub) {
            const long long lim = 8LL * 2 * ((long long)G.NX * G.NY + (long long)G.NY * G.NZ + (long long)G.NZ * G.NX);
            if ((long long)S.gid0 >= lim) c = 0;
            else if ((long long)S.gid0 + c > lim) c = (uint32_t)(lim - S.gid0);
        }
        if (S.SOURCE == 3) {                                          // 100 work items per surface element of the loaded record (kernel_ASOC.c:97-105)
            if (S.ROILOAD < 1 || !S.ROI) return hipErrorNotSupported;
            const long long lim = 100LL * S.ROILOAD;
            if ((long long)S.gid0 >= lim) c = 0;
            else if ((long long)S.gid0 + c > lim) c = (uint32_t)(lim - S.gid0);
        }
        if (S.SOURCE == SOC_SOURCE_CL) {                              // work items beyond the cells do nothing (kernel_ASOC.c:1273)
            if ((long long)S.gid0 >= G.CELLS) c = 0;
            else if ((long long)S.gid0 + c > G.CELLS) c = (uint32_t)(G.CELLS - S.gid0);
            if (S.USE_EMWEIGHT == 2) return hipErrorNotSupported;
            if (S.XAB && !A.ali) return hipErrorNotSupported;
        }
        if (S.BATCH <= 0 && S.SOURCE != SOC_SOURCE_CL) c = 0;
        if (c == 0) continue;                                         // nothing to do for this launch
        if ((unsigned long long)count + c > 0x7fffffffull) return hipErrorNotSupported;
        K.S[K.n] = S;
        K.S[K.n].gid_count = c;
        K.first[K.n] = count;
        count += c;
        K.n++;
    }
    if (K.n == 0) { if (passes_out) *passes_out = 0;  return hipSuccess; }
    for (int l = K.n; l <= SOC_MAXLAUNCH; l++) K.first[l] = count;
    // several launches with per-cell opacities: their OPT arrays are slots of one buffer (soc_capi.hip)
    A.opt_stride = 0;
    if (V.abu && K.n > 1) {
        A.opt_stride = (long long)(K.S[1].OPT - K.S[0].OPT);
        for (int l = 1; l < K.n; l++)
            if ((long long)(K.S[l].OPT - K.S[0].OPT) != l * A.opt_stride) return hipErrorInvalidValue;
    }
    // launches with the INT tally: brick queues per launch, so that a workgroup's LDS tallies belong to one launch
    // (launches that tally into ONE INT array -- the source blocks of one frequency, soc_batch_begin_shared_int -- share the queues)
    int ngrp = 0;
    for (int l = 0; l < K.n; l++) {
        int g = 0;
        while (g < ngrp && K.S[K.gfirst[g]].INT != K.S[l].INT) g++;
        if (g == ngrp) K.gfirst[ngrp++] = l;
        K.grp[l] = g;
    }
    if (!V.wint) { ngrp = 1;  for (int l = 0; l < K.n; l++) K.grp[l] = 0; }
    A.NBQ = (ngrp > 1) ? A.NB * ngrp : A.NB;
    if ((long long)A.NB * ngrp > (1 << 20)) return hipErrorNotSupported;
    const int NQ = A.NBQ + A.EQ * K.n + 1;
    // packets in flight: `population` of them (0: all work items at once); the other work items are admitted, in
    // order, as earlier ones finish.  Queues, descriptors and the grids of the passes are sized for that many.
    if (population < 0) {
        // measured: about 5300 packets per brick on Cartesian grids (C2, 512 bricks: 2.7e6 -> 8.5e8 packets/s, 2.1e6 and
        // 3.1e6 -> 8.1e8; 256^3, 4096 bricks: 2.6e7 -> 1.09e11 steps/s, 2.7e6 -> 8.1e10), 2.6e7 on the 256^3-root
        // hierarchy (1.3e7 -> 3.7e10 steps/s, 5.0e7 -> 4.0e10)
        // brick-local hierarchies: 3e8 (256^3 roots, 4 levels, 8194 bricks: 5.0e7 -> 4.6e10 steps/s, 1.0e8 -> 5.0e10, 3.0e8 -> 5.3e10:
        // the longer a workgroup lives, the less its last iterations -- lanes running dry -- weigh)
        // (round 3, the bench's 20 launches = 3.36e8 work items: all at once 3.87e8 packets/s, capped at 3e8 3.79e8, at 2.4e8 3.79e8 -> 6e8)
        const long long p = A.LT ? 600000000LL : V.octree ? 26000000LL : std::max(2700000LL, 5300LL * A.NBQ);
        population = (int)std::min(p, 2000000000LL);
    }
    if (tune.POP > 0) population = tune.POP;
    A.target = (population > 0 && (uint32_t)population < count) ? population : (int)count;
    const uint32_t live = (uint32_t)A.target;
    const int maxdesc = (int)((live + A.P - 1) / A.P) + NQ + K.n;
    A.HS = (NQ > 4096) ? 1024 : 0;
    if (tune.HS > 0) A.HS = tune.HS;
    if (A.HS & (A.HS - 1)) return hipErrorInvalidValue;

    SocBrickBuffers &bb = g_bb[device];
    if (bb.cap_items < count) {
        BCHK(hipStreamSynchronize(st));
        BCHK(brick_alloc(&bb.pk, count));
        BCHK(brick_alloc(&bb.idq[0], count));
        BCHK(brick_alloc(&bb.idq[1], count));
        BCHK(brick_alloc(&bb.keyq, count));
        BCHK(brick_alloc(&bb.posq, count));
        bb.cap_items = count;
    }
    if (bb.cap_nq < NQ) {
        BCHK(hipStreamSynchronize(st));
        BCHK(brick_alloc(&bb.hist, NQ));
        BCHK(brick_alloc(&bb.off, NQ));
        bb.cap_nq = NQ;
    }
    if (bb.cap_desc < maxdesc) {
        BCHK(hipStreamSynchronize(st));
        BCHK(brick_alloc(&bb.desc[0], maxdesc));
        BCHK(brick_alloc(&bb.desc[1], maxdesc));
        bb.cap_desc = maxdesc;
    }
    if (sca && bb.cap_park < count) {
        BCHK(hipStreamSynchronize(st));
        BCHK(brick_alloc(&bb.park, count));
        bb.cap_park = count;
    }
    A.park = sca ? bb.park : nullptr;
    if (sca) A.sca = *sca;
    if (!bb.pack) BCHK(brick_alloc(&bb.pack, 1));
    if (!bb.ndesc) { BCHK(brick_alloc(&bb.ndesc, 4));  BCHK(brick_alloc(&bb.total, 1));  BCHK(brick_alloc(&bb.admit, 1 + 3 * SOC_MAXLAUNCH)); }
    A.pk = bb.pk;  A.keyq = bb.keyq;  A.posq = bb.posq;  A.hist = bb.hist;  A.off = bb.off;  A.total = bb.total;
    A.admit = bb.admit;
    A.nl = K.n;
    A.first = (const uint32_t *)((const char *)bb.pack + offsetof(SocSimPack, first));

    const int BV = V.octree ? A.CAP : (1 << (3 * LB));
    const int nh = A.HS ? 2 * A.HS : NQ;
    const size_t lds_walk = A.LT ? (size_t)(BV * (sca ? 1 : (2 + (V.wint == 2 ? 4 : (V.wint && !A.int_only) ? 1 : 0) + (A.ali ? 2 : 0))) + ((nh + 3) & ~3) + 4 + 4 * SOC_MAXLAUNCH) * 4
                                 : (size_t)(BV * (1 + (V.wint ? 1 : 0)) + nh + 2 + 3 * SOC_MAXLAUNCH + SOC_MAXL + A.P) * 4;
    const size_t lds_ev = (size_t)(nh + 4 + SOC_MAXL) * 4;
    const size_t lds = lds_walk > lds_ev ? lds_walk : lds_ev;
    if (lds > 160 * 1024) return hipErrorNotSupported;
    const int vkey = (V.abu ? 2 : 0) | (V.wint ? 1 : 0);
    int kind = (K.S[0].SOURCE == SOC_SOURCE_CL) ? 2 : (K.S[0].SOURCE == SOC_SOURCE_HP) ? 1 : 0;
    bool all_bg = (kind == 0);
    for (int l = 0; l < K.n; l++) {
        const int kl = (K.S[l].SOURCE == SOC_SOURCE_CL) ? 2 : (K.S[l].SOURCE == SOC_SOURCE_HP) ? 1 : 0;
        if (kl != kind) {                                             // launches of several kinds: brick-local hierarchies only, where the walk
            if (!A.LT) return hipErrorInvalidValue;                   // takes the kind from the launch (soc_capi.hip sees to it elsewhere)
            kind = 4;
            break;
        }
        all_bg = all_bg && (K.S[l].SOURCE == 1);
    }
    if (kind != 4 && all_bg && !tune.nolean) kind = 3;  // background packets only: the lean kernel
    if (!A.LT) for (int l = 0; l < K.n; l++) if (K.S[l].MIRROR) return hipErrorNotSupported;      // reflecting faces: the event workgroups of brick-local hierarchies only
    const int slices = (A.P + A.T - 1) / A.T;
    const int nev = ((int)((live + A.P - 1) / A.P) + (A.EQ + 1) * K.n) * slices;

    BCHK(hipMemcpyAsync(bb.pack, &K, sizeof(SocSimPack), hipMemcpyHostToDevice, st));
    BCHK(hipStreamSynchronize(st));                                   // K is on this stack
    soc_brick2_init<<<(max(count, (uint32_t)NQ) + 255) / 256, 256, 0, st>>>(bb.pack, A, count, bb.idq[0], bb.desc[0], bb.ndesc, bb.hist);
    BCHK(hipGetLastError());
    int passes = 0, total = 1;
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_last = t_begin;
    // The grids of a pass are sized for the packets that can be in the queues.  Once every work item has been admitted that number only
    // falls, and the host learns it every 64 passes anyway: the grids follow it down -- the long tail of a sweep (the few work items with
    // the longest chains of packets, scatterings, rays) otherwise launches the workgroups of the full population pass after pass, a few
    // hundred thousand of them to find nothing to do (0.4 ms per pass on config 3).
    long long live_now = (long long)live;
    while (total > 0) {
        const int chunks_now = (int)((live_now + A.P - 1) / A.P);
        const int maxdesc_now = std::min<long long>(maxdesc, chunks_now + std::min<long long>(NQ, live_now) + K.n);
        const int nev_now = std::min<long long>(nev, (long long)(chunks_now + (A.EQ + 1) * K.n) * slices);
        for (int k = 0; k < 64; k++, passes++) {
            const int c = k & 1;
            A.idq = bb.idq[c];  A.idq_next = bb.idq[1 - c];
            A.desc = bb.desc[c];  A.ndesc = bb.ndesc + c;
            A.desc_next = bb.desc[1 - c];  A.ndesc_next = bb.ndesc + (1 - c);
            if (sca)         BCHK(soc_lray_launch_pass(maxdesc_now + nev_now, A.T, lds, st, G, bb.pack, A, maxdesc_now, slices));
            else if (A.LT)   soc_lbrick_launch_pass(A.int_only ? 3 : V.wint, kind, maxdesc_now + nev_now, A.T, lds, st, G, bb.pack, A, maxdesc_now, slices);
            else if (!V.octree) soc_brick_launch_pass<false, false>(vkey, kind, maxdesc_now + nev_now, A.T, lds, st, G, bb.pack, A, maxdesc_now, slices);
            else if (!V.dbl) soc_brick_launch_pass<true, false>(vkey, kind, maxdesc_now + nev_now, A.T, lds, st, G, bb.pack, A, maxdesc_now, slices);
            else             soc_brick_launch_pass<true, true>(vkey, kind, maxdesc_now + nev_now, A.T, lds, st, G, bb.pack, A, maxdesc_now, slices);
            SocBrickArgs Q = A;                           // the sort sees NQ - 1 live queues; the last one = finished
            Q.NB = NQ - 1;
            Q.ev_brick = A.NBQ;
            soc_brick_scan<<<1, 1024, 0, st>>>(Q);
            soc_brick_scatter<<<maxdesc_now + 16 * K.n, SOC_BRICK_T, 0, st>>>(Q, maxdesc_now);
        }
        BCHK(hipGetLastError());
        int admitted = 0;
        BCHK(hipMemcpyAsync(&total, bb.total, sizeof(int), hipMemcpyDeviceToHost, st));
        BCHK(hipMemcpyAsync(&admitted, bb.admit, sizeof(int), hipMemcpyDeviceToHost, st));
        BCHK(hipStreamSynchronize(st));
        if ((uint32_t)admitted >= count) live_now = std::min<long long>(live_now, std::max(total, 1));
        if (tune.verbose > 1) {                                       // the course of a sweep: packets in the queues every 64 passes
            const auto t_now = std::chrono::steady_clock::now();
            fprintf(stderr, "soc_brick: pass %6d  packets in queues %10d  %8.2f ms per pass\n", passes, total,
                    std::chrono::duration<double, std::milli>(t_now - t_last).count() / 64.0);
            t_last = t_now;
        }
        if (passes > 4000000) return hipErrorUnknown;                 // cannot happen: every pass retires work
    }
    if (passes_out) *passes_out = passes;
    if (form_out) *form_out = A.LT ? 3 : (V.octree ? 2 : 1);
    return hipSuccess;
}
