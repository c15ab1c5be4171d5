// soc_capi.hip -- host side of libsoc_hip.so: the C ABI declared in include/soc_hip.h.
// Owns device memory, validates the model on the host before anything reaches a kernel,
// derives the per-launch seed constants and dispatches the kernels of soc_kernels.hip.
#include "../../include/soc_hip.h"
#include "soc_dev.h"
#include "soc_rng.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>


struct soc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    // model
    bool have_grid = false;
    SocGrid G{};
    float *dDENS = nullptr;
    int   *dPAR = nullptr;
    int64_t npar = 0;
    // tables / per-frequency data
    float *dCSC = nullptr, *dDSC = nullptr;
    int    BINS = 0;
    bool   have_dsc = false;
    // scattered-light view (soc_sca_*)
    SocSca view{};
    float4 *dODIR = nullptr, *dORA = nullptr, *dODE = nullptr;
    float  *dOUT = nullptr;
    bool    own_OUT = false, have_view = false;
    float  ABS = 0.0f, SCA = 0.0f;
    bool   have_optical = false;
    float2 *dOPT = nullptr;
    float *dEMIT = nullptr, *dEMWEI = nullptr, *dXAB = nullptr;
    float *dINTV = nullptr;            // -D SAVE_INTENSITY=2: INTX | INTY | INTZ, CELLS floats each (with_int == 2)
    size_t intv_cells = 0;
    int   *dEMINDEX = nullptr;
    bool   have_emit = false, have_emindex = false, with_ali = false;
    float *dHPBG = nullptr, *dHPBGP = nullptr;    // Healpix sky of the current frequency (NSIDE 64)
    float *dABU = nullptr, *dAF = nullptr;        // abundances [CELLS, NDUST] (or [CELLS]), cross sections of the frequency
    int    abu_ndust = 0, abu_single = 0;
    int    map_level_threshold = 0;    // -D LEVEL_THRESHOLD (soc_set_map_threshold)
    int    map_interpolation = 0;      // -D MAP_INTERPOLATION (soc_set_map_interpolation)
    int    map_roi_on = 0, map_roi[6] = { 0, 0, 0, 0, 0, 0 };   // -D ROI_MAP (soc_set_map_roi)
    float  cr_rate = 0.0f;             // -D CR_HEATING_RATE with -D CR_HEATING=1 (soc_set_cr_heating); 0 = off
    bool   opt_half = false;          // -D OPT_IS_HALF: OPT rounded through fp16 (soc_set_opt_half)
    bool   opt_from_abu = false;       // dOPT and dAF hold the current frequency's soc_set_optical_abu values
    int    msf_ndust = 1;              // > 1: -D WITH_MSF, dCSC/dDSC hold [msf_ndust][BINS] (soc_set_scatter_tables)
    int    step_weight = 0;            // -D STEP_WEIGHT (soc_set_step_weight)
    float  sw_a = 0.0f, sw_b = 0.0f;
    size_t abu_cells = 0;
    SocRoi roi{};                                 // region of interest (host copy of *dRoi)
    SocRoi *dRoi = nullptr;
    float *dRoiSave = nullptr, *dRoiLoad = nullptr;
    size_t roi_save_n = 0, roi_load_cap = 0;
    bool   have_hpbg = false, hpbg_weighted = false;
    // tallies
    float *dTABS = nullptr, *dINT = nullptr;
    bool   own_TABS = false, own_INT = false;
    // point-source scratch: slot 0 for immediate launches, one slot per deferred launch of a batch
    struct SrcBuf {
        float4 *PSPOS = nullptr;
        float  *PS = nullptr, *XPS_AREA = nullptr;
        int    *XPS_NSIDE = nullptr, *XPS_SIDE = nullptr;
        int     cap = 0;
    } src[SOC_MAXLAUNCH];
    // deferred launches (soc_batch_begin .. soc_batch_end): executed together in one brick sweep
    bool   batching = false;
    int    batch_max = 4;
    std::vector<SocSim> pending;
    bool   pending_sca = false;                               // the deferred launches are scattered-light ones (rays; soc_sca_sim_*)
    float *dDSCslot[SOC_MAXLAUNCH] = {};                      // their discrete scattering functions
    int    dsc_slot_bins = 0;
    float *dOUTslots = nullptr;                               // soc_sca_batch_images: several images, one per frequency of a batch
    int    out_slots = 0, out_slot_cur = 0;
    size_t out_slot_pixels = 0;
    float *dCSCslot[SOC_MAXLAUNCH] = {};
    float2 *dOPTslots = nullptr;                  // [SOC_OPT_SLOTS][CELLS] per-cell opacities of deferred launches (one buffer: the sweep strides through it)
    float *dHPslots = nullptr;                    // [SOC_MAXLAUNCH][2][49152] Healpix skies of deferred SimRAM_HP launches
    // EMIT | EMWEI copies of deferred SimRAM_CL launches and INT tallies of deferred launches (soc_batch_read_int): one buffer per
    // launch slot, allocated when a batch first reaches that slot (128 slots of a 5e7-cell model up front would be 50 GB)
    float *dEMITslot[SOC_MAXLAUNCH] = {};
    float *dINTslot[SOC_MAXLAUNCH] = {};
    size_t intslot_cells = 0;
    unsigned long long emit_gen = 0;              // bumped by soc_set_emission: launches deferred without a change in between share one copy
    unsigned long long emit_slot_gen = 0;
    int    emit_slot_last = -1;
    int    int_slots_done = 0;                    // launches of the last executed sweep whose INT can be read
    bool   batch_keep_int = false;                // soc_batch_begin_int: deferred launches keep their own INT tally
    bool   batch_share_int = false;               // soc_batch_begin_shared_int: deferred launches tally into the handle's INT together
    bool   batch_group_int = false;               // soc_batch_begin_int_groups: the launches between two soc_batch_next_int calls share an INT tally
    bool   int_group_open = false;                // ... and the current group has its tally
    size_t emitslot_cells = 0;
    size_t optslot_cells = 0;
    int    csc_slot_bins = 0;
    // rng
    uint64_t *dSeedTab = nullptr;
    unsigned long long *dStats = nullptr;
    unsigned long long ray_steps = 0;                        // cell steps of the rays of the scattered-light sweeps, as of the last soc_stats
    // features
    int with_int = 0, ps_method = 0, use_emweight = 0, mirror = 0;
    // execution
    int exec_mode = -1, brick_log2 = 4, last_passes = 0, last_form = 0;
    SocBrickTune tune{};
    // equilibrium temperature / emission (soc_emit.hip)
    float *dT = nullptr, *dTTT = nullptr, *dEbuf = nullptr, *dEF = nullptr;
    int    ttt_cap = 0, ef_cap = 0;
    size_t ebuf_cap = 0;
    bool   have_T = false;
    // map making (soc_map.hip)
    float *dMapEmit = nullptr, *dMap = nullptr, *dMapTau = nullptr;
    size_t map_cap = 0, mapemit_cap = 0;
    // A2E
    int a2e_NE = 0, a2e_NFREQ = 0, a2e_npair = 0, a2e_cap = 0, a2e_noIw = 0;
    float *aIw = nullptr, *aTdown = nullptr, *aEA = nullptr, *aAF = nullptr, *aABS = nullptr, *aEMIT = nullptr;
    float *aAll = nullptr, *aSum = nullptr;                  // soc_a2e_resident_*: absorptions of all cells, emission summed over the sizes
    int64_t a2e_cells = 0;  int a2e_res_nfreq = 0;
    int   *aFirst = nullptr, *aLast = nullptr, *aIwOff = nullptr, *aDst = nullptr, *aIbeg = nullptr;
};

static std::string g_create_err;

static int fail(soc_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else   g_create_err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), SOC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
    } while (0)

template <typename T>
static hipError_t dev_alloc(T **p, size_t n)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    return hipMalloc((void **)p, (n ? n : 1) * sizeof(T));
}

// number of floats of the scattered-light image: NDIR maps of NPIX_X x NPIX_Y pixels, or one Healpix map
static size_t view_pixels(const soc_ctx *c)
{
    if (!c->have_view) return 0;
    if (c->view.NDIR < 0) return (size_t)12 * c->view.NDIR * c->view.NDIR;
    return (size_t)c->view.NDIR * c->view.NPIX_X * c->view.NPIX_Y;
}

// Hierarchies on which the brick sweep keeps the brick's cells in LDS (soc_brick.hip, soc_ltree.h): there a lone
// launch with enough work items pays too -- also with the INT tally, which lives in LDS beside TABS.
static bool lt_capable(const soc_ctx *c, bool abu)
{
    const SocGrid &G = c->G;
    const int n = G.NX > G.NY ? (G.NX > G.NZ ? G.NX : G.NZ) : (G.NY > G.NZ ? G.NY : G.NZ);
    return G.LEVELS > 1 && G.LEVELS <= 8 && G.NX > ((G.LEVELS < 3) ? 399 : 100) && !abu && !c->tune.global_tree
           && (((long long)n << (G.LEVELS - 1)) < (1LL << 24)) && n < 4096;
}
// ... the launches of the scattered-light kernels (alone, or those of a batch together) go to the sweep of rays.  Measured on the 256^3-root
// hierarchy (tools/exp_sca.py): 1.0e6 work items 0.47x the direct kernel, 3.1e6 0.92x, 8.4e6 1.3x, 5.0e7 1.6x (best direct launch shape), 32
// launches of 3.1e6 in one batch 4.4x
#define SOC_SCA_RAYS_LAUNCH 4000000
#define SOC_LT_LONE_LAUNCH 1000000                           // work items from which a lone launch goes to the sweep there

// Execute the launches deferred since soc_batch_begin: one brick sweep for all of them.
static int flush_pending(soc_ctx *c)
{
    if (c->pending.empty()) return SOC_OK;
    std::vector<SocSim> todo;
    todo.swap(c->pending);
    SocVariant V;
    V.octree = c->G.LEVELS > 1;  V.dbl = c->G.NX > ((c->G.LEVELS < 3) ? 399 : 100);
    if (c->pending_sca) {
        // deferred launches of the scattered-light kernels: one sweep of rays for all of them; where the sweep does not apply
        // (it did when they were deferred: the grid has not changed since) each runs through the direct kernel
        c->pending_sca = false;
        V.abu = 0;  V.wint = 0;
        HIPCHK(c, hipSetDevice(c->device));
        SocSca X = c->view;
        X.kind = todo[0].SCAKIND - 1;  X.DSC = todo[0].DSC;  X.OUT = todo[0].OUT;
        unsigned long long items = 0;
        for (const SocSim &S1 : todo) items += S1.gid_count;
        hipError_t e = hipErrorNotSupported;
        if (c->exec_mode == 1 || items >= SOC_SCA_RAYS_LAUNCH)     // (too few rays to fill the brick queues: the direct kernel, launch by launch)
            e = soc_brick_run_pb(c->device, c->G, todo.data(), (int)todo.size(), V, c->brick_log2, -1, c->tune, c->stream, &c->last_passes, &c->last_form, &X);
        if (e == hipErrorNotSupported) {
            for (SocSim &S1 : todo) {
                X.kind = S1.SCAKIND - 1;  X.DSC = S1.DSC;  X.OUT = S1.OUT;
                if (S1.SOURCE == SOC_SOURCE_CL) S1.SOURCE = 2;
                HIPCHK(c, soc_launch_sca(c->G, S1, X, V, c->stream));
            }
            return SOC_OK;
        }
        if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "sweep of the rays of %d deferred scattered-light launches failed: %s", (int)todo.size(), hipGetErrorString(e));
        return SOC_OK;
    }
    V.abu = todo[0].OPT != nullptr;                          // what makes a launch deferrable (see soc_sim_pb)
    V.wint = ((c->batch_keep_int || c->batch_share_int) && c->with_int) ? c->with_int : 0;
    HIPCHK(c, hipSetDevice(c->device));
    if (V.octree && todo.size() == 1 && c->exec_mode < 0 && !(lt_capable(c, V.abu != 0) && todo[0].gid_count >= SOC_LT_LONE_LAUNCH)) {
        // a single launch on a hierarchy: the direct kernel is as fast (1.9e10 vs 2.0e10 steps/s at 256^3, 4 levels)
        c->last_passes = 0;
        if (todo[0].SOURCE == SOC_SOURCE_HP) {
            SocSim S1 = todo[0];
            S1.SOURCE = 1;
            HIPCHK(c, soc_launch_sim_hp(c->G, S1, V, c->stream));
        } else if (todo[0].SOURCE == SOC_SOURCE_CL) {
            SocSim S1 = todo[0];
            S1.SOURCE = 2;
            HIPCHK(c, soc_launch_sim_cl(c->G, S1, V, c->stream));
        } else {
            HIPCHK(c, soc_launch_sim_pb(c->G, todo[0], V, c->stream));
        }
        return SOC_OK;
    }
    // packets in flight: chosen by the sweep from the number of bricks (-1)
    hipError_t e = soc_brick_run_pb(c->device, c->G, todo.data(), (int)todo.size(), V, c->brick_log2, -1, c->tune, c->stream, &c->last_passes, &c->last_form);
    if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "brick sweep of %d deferred launches failed: %s", (int)todo.size(), hipGetErrorString(e));
    return SOC_OK;
}
#define FLUSH(c)                                    \
    do {                                            \
        int f_ = flush_pending(c);                  \
        if (f_) return f_;                          \
    } while (0)

#pragma GCC visibility push(default)
extern "C" {

const char *soc_version(void) { return "soc_hip 0.1 (gfx950)"; }

const char *soc_last_error(const soc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

static int g_handles[64];              // live handles per GPU: the brick scratch of a GPU goes with the last one

int soc_create(int device, soc_ctx **out)
{
    if (!out) return fail(nullptr, SOC_ERR_ARG, "soc_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        // two HIP runtimes in one process (torch ships its own libamdhip64.so): the one initialised second finds no device
        std::string first, second;
        if (FILE *fp = fopen("/proc/self/maps", "r")) {
            char line[1024];
            while (fgets(line, sizeof line, fp)) {
                const char *q = strstr(line, "libamdhip64");
                if (!q) continue;
                const char *b = strchr(line, '/');
                if (!b) continue;
                std::string path(b);
                while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) path.pop_back();
                if (first.empty()) first = path;
                else if (path != first && second.empty()) second = path;
            }
            fclose(fp);
        }
        if (!second.empty())
            return fail(nullptr, SOC_ERR_HIP, "soc_create: no HIP device available (%s): two HIP runtimes are loaded in this process (%s and %s) and the "
                        "one initialised second finds no device -- load torch's first (import torch before libsoc_hip.so is opened; soc_amd.lib does)",
                        hipGetErrorString(e), first.c_str(), second.c_str());
        return fail(nullptr, SOC_ERR_HIP, "soc_create: no HIP device available (%s)", hipGetErrorString(e));
    }
    if (device < 0 || device >= ndev)
        return fail(nullptr, SOC_ERR_ARG, "soc_create: device %d out of range (0..%d)", device, ndev - 1);
    soc_ctx *c = new soc_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        int r = fail(nullptr, SOC_ERR_HIP, "soc_create: %s", hipGetErrorString(e));
        delete c;
        return r;
    }
    c->stream = c->own_stream;
    if (device < 64) g_handles[device]++;
    // seed tables: T[k][b] = G^(b*256^k) mod M with G = A^(2^38) mod M  (soc_rng.h)
    std::vector<uint64_t> tab(1024);
    soc_build_seed_table(tab.data());
    if ((e = hipMalloc((void **)&c->dSeedTab, 1024 * sizeof(uint64_t))) != hipSuccess ||
        (e = hipMemcpy(c->dSeedTab, tab.data(), 1024 * sizeof(uint64_t), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMalloc((void **)&c->dStats, 4 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemsetAsync(c->dStats, 0, 4 * sizeof(unsigned long long), c->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
        int r = fail(nullptr, SOC_ERR_HIP, "soc_create: %s", hipGetErrorString(e));
        soc_destroy(c);
        return r;
    }
    *out = c;
    return SOC_OK;
}

void soc_destroy(soc_ctx *c)
{
    if (!c) return;
    (void)flush_pending(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &b : c->src) {
        void *sb[] = { b.PSPOS, b.PS, b.XPS_AREA, b.XPS_NSIDE, b.XPS_SIDE };
        for (void *q : sb) if (q) (void)hipFree(q);
    }
    for (float *q : c->dCSCslot) if (q) (void)hipFree(q);
    for (float *q : c->dDSCslot) if (q) (void)hipFree(q);
    if (c->dOUTslots) (void)hipFree(c->dOUTslots);
    for (float *q : c->dEMITslot) if (q) (void)hipFree(q);
    for (float *q : c->dINTslot) if (q) (void)hipFree(q);
    void *bufs[] = { c->dHPslots, c->dOPTslots, c->dABU, c->dAF, c->dRoi, c->dRoiSave, c->dRoiLoad, c->dDENS, c->dPAR, c->dCSC, c->dDSC, c->dOPT, c->dEMIT, c->dEMWEI, c->dXAB, c->dINTV, c->dEMINDEX, c->dSeedTab, c->dStats, c->dODIR, c->dORA, c->dODE, c->dHPBG, c->dHPBGP, c->dT, c->dTTT, c->dEbuf, c->dEF, c->dMapEmit, c->dMap, c->dMapTau,
                     c->aIw, c->aTdown, c->aEA, c->aAF, c->aABS, c->aEMIT, c->aAll, c->aSum, c->aFirst, c->aLast, c->aIwOff, c->aDst, c->aIbeg };
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (c->own_TABS && c->dTABS) (void)hipFree(c->dTABS);
    if (c->own_INT && c->dINT) (void)hipFree(c->dINT);
    if (c->own_OUT && c->dOUT) (void)hipFree(c->dOUT);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->device >= 64 || --g_handles[c->device] <= 0) soc_brick_release(c->device);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int soc_set_stream(soc_ctx *c, void *hip_stream)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return SOC_OK;
}

int soc_set_grid(soc_ctx *c, int NX, int NY, int NZ, int LEVELS, const int32_t *LCELLS, const float *DENS)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!LCELLS || !DENS) return fail(c, SOC_ERR_ARG, "soc_set_grid: NULL array");
    if (NX < 1 || NY < 1 || NZ < 1 || NX > 9999) return fail(c, SOC_ERR_ARG, "soc_set_grid: bad dimensions %d %d %d", NX, NY, NZ);
    if (LEVELS < 1 || LEVELS > SOC_MAXL) return fail(c, SOC_ERR_ARG, "soc_set_grid: LEVELS=%d unsupported (1..%d)", LEVELS, SOC_MAXL);
    const int64_t nxyz = (int64_t)NX * NY * NZ;
    if (nxyz > 2147483647LL || LCELLS[0] != nxyz)
        return fail(c, SOC_ERR_ARG, "soc_set_grid: LCELLS[0]=%d does not match NX*NY*NZ=%lld", LCELLS[0], (long long)nxyz);
    SocGrid G{};
    G.NX = NX; G.NY = NY; G.NZ = NZ; G.LEVELS = LEVELS; G.NXYZ = (int)nxyz;
    int64_t cells = 0;
    for (int l = 0; l < LEVELS; l++) {
        if (LCELLS[l] < 0 || (l > 0 && LCELLS[l] % 8 != 0))
            return fail(c, SOC_ERR_ARG, "soc_set_grid: LCELLS[%d]=%d is not a whole number of octets", l, LCELLS[l]);
        G.OFF[l] = (int)cells;
        G.LCELLS[l] = LCELLS[l];
        cells += LCELLS[l];
        if (cells > 2147483647LL) return fail(c, SOC_ERR_ARG, "soc_set_grid: more than 2^31-1 cells");
    }
    G.CELLS = (int)cells;
    // validate links on the host: every parent must point at an aligned octet of the next level
    for (int l = 0; l < LEVELS; l++) {
        const float *d = DENS + G.OFF[l];
        const int nchild = (l + 1 < LEVELS) ? LCELLS[l + 1] : 0;
        for (int i = 0; i < LCELLS[l]; i++) {
            float v = d[i];
            if (v != v) return fail(c, SOC_ERR_ARG, "soc_set_grid: NaN density at level %d cell %d", l, i);
            if (!(v > 0.0f)) {
                uint32_t bits;
                memcpy(&bits, &v, 4);
                int first = (int)(bits ^ 0x80000000u);
                if (first < 0 || first % 8 != 0 || first + 8 > nchild)
                    return fail(c, SOC_ERR_ARG, "soc_set_grid: level %d cell %d: value %g is not a density and not a valid child link", l, i, (double)v);
            }
        }
    }
    // refused before anything of the handle changes: a refused call leaves the old grid usable
    if (c->have_grid && (int64_t)G.CELLS != c->G.CELLS && ((c->dTABS && !c->own_TABS) || (c->dINT && !c->own_INT)))
        return fail(c, SOC_ERR_STATE, "soc_set_grid: a caller-owned tally of %d cells is bound; soc_bind_tally(ctx, which, NULL, 0) first, re-bind after", c->G.CELLS);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, dev_alloc(&c->dDENS, (size_t)cells));
    HIPCHK(c, hipMemcpy(c->dDENS, DENS, (size_t)cells * 4, hipMemcpyHostToDevice));
    c->npar = cells - nxyz;
    HIPCHK(c, dev_alloc(&c->dPAR, (size_t)c->npar));
    HIPCHK(c, hipMemsetAsync(c->dPAR, 0, (size_t)(c->npar ? c->npar : 1) * 4, c->stream));
    G.DENS = c->dDENS;
    G.PAR = c->dPAR;
    if ((int64_t)G.CELLS != c->G.CELLS || !c->have_grid) {
        // tallies follow the cell count
        if (c->own_TABS || !c->dTABS) { c->dTABS = nullptr; HIPCHK(c, dev_alloc(&c->dTABS, (size_t)cells)); c->own_TABS = true; HIPCHK(c, hipMemsetAsync(c->dTABS, 0, (size_t)cells * 4, c->stream)); }
        if (c->own_INT || !c->dINT) { c->dINT = nullptr; HIPCHK(c, dev_alloc(&c->dINT, (size_t)cells)); c->own_INT = true; HIPCHK(c, hipMemsetAsync(c->dINT, 0, (size_t)cells * 4, c->stream)); }
        if (c->dOPT) { (void)hipFree(c->dOPT); c->dOPT = nullptr; }
        c->have_emit = false;
        // everything else that is sized by the cell count
        if (c->dT) { (void)hipFree(c->dT); c->dT = nullptr; }
        if (c->dXAB) { (void)hipFree(c->dXAB); c->dXAB = nullptr; }
        if (c->dINTV) {                                       // INTX, INTY, INTZ follow the cell count like TABS and INT (with_int stays 2)
            (void)hipFree(c->dINTV);  c->dINTV = nullptr;  c->intv_cells = 0;
            if (c->with_int == 2) {
                HIPCHK(c, dev_alloc(&c->dINTV, (size_t)3 * cells));
                c->intv_cells = (size_t)cells;
                HIPCHK(c, hipMemsetAsync(c->dINTV, 0, (size_t)3 * cells * 4, c->stream));
            }
        }
        if (c->dEMINDEX) { (void)hipFree(c->dEMINDEX); c->dEMINDEX = nullptr; }
        c->have_T = false;  c->with_ali = false;  c->have_emindex = false;
        c->abu_ndust = 0;  c->abu_cells = 0;
    }
    c->G = G;
    c->have_grid = true;
    soc_brick_invalidate(c->device);
    HIPCHK(c, soc_launch_parents(c->G, c->dPAR, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_set_features(soc_ctx *c, int with_int, int ps_method, int use_emweight)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!(ps_method == 0 || ps_method == 1 || ps_method == 2 || ps_method == 4 || ps_method == 5))
        return fail(c, SOC_ERR_ARG, "soc_set_features: PS_METHOD %d not supported (0,1,2,4,5)", ps_method);
    if (use_emweight < 0 || use_emweight > 2)
        return fail(c, SOC_ERR_ARG, "soc_set_features: USE_EMWEIGHT %d not supported (0,1,2)", use_emweight);
    if (with_int == 2) {                                    // SAVE_INTENSITY == 2: three more tallies (kernel_ASOC.c:604-612)
        if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_set_features: with_int 2 needs soc_set_grid first (it allocates INTX, INTY, INTZ)");
        HIPCHK(c, hipSetDevice(c->device));
        if (c->intv_cells != (size_t)c->G.CELLS) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, dev_alloc(&c->dINTV, (size_t)3 * c->G.CELLS));
            c->intv_cells = (size_t)c->G.CELLS;
            HIPCHK(c, hipMemsetAsync(c->dINTV, 0, (size_t)3 * c->G.CELLS * 4, c->stream));
        }
    }
    c->with_int = (with_int == 2) ? 2 : (with_int ? 1 : 0);
    c->ps_method = ps_method;
    c->use_emweight = use_emweight;
    return SOC_OK;
}

int soc_set_mirror(soc_ctx *c, int mask)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (mask < 0 || mask > 63) return fail(c, SOC_ERR_ARG, "soc_set_mirror: mask %d (bits x,X,y,Y,z,Z = 1,2,4,8,16,32)", mask);
    c->mirror = mask;
    return SOC_OK;
}

int soc_set_exec(soc_ctx *c, int mode, int brick_log2)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (mode < -1 || mode > 1 || brick_log2 < 2 || brick_log2 > 4)
        return fail(c, SOC_ERR_ARG, "soc_set_exec: mode %d (-1,0,1), brick_log2 %d (2..4)", mode, brick_log2);
    c->exec_mode = mode;
    c->brick_log2 = brick_log2;
    return SOC_OK;
}

int soc_set_tuning(soc_ctx *c, const char *name, int value)
{
    if (!c || !name) return SOC_ERR_ARG;
    FLUSH(c);
    if (value < 0) return fail(c, SOC_ERR_ARG, "soc_set_tuning: %s = %d (0 = built-in choice)", name, value);
    struct { const char *n; int *p; } tab[] = {
        { "threads", &c->tune.T }, { "chunk", &c->tune.P }, { "steps_per_visit", &c->tune.KCAP }, { "swap_lanes", &c->tune.FTH },
        { "climb_lanes", &c->tune.CTH }, { "brick_cells", &c->tune.CAP }, { "tail_lanes", &c->tune.TAIL }, { "park_below", &c->tune.park }, { "population", &c->tune.POP },
        { "hash_slots", &c->tune.HS }, { "global_tree", &c->tune.global_tree }, { "slow_every", &c->tune.slow_every },
        { "general_kernel", &c->tune.nolean }, { "oversubscribe", &c->tune.oversub }, { "verbose", &c->tune.verbose } };
    for (auto &t : tab)
        if (!strcmp(name, t.n)) {
            if (t.p == &c->tune.CAP && value != c->tune.CAP) soc_brick_invalidate(c->device);
            *t.p = value;
            return SOC_OK;
        }
    return fail(c, SOC_ERR_ARG, "soc_set_tuning: unknown parameter '%s'", name);
}

int soc_last_passes(soc_ctx *c) { return c ? c->last_passes : 0; }
int soc_last_form(soc_ctx *c) { return (c && c->last_passes > 0) ? c->last_form : 0; }

int soc_set_optical(soc_ctx *c, const float *ABS, const float *SCA, int ndust)
{
    if (!c) return SOC_ERR_ARG;
    if (!ABS || !SCA || ndust != 1) return fail(c, SOC_ERR_ARG, "soc_set_optical: need ABS, SCA with ndust==1 (WITH_MSF is not supported)");
    c->ABS = ABS[0];
    c->SCA = SCA[0];
    c->have_optical = true;
    return SOC_OK;
}

int soc_set_opt(soc_ctx *c, const float *OPT)
{
    if (!c) return SOC_ERR_ARG;
    // no flush: a deferred launch keeps its own copy of the opacities (soc_sim_pb)
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_set_opt: call soc_set_grid first");
    HIPCHK(c, hipSetDevice(c->device));
    c->opt_from_abu = false;
    if (!OPT) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->dOPT) { (void)hipFree(c->dOPT); c->dOPT = nullptr; }
        return SOC_OK;
    }
    if (!c->dOPT) HIPCHK(c, dev_alloc(&c->dOPT, (size_t)c->G.CELLS));
    HIPCHK(c, hipMemcpyAsync(c->dOPT, OPT, (size_t)c->G.CELLS * 8, hipMemcpyHostToDevice, c->stream));
    if (c->opt_half) HIPCHK(c, soc_launch_opt_half(c->G.CELLS, c->dOPT, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_set_opt_half(soc_ctx *c, int on)
{
    if (!c) return SOC_ERR_ARG;
    c->opt_half = on != 0;                                  // applies to the next soc_set_opt / soc_set_optical_abu
    return SOC_OK;
}

int soc_set_abundances(soc_ctx *c, int NDUST, int single, const float *ABU)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_set_abundances: call soc_set_grid first");
    HIPCHK(c, hipSetDevice(c->device));
    c->opt_from_abu = false;
    if (!ABU) {                                             // off
        c->abu_ndust = 0;
        c->abu_cells = 0;
        return SOC_OK;
    }
    if (NDUST < 1 || NDUST > 64 || (single && NDUST != 2))
        return fail(c, SOC_ERR_ARG, "soc_set_abundances: NDUST %d (1..64; the one-abundance form describes exactly two species)", NDUST);
    const size_t n = (size_t)c->G.CELLS * (single ? 1 : NDUST);
    for (size_t i = 0; i < n; i++)
        if (!std::isfinite(ABU[i])) return fail(c, SOC_ERR_ARG, "soc_set_abundances: ABU[%zu] is not finite", i);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, dev_alloc(&c->dABU, n));
    HIPCHK(c, dev_alloc(&c->dAF, (size_t)2 * NDUST));
    HIPCHK(c, hipMemcpyAsync(c->dABU, ABU, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->abu_ndust = NDUST;  c->abu_single = single ? 1 : 0;  c->abu_cells = (size_t)c->G.CELLS;
    return SOC_OK;
}

int soc_set_optical_abu(soc_ctx *c, const float *AFABS, const float *AFSCA, int ndust)
{
    if (!c) return SOC_ERR_ARG;
    // no flush: a deferred launch keeps its own copy of the opacities (soc_sim_pb)
    if (!c->abu_ndust || c->abu_cells != (size_t)c->G.CELLS) return fail(c, SOC_ERR_STATE, "soc_set_optical_abu: call soc_set_abundances (after soc_set_grid) first");
    if (!AFABS || !AFSCA || ndust != c->abu_ndust) return fail(c, SOC_ERR_ARG, "soc_set_optical_abu: need the cross sections of the %d species", c->abu_ndust);
    HIPCHK(c, hipSetDevice(c->device));
    float af[128];
    for (int d = 0; d < ndust; d++) { af[d] = AFABS[d];  af[ndust + d] = AFSCA[d]; }
    if (!c->dOPT) HIPCHK(c, dev_alloc(&c->dOPT, (size_t)c->G.CELLS));
    HIPCHK(c, hipMemcpyAsync(c->dAF, af, (size_t)2 * ndust * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));             // af is on the stack
    HIPCHK(c, soc_launch_opt(c->G.CELLS, ndust, c->abu_single, c->dABU, c->dAF, c->dOPT, c->stream));
    if (c->opt_half) HIPCHK(c, soc_launch_opt_half(c->G.CELLS, c->dOPT, c->stream));
    c->opt_from_abu = true;
    return SOC_OK;
}

int soc_read_opt(soc_ctx *c, float *OPT)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->dOPT || !OPT) return fail(c, SOC_ERR_STATE, "soc_read_opt: no per-cell opacities are set");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(OPT, c->dOPT, (size_t)c->G.CELLS * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_set_scatter_tables(soc_ctx *c, int NDUST, const float *DSC, const float *CSC, int BINS)
{
    if (!c) return SOC_ERR_ARG;
    if (!CSC || BINS < 1 || BINS > 16000) return fail(c, SOC_ERR_ARG, "soc_set_scatter_tables: need CSC and 1 <= BINS <= 16000 (got %d)", BINS);
    if (NDUST < 1 || NDUST > 64) return fail(c, SOC_ERR_ARG, "soc_set_scatter_tables: NDUST %d (1..64)", NDUST);
    HIPCHK(c, hipSetDevice(c->device));
    if (NDUST > 1 || c->msf_ndust > 1) FLUSH(c);            // a deferred launch snapshots one table only: run what is pending first
    const size_t n = (size_t)NDUST * BINS;
    if (BINS != c->BINS || NDUST != c->msf_ndust || !c->dCSC) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_alloc(&c->dCSC, n));
        HIPCHK(c, dev_alloc(&c->dDSC, n));
        c->BINS = BINS;
        c->msf_ndust = NDUST;
        c->have_dsc = false;
    }
    if (DSC) c->have_dsc = true;
    HIPCHK(c, hipMemcpyAsync(c->dCSC, CSC, n * 4, hipMemcpyHostToDevice, c->stream));
    if (DSC) HIPCHK(c, hipMemcpyAsync(c->dDSC, DSC, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // host buffer may be reused by the caller
    return SOC_OK;
}

int soc_set_scatter_table(soc_ctx *c, const float *DSC, const float *CSC, int BINS)
{
    return soc_set_scatter_tables(c, 1, DSC, CSC, BINS);
}

int soc_set_step_weight(soc_ctx *c, int mode, float SW_A, float SW_B)
{
    if (!c) return SOC_ERR_ARG;
    if (mode <= 0) { c->step_weight = 0;  c->sw_a = c->sw_b = 0.0f;  return SOC_OK; }
    if (mode > 2) return fail(c, SOC_ERR_ARG, "soc_set_step_weight: mode %d (0 off, 1 exp(-A*t), 2 B*exp(-A*t)+(1-B)*exp(-2*A*t))", mode);
    if (!(SW_A > 0.0f) || !std::isfinite(SW_A)) return fail(c, SOC_ERR_ARG, "soc_set_step_weight: SW_A=%g must be positive", (double)SW_A);
    if (mode == 2 && !(SW_B > 0.0f && SW_B < 1.0f)) return fail(c, SOC_ERR_ARG, "soc_set_step_weight: mode 2 needs 0 < SW_B < 1, got %g", (double)SW_B);
    c->step_weight = mode;  c->sw_a = SW_A;  c->sw_b = SW_B;
    return SOC_OK;
}

int soc_set_emission(soc_ctx *c, const float *EMIT, const float *EMWEI)
{
    if (!c) return SOC_ERR_ARG;
    // no flush: a deferred SimRAM_CL launch keeps its own copy of EMIT and EMWEI (soc_sim_cl)
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_set_emission: call soc_set_grid first");
    if (!EMIT) return fail(c, SOC_ERR_ARG, "soc_set_emission: EMIT is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->G.CELLS;
    if (!c->have_emit) {
        HIPCHK(c, dev_alloc(&c->dEMIT, n));
        HIPCHK(c, dev_alloc(&c->dEMWEI, n));
        HIPCHK(c, hipMemsetAsync(c->dEMWEI, 0, n * 4, c->stream));
        c->have_emit = true;
    }
    HIPCHK(c, hipMemcpyAsync(c->dEMIT, EMIT, n * 4, hipMemcpyHostToDevice, c->stream));
    if (EMWEI) HIPCHK(c, hipMemcpyAsync(c->dEMWEI, EMWEI, n * 4, hipMemcpyHostToDevice, c->stream));
    c->emit_gen++;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_set_emindex(soc_ctx *c, const int32_t *EMINDEX)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid || !EMINDEX) return fail(c, SOC_ERR_STATE, "soc_set_emindex: needs a grid and EMINDEX[CELLS]");
    for (int i = 0; i < c->G.CELLS; i++)
        if (EMINDEX[i] >= c->G.CELLS) return fail(c, SOC_ERR_ARG, "soc_set_emindex: EMINDEX[%d] = %d is not a cell", i, EMINDEX[i]);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->dEMINDEX) HIPCHK(c, dev_alloc(&c->dEMINDEX, (size_t)c->G.CELLS));
    HIPCHK(c, hipMemcpyAsync(c->dEMINDEX, EMINDEX, (size_t)c->G.CELLS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_emindex = true;
    return SOC_OK;
}

int soc_set_ali(soc_ctx *c, int with_ali)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_set_ali: call soc_set_grid first");
    HIPCHK(c, hipSetDevice(c->device));
    if (with_ali && !c->dXAB) {
        HIPCHK(c, dev_alloc(&c->dXAB, (size_t)c->G.CELLS));
        HIPCHK(c, hipMemsetAsync(c->dXAB, 0, (size_t)c->G.CELLS * 4, c->stream));
    }
    c->with_ali = with_ali != 0;
    return SOC_OK;
}

static float *tally_buf(soc_ctx *c, int which)
{
    if (which == SOC_TALLY_TABS) return c->dTABS;
    if (which == SOC_TALLY_INT) return c->dINT;
    if (which == SOC_TALLY_XAB) return c->with_ali ? c->dXAB : nullptr;
    if (which >= SOC_TALLY_INTX && which <= SOC_TALLY_INTZ)
        return (c->with_int == 2 && c->dINTV) ? c->dINTV + (size_t)(which - SOC_TALLY_INTX) * c->G.CELLS : nullptr;
    return nullptr;
}

int soc_zero(soc_ctx *c, int tag)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_zero: call soc_set_grid first");
    float *b = tally_buf(c, tag);
    if (!b) return fail(c, SOC_ERR_ARG, "soc_zero: tag %d", tag);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(b, 0, (size_t)c->G.CELLS * 4, c->stream));
    if (tag == SOC_TALLY_TABS && c->with_ali && c->dXAB)           // ZeroAMC tag 0 clears TABS and XAB (kernel_ASOC_aux.c:664-668)
        HIPCHK(c, hipMemsetAsync(c->dXAB, 0, (size_t)c->G.CELLS * 4, c->stream));
    if (tag == SOC_TALLY_INT && c->with_int == 2 && c->dINTV)      // ... tag 1 INT and the three vector sums (:676-681)
        HIPCHK(c, hipMemsetAsync(c->dINTV, 0, (size_t)3 * c->G.CELLS * 4, c->stream));
    return SOC_OK;
}

static int check_launch(soc_ctx *c, const char *who, int BATCH, int GLOBAL, int gid_first, int gid_count)
{
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "%s: call soc_set_grid first", who);
    if (!c->dCSC) return fail(c, SOC_ERR_STATE, "%s: call soc_set_scatter_table first", who);
    if (!c->dOPT && !c->have_optical) return fail(c, SOC_ERR_STATE, "%s: call soc_set_optical or soc_set_opt first", who);
    if (c->msf_ndust > 1 && !(c->opt_from_abu && c->dOPT && c->abu_ndust == c->msf_ndust && !c->abu_single && c->abu_cells == (size_t)c->G.CELLS))
        return fail(c, SOC_ERR_STATE, "%s: %d scattering functions (WITH_MSF) need soc_set_abundances with %d species and this frequency's soc_set_optical_abu",
                    who, c->msf_ndust, c->msf_ndust);
    if (BATCH < 0) return fail(c, SOC_ERR_ARG, "%s: BATCH=%d", who, BATCH);
    if (GLOBAL < 1 || gid_first < 0 || gid_count < 0 || (int64_t)gid_first + gid_count > GLOBAL)
        return fail(c, SOC_ERR_ARG, "%s: work-item range [%d,+%d) outside GLOBAL=%d", who, gid_first, gid_count, GLOBAL);
    return SOC_OK;
}

static void fill_sim(soc_ctx *c, SocSim &S, SocVariant &V, int SOURCE, int BATCH, float SEED, float BG, float TW,
                     int GLOBAL, int gid_first, int gid_count)
{
    memset(&S, 0, sizeof S);
    S.SOURCE = SOURCE; S.BATCH = BATCH; S.GLOBAL = GLOBAL;
    S.PS_METHOD = c->ps_method; S.BINS = c->BINS; S.USE_EMWEIGHT = c->use_emweight; S.MIRROR = c->mirror;
    S.gid0 = (uint32_t)gid_first; S.gid_count = (uint32_t)gid_count;
    S.seed_mul = soc_seed_mul(SEED); S.seed_tab = c->dSeedTab;
    S.ABS = c->ABS; S.SCA = c->SCA; S.BG = BG; S.TW = TW;
    S.CSC = c->dCSC; S.OPT = c->dOPT;
    S.EMIT = c->dEMIT; S.EMWEI = c->dEMWEI;
    S.EMINDEX = c->dEMINDEX; S.XAB = nullptr;
    S.HPBG = c->dHPBG; S.HPBGP = c->dHPBGP; S.HPBG_WEIGHTED = c->hpbg_weighted ? 1 : 0;
    S.TABS = c->dTABS; S.INT = c->dINT;
    S.stats = c->dStats;
    S.STEP_WEIGHT = c->step_weight;  S.SW_A = c->sw_a;  S.SW_B = c->sw_b;
    S.INTV = (c->with_int == 2) ? c->dINTV : nullptr;  S.CELLS = c->G.CELLS;
    S.NDUST = c->msf_ndust;
    if (c->msf_ndust > 1) { S.MSF_SCA = c->dAF + c->msf_ndust;  S.ABU = c->dABU; }
    V.octree = c->G.LEVELS > 1;
    V.dbl = c->G.NX > ((c->G.LEVELS < 3) ? 399 : 100);   // DIMLIM, kernel_ASOC_aux.c:25-37
    V.abu = c->dOPT != nullptr;
    V.wint = c->with_int;                                  // 0, 1, or 2: INT and the vector sums (the brick-local sweep and the direct kernels)
}

// Point sources of one launch -> device.  xps_as_float: the scattered-light kernels declare
// XPS_NSIDE and XPS_SIDE as "__global float *" (kernel_ASOC_sca.c:495-496, :1486-1487) while
// ASOCS.py uploads the int32 arrays of AnalyseExternalPointSources (ASOCS.py:267-268), so the
// reference reads the integer bit patterns as floats: floor(u*asfloat(nside)*0.999999f) and
// (int)asfloat(side).  The values the reference kernel ends up with are computed here.
static int upload_sources(soc_ctx *c, const char *who, SocSim &S, const float *PSPOS, const float *PS, int NO_PS,
                          const int32_t *XPS_NSIDE, const int32_t *XPS_SIDE, const float *XPS_AREA, bool xps_as_float, int slot = 0)
{
    soc_ctx::SrcBuf &B = c->src[slot];
    if (NO_PS < 1 || !PSPOS || !PS) return fail(c, SOC_ERR_ARG, "%s: point sources need PSPOS, PS and NO_PS>=1", who);
    if ((c->ps_method == 2 || c->ps_method == 5) && (!XPS_NSIDE || !XPS_SIDE || !XPS_AREA))
        return fail(c, SOC_ERR_ARG, "%s: PS_METHOD %d needs XPS_NSIDE/XPS_SIDE/XPS_AREA", who, c->ps_method);
    std::vector<int32_t> nside((size_t)NO_PS, 0), side((size_t)3 * NO_PS, 0);
    if (XPS_NSIDE) nside.assign(XPS_NSIDE, XPS_NSIDE + NO_PS);
    if (XPS_SIDE)  side.assign(XPS_SIDE, XPS_SIDE + 3 * NO_PS);
    if (c->ps_method == 2) {
        for (int i = 0; i < NO_PS; i++) {
            if (nside[i] < 0 || nside[i] > 3) return fail(c, SOC_ERR_ARG, "%s: XPS_NSIDE[%d]=%d", who, i, nside[i]);
            for (int k = 0; k < 3; k++)
                if (side[3 * i + k] < 0 || side[3 * i + k] > 5) return fail(c, SOC_ERR_ARG, "%s: XPS_SIDE[%d]=%d", who, 3 * i + k, side[3 * i + k]);
        }
    }
    if (xps_as_float) {
        // 0 <= v <= 5 as a float bit pattern is a denormal: u*v*0.999999f < 1 and (int)v == 0
        for (auto &v : nside) { float f;  memcpy(&f, &v, 4);  v = (f * 0.999999f < 1.0f) ? 0 : (int32_t)f; }
        for (auto &v : side)  { float f;  memcpy(&f, &v, 4);  v = (int32_t)f; }
    }
    if (NO_PS > B.cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_alloc(&B.PSPOS, (size_t)NO_PS));
        HIPCHK(c, dev_alloc(&B.PS, (size_t)NO_PS));
        HIPCHK(c, dev_alloc(&B.XPS_NSIDE, (size_t)NO_PS));
        HIPCHK(c, dev_alloc(&B.XPS_SIDE, (size_t)3 * NO_PS));
        HIPCHK(c, dev_alloc(&B.XPS_AREA, (size_t)3 * NO_PS));
        B.cap = NO_PS;
    }
    HIPCHK(c, hipMemcpyAsync(B.PSPOS, PSPOS, (size_t)NO_PS * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(B.PS, PS, (size_t)NO_PS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(B.XPS_NSIDE, nside.data(), (size_t)NO_PS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(B.XPS_SIDE, side.data(), (size_t)NO_PS * 12, hipMemcpyHostToDevice, c->stream));
    if (XPS_AREA)  HIPCHK(c, hipMemcpyAsync(B.XPS_AREA, XPS_AREA, (size_t)NO_PS * 12, hipMemcpyHostToDevice, c->stream));
    else           HIPCHK(c, hipMemsetAsync(B.XPS_AREA, 0, (size_t)NO_PS * 12, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    S.NO_PS = NO_PS;
    S.PSPOS = B.PSPOS; S.PS = B.PS;
    S.XPS_NSIDE = B.XPS_NSIDE; S.XPS_SIDE = B.XPS_SIDE; S.XPS_AREA = B.XPS_AREA;
    return SOC_OK;
}

static int snapshot_inputs(soc_ctx *c, SocSim &S, const SocVariant &V, int slot);
#define SOC_OPT_SLOTS 16        // launches with per-cell opacities per sweep (8 B per cell and launch, in one buffer)

// a sweep runs one kernel variant: launches of one kind (SimRAM_PB, _HP or _CL), all with or all without per-cell opacities
// soc_batch_begin_int: the next launch of the batch -- deferred or not -- gets its own, zeroed INT tally
static int take_int_slot(soc_ctx *c, const char *who, SocSim &S)
{
    if (!(c->batching && c->batch_keep_int && c->with_int)) return SOC_OK;
    if (c->batch_group_int && c->int_group_open) {           // a further launch of the current group: the group's tally
        S.INT = c->dINTslot[c->int_slots_done - 1];
        return SOC_OK;
    }
    if (c->int_slots_done >= c->batch_max)
        return fail(c, SOC_ERR_STATE, "%s: %d launches of this batch hold an INT tally; soc_batch_end and soc_batch_read_int first", who, c->batch_max);
    const size_t cells = (size_t)c->G.CELLS;
    if (c->intslot_cells != cells) {                       // another grid: the slots are re-made as they are reached
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (float *&q : c->dINTslot) if (q) { (void)hipFree(q);  q = nullptr; }
        c->intslot_cells = cells;
    }
    if (!c->dINTslot[c->int_slots_done]) HIPCHK(c, dev_alloc(&c->dINTslot[c->int_slots_done], cells));
    S.INT = c->dINTslot[c->int_slots_done];
    c->int_group_open = true;
    HIPCHK(c, hipMemsetAsync(S.INT, 0, cells * 4, c->stream));
    c->int_slots_done++;
    return SOC_OK;
}

// ... except on brick-local hierarchies, where the walk and the event workgroups take the kind from the launch: there the
// point-source, background, Healpix and cell-emission launches of a TABS-only run share one sweep
static bool same_sweep(const soc_ctx *c, int source, bool abu)
{
    if (c->pending.empty()) return true;
    if (c->pending_sca) return false;                        // deferred scattered-light launches: another kind of sweep
    const SocSim &P = c->pending[0];
    const int kp = (P.SOURCE == SOC_SOURCE_CL) ? 2 : (P.SOURCE == SOC_SOURCE_HP) ? 1 : 0;
    const int kn = (source == SOC_SOURCE_CL) ? 2 : (source == SOC_SOURCE_HP) ? 1 : 0;
    if ((P.OPT != nullptr) != abu) return false;
    return kp == kn || lt_capable(c, abu);      // (brick-local hierarchies: the kinds share sweeps, also with per-group INT tallies)
}

int soc_sim_pb(soc_ctx *c, int SOURCE, int PACKETS, int BATCH, float SEED, float BG, float TW,
               const float *PSPOS, const float *PS, int NO_PS,
               const int32_t *XPS_NSIDE, const int32_t *XPS_SIDE, const float *XPS_AREA,
               int GLOBAL, int gid_first, int gid_count)
{
    (void)PACKETS;
    if (!c) return SOC_ERR_ARG;
    int r = check_launch(c, "soc_sim_pb", BATCH, GLOBAL, gid_first, gid_count);
    if (r) return r;
    if (SOURCE != 0 && SOURCE != 1 && SOURCE != 3)
        return fail(c, SOC_ERR_ARG, "soc_sim_pb: SOURCE=%d (0 point sources, 1 background, 3 packets of soc_set_roi_load)", SOURCE);
    if (SOURCE == 3) {
        if (!c->roi.load) return fail(c, SOC_ERR_STATE, "soc_sim_pb: SOURCE 3 needs soc_set_roi_load");
        const int npix = 12 * c->roi.NSIDE * c->roi.NSIDE;
        if (PACKETS != c->roi.NELEM) return fail(c, SOC_ERR_ARG, "soc_sim_pb: SOURCE 3 takes PACKETS = %d surface elements, got %d", c->roi.NELEM, PACKETS);
        if (BATCH % npix) return fail(c, SOC_ERR_ARG, "soc_sim_pb: SOURCE 3 takes BATCH = a multiple of the %d Healpix pixels, got %d", npix, BATCH);
    }
    HIPCHK(c, hipSetDevice(c->device));
    SocSim S;
    SocVariant V;
    fill_sim(c, S, V, SOURCE, BATCH, SEED, BG, TW, GLOBAL, gid_first, gid_count);
    S.ROI = (c->roi.save || c->roi.load) ? c->dRoi : nullptr;
    S.ROISAVE = c->roi.save;
    S.ROILOAD = c->roi.load ? c->roi.NELEM : 0;
    // brick sweep: enough work items to fill the chip.  Hierarchies: it pays from two launches per sweep on
    // (256^3 roots, 4 levels: 1.9e10 steps/s with one launch, 2.8e10 with two, 4.4e10 with eight; direct kernel
    // 2.0e10), so in automatic mode only deferred launches use it (see flush_pending)
    const int B = 1 << c->brick_log2;
    const long long nb = (long long)((c->G.NX + B - 1) / B) * ((c->G.NY + B - 1) / B) * ((c->G.NZ + B - 1) / B);
    bool bricks = (c->exec_mode != 0) && nb <= (1 << 18) && c->G.LEVELS <= 15 && c->device < 16 && (c->mirror == 0 || lt_capable(c, V.abu != 0)) && (c->with_int != 2 || lt_capable(c, V.abu != 0))
                  && (!c->roi.save || (lt_capable(c, V.abu != 0) && c->mirror == 0));      // region-of-interest records: the brick-local sweep's event workgroups (packets of a loaded record, SOURCE 3: any sweep)
    if (c->exec_mode < 0) bricks = bricks && gid_count >= 65536 && nb >= 8
                                   && (!V.octree || (c->batching && (!V.wint || c->batch_keep_int || c->batch_share_int)) || (lt_capable(c, V.abu != 0) && gid_count >= SOC_LT_LONE_LAUNCH));
    if (c->exec_mode == 1 && !bricks)
        return fail(c, SOC_ERR_ARG, "soc_sim_pb: brick sweep requested but not applicable (mirror, with_int 2, roisave/roiload, > 15 levels or > 2^18 bricks)");
    // inside soc_batch_begin/end a brick launch with scalar opacities and no INT tally is deferred:
    // its per-launch inputs are snapshotted (scattering table, sources) and it runs with the others
    const bool defer = c->batching && bricks && (!V.wint || (c->batch_keep_int && V.wint != 2) || c->batch_share_int) && c->msf_ndust <= 1;   // WITH_MSF: per-species tables are not snapshotted
    if (!defer) FLUSH(c);
    if (defer && c->batch_keep_int && !same_sweep(c, SOURCE, V.abu != 0))
        return fail(c, SOC_ERR_STATE, "soc_sim_pb: a batch with the INT tally holds launches of one kind");
    if (defer && !same_sweep(c, SOURCE, V.abu != 0)) FLUSH(c);
    r = take_int_slot(c, "soc_sim_pb", S);
    if (r) return r;
    const int slot = defer ? (int)c->pending.size() : 0;
    if (SOURCE == 0) {
        r = upload_sources(c, "soc_sim_pb", S, PSPOS, PS, NO_PS, XPS_NSIDE, XPS_SIDE, XPS_AREA, false, slot);
        if (r) return r;
    } else {
        S.NO_PS = 1;
    }
    c->last_passes = 0;
    if (defer) {
        r = snapshot_inputs(c, S, V, slot);
        if (r) return r;
        c->pending.push_back(S);
        if ((!c->batch_keep_int && (int)c->pending.size() >= (V.abu ? std::min(c->batch_max, SOC_OPT_SLOTS) : c->batch_max))
            || (int)c->pending.size() >= SOC_MAXLAUNCH) FLUSH(c);      // (with INT tallies per launch or group: a sweep's worth of launches)
        return SOC_OK;
    }
    if (bricks) {
        hipError_t e = soc_brick_run_pb(c->device, c->G, &S, 1, V, c->brick_log2, -1, c->tune, c->stream, &c->last_passes, &c->last_form);
        if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "brick sweep failed: %s", hipGetErrorString(e));
        return SOC_OK;
    }
    HIPCHK(c, soc_launch_sim_pb(c->G, S, V, c->stream));
    return SOC_OK;
}

int soc_batch_begin(soc_ctx *c, int max_launches)
{
    if (!c) return SOC_ERR_ARG;
    if (max_launches < 0 || max_launches > SOC_MAXLAUNCH)
        return fail(c, SOC_ERR_ARG, "soc_batch_begin: max_launches %d (1..%d, 0 = default)", max_launches, SOC_MAXLAUNCH);
    FLUSH(c);
    c->batching = true;
    c->batch_keep_int = false;
    c->batch_share_int = false;
    c->batch_group_int = false;
    c->int_group_open = false;
    c->int_slots_done = 0;
    // default: as many as one sweep takes (the packets in flight are limited separately, see flush_pending)
    c->batch_max = max_launches ? max_launches : SOC_MAXLAUNCH;
    return SOC_OK;
}

int soc_batch_begin_int(soc_ctx *c, int max_launches)
{
    int r = soc_batch_begin(c, max_launches);
    if (r) return r;
    c->batch_keep_int = true;
    return SOC_OK;
}

int soc_batch_begin_int_groups(soc_ctx *c, int max_groups)
{
    int r = soc_batch_begin(c, max_groups);
    if (r) return r;
    c->batch_keep_int = true;
    c->batch_group_int = true;
    return SOC_OK;
}

int soc_batch_next_int(soc_ctx *c)
{
    if (!c) return SOC_ERR_ARG;
    if (!(c->batching && c->batch_group_int)) return fail(c, SOC_ERR_STATE, "soc_batch_next_int: call soc_batch_begin_int_groups first");
    c->int_group_open = false;                               // the next launch takes a new, zeroed INT tally
    return SOC_OK;
}

int soc_batch_begin_shared_int(soc_ctx *c, int max_launches)
{
    int r = soc_batch_begin(c, max_launches);
    if (r) return r;
    c->batch_share_int = true;
    return SOC_OK;
}

int soc_batch_read_int(soc_ctx *c, int k, float *out, long n)
{
    if (!c) return SOC_ERR_ARG;
    if (c->batching || !c->pending.empty()) return fail(c, SOC_ERR_STATE, "soc_batch_read_int: call soc_batch_end first");
    if (k < 0 || k >= c->int_slots_done) return fail(c, SOC_ERR_ARG, "soc_batch_read_int: launch %d of %d deferred with the INT tally", k, c->int_slots_done);
    if (!out || n != (long)c->G.CELLS) return fail(c, SOC_ERR_ARG, "soc_batch_read_int: the tally has %d cells, buffer %ld", c->G.CELLS, n);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->dINTslot[k], (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_batch_end(soc_ctx *c)
{
    if (!c) return SOC_ERR_ARG;
    c->batching = false;
    FLUSH(c);
    return SOC_OK;
}

int soc_set_hpbg(soc_ctx *c, const float *BG, const float *HPBGP)
{
    if (!c) return SOC_ERR_ARG;
    // no flush: a deferred SimRAM_HP launch keeps its own copy of the sky (soc_sim_hp)
    if (!BG) return fail(c, SOC_ERR_ARG, "soc_set_hpbg: BG is NULL");
    const int NPIX = 49152;                                 // NSIDE = 64, fixed in the reference (ASOC.py:297)
    for (int i = 0; i < NPIX; i++)
        if (!std::isfinite(BG[i])) return fail(c, SOC_ERR_ARG, "soc_set_hpbg: BG[%d] is not finite", i);
    if (HPBGP) {
        // the pixel search relies on a non-decreasing table that ends above every random number (ASOC.py:1208)
        for (int i = 1; i < NPIX; i++)
            if (!(HPBGP[i] >= HPBGP[i - 1])) return fail(c, SOC_ERR_ARG, "soc_set_hpbg: HPBGP decreases at pixel %d", i);
        if (!(HPBGP[NPIX - 1] >= 1.0f)) return fail(c, SOC_ERR_ARG, "soc_set_hpbg: HPBGP[last]=%g must be >= 1", (double)HPBGP[NPIX - 1]);
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->dHPBG) {
        HIPCHK(c, dev_alloc(&c->dHPBG, (size_t)NPIX));
        HIPCHK(c, dev_alloc(&c->dHPBGP, (size_t)NPIX));
    }
    HIPCHK(c, hipMemcpyAsync(c->dHPBG, BG, (size_t)NPIX * 4, hipMemcpyHostToDevice, c->stream));
    if (HPBGP) HIPCHK(c, hipMemcpyAsync(c->dHPBGP, HPBGP, (size_t)NPIX * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_hpbg = true;
    c->hpbg_weighted = (HPBGP != nullptr);
    return SOC_OK;
}

// ---- region of interest (roi / roisave / roiload keys) ----

static int roi_upload(soc_ctx *c)
{
    if (!c->dRoi) HIPCHK(c, hipMalloc((void **)&c->dRoi, sizeof(SocRoi)));
    c->roi.SAVE = c->dRoiSave;
    c->roi.LOAD = c->dRoiLoad;
    HIPCHK(c, hipMemcpyAsync(c->dRoi, &c->roi, sizeof(SocRoi), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_set_roi_save(soc_ctx *c, const int32_t *ROI, int ROI_STEP, int ROI_NSIDE)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (!ROI) {                                             // off
        c->roi.save = 0;
        c->roi_save_n = 0;
        return roi_upload(c);
    }
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_set_roi_save: call soc_set_grid first");
    const int lim[3] = { c->G.NX, c->G.NY, c->G.NZ };
    for (int a = 0; a < 3; a++)
        if (ROI[2 * a] < 0 || ROI[2 * a + 1] < ROI[2 * a] || ROI[2 * a + 1] >= lim[a])
            return fail(c, SOC_ERR_ARG, "soc_set_roi_save: ROI[%d..%d] = %d..%d outside the %d root cells of that axis", 2 * a, 2 * a + 1, ROI[2 * a], ROI[2 * a + 1], lim[a]);
    if (ROI_STEP < 1 || ROI_NSIDE < 1 || ROI_NSIDE > 1024) return fail(c, SOC_ERR_ARG, "soc_set_roi_save: ROI_STEP %d, ROI_NSIDE %d", ROI_STEP, ROI_NSIDE);
    if (c->roi.load && c->roi.NSIDE != ROI_NSIDE)
        return fail(c, SOC_ERR_ARG, "soc_set_roi_save: ROI_NSIDE %d differs from the loaded record's %d (one -D ROI_NSIDE in the reference)", ROI_NSIDE, c->roi.NSIDE);
    const int64_t n[3] = { (int64_t)(ROI[1] - ROI[0] + 1) * ROI_STEP, (int64_t)(ROI[3] - ROI[2] + 1) * ROI_STEP, (int64_t)(ROI[5] - ROI[4] + 1) * ROI_STEP };
    const int64_t total = (n[0] * n[1] + n[1] * n[2] + n[2] * n[0]) * 12 * ROI_NSIDE * ROI_NSIDE;
    if (total > 2147483647LL) return fail(c, SOC_ERR_ARG, "soc_set_roi_save: %lld record entries (int32 indices in the kernel)", (long long)total);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, dev_alloc(&c->dRoiSave, (size_t)total));
    HIPCHK(c, hipMemsetAsync(c->dRoiSave, 0, (size_t)total * 4, c->stream));
    for (int i = 0; i < 6; i++) c->roi.ROI[i] = ROI[i];
    c->roi.STEP = ROI_STEP;  c->roi.NSIDE = ROI_NSIDE;  c->roi.save = 1;
    c->roi_save_n = (size_t)total;
    return roi_upload(c);
}

int soc_roi_zero(soc_ctx *c)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->roi.save) return fail(c, SOC_ERR_STATE, "soc_roi_zero: call soc_set_roi_save first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->dRoiSave, 0, c->roi_save_n * 4, c->stream));
    return SOC_OK;
}

int soc_roi_read(soc_ctx *c, float *out, long n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->roi.save) return fail(c, SOC_ERR_STATE, "soc_roi_read: call soc_set_roi_save first");
    if (!out || n != (long)c->roi_save_n) return fail(c, SOC_ERR_ARG, "soc_roi_read: the record has %zu entries, buffer %ld", c->roi_save_n, n);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->dRoiSave, c->roi_save_n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_set_roi_load(soc_ctx *c, const int32_t *DIM, int ROI_NSIDE, const float *LOAD)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (!LOAD) {                                            // off
        c->roi.load = 0;
        return roi_upload(c);
    }
    if (!DIM || DIM[0] < 1 || DIM[1] < 1 || DIM[2] < 1 || ROI_NSIDE < 1 || ROI_NSIDE > 1024)
        return fail(c, SOC_ERR_ARG, "soc_set_roi_load: DIM / ROI_NSIDE");
    if (c->roi.save && c->roi.NSIDE != ROI_NSIDE)
        return fail(c, SOC_ERR_ARG, "soc_set_roi_load: ROI_NSIDE %d differs from the saved record's %d (one -D ROI_NSIDE in the reference)", ROI_NSIDE, c->roi.NSIDE);
    const int64_t nelem = (int64_t)DIM[0] * DIM[1] + (int64_t)DIM[1] * DIM[2] + (int64_t)DIM[2] * DIM[0];
    const int64_t total = nelem * 12 * ROI_NSIDE * ROI_NSIDE;
    if (nelem > 21474836LL || total > 2147483647LL) return fail(c, SOC_ERR_ARG, "soc_set_roi_load: %lld surface elements", (long long)nelem);
    for (int64_t i = 0; i < total; i++)
        if (!std::isfinite(LOAD[i])) return fail(c, SOC_ERR_ARG, "soc_set_roi_load: LOAD[%lld] is not finite", (long long)i);
    if (c->roi_load_cap < (size_t)total) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_alloc(&c->dRoiLoad, (size_t)total));
        c->roi_load_cap = (size_t)total;
    }
    HIPCHK(c, hipMemcpyAsync(c->dRoiLoad, LOAD, (size_t)total * 4, hipMemcpyHostToDevice, c->stream));
    for (int i = 0; i < 3; i++) c->roi.DIM[i] = DIM[i];
    c->roi.NELEM = (int)nelem;  c->roi.NSIDE = ROI_NSIDE;  c->roi.load = 1;
    return roi_upload(c);
}

// what a deferred launch needs besides its SocSim: its own copies of the scattering table and, with abundances,
// of the per-cell opacities (the caller overwrites both for the next frequency)
static int snapshot_inputs(soc_ctx *c, SocSim &S, const SocVariant &V, int slot)
{
    if (c->csc_slot_bins != c->BINS) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < SOC_MAXLAUNCH; k++) HIPCHK(c, dev_alloc(&c->dCSCslot[k], (size_t)c->BINS));
        c->csc_slot_bins = c->BINS;
    }
    HIPCHK(c, hipMemcpyAsync(c->dCSCslot[slot], c->dCSC, (size_t)c->BINS * 4, hipMemcpyDeviceToDevice, c->stream));
    S.CSC = c->dCSCslot[slot];
    if (V.abu) {                                            // the per-cell opacities of this launch: slot of one buffer
        if (slot >= SOC_OPT_SLOTS) return fail(c, SOC_ERR_STATE, "a batch holds at most %d launches with per-cell opacities", SOC_OPT_SLOTS);
        const size_t cells = (size_t)c->G.CELLS;
        if (c->optslot_cells != cells) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, dev_alloc(&c->dOPTslots, cells * SOC_OPT_SLOTS));
            c->optslot_cells = cells;
        }
        HIPCHK(c, hipMemcpyAsync(c->dOPTslots + (size_t)slot * cells, c->dOPT, cells * 8, hipMemcpyDeviceToDevice, c->stream));
        S.OPT = c->dOPTslots + (size_t)slot * cells;
    }
    return SOC_OK;
}

int soc_sim_hp(soc_ctx *c, int PACKETS, int BATCH, float SEED, float TW, int GLOBAL, int gid_first, int gid_count)
{
    (void)PACKETS;
    if (!c) return SOC_ERR_ARG;
    int r = check_launch(c, "soc_sim_hp", BATCH, GLOBAL, gid_first, gid_count);
    if (r) return r;
    if (!c->have_hpbg) return fail(c, SOC_ERR_STATE, "soc_sim_hp: call soc_set_hpbg first");
    HIPCHK(c, hipSetDevice(c->device));
    SocSim S;
    SocVariant V;
    fill_sim(c, S, V, 1, BATCH, SEED, 0.0f, TW, GLOBAL, gid_first, gid_count);
    S.NO_PS = 1;
    // the brick sweep as for soc_sim_pb: the walk is SimRAM_PB's, only the creation of a packet differs
    const int B = 1 << c->brick_log2;
    const long long nb = (long long)((c->G.NX + B - 1) / B) * ((c->G.NY + B - 1) / B) * ((c->G.NZ + B - 1) / B);
    bool bricks = (c->exec_mode != 0) && nb <= (1 << 18) && c->G.LEVELS <= 15 && c->device < 16 && (c->mirror == 0 || lt_capable(c, V.abu != 0)) && (c->with_int != 2 || lt_capable(c, V.abu != 0));
    if (c->exec_mode < 0) bricks = bricks && gid_count >= 65536 && nb >= 8
                                   && (!V.octree || (c->batching && (!V.wint || c->batch_keep_int || c->batch_share_int)) || (lt_capable(c, V.abu != 0) && gid_count >= SOC_LT_LONE_LAUNCH));
    if (c->exec_mode == 1 && !bricks)
        return fail(c, SOC_ERR_ARG, "soc_sim_hp: brick sweep requested but not applicable (mirror, with_int 2, > 15 levels or > 2^18 bricks)");
    const bool defer = c->batching && bricks && (!V.wint || (c->batch_keep_int && V.wint != 2) || c->batch_share_int) && c->msf_ndust <= 1;   // WITH_MSF: per-species tables are not snapshotted
    if (!defer) FLUSH(c);
    if (defer && c->batch_keep_int && !same_sweep(c, SOC_SOURCE_HP, V.abu != 0))
        return fail(c, SOC_ERR_STATE, "soc_sim_hp: a batch with the INT tally holds launches of one kind");
    if (defer && !same_sweep(c, SOC_SOURCE_HP, V.abu != 0)) FLUSH(c);
    r = take_int_slot(c, "soc_sim_hp", S);
    if (r) return r;
    c->last_passes = 0;
    if (bricks) S.SOURCE = SOC_SOURCE_HP;
    if (defer) {
        const int slot = (int)c->pending.size();
        r = snapshot_inputs(c, S, V, slot);
        if (r) return r;
        if (!c->dHPslots) HIPCHK(c, dev_alloc(&c->dHPslots, (size_t)SOC_MAXLAUNCH * 2 * 49152));
        float *sky = c->dHPslots + (size_t)slot * 2 * 49152;
        HIPCHK(c, hipMemcpyAsync(sky, c->dHPBG, 49152 * 4, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(sky + 49152, c->dHPBGP, 49152 * 4, hipMemcpyDeviceToDevice, c->stream));
        S.HPBG = sky;  S.HPBGP = sky + 49152;
        c->pending.push_back(S);
        if ((!c->batch_keep_int && (int)c->pending.size() >= (V.abu ? std::min(c->batch_max, SOC_OPT_SLOTS) : c->batch_max))
            || (int)c->pending.size() >= SOC_MAXLAUNCH) FLUSH(c);      // (with INT tallies per launch or group: a sweep's worth of launches)
        return SOC_OK;
    }
    if (bricks) {
        hipError_t e = soc_brick_run_pb(c->device, c->G, &S, 1, V, c->brick_log2, -1, c->tune, c->stream, &c->last_passes, &c->last_form);
        if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "brick sweep failed: %s", hipGetErrorString(e));
        return SOC_OK;
    }
    HIPCHK(c, soc_launch_sim_hp(c->G, S, V, c->stream));
    return SOC_OK;
}

// a deferred cell-emission launch keeps its own copy of the emission (and of the packet weights): the caller uploads the next frequency's
static int snapshot_emission(soc_ctx *c, SocSim &S, int slot)
{
    const size_t cells = (size_t)c->G.CELLS;
    if (c->emitslot_cells != cells) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (float *&q : c->dEMITslot) if (q) { (void)hipFree(q);  q = nullptr; }
        c->emitslot_cells = cells;
        c->emit_slot_last = -1;
    }
    // (the copy of an earlier launch of this batch serves when soc_set_emission has not been called since)
    int es = slot;
    if (c->emit_slot_last >= 0 && c->emit_slot_last < slot && c->emit_slot_gen == c->emit_gen) {
        es = c->emit_slot_last;
    } else {
        if (!c->dEMITslot[slot]) HIPCHK(c, dev_alloc(&c->dEMITslot[slot], cells * 2));
        HIPCHK(c, hipMemcpyAsync(c->dEMITslot[slot], c->dEMIT, cells * 4, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->dEMITslot[slot] + cells, c->dEMWEI, cells * 4, hipMemcpyDeviceToDevice, c->stream));
        c->emit_slot_last = slot;  c->emit_slot_gen = c->emit_gen;
    }
    float *em = c->dEMITslot[es];
    S.EMIT = em;  S.EMWEI = em + cells;
    return SOC_OK;
}

int soc_sim_cl(soc_ctx *c, int SOURCE, int PACKETS, int BATCH, float SEED, float TW,
               int GLOBAL, int gid_first, int gid_count)
{
    (void)PACKETS;
    if (!c) return SOC_ERR_ARG;
    int r = check_launch(c, "soc_sim_cl", BATCH, GLOBAL, gid_first, gid_count);
    if (r) return r;
    if (!c->have_emit) return fail(c, SOC_ERR_STATE, "soc_sim_cl: call soc_set_emission first");
    HIPCHK(c, hipSetDevice(c->device));
    SocSim S;
    SocVariant V;
    fill_sim(c, S, V, SOURCE, BATCH, SEED, 0.0f, TW, GLOBAL, gid_first, gid_count);
    S.NO_PS = 1;
    if (c->use_emweight == 2 && !c->have_emindex) return fail(c, SOC_ERR_STATE, "soc_sim_cl: USE_EMWEIGHT 2 needs soc_set_emindex");
    if (c->with_ali) S.XAB = c->dXAB;
    S.ROI = c->roi.save ? c->dRoi : nullptr;                // SimRAM_CL records too (kernel_ASOC.c:1250-1254)
    S.ROISAVE = c->roi.save;
    // The brick sweep: the same walk, the event workgroups step through the work item's cells.  It needs packets in
    // flight to sort -- one per work item that has a cell, min(GLOBAL, CELLS): with the reference's GLOBAL = 32768
    // the direct kernel (1.5e10 steps/s at C2, the rate of the fabric atomics) stays; `global` in the ini file
    // raises it.  Host-listed cells (USE_EMWEIGHT 2), ALI and region-of-interest records: direct kernel.
    const int B = 1 << c->brick_log2;
    const long long nb = (long long)((c->G.NX + B - 1) / B) * ((c->G.NY + B - 1) / B) * ((c->G.NZ + B - 1) / B);
    const long long inflight = std::min<long long>((long long)gid_first + gid_count, c->G.CELLS) - gid_first;
    bool bricks = (c->exec_mode != 0) && nb <= (1 << 18) && c->G.LEVELS <= 15 && c->device < 16 && (c->mirror == 0 || lt_capable(c, V.abu != 0)) && (c->with_int != 2 || lt_capable(c, V.abu != 0))
                  && c->use_emweight != 2 && (!c->with_ali || (lt_capable(c, V.abu != 0) && c->with_int != 2)) && (!c->roi.save || (lt_capable(c, V.abu != 0) && c->mirror == 0));
    if (c->exec_mode < 0) bricks = bricks && inflight >= 262144 && nb >= 8
                                   && (!V.octree || (c->batching && (!V.wint || c->batch_keep_int || c->batch_share_int)) || (lt_capable(c, V.abu != 0) && inflight >= SOC_LT_LONE_LAUNCH));
    if (c->exec_mode == 1 && !bricks)
        return fail(c, SOC_ERR_ARG, "soc_sim_cl: brick sweep requested but not applicable (mirror, with_int 2, USE_EMWEIGHT 2, ALI, roisave, > 15 levels or > 2^18 bricks)");
    const bool defer = c->batching && bricks && (!V.wint || (c->batch_keep_int && V.wint != 2) || c->batch_share_int) && c->msf_ndust <= 1;   // WITH_MSF: per-species tables are not snapshotted
    if (!defer) FLUSH(c);
    if (defer && c->batch_keep_int && !same_sweep(c, SOC_SOURCE_CL, V.abu != 0))
        return fail(c, SOC_ERR_STATE, "soc_sim_cl: a batch with the INT tally holds launches of one kind");
    if (defer && !same_sweep(c, SOC_SOURCE_CL, V.abu != 0)) FLUSH(c);
    r = take_int_slot(c, "soc_sim_cl", S);
    if (r) return r;
    c->last_passes = 0;
    if (bricks) S.SOURCE = SOC_SOURCE_CL;
    if (defer) {
        const int slot = (int)c->pending.size();
        r = snapshot_inputs(c, S, V, slot);
        if (r) return r;
        r = snapshot_emission(c, S, slot);
        if (r) return r;
        c->pending.push_back(S);
        if ((!c->batch_keep_int && (int)c->pending.size() >= (V.abu ? std::min(c->batch_max, SOC_OPT_SLOTS) : c->batch_max))
            || (int)c->pending.size() >= SOC_MAXLAUNCH) FLUSH(c);      // (with INT tallies per launch or group: a sweep's worth of launches)
        return SOC_OK;
    }
    if (bricks) {
        hipError_t e = soc_brick_run_pb(c->device, c->G, &S, 1, V, c->brick_log2, -1, c->tune, c->stream, &c->last_passes, &c->last_form);
        if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "brick sweep failed: %s", hipGetErrorString(e));
        return SOC_OK;
    }
    HIPCHK(c, soc_launch_sim_cl(c->G, S, V, c->stream));
    return SOC_OK;
}

// ------------------------------------------------------------------------------------
// scattered-light images (ASOCS.py / kernel_ASOC_sca.c)
// ------------------------------------------------------------------------------------

int soc_sca_set_view(soc_ctx *c, int NDIR, const float *ODIR, const float *RA, const float *DE,
                     int NPIX_X, int NPIX_Y, float MAP_DX, const float *CENTRE, int FFS)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (NDIR < 1 || NDIR > 4096 || !ODIR || !RA || !DE || !CENTRE)
        return fail(c, SOC_ERR_ARG, "soc_sca_set_view: need 1 <= NDIR <= 4096 observer directions (Healpix output, NDIR<0, is not supported)");
    if (NPIX_X < 1 || NPIX_Y < 1 || (int64_t)NDIR * NPIX_X * NPIX_Y > 2147483647LL || !(MAP_DX > 0.0f))
        return fail(c, SOC_ERR_ARG, "soc_sca_set_view: NPIX %d x %d, MAP_DX %g", NPIX_X, NPIX_Y, (double)MAP_DX);
    for (int i = 0; i < NDIR; i++) {
        // GetStep divides by the components of the direction: the host makes them non-zero (ASOC_aux.py:1177-1181)
        for (int k = 0; k < 3; k++) {
            const float v = ODIR[4 * i + k];
            if (!(std::fabs(v) > 0.0f) || !std::isfinite(v))
                return fail(c, SOC_ERR_ARG, "soc_sca_set_view: ODIR[%d].%c = %g (must be finite and non-zero)", i, "xyz"[k], (double)v);
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t npix = (size_t)NDIR * NPIX_X * NPIX_Y;
    const size_t old = view_pixels(c);
    if (NDIR != c->view.NDIR || !c->dODIR) {
        c->view.NDIR = 0;
        HIPCHK(c, dev_alloc(&c->dODIR, (size_t)NDIR));
        HIPCHK(c, dev_alloc(&c->dORA, (size_t)NDIR));
        HIPCHK(c, dev_alloc(&c->dODE, (size_t)NDIR));
    }
    if (npix != old || !c->dOUT) {
        if (!c->own_OUT && c->dOUT && npix != old)
            return fail(c, SOC_ERR_STATE, "soc_sca_set_view: image size changed while a caller-owned image is bound");
        if (c->own_OUT || !c->dOUT) {
            c->dOUT = nullptr;
            HIPCHK(c, dev_alloc(&c->dOUT, npix));
            c->own_OUT = true;
            HIPCHK(c, hipMemsetAsync(c->dOUT, 0, npix * 4, c->stream));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));            // launches in flight still read the previous view
    HIPCHK(c, hipMemcpy(c->dODIR, ODIR, (size_t)NDIR * 16, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dORA, RA, (size_t)NDIR * 16, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dODE, DE, (size_t)NDIR * 16, hipMemcpyHostToDevice));
    c->view.NDIR = NDIR;  c->view.NPIX_X = NPIX_X;  c->view.NPIX_Y = NPIX_Y;  c->view.FFS = FFS ? 1 : 0;
    c->view.MAP_DX = MAP_DX;  c->view.CX = CENTRE[0];  c->view.CY = CENTRE[1];  c->view.CZ = CENTRE[2];
    c->view.ODIRS = c->dODIR;  c->view.ORA = c->dORA;  c->view.ODE = c->dODE;
    c->have_view = true;
    c->out_slots = 0;                                        // (images of soc_sca_batch_images belonged to the old view)
    return SOC_OK;
}

int soc_sca_set_healpix(soc_ctx *c, int NSIDE, const float *OBSERVER, int FFS)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (NSIDE < 1 || NSIDE > 8192 || (NSIDE & (NSIDE - 1)) || !OBSERVER)
        return fail(c, SOC_ERR_ARG, "soc_sca_set_healpix: NSIDE %d (power of two up to 8192) and the observer position are needed", NSIDE);
    for (int k = 0; k < 3; k++)
        if (!std::isfinite(OBSERVER[k])) return fail(c, SOC_ERR_ARG, "soc_sca_set_healpix: observer position is not finite");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t npix = (size_t)12 * NSIDE * NSIDE, old = view_pixels(c);
    if (!c->dODIR || c->view.NDIR < 1) {
        HIPCHK(c, dev_alloc(&c->dODIR, (size_t)1));
        HIPCHK(c, dev_alloc(&c->dORA, (size_t)1));
        HIPCHK(c, dev_alloc(&c->dODE, (size_t)1));
    }
    if (npix != old || !c->dOUT) {
        if (!c->own_OUT && c->dOUT && npix != old)
            return fail(c, SOC_ERR_STATE, "soc_sca_set_healpix: image size changed while a caller-owned image is bound");
        if (c->own_OUT || !c->dOUT) {
            c->dOUT = nullptr;
            HIPCHK(c, dev_alloc(&c->dOUT, npix));
            c->own_OUT = true;
            HIPCHK(c, hipMemsetAsync(c->dOUT, 0, npix * 4, c->stream));
        }
    }
    const float obs[4] = { OBSERVER[0], OBSERVER[1], OBSERVER[2], 0.0f };
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->dODIR, obs, 16, hipMemcpyHostToDevice));
    c->view.NDIR = -NSIDE;  c->view.NPIX_X = 1;  c->view.NPIX_Y = 1;  c->view.FFS = FFS ? 1 : 0;
    c->view.MAP_DX = 1.0f;  c->view.CX = c->view.CY = c->view.CZ = 0.0f;
    c->view.ODIRS = c->dODIR;  c->view.ORA = c->dORA;  c->view.ODE = c->dODE;
    c->have_view = true;
    c->out_slots = 0;                                        // (images of soc_sca_batch_images belonged to the old view)
    return SOC_OK;
}

int soc_sca_zero(soc_ctx *c)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_view) return fail(c, SOC_ERR_STATE, "soc_sca_zero: call soc_sca_set_view first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->dOUT, 0, view_pixels(c) * 4, c->stream));
    return SOC_OK;
}

// Rays on brick-local hierarchies (soc_brick.hip: soc_sca_events) take flat images, scalar opacities and one scattering function
static bool sca_rays_ok(const soc_ctx *c, int kind)
{
    return c->have_view && c->view.NDIR > 0 && kind != SOC_SCA_HP && lt_capable(c, c->dOPT != nullptr) && c->msf_ndust <= 1 && c->device < 16;
}

// Start of a soc_sca_sim_* call: inside soc_batch_begin/end a launch that can run as rays is deferred -- the slot its inputs are
// kept in is returned -- and runs with the others of the batch in one sweep; otherwise (-1) what is pending runs first.
static int sca_begin(soc_ctx *c, int kind, int *slot)
{
    *slot = -1;
    const bool defer = c->batching && c->exec_mode != 0 && sca_rays_ok(c, kind);
    if (!defer || !c->pending_sca || (int)c->pending.size() >= c->batch_max) FLUSH(c);
    if (defer) *slot = (int)c->pending.size();
    return SOC_OK;
}

static int sca_launch(soc_ctx *c, const char *who, int kind, SocSim &S, SocVariant &V, int slot)
{
    if (!c->have_view) return fail(c, SOC_ERR_STATE, "%s: call soc_sca_set_view first", who);
    if (kind != SOC_SCA_CL && kind != SOC_SCA_HP && !c->have_dsc) return fail(c, SOC_ERR_STATE, "%s: soc_set_scatter_table was called without DSC", who);
    if (c->BINS > 8000) return fail(c, SOC_ERR_ARG, "%s: BINS=%d > 8000", who, c->BINS);
    SocSca X = c->view;
    X.kind = kind;
    X.DSC = c->dDSC;
    X.OUT = c->out_slots ? c->dOUTslots + (size_t)c->out_slot_cur * c->out_slot_pixels : c->dOUT;
    S.TABS = nullptr;  S.INT = nullptr;
    S.SCAKIND = kind + 1;  S.DSC = X.DSC;  S.OUT = X.OUT;
    c->last_passes = 0;  c->last_form = 0;
    const bool rays_ok = sca_rays_ok(c, kind);
    if (c->exec_mode == 1 && !rays_ok)
        return fail(c, SOC_ERR_ARG, "%s: brick sweep requested but not applicable (needs a hierarchy walked in double, a flat image, scalar opacities, one scattering function)", who);
    if (slot >= 0) {
        // deferred: the launch keeps its own copies of the scattering functions (and of the emission; the point sources are in their slot already)
        SocVariant W = V;
        W.abu = 0;
        int r = snapshot_inputs(c, S, W, slot);
        if (r) return r;
        if (c->dsc_slot_bins != c->BINS) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int k = 0; k < SOC_MAXLAUNCH; k++) HIPCHK(c, dev_alloc(&c->dDSCslot[k], (size_t)c->BINS));
            c->dsc_slot_bins = c->BINS;
        }
        if (c->have_dsc) HIPCHK(c, hipMemcpyAsync(c->dDSCslot[slot], c->dDSC, (size_t)c->BINS * 4, hipMemcpyDeviceToDevice, c->stream));
        S.DSC = c->dDSCslot[slot];
        if (kind == SOC_SCA_CL) {
            r = snapshot_emission(c, S, slot);
            if (r) return r;
            S.SOURCE = SOC_SOURCE_CL;
        }
        c->pending.push_back(S);
        c->pending_sca = true;
        return SOC_OK;
    }
    // a lone launch: as rays when it is large enough to fill the brick queues; soc_set_exec(1) asks for it, (0) for the direct kernel
    if (rays_ok && (c->exec_mode == 1 || (c->exec_mode < 0 && S.gid_count >= SOC_SCA_RAYS_LAUNCH))) {
        SocSim R = S;
        if (kind == SOC_SCA_CL) R.SOURCE = SOC_SOURCE_CL;
        SocVariant W = V;
        W.wint = 0;
        hipError_t e = soc_brick_run_pb(c->device, c->G, &R, 1, W, c->brick_log2, -1, c->tune, c->stream, &c->last_passes, &c->last_form, &X);
        if (e == hipSuccess) return SOC_OK;
        if (e != hipErrorNotSupported || c->exec_mode == 1) return fail(c, SOC_ERR_HIP, "%s: brick sweep failed: %s", who, hipGetErrorString(e));
    }
    HIPCHK(c, soc_launch_sca(c->G, S, X, V, c->stream));
    return SOC_OK;
}

int soc_sca_sim_ps(soc_ctx *c, int PACKETS, int BATCH, float SEED, float BG, const float *PSPOS, const float *PS, int NO_PS,
                   const int32_t *XPS_NSIDE, const int32_t *XPS_SIDE, const float *XPS_AREA,
                   int GLOBAL, int gid_first, int gid_count)
{
    (void)PACKETS;
    if (!c) return SOC_ERR_ARG;
    int slot = -1;
    int r = sca_begin(c, SOC_SCA_PS, &slot);
    if (r) return r;
    r = check_launch(c, "soc_sca_sim_ps", BATCH, GLOBAL, gid_first, gid_count);
    if (r) return r;
    HIPCHK(c, hipSetDevice(c->device));
    SocSim S;
    SocVariant V;
    fill_sim(c, S, V, 0, BATCH, SEED, BG, 0.0f, GLOBAL, gid_first, gid_count);
    r = upload_sources(c, "soc_sca_sim_ps", S, PSPOS, PS, NO_PS, XPS_NSIDE, XPS_SIDE, XPS_AREA, true, slot < 0 ? 0 : slot);
    if (r) return r;
    return sca_launch(c, "soc_sca_sim_ps", SOC_SCA_PS, S, V, slot);
}

int soc_sca_sim_pb(soc_ctx *c, int SOURCE, int PACKETS, int BATCH, float SEED, float BG, const float *PSPOS, const float *PS,
                   int NO_PS, const int32_t *XPS_NSIDE, const int32_t *XPS_SIDE, const float *XPS_AREA,
                   int GLOBAL, int gid_first, int gid_count)
{
    (void)PACKETS;
    if (!c) return SOC_ERR_ARG;
    int slot = -1;
    int r = sca_begin(c, SOC_SCA_PB, &slot);
    if (r) return r;
    r = check_launch(c, "soc_sca_sim_pb", BATCH, GLOBAL, gid_first, gid_count);
    if (r) return r;
    if (SOURCE != 0 && SOURCE != 1)
        return fail(c, SOC_ERR_ARG, "soc_sca_sim_pb: SOURCE=%d (0 point sources, 1 background; ROI_LOAD is not supported)", SOURCE);
    HIPCHK(c, hipSetDevice(c->device));
    SocSim S;
    SocVariant V;
    fill_sim(c, S, V, SOURCE, BATCH, SEED, BG, 0.0f, GLOBAL, gid_first, gid_count);
    if (SOURCE == 0) {
        r = upload_sources(c, "soc_sca_sim_pb", S, PSPOS, PS, NO_PS, XPS_NSIDE, XPS_SIDE, XPS_AREA, true, slot < 0 ? 0 : slot);
        if (r) return r;
    } else {
        S.NO_PS = 1;
    }
    return sca_launch(c, "soc_sca_sim_pb", SOC_SCA_PB, S, V, slot);
}

int soc_sca_sim_cl(soc_ctx *c, int SOURCE, int PACKETS, int BATCH, float SEED, int GLOBAL, int gid_first, int gid_count)
{
    (void)PACKETS;  (void)SOURCE;
    if (!c) return SOC_ERR_ARG;
    int slot = -1;
    int r = sca_begin(c, SOC_SCA_CL, &slot);
    if (r) return r;
    r = check_launch(c, "soc_sca_sim_cl", BATCH, GLOBAL, gid_first, gid_count);
    if (r) return r;
    if (!c->have_emit) return fail(c, SOC_ERR_STATE, "soc_sca_sim_cl: call soc_set_emission first");
    HIPCHK(c, hipSetDevice(c->device));
    SocSim S;
    SocVariant V;
    fill_sim(c, S, V, 2, BATCH, SEED, 0.0f, 0.0f, GLOBAL, gid_first, gid_count);
    S.NO_PS = 1;
    return sca_launch(c, "soc_sca_sim_cl", SOC_SCA_CL, S, V, slot);
}

int soc_sca_sim_hp(soc_ctx *c, int PACKETS, int BATCH, float SEED, int GLOBAL, int gid_first, int gid_count)
{
    (void)PACKETS;
    if (!c) return SOC_ERR_ARG;
    int slot = -1;
    int r = sca_begin(c, SOC_SCA_HP, &slot);
    if (r) return r;
    r = check_launch(c, "soc_sca_sim_hp", BATCH, GLOBAL, gid_first, gid_count);
    if (r) return r;
    if (!c->have_hpbg) return fail(c, SOC_ERR_STATE, "soc_sca_sim_hp: call soc_set_hpbg first");
    HIPCHK(c, hipSetDevice(c->device));
    SocSim S;
    SocVariant V;
    fill_sim(c, S, V, 1, BATCH, SEED, 0.0f, 0.0f, GLOBAL, gid_first, gid_count);
    S.NO_PS = 1;
    return sca_launch(c, "soc_sca_sim_hp", SOC_SCA_HP, S, V, slot);
}

int soc_sca_read_out(soc_ctx *c, float *out, int64_t n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_view) return fail(c, SOC_ERR_STATE, "soc_sca_read_out: call soc_sca_set_view first");
    const int64_t npix = (int64_t)view_pixels(c);
    if (!out || n < 0 || n > npix) return fail(c, SOC_ERR_ARG, "soc_sca_read_out: n=%lld (image has %lld values)", (long long)n, (long long)npix);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->dOUT, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

void *soc_sca_out_ptr(soc_ctx *c) { return (c && c->have_view) ? (void *)c->dOUT : nullptr; }

int soc_sca_batch_images(soc_ctx *c, int n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (n < 0 || n > 4096) return fail(c, SOC_ERR_ARG, "soc_sca_batch_images: n=%d (0 = the image of soc_sca_set_view again, at most 4096)", n);
    if (n > 0 && !c->have_view) return fail(c, SOC_ERR_STATE, "soc_sca_batch_images: call soc_sca_set_view first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t pix = view_pixels(c);
    if (n > 0) {
        if ((size_t)n * pix > (size_t)c->out_slots * c->out_slot_pixels || !c->dOUTslots) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, dev_alloc(&c->dOUTslots, (size_t)n * pix));
        }
        HIPCHK(c, hipMemsetAsync(c->dOUTslots, 0, (size_t)n * pix * 4, c->stream));
    }
    c->out_slots = n;  c->out_slot_pixels = pix;  c->out_slot_cur = 0;
    return SOC_OK;
}

int soc_sca_batch_select(soc_ctx *c, int k)
{
    if (!c) return SOC_ERR_ARG;
    if (k < 0 || k >= c->out_slots) return fail(c, SOC_ERR_ARG, "soc_sca_batch_select: image %d of %d (soc_sca_batch_images)", k, c->out_slots);
    c->out_slot_cur = k;                                     // (launches already deferred keep the image they were given)
    return SOC_OK;
}

int soc_sca_batch_read(soc_ctx *c, int k, float *out, int64_t n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (k < 0 || k >= c->out_slots) return fail(c, SOC_ERR_ARG, "soc_sca_batch_read: image %d of %d (soc_sca_batch_images)", k, c->out_slots);
    if (!out || n < 0 || (size_t)n > c->out_slot_pixels) return fail(c, SOC_ERR_ARG, "soc_sca_batch_read: n=%lld (an image has %lld values)", (long long)n, (long long)c->out_slot_pixels);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->dOUTslots + (size_t)k * c->out_slot_pixels, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_sca_bind_out(soc_ctx *c, void *device_ptr)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_view) return fail(c, SOC_ERR_STATE, "soc_sca_bind_out: call soc_sca_set_view first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_OUT && c->dOUT) (void)hipFree(c->dOUT);
    c->dOUT = nullptr;
    if (device_ptr) {
        c->dOUT = (float *)device_ptr;
        c->own_OUT = false;
    } else {                                                // back to memory of the library
        const size_t npix = view_pixels(c);
        HIPCHK(c, dev_alloc(&c->dOUT, npix));
        c->own_OUT = true;
        HIPCHK(c, hipMemsetAsync(c->dOUT, 0, npix * 4, c->stream));
    }
    return SOC_OK;
}

int soc_sync(soc_ctx *c)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_read_tally(soc_ctx *c, int which, float *out, int64_t n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_read_tally: call soc_set_grid first");
    float *b = tally_buf(c, which);
    if (!b || !out || n < 0 || n > c->G.CELLS) return fail(c, SOC_ERR_ARG, "soc_read_tally: which=%d n=%lld", which, (long long)n);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, b, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_write_tally(soc_ctx *c, int which, const float *in, int64_t n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_write_tally: call soc_set_grid first");
    float *b = tally_buf(c, which);
    if (!b || !in || n < 0 || n > c->G.CELLS) return fail(c, SOC_ERR_ARG, "soc_write_tally: which=%d n=%lld", which, (long long)n);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(b, in, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

void *soc_tally_ptr(soc_ctx *c, int which) { return c ? (void *)tally_buf(c, which) : nullptr; }

int soc_bind_tally(soc_ctx *c, int which, void *device_ptr, int64_t n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (which != SOC_TALLY_TABS && which != SOC_TALLY_INT) return fail(c, SOC_ERR_ARG, "soc_bind_tally: which=%d", which);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_bind_tally: call soc_set_grid first (the tally has CELLS elements)");
    if (device_ptr && n != (int64_t)c->G.CELLS)
        return fail(c, SOC_ERR_ARG, "soc_bind_tally: the buffer holds %lld floats, the grid has %d cells", (long long)n, c->G.CELLS);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float **buf = (which == SOC_TALLY_TABS) ? &c->dTABS : &c->dINT;
    bool  *own = (which == SOC_TALLY_TABS) ? &c->own_TABS : &c->own_INT;
    if (*own && *buf) (void)hipFree(*buf);
    *buf = nullptr;
    if (device_ptr) {
        *buf = (float *)device_ptr;
        *own = false;
    } else {                                                // back to memory of the library
        HIPCHK(c, dev_alloc(buf, (size_t)c->G.CELLS));
        *own = true;
        HIPCHK(c, hipMemsetAsync(*buf, 0, (size_t)c->G.CELLS * 4, c->stream));
    }
    return SOC_OK;
}

int soc_read_par(soc_ctx *c, int32_t *out, int64_t n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_read_par: call soc_set_grid first");
    if (!out || n < 0 || n > c->npar) return fail(c, SOC_ERR_ARG, "soc_read_par: n=%lld (have %lld)", (long long)n, (long long)c->npar);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n) HIPCHK(c, hipMemcpy(out, c->dPAR, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SOC_OK;
}

int soc_stats(soc_ctx *c, uint64_t out[3], int reset)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long h[4];
    HIPCHK(c, hipMemcpy(h, c->dStats, sizeof h, hipMemcpyDeviceToHost));
    if (out) for (int i = 0; i < 3; i++) out[i] = h[i];
    c->ray_steps = h[3];
    if (reset) HIPCHK(c, hipMemsetAsync(c->dStats, 0, sizeof h, c->stream));
    return SOC_OK;
}

int64_t soc_sca_ray_steps(soc_ctx *c) { return c ? (int64_t)c->ray_steps : -1; }

int soc_timer_start(soc_ctx *c)
{
    if (!c) return SOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return SOC_OK;
}

int soc_timer_stop(soc_ctx *c, float *elapsed_ms)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return SOC_OK;
}

// ---------------------------------------------------------------------------------------
// A2E: stochastically heated grains
// ---------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------
// equilibrium temperature and emission (ASOC.py `CLT` / `CLE` paths)
// ------------------------------------------------------------------------------------

int soc_solve_temperature(soc_ctx *c, float adhoc, float kE, float Emin, int NE, const float *TTT, float FACTOR, float LENGTH,
                          const float *EABS, float *TNEW)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_solve_temperature: call soc_set_grid first");
    if (!TTT || !EABS || NE < 2 || !(kE > 1.0f) || !(Emin > 0.0f) || !(adhoc > 0.0f) || !(LENGTH > 0.0f))
        return fail(c, SOC_ERR_ARG, "soc_solve_temperature: need TTT[NE>=2], EABS, kE>1, Emin>0, adhoc>0, LENGTH>0");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t cells = (size_t)c->G.CELLS;
    if (!c->dT) HIPCHK(c, dev_alloc(&c->dT, cells));
    if (NE > c->ttt_cap) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, dev_alloc(&c->dTTT, (size_t)NE)); c->ttt_cap = NE; }
    if (c->ebuf_cap < cells) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, dev_alloc(&c->dEbuf, cells)); c->ebuf_cap = cells; }
    HIPCHK(c, hipMemcpyAsync(c->dTTT, TTT, (size_t)NE * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dEbuf, EABS, cells * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, soc_launch_eqtemp(c->G, adhoc, kE, Emin, NE, FACTOR, LENGTH, c->cr_rate, c->dTTT, c->dEbuf, c->dT, c->stream));
    if (TNEW) HIPCHK(c, hipMemcpyAsync(TNEW, c->dT, cells * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_T = true;
    return SOC_OK;
}

int soc_set_cr_heating(soc_ctx *c, float rate)
{
    if (!c) return SOC_ERR_ARG;
    if (!(rate >= 0.0f) || !std::isfinite(rate)) return fail(c, SOC_ERR_ARG, "soc_set_cr_heating: rate %g (>= 0; 0 switches it off)", (double)rate);
    c->cr_rate = rate;
    return SOC_OK;
}

int soc_set_map_roi(soc_ctx *c, const int32_t *ROI)
{
    if (!c) return SOC_ERR_ARG;
    if (!ROI) { c->map_roi_on = 0;  return SOC_OK; }
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_set_map_roi: call soc_set_grid first");
    const int N[3] = { c->G.NX, c->G.NY, c->G.NZ };
    for (int k = 0; k < 3; k++)
        if (ROI[2 * k] < 0 || ROI[2 * k + 1] < ROI[2 * k] || ROI[2 * k + 1] >= N[k])
            return fail(c, SOC_ERR_ARG, "soc_set_map_roi: limits %d..%d on axis %d of a grid of %d root cells", ROI[2 * k], ROI[2 * k + 1], k, N[k]);
    for (int k = 0; k < 6; k++) c->map_roi[k] = ROI[k];
    c->map_roi_on = 1;
    return SOC_OK;
}

int soc_set_map_threshold(soc_ctx *c, int level)
{
    if (!c) return SOC_ERR_ARG;
    if (level < 0 || level > SOC_MAXL) return fail(c, SOC_ERR_ARG, "soc_set_map_threshold: level %d", level);
    c->map_level_threshold = level;
    return SOC_OK;
}

int soc_set_map_interpolation(soc_ctx *c, int mode)
{
    if (!c) return SOC_ERR_ARG;
    if (mode < 0 || mode > 2) return fail(c, SOC_ERR_ARG, "soc_set_map_interpolation: mode %d (0, 1 or 2)", mode);
    c->map_interpolation = mode;
    return SOC_OK;
}

int soc_set_temperature(soc_ctx *c, const float *T)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid || !T) return fail(c, SOC_ERR_STATE, "soc_set_temperature: needs a grid and T[CELLS]");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->dT) HIPCHK(c, dev_alloc(&c->dT, (size_t)c->G.CELLS));
    HIPCHK(c, hipMemcpyAsync(c->dT, T, (size_t)c->G.CELLS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_T = true;
    return SOC_OK;
}

int soc_emission(soc_ctx *c, int nfreq, const float *FREQ, const float *FABS, float FACTOR, float LENGTH, float *EMITTED)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_T) return fail(c, SOC_ERR_STATE, "soc_emission: call soc_solve_temperature or soc_set_temperature first");
    if (nfreq < 1 || !FREQ || !FABS || !EMITTED || !(LENGTH > 0.0f)) return fail(c, SOC_ERR_ARG, "soc_emission: nfreq %d", nfreq);
    HIPCHK(c, hipSetDevice(c->device));
    if (2 * nfreq > c->ef_cap) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, dev_alloc(&c->dEF, (size_t)2 * nfreq)); c->ef_cap = 2 * nfreq; }
    HIPCHK(c, hipMemcpyAsync(c->dEF, FREQ, (size_t)nfreq * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dEF + nfreq, FABS, (size_t)nfreq * 4, hipMemcpyHostToDevice, c->stream));
    // batches of cells, all frequencies (the layout of the emitted file: EMITTED[CELLS][nfreq])
    const int cells = c->G.CELLS;
    int batch = (int)(((size_t)64 << 20) / (size_t)nfreq);            // <= 256 MB of floats per batch
    if (batch < 1) batch = 1;
    if (batch > cells) batch = cells;
    const size_t need = (size_t)batch * nfreq;
    if (c->ebuf_cap < need) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, dev_alloc(&c->dEbuf, need)); c->ebuf_cap = need; }
    for (int a = 0; a < cells; a += batch) {
        const int b = (a + batch < cells) ? a + batch : cells;
        HIPCHK(c, soc_launch_emission(a, b, nfreq, FACTOR, LENGTH, c->dEF, c->dEF + nfreq, c->dT, c->dEbuf, c->stream));
        HIPCHK(c, hipMemcpyAsync(EMITTED + (size_t)a * nfreq, c->dEbuf, (size_t)(b - a) * nfreq * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return SOC_OK;
}

// ------------------------------------------------------------------------------------
// map making (ASOC.py:2924-3177 -> kernel_ASOC_map.c Mapping / HealpixMapping)
// ------------------------------------------------------------------------------------

int soc_map(soc_ctx *c, int healpix, int NPIX_X, int NPIX_Y, float MAP_DX, const float *EMIT, const float *DIR, const float *RA,
            const float *DE, const float *CENTRE, const float *INTOBS, float ABS, float SCA, int save_colden, float LENGTH,
            float *MAP, float *SAVETAU)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_map: call soc_set_grid first");
    if (!EMIT || !MAP || !SAVETAU) return fail(c, SOC_ERR_ARG, "soc_map: EMIT, MAP and SAVETAU are needed");
    const bool inside = INTOBS && INTOBS[0] > -1e10f;
    if (healpix) {
        if (NPIX_X < 1 || NPIX_X > 8192 || !inside) return fail(c, SOC_ERR_ARG, "soc_map: Healpix maps need NSIDE (NPIX_X) and an observer position");
    } else {
        if (NPIX_X < 1 || NPIX_Y < 1 || (int64_t)NPIX_X * NPIX_Y > 2147483647LL) return fail(c, SOC_ERR_ARG, "soc_map: NPIX %d x %d", NPIX_X, NPIX_Y);
        if (!inside && (!DIR || !RA || !DE || !CENTRE || !(MAP_DX > 0.0f))) return fail(c, SOC_ERR_ARG, "soc_map: DIR, RA, DE, CENTRE and MAP_DX > 0 are needed");
    }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = healpix ? (size_t)12 * NPIX_X * NPIX_X : (size_t)NPIX_X * NPIX_Y;
    const size_t cells = (size_t)c->G.CELLS;
    if (c->mapemit_cap < cells) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_alloc(&c->dMapEmit, cells));
        c->mapemit_cap = cells;
    }
    if (c->map_cap < npix) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_alloc(&c->dMap, npix));
        HIPCHK(c, dev_alloc(&c->dMapTau, npix));
        c->map_cap = npix;
    }
    SocMapArgs A;
    memset(&A, 0, sizeof A);
    A.mode = healpix ? 1 : 0;
    A.NPIX_X = NPIX_X;  A.NPIX_Y = healpix ? 1 : NPIX_Y;  A.SAVE_COLDEN = save_colden;
    A.LEVEL_THRESHOLD = c->map_level_threshold;
    A.MAPINT = healpix ? 0 : c->map_interpolation;
    A.ROI_MAP = c->map_roi_on;
    for (int k = 0; k < 6; k++) A.ROI[k] = c->map_roi[k];
    A.MAP_DX = MAP_DX;  A.ABS = ABS;  A.SCA = SCA;  A.LENGTH = LENGTH;
    for (int k = 0; k < 3; k++) {
        A.DIR[k] = DIR ? DIR[k] : 0.0f;  A.RA[k] = RA ? RA[k] : 0.0f;  A.DE[k] = DE ? DE[k] : 0.0f;
        A.CENTRE[k] = CENTRE ? CENTRE[k] : 0.0f;
        A.INTOBS[k] = inside ? INTOBS[k] : (k == 0 ? -1.0e12f : 0.0f);
    }
    A.EMIT = c->dMapEmit;  A.OPT = c->dOPT;  A.MAP = c->dMap;  A.SAVETAU = c->dMapTau;
    HIPCHK(c, hipMemcpyAsync(c->dMapEmit, EMIT, cells * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, soc_launch_map(c->G, A, c->dOPT != nullptr, c->stream));
    HIPCHK(c, hipMemcpyAsync(MAP, c->dMap, npix * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(SAVETAU, c->dMapTau, npix * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_ps_tau(soc_ctx *c, int NO_PS, const float *PSPOS, const float *DIR, float ABS, float SCA, float LENGTH, float *pscolden, float *pstau)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_ps_tau: call soc_set_grid first");
    if (NO_PS < 1 || NO_PS > 1000000 || !PSPOS || !DIR || !pscolden || !pstau) return fail(c, SOC_ERR_ARG, "soc_ps_tau: need NO_PS >= 1 sources, DIR and the two output arrays");
    for (int k = 0; k < 3; k++) if (!std::isfinite(DIR[k]) || DIR[k] == 0.0f) return fail(c, SOC_ERR_ARG, "soc_ps_tau: DIR[%d] = %g", k, (double)DIR[k]);
    HIPCHK(c, hipSetDevice(c->device));
    float *d = nullptr;                                     // PSPOS (4 floats per source) | colden | tau
    const size_t n = (size_t)NO_PS;
    if (hipMalloc((void **)&d, n * 6 * 4) != hipSuccess) return fail(c, SOC_ERR_HIP, "soc_ps_tau: allocation");
    hipError_t e = hipMemcpyAsync(d, PSPOS, n * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = soc_launch_pstau(c->G, NO_PS, (const float4 *)d, DIR, ABS, SCA, c->dOPT, LENGTH, d + 4 * n, d + 5 * n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pscolden, d + 4 * n, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pstau, d + 5 * n, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "soc_ps_tau: %s", hipGetErrorString(e));
    return SOC_OK;
}

int soc_a2e_set_size(soc_ctx *c, int NE, int NFREQ, int noIw, const float *Iw, const int32_t *L1,
                     const int32_t *L2, const float *Tdown, const float *EA, const int32_t *Ibeg, const float *AF)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (NE < 3 || NE > 280 || NFREQ < 2 || noIw < 0 || !Iw || !L1 || !L2 || !Tdown || !EA || !Ibeg || !AF)
        return fail(c, SOC_ERR_ARG, "soc_a2e_set_size: bad arguments (3 <= NE <= 280, NFREQ >= 2)");
    // pair tables in the reference's (l, u) loop order; validate every window on the host
    const int npair = (NE * NE - NE) / 2;
    std::vector<int> first(npair), last(npair), off(npair), dst(npair);
    long long iw = 0;
    int e = 0;
    for (int l = 0; l < NE - 1; l++) {
        for (int u = l + 1; u < NE; u++, e++) {
            const int i0 = L1[l * NE + u], i1 = L2[l * NE + u];
            if (i1 >= i0 && (i0 < 0 || i1 >= NFREQ))
                return fail(c, SOC_ERR_ARG, "soc_a2e_set_size: window [%d,%d] of pair (l=%d,u=%d) outside 0..%d", i0, i1, l, u, NFREQ - 1);
            first[e] = i0;  last[e] = i1;  off[e] = (int)iw;  dst[e] = (u * u - u) / 2 + l;
            if (i1 >= i0) iw += i1 - i0 + 1;
        }
    }
    if (iw != noIw) return fail(c, SOC_ERR_ARG, "soc_a2e_set_size: windows need %lld weights, noIw = %d", iw, noIw);
    for (int f = 0; f < NFREQ; f++)
        if (Ibeg[f] < 0 || Ibeg[f] > NE) return fail(c, SOC_ERR_ARG, "soc_a2e_set_size: Ibeg[%d] = %d", f, Ibeg[f]);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, dev_alloc(&c->aIw, (size_t)noIw));
    HIPCHK(c, dev_alloc(&c->aFirst, (size_t)npair));
    HIPCHK(c, dev_alloc(&c->aLast, (size_t)npair));
    HIPCHK(c, dev_alloc(&c->aIwOff, (size_t)npair));
    HIPCHK(c, dev_alloc(&c->aDst, (size_t)npair));
    HIPCHK(c, dev_alloc(&c->aTdown, (size_t)NE));
    HIPCHK(c, dev_alloc(&c->aEA, (size_t)NE * NFREQ));
    HIPCHK(c, dev_alloc(&c->aIbeg, (size_t)NFREQ));
    HIPCHK(c, dev_alloc(&c->aAF, (size_t)NFREQ));
    if (noIw) HIPCHK(c, hipMemcpy(c->aIw, Iw, (size_t)noIw * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->aFirst, first.data(), (size_t)npair * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->aLast, last.data(), (size_t)npair * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->aIwOff, off.data(), (size_t)npair * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->aDst, dst.data(), (size_t)npair * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->aTdown, Tdown, (size_t)NE * 4, hipMemcpyHostToDevice));
    {   // transposed on the way: EAT[i * NFREQ + f] = EA[f * NE + i], so that the lanes of the emission loop (one
        // frequency each) read neighbouring words (the sum over the enthalpy bins keeps its order)
        std::vector<float> eat((size_t)NE * NFREQ);
        for (int f = 0; f < NFREQ; f++)
            for (int i = 0; i < NE; i++) eat[(size_t)i * NFREQ + f] = EA[(size_t)f * NE + i];
        HIPCHK(c, hipMemcpy(c->aEA, eat.data(), (size_t)NE * NFREQ * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMemcpy(c->aIbeg, Ibeg, (size_t)NFREQ * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->aAF, AF, (size_t)NFREQ * 4, hipMemcpyHostToDevice));
    if (NFREQ != c->a2e_NFREQ) c->a2e_cap = 0;
    c->a2e_NE = NE;  c->a2e_NFREQ = NFREQ;  c->a2e_npair = npair;  c->a2e_noIw = noIw;
    return SOC_OK;
}

static int a2e_reserve(soc_ctx *c, int batch)
{
    if (c->a2e_NE == 0) return fail(c, SOC_ERR_STATE, "A2E: call soc_a2e_set_size first");
    if (batch < 1) return fail(c, SOC_ERR_ARG, "A2E: batch = %d", batch);
    if (batch > c->a2e_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, dev_alloc(&c->aABS, (size_t)batch * c->a2e_NFREQ));
        HIPCHK(c, dev_alloc(&c->aEMIT, (size_t)batch * c->a2e_NFREQ));
        c->a2e_cap = batch;
    }
    return SOC_OK;
}

int soc_a2e_upload(soc_ctx *c, int batch, const float *AABS)
{
    if (!c || !AABS) return SOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int r = a2e_reserve(c, batch);
    if (r) return r;
    HIPCHK(c, hipMemcpyAsync(c->aABS, AABS, (size_t)batch * c->a2e_NFREQ * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_a2e_run(soc_ctx *c, int batch)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->a2e_NE == 0 || batch < 1 || batch > c->a2e_cap) return fail(c, SOC_ERR_STATE, "soc_a2e_run: upload a batch first");
    SocA2EArgs A{};
    A.NE = c->a2e_NE;  A.NFREQ = c->a2e_NFREQ;  A.npair = c->a2e_npair;  A.batch = batch;
    A.Iw = c->aIw;  A.pair_first = c->aFirst;  A.pair_last = c->aLast;  A.pair_iw = c->aIwOff;  A.pair_dst = c->aDst;
    A.Tdown = c->aTdown;  A.EA = c->aEA;  A.Ibeg = c->aIbeg;  A.AF = c->aAF;  A.AABS = c->aABS;  A.AEMIT = c->aEMIT;
    HIPCHK(c, soc_launch_a2e_dosolve(A, c->stream));
    return SOC_OK;
}

int soc_a2e_download(soc_ctx *c, int batch, float *AEMIT)
{
    if (!c || !AEMIT) return SOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (batch < 1 || batch > c->a2e_cap) return fail(c, SOC_ERR_ARG, "soc_a2e_download: batch = %d", batch);
    HIPCHK(c, hipMemcpyAsync(AEMIT, c->aEMIT, (size_t)batch * c->a2e_NFREQ * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

// ---- config 5 with the cells resident in HBM ----
int soc_a2e_resident_begin(soc_ctx *c, int64_t cells, int NFREQ)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (cells < 1 || NFREQ < 2 || cells > (int64_t)2147483647) return fail(c, SOC_ERR_ARG, "soc_a2e_resident_begin: cells=%lld NFREQ=%d", (long long)cells, NFREQ);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    const size_t need = (size_t)cells * NFREQ * 8;
    if (need + ((size_t)1 << 30) > free_b + (c->aAll ? (size_t)c->a2e_cells * c->a2e_res_nfreq * 8 : 0))
        return fail(c, SOC_ERR_STATE, "soc_a2e_resident_begin: %lld cells x %d frequencies need %.1f GB of device memory, %.1f GB are free (use soc_a2e_solve in batches)",
                    (long long)cells, NFREQ, need * 1e-9, free_b * 1e-9);
    HIPCHK(c, dev_alloc(&c->aAll, (size_t)cells * NFREQ));
    HIPCHK(c, dev_alloc(&c->aSum, (size_t)cells * NFREQ));
    HIPCHK(c, hipMemsetAsync(c->aSum, 0, (size_t)cells * NFREQ * 4, c->stream));
    c->a2e_cells = cells;  c->a2e_res_nfreq = NFREQ;
    return SOC_OK;
}

int soc_a2e_resident_upload(soc_ctx *c, int64_t c0, int64_t n, const float *AABS)
{
    if (!c || !AABS) return SOC_ERR_ARG;
    if (!c->aAll || c0 < 0 || n < 1 || c0 + n > c->a2e_cells) return fail(c, SOC_ERR_ARG, "soc_a2e_resident_upload: cells [%lld, %lld) of %lld", (long long)c0, (long long)(c0 + n), (long long)c->a2e_cells);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->aAll + (size_t)c0 * c->a2e_res_nfreq, AABS, (size_t)n * c->a2e_res_nfreq * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));             // (the host buffer may be a temporary)
    return SOC_OK;
}

int soc_a2e_resident_solve(soc_ctx *c)
{
    if (!c) return SOC_ERR_ARG;
    if (!c->aAll) return fail(c, SOC_ERR_STATE, "soc_a2e_resident_solve: call soc_a2e_resident_begin first");
    if (c->a2e_NE == 0 || c->a2e_NFREQ != c->a2e_res_nfreq) return fail(c, SOC_ERR_STATE, "soc_a2e_resident_solve: soc_a2e_set_size with NFREQ = %d first", c->a2e_res_nfreq);
    HIPCHK(c, hipSetDevice(c->device));
    SocA2EArgs A{};
    A.NE = c->a2e_NE;  A.NFREQ = c->a2e_NFREQ;  A.npair = c->a2e_npair;
    A.Iw = c->aIw;  A.pair_first = c->aFirst;  A.pair_last = c->aLast;  A.pair_iw = c->aIwOff;  A.pair_dst = c->aDst;
    A.Tdown = c->aTdown;  A.EA = c->aEA;  A.Ibeg = c->aIbeg;  A.AF = c->aAF;
    A.accumulate = 1;
    const int64_t step = 1 << 20;                           // cells per launch (the grid is one workgroup per four cells)
    for (int64_t c0 = 0; c0 < c->a2e_cells; c0 += step) {
        A.batch = (int)std::min<int64_t>(step, c->a2e_cells - c0);
        A.AABS = c->aAll + (size_t)c0 * A.NFREQ;  A.AEMIT = c->aSum + (size_t)c0 * A.NFREQ;
        HIPCHK(c, soc_launch_a2e_dosolve(A, c->stream));
    }
    return SOC_OK;
}

int soc_a2e_resident_download(soc_ctx *c, int64_t c0, int64_t n, float *AEMIT)
{
    if (!c || !AEMIT) return SOC_ERR_ARG;
    if (!c->aSum || c0 < 0 || n < 1 || c0 + n > c->a2e_cells) return fail(c, SOC_ERR_ARG, "soc_a2e_resident_download: cells [%lld, %lld) of %lld", (long long)c0, (long long)(c0 + n), (long long)c->a2e_cells);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(AEMIT, c->aSum + (size_t)c0 * c->a2e_res_nfreq, (size_t)n * c->a2e_res_nfreq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SOC_OK;
}

int soc_a2e_resident_end(soc_ctx *c)
{
    if (!c) return SOC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->aAll) { (void)hipFree(c->aAll);  c->aAll = nullptr; }
    if (c->aSum) { (void)hipFree(c->aSum);  c->aSum = nullptr; }
    c->a2e_cells = 0;
    return SOC_OK;
}

int soc_a2e_solve(soc_ctx *c, int batch, const float *AABS, float *AEMIT)
{
    int r = soc_a2e_upload(c, batch, AABS);
    if (r) return r;
    r = soc_a2e_run(c, batch);
    if (r) return r;
    return soc_a2e_download(c, batch, AEMIT);
}

static int eqtemp_common(soc_ctx *c, const char *who, bool eqsolver, int batch, int icell, int CELLS, int NFREQ, int NIP, float FACTOR, float kE,
                   float oplgkE, float Emin, const float *FREQ, const float *KABS, const float *TTT,
                   const float *ABS, float *T, float *EMIT)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (batch < 1 || NFREQ < 2 || NIP < 2 || !FREQ || !KABS || !TTT || !ABS || !T || !EMIT)
        return fail(c, SOC_ERR_ARG, "%s: bad arguments", who);
    HIPCHK(c, hipSetDevice(c->device));
    float *d = nullptr;
    const size_t n = (size_t)2 * NFREQ + NIP + (size_t)2 * batch * NFREQ + batch;
    HIPCHK(c, hipMalloc((void **)&d, n * 4));
    float *dF = d, *dK = dF + NFREQ, *dT3 = dK + NFREQ, *dA = dT3 + NIP, *dE = dA + (size_t)batch * NFREQ, *dT = dE + (size_t)batch * NFREQ;
    hipError_t e = hipMemcpy(dF, FREQ, (size_t)NFREQ * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dK, KABS, (size_t)NFREQ * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dT3, TTT, (size_t)NIP * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dA, ABS, (size_t)batch * NFREQ * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(dE, 0, ((size_t)batch * NFREQ + batch) * 4, c->stream);
    SocEqTArgs A{};
    A.batch = batch;  A.icell = icell;  A.CELLS = CELLS;  A.NFREQ = NFREQ;  A.NIP = NIP;
    A.FACTOR = FACTOR;  A.kE = kE;  A.oplgkE = oplgkE;  A.Emin = Emin;
    A.FREQ = dF;  A.KABS = dK;  A.TTT = dT3;  A.ABS = dA;  A.T = dT;  A.EMIT = dE;
    if (e == hipSuccess) e = eqsolver ? soc_launch_eqsolver(A, c->stream) : soc_launch_a2e_eqtemp(A, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(T, dT, (size_t)batch * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(EMIT, dE, (size_t)batch * NFREQ * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    return SOC_OK;
}

int soc_a2e_eqtemp(soc_ctx *c, int batch, int icell, int CELLS, int NFREQ, int NIP, float FACTOR, float kE,
                   float oplgkE, float Emin, const float *FREQ, const float *KABS, const float *TTT,
                   const float *ABS, float *T, float *EMIT)
{
    return eqtemp_common(c, "soc_a2e_eqtemp", false, batch, icell, CELLS, NFREQ, NIP, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS, T, EMIT);
}

int soc_eqsolver(soc_ctx *c, int batch, int icell, int CELLS, int NFREQ, int NE, float FACTOR, float kE,
                 float oplgkE, float Emin, const float *FREQ, const float *KABS, const float *TTT,
                 const float *ABS, float *T, float *EMIT)
{
    return eqtemp_common(c, "soc_eqsolver", true, batch, icell, CELLS, NFREQ, NE, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS, T, EMIT);
}

int soc_a2e_pre(soc_ctx *c, int NFREQ, int NE, float FACTOR, const float *FREQ, const float *Ef, const float *SKABS, const float *E,
                const float *T, int32_t *L1, int32_t *L2, float *Iw, int32_t *noIw, float *Tdown)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (NFREQ < 2 || NE < 2 || NE > 4096 || !FREQ || !Ef || !SKABS || !E || !T || !L1 || !L2 || !Iw || !noIw || !Tdown)
        return fail(c, SOC_ERR_ARG, "soc_a2e_pre: NFREQ %d, NE %d or a NULL array", NFREQ, NE);
    for (int i = 1; i < NFREQ; i++)
        if (!(FREQ[i] > FREQ[i - 1]) || !(Ef[i] > Ef[i - 1])) return fail(c, SOC_ERR_ARG, "soc_a2e_pre: FREQ, Ef must increase (entry %d)", i);
    for (int i = 1; i <= NE; i++)
        if (!(E[i] > E[i - 1])) return fail(c, SOC_ERR_ARG, "soc_a2e_pre: the enthalpy grid E[NE+1] must increase (entry %d)", i);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nIw = (size_t)NE * NE * NFREQ, nW = (size_t)NE * NFREQ;
    float *d = nullptr, *dIw = nullptr;
    int   *dL = nullptr;
    // one block of floats: FREQ | Ef | SKABS (NFREQ each) | E | T (NE+1 each) | Tdown (NE) | wrk (NE*NFREQ)
    const size_t nf = 3 * (size_t)NFREQ + 2 * (size_t)(NE + 1) + NE + nW;
    hipError_t e = hipMalloc((void **)&d, nf * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dIw, nIw * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dL, (2 * (size_t)NE * NE + NE) * 4);
    if (e == hipSuccess) {
        float *dF = d, *dEf = dF + NFREQ, *dSK = dEf + NFREQ, *dE = dSK + NFREQ, *dT = dE + NE + 1, *dTd = dT + NE + 1, *dW = dTd + NE;
        int *dL1 = dL, *dL2 = dL1 + (size_t)NE * NE, *dN = dL2 + (size_t)NE * NE;
        e = hipMemcpyAsync(dF, FREQ, (size_t)NFREQ * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dEf, Ef, (size_t)NFREQ * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dSK, SKABS, (size_t)NFREQ * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dE, E, (size_t)(NE + 1) * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dT, T, (size_t)(NE + 1) * 4, hipMemcpyHostToDevice, c->stream);
        // entries the kernels do not write (pairs with u <= l, the unused tail of Iw) are 0 here; the reference leaves them to chance
        if (e == hipSuccess) e = hipMemsetAsync(dL, 0, (2 * (size_t)NE * NE + NE) * 4, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(dIw, 0, nIw * 4, c->stream);
        if (e == hipSuccess) e = soc_launch_a2e_pre(NFREQ, NE, FACTOR, dF, dEf, dSK, dE, dT, dL1, dL2, dIw, dW, dN, dTd, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(L1, dL1, (size_t)NE * NE * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(L2, dL2, (size_t)NE * NE * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(noIw, dN, (size_t)(NE - 1) * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(Iw, dIw, nIw * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(Tdown, dTd, (size_t)NE * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d);
    (void)hipFree(dIw);
    (void)hipFree(dL);
    if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "soc_a2e_pre: %s", hipGetErrorString(e));
    return SOC_OK;
}

// ---------------------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------------------

int soc_probe_rng(soc_ctx *c, float SEED, uint32_t gid_first, uint32_t n, int ndraw, uint32_t *state_xc, uint32_t *draws)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!state_xc || !draws || ndraw < 0 || n == 0) return fail(c, SOC_ERR_ARG, "soc_probe_rng: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t *dS = nullptr, *dD = nullptr;
    HIPCHK(c, hipMalloc((void **)&dS, (size_t)n * 8));
    HIPCHK(c, hipMalloc((void **)&dD, (size_t)n * (ndraw ? ndraw : 1) * 4));
    hipError_t e = soc_launch_seed_probe(soc_seed_mul(SEED), c->dSeedTab, gid_first, n, ndraw, dS, dD, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(state_xc, dS, (size_t)n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && ndraw) e = hipMemcpy(draws, dD, (size_t)n * ndraw * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dS);
    (void)hipFree(dD);
    if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "soc_probe_rng: %s", hipGetErrorString(e));
    return SOC_OK;
}

int soc_probe_math(soc_ctx *c, int fn, const float *x, float *y, int64_t n)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!x || !y || n <= 0) return fail(c, SOC_ERR_ARG, "soc_probe_math: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    float *dx = nullptr, *dy = nullptr;
    HIPCHK(c, hipMalloc((void **)&dx, (size_t)n * 4));
    HIPCHK(c, hipMalloc((void **)&dy, (size_t)n * 4));
    hipError_t e = hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = soc_launch_math_probe(fn, dx, dy, (long)n, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(y, dy, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dx);
    (void)hipFree(dy);
    if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "soc_probe_math: %s", hipGetErrorString(e));
    return SOC_OK;
}

int soc_probe_trace(soc_ctx *c, const float pos[3], const float dir[3], int maxsteps,
                    int32_t *levels, int32_t *inds, float *ds, float endpos[3], int32_t *nsteps)
{
    if (!c) return SOC_ERR_ARG;
    FLUSH(c);
    if (!c->have_grid) return fail(c, SOC_ERR_STATE, "soc_probe_trace: call soc_set_grid first");
    if (!pos || !dir || maxsteps < 1 || !levels || !inds || !ds || !endpos || !nsteps) return fail(c, SOC_ERR_ARG, "soc_probe_trace: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    float *dIn = nullptr, *dDs = nullptr;
    int *dLev = nullptr, *dN = nullptr;
    HIPCHK(c, hipMalloc((void **)&dIn, 9 * 4));
    HIPCHK(c, hipMalloc((void **)&dDs, (size_t)maxsteps * 4));
    HIPCHK(c, hipMalloc((void **)&dLev, (size_t)maxsteps * 8));
    HIPCHK(c, hipMalloc((void **)&dN, 4));
    float h[9] = { pos[0], pos[1], pos[2], dir[0], dir[1], dir[2], 0, 0, 0 };
    SocVariant V;
    V.octree = c->G.LEVELS > 1;
    V.dbl = c->G.NX > ((c->G.LEVELS < 3) ? 399 : 100);
    V.abu = 0; V.wint = 0;
    hipError_t e = hipMemcpy(dIn, h, sizeof h, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = soc_launch_trace(c->G, V, dIn, dIn + 3, maxsteps, dLev, dLev + maxsteps, dDs, dIn + 6, dN, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(nsteps, dN, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(levels, dLev, (size_t)maxsteps * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(inds, dLev + maxsteps, (size_t)maxsteps * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ds, dDs, (size_t)maxsteps * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(endpos, dIn + 6, 12, hipMemcpyDeviceToHost);
    (void)hipFree(dIn);
    (void)hipFree(dDs);
    (void)hipFree(dLev);
    (void)hipFree(dN);
    if (e != hipSuccess) return fail(c, SOC_ERR_HIP, "soc_probe_trace: %s", hipGetErrorString(e));
    return SOC_OK;
}

}  // extern "C"
#pragma GCC visibility pop
