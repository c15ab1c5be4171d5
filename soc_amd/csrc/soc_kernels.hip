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
#include "soc_dev.h"
#include "soc_math.h"
#include "soc_rng.h"

#define SOC_TWOPI  6.28318531f
#define SOC_TAULIM 5.0e-4f
#define SOC_PI     3.1415926535897f
#define SOC_PEPS   1.0e-4f
#define SOC_DEPS   5.0e-5f

// ------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------

// child link of a parent cell: DENS holds -(float with the bits of the index)
__device__ __forceinline__ int soc_link_index(float d)
{
    return (int)(__float_as_uint(d) ^ 0x80000000u);
}

__device__ __forceinline__ void soc_normalize(float &x, float &y, float &z)
{
    float s = 1.0f / soc_sqrtf(x * x + y * y + z * z);
    x = x * s;
    y = y * s;
    z = z * s;
}

__device__ __forceinline__ void soc_tally(float *buf, int oind, float v)
{
#if defined(SOC_EXPERIMENT_NO_TALLY)
    // timing experiment only (never shipped): keep the value alive without touching memory
    if (v == 1.2345e-30f) buf[oind] = v;
#else
    __hip_atomic_fetch_add(buf + oind, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

template <typename T> struct SocReal;
template <> struct SocReal<float> {
    static __device__ __forceinline__ float fmod1(float x) { return soc_fmod1f(x); }
    static __device__ __forceinline__ float floorr(float x) { return soc_floorf(x); }
};
template <> struct SocReal<double> {
    static __device__ __forceinline__ double fmod1(double x) { return soc_fmod1d(x); }
    static __device__ __forceinline__ double floorr(double x) { return __builtin_floor(x); }
};

// ------------------------------------------------------------------------------------
// grid traversal
// ------------------------------------------------------------------------------------

// IndexG (kernel_ASOC_aux.c:131-165): global position -> leaf (level, ind); position is
// converted to the leaf's local octet coordinates.  `dens` returns the leaf's density.
template <bool OCT>
__device__ __forceinline__ void soc_indexg(const SocGrid &G, const int *sOFF, float &px, float &py, float &pz,
                                           int &level, int &ind, float &dens)
{
    ind = -1;
    if ((px <= 0.0f) || (py <= 0.0f) || (pz <= 0.0f)) return;
    if ((px >= G.NX) || (py >= G.NY) || (pz >= G.NZ)) return;
    level = 0;
    ind   = (int)soc_floorf(pz) * G.NX * G.NY + (int)soc_floorf(py) * G.NX + (int)soc_floorf(px);
    dens  = G.DENS[ind];
    if (!OCT) return;
    while (!(dens > 0.0f)) {
        px = 2.0f * soc_fmod1f(px);
        py = 2.0f * soc_fmod1f(py);
        pz = 2.0f * soc_fmod1f(pz);
        ind = soc_link_index(dens);
        level++;
        ind += 4 * (int)soc_floorf(pz) + 2 * (int)soc_floorf(py) + (int)soc_floorf(px);
        dens = G.DENS[sOFF[level] + ind];
    }
}

// Index (kernel_ASOC_aux.c:198-278): neighbour lookup after a step.  T = float, or double
// when NX > DIMLIM.  On return ind < 0 means the packet left the model.
template <bool OCT, typename T>
__device__ __forceinline__ void soc_index(const SocGrid &G, const int *sOFF, float &px, float &py, float &pz,
                                          int &level, int &ind, float &dens)
{
    const int NX = G.NX, NY = G.NY, NZ = G.NZ;
    if (!OCT || (level == 0)) {
        if ((px <= 0.0f) || (px >= NX) || (py <= 0.0f) || (py >= NY) || (pz <= 0.0f) || (pz >= NZ)) {
            ind = -1;
            return;
        }
        ind  = (int)soc_floorf(pz) * NX * NY + (int)soc_floorf(py) * NX + (int)soc_floorf(px);
        dens = G.DENS[ind];
        if (!OCT) return;
        if (dens > 0.0f) return;
    }
    if (OCT) {
        T PX = px, PY = py, PZ = pz;
        const T HALF = (T)0.5, TWO = (T)2.0, ZERO = (T)0.0;
        while (level > 0) {                                  // climb until inside an octet
            ind = G.PAR[sOFF[level] + ind - G.NXYZ];
            level--;
            if (level == 0) {
                PX *= HALF;  PY *= HALF;  PZ *= HALF;
                PX += ind % NX;
                PY += (ind / NX) % NY;
                PZ += ind / (NX * NY);
                if ((PX <= ZERO) || (PX >= NX) || (PY <= ZERO) || (PY >= NY) || (PZ <= ZERO) || (PZ >= NZ)) {
                    ind = -1;
                    px = (float)PX;  py = (float)PY;  pz = (float)PZ;
                    return;
                }
                ind  = (int)SocReal<T>::floorr(PZ) * NX * NY + (int)SocReal<T>::floorr(PY) * NX + (int)SocReal<T>::floorr(PX);
                dens = G.DENS[ind];
                if (dens > 0.0f) {
                    px = (float)PX;  py = (float)PY;  pz = (float)PZ;
                    return;
                }
                break;
            } else {
                int sid = ind % 8;
                PX *= HALF;  PY *= HALF;  PZ *= HALF;
                PX += sid % 2;  PY += (sid / 2) % 2;  PZ += sid / 4;
                if ((PX >= ZERO) && (PX <= TWO) && (PY >= ZERO) && (PY <= TWO) && (PZ >= ZERO) && (PZ <= TWO)) {
                    ind += -sid + 4 * (int)SocReal<T>::floorr(PZ) + 2 * (int)SocReal<T>::floorr(PY) + (int)SocReal<T>::floorr(PX);
                    dens = G.DENS[sOFF[level] + ind];
                    break;
                }
            }
        }
        while (!(dens > 0.0f)) {                             // descend to the leaf
            PX = TWO * SocReal<T>::fmod1(PX);
            PY = TWO * SocReal<T>::fmod1(PY);
            PZ = TWO * SocReal<T>::fmod1(PZ);
            ind = soc_link_index(dens);
            level++;
            ind += 4 * (int)SocReal<T>::floorr(PZ) + 2 * (int)SocReal<T>::floorr(PY) + (int)SocReal<T>::floorr(PX);
            dens = G.DENS[sOFF[level] + ind];
        }
        px = (float)PX;  py = (float)PY;  pz = (float)PZ;
    }
}

// GetStep (kernel_ASOC_aux.c:282-315, float branch): distance to the next cell face in local
// coordinates, overstep by PEPS, advance; returns the step in root-grid units.
template <bool OCT, bool DBL>
__device__ __forceinline__ float soc_getstep(const SocGrid &G, const int *sOFF, float &px, float &py, float &pz,
                                             float ux, float uy, float uz, int &level, int &ind, float &dens)
{
    float ax = (ux > 0.0f) ? (((1.0f + SOC_PEPS) - soc_fmod1f(px)) / ux) : ((-SOC_PEPS - soc_fmod1f(px)) / ux);
    float ay = (uy > 0.0f) ? (((1.0f + SOC_PEPS) - soc_fmod1f(py)) / uy) : ((-SOC_PEPS - soc_fmod1f(py)) / uy);
    float az = (uz > 0.0f) ? (((1.0f + SOC_PEPS) - soc_fmod1f(pz)) / uz) : ((-SOC_PEPS - soc_fmod1f(pz)) / uz);
    float s = __builtin_fminf(ax, __builtin_fminf(ay, az));
    px += s * ux;
    py += s * uy;
    pz += s * uz;
    s = soc_scale_down(s, level);
    if (DBL) soc_index<OCT, double>(G, sOFF, px, py, pz, level, ind, dens);
    else     soc_index<OCT, float>(G, sOFF, px, py, pz, level, ind, dens);
    return s;
}

// Deflect (kernel_ASOC_aux.c:499-533)
__device__ __forceinline__ void soc_deflect(float &ux, float &uy, float &uz, const float COS_THETA, const float phi)
{
    float cx = ux, cy = uy, cz = uz;
    float sin_theta, cos_theta, sin_phi, cos_phi;
    sin_theta = soc_sqrtf(1.0f - COS_THETA * COS_THETA);
    soc_sincosf(phi, &sin_phi, &cos_phi);
    float ox = sin_theta * cos_phi;
    float oy = sin_theta * sin_phi;
    float oz = COS_THETA;
    float theta0 = soc_acosf(cz / soc_sqrtf(cx * cx + cy * cy + cz * cz + SOC_DEPS));
    float phi0   = soc_acosf(cx / soc_sqrtf(cx * cx + cy * cy + SOC_DEPS));
    if (uy < 0.0f) phi0 = (SOC_TWOPI - phi0);
    theta0 = -theta0;
    phi0   = -phi0;
    soc_sincosf(theta0, &sin_theta, &cos_theta);
    soc_sincosf(phi0, &sin_phi, &cos_phi);
    ux = +ox * cos_theta * cos_phi + oy * sin_phi - oz * sin_theta * cos_phi;
    uy = -ox * cos_theta * sin_phi + oy * cos_phi + oz * sin_theta * sin_phi;
    uz = +ox * sin_theta + oz * cos_theta;
}

// Scatter (kernel_ASOC_aux.c:540-561), CSC read from LDS
__device__ __forceinline__ void soc_scatter(float &ux, float &uy, float &uz, const float *sCSC, int BINS, soc_rng_t *rng)
{
    int bin = (int)soc_floorf(soc_rand(rng) * BINS);
    bin = bin < 0 ? 0 : (bin > BINS - 1 ? BINS - 1 : bin);
    float cos_theta = sCSC[bin];
    float phi = SOC_TWOPI * soc_rand(rng);
    soc_deflect(ux, uy, uz, cos_theta, phi);
    if (soc_fabsf(ux) < SOC_DEPS) ux = SOC_DEPS;
    if (soc_fabsf(uy) < SOC_DEPS) uy = SOC_DEPS;
    if (soc_fabsf(uz) < SOC_DEPS) uz = SOC_DEPS;
    soc_normalize(ux, uy, uz);
}

// Surface (kernel_ASOC_aux.c:912-940): step from outside the model to its boundary
__device__ __forceinline__ void soc_surface(const SocGrid &G, float &px, float &py, float &pz, float ux, float uy, float uz)
{
    float dx, dy, dz;
    if (ux > 0.0f) dx = (px < 0.0f) ? ((SOC_PEPS - px) / ux) : -1.0e10f;
    else           dx = (px > G.NX) ? (((G.NX - SOC_PEPS) - px) / ux) : -1.0e10f;
    if (uy > 0.0f) dy = (py < 0.0f) ? ((SOC_PEPS - py) / uy) : -1.0e10f;
    else           dy = (py > G.NY) ? (((G.NY - SOC_PEPS) - py) / uy) : -1.0e10f;
    if (uz > 0.0f) dz = (pz < 0.0f) ? ((SOC_PEPS - pz) / uz) : -1.0e10f;
    else           dz = (pz > G.NZ) ? (((G.NZ - SOC_PEPS) - pz) / uz) : -1.0e10f;
    dx = __builtin_fmaxf(dx, __builtin_fmaxf(dy, dz));
    px += dx * ux;
    py += dx * uy;
    pz += dx * uz;
}

// ------------------------------------------------------------------------------------
// per-lane packet state and the cell step
// ------------------------------------------------------------------------------------

template <bool OCT, bool DBL, bool ABU, bool WINT>
struct SocWalker {
    float px, py, pz, ux, uy, uz;
    float photons, free_path, tau, dens;
    int   level, ind, scat;
    soc_rng_t rng;
    unsigned int n_tally, n_scat;

    // after creation: kernel_ASOC.c:508-519
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

    // One pass of the inner loop body (kernel_ASOC.c:565-683).  Returns true when the free
    // path ends inside the cell: the lane is then put back to the state it had at the
    // beginning of the step (what the reference does with ind0/level0/POS0) and the
    // scattering block is left to scatter(), which may run later -- nothing it needs is lost.
    template <bool CL_ORDER>
    __device__ __forceinline__ bool step(const SocGrid &G, const SocSim &S, const int *sOFF)
    {
        const int   oind = sOFF[level] + ind;
        const int   ind0 = ind, level0 = level;
        const float p0x = px, p0y = py, p0z = pz;
        const float d0 = dens;
        float kabs, ksca;
        if (ABU) {
            float2 o = S.OPT[oind];
            kabs = o.x;
            ksca = o.y;
        } else {
            kabs = S.ABS;
            ksca = S.SCA;
        }
        float ds   = soc_getstep<OCT, DBL>(G, sOFF, px, py, pz, ux, uy, uz, level, ind, dens);
        float tauA = ds * d0 * kabs;
        float dtau = ds * d0 * ksca;
        if (free_path < (tau + dtau)) {
            px = p0x;  py = p0y;  pz = p0z;
            ind = ind0;  level = level0;  dens = d0;
            return true;
        }
        float e = soc_expf(-tauA);
        float delta = (tauA > SOC_TAULIM) ? (photons * (1.0f - e)) : (photons * tauA * (1.0f - 0.5f * tauA));
        soc_tally(S.TABS, oind, delta * S.TW);
        if (WINT) soc_tally(S.INT, oind, delta);
        n_tally++;
        photons *= e;
        tau += dtau;
        if (!CL_ORDER) {
            if ((level == level0) && (ind == ind0)) {       // failed step: nudge (kernel_ASOC.c:649-653)
                px += SOC_PEPS * ux;
                py += SOC_PEPS * uy;
                pz += SOC_PEPS * uz;
            }
        }
        return false;
    }

    // The scattering block (kernel_ASOC.c:700-804, SimRAM_CL: 1545-1676) for a lane that
    // step() returned true for.  Leaves ind < 0 when the packet is dropped (> 20 scatterings).
    template <bool CL_ORDER>
    __device__ __forceinline__ void scatter(const SocSim &S, const float *sCSC, const int *sOFF)
    {
        const int oind = sOFF[level] + ind;
        float kabs, ksca;
        if (ABU) {
            float2 o = S.OPT[oind];
            kabs = o.x;
            ksca = o.y;
        } else {
            kabs = S.ABS;
            ksca = S.SCA;
        }
        scat++;
        if (CL_ORDER && (scat > 20)) { ind = -1; return; }
        float dt = free_path - tau;
        float dx = dt / (ksca * dens);
        float tauA = dx * dens * kabs;
        float e = soc_expf(-tauA);
        float delta = (tauA > SOC_TAULIM) ? (photons * (1.0f - e)) : (photons * tauA * (1.0f - 0.5f * tauA));
        soc_tally(S.TABS, oind, delta * S.TW);
        if (WINT) soc_tally(S.INT, oind, delta);
        n_tally++;
        n_scat++;
        dx = soc_scale_up(dx, level);
        dx = __builtin_fmaxf(0.0f, dx - 2.0f * SOC_PEPS);
        px = px + dx * ux;
        py = py + dx * uy;
        pz = pz + dx * uz;
        photons *= e;
        free_path = -soc_logf(soc_rand(&rng));
        soc_scatter(ux, uy, uz, sCSC, S.BINS, &rng);
        if (!CL_ORDER && (scat > 20)) ind = -1;
        tau = 0.0f;
    }
};

// Lane modes of the per-lane state machine.  Rare arms (packet creation, scattering) are
// not entered the moment ONE lane needs them -- that ran them at 1-2 active lanes on nearly
// every iteration (measured: 927 VALU instructions per wave-iteration at 15 % lane
// utilisation) -- but when a wave ballot shows SOC_SERVICE_LANES lanes waiting, or nobody
// can step.  A lane's own sequence of operations (and RNG draws) is unchanged.
enum { SOC_M_STEP = 0, SOC_M_CREATE = 1, SOC_M_SCATTER = 2, SOC_M_DONE = 3 };
#ifndef SOC_SERVICE_LANES
#define SOC_SERVICE_LANES 12
#endif

__device__ __forceinline__ bool soc_service_now(bool waiting, bool nobody_steps)
{
    unsigned long long m = __ballot(waiting);
    return (m != 0ull) && (nobody_steps || (__popcll(m) >= SOC_SERVICE_LANES));
}

__device__ __forceinline__ void soc_stage_lds(const SocGrid &G, const SocSim &S, float *sCSC, int *sOFF, int *sLC)
{
    for (int i = threadIdx.x; i < S.BINS; i += blockDim.x) sCSC[i] = S.CSC[i];
    if (threadIdx.x < SOC_MAXL) {
        sOFF[threadIdx.x] = G.OFF[threadIdx.x];
        sLC[threadIdx.x]  = G.LCELLS[threadIdx.x];
    }
    __syncthreads();
}

__device__ __forceinline__ void soc_flush_stats(const SocSim &S, unsigned int n_tally, unsigned int n_pkt, unsigned int n_scat)
{
    if (S.stats) {
        atomicAdd(S.stats + 0, (unsigned long long)n_tally);
        atomicAdd(S.stats + 1, (unsigned long long)n_pkt);
        atomicAdd(S.stats + 2, (unsigned long long)n_scat);
    }
}

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
    const int NX = G.NX, NY = G.NY, NZ = G.NZ;
    const int AREA = 2 * (NX * NY + NY * NZ + NZ * NX);
    if ((S.SOURCE == 1) && (id >= 8 * AREA)) return;

    SocWalker<OCT, DBL, ABU, WINT> w;
    w.rng = soc_seed_stream(S.seed_mul, S.seed_tab, (uint32_t)id);
    w.ind = -1;  w.level = 0;  w.n_tally = 0;  w.n_scat = 0;
    w.ux = w.uy = w.uz = 0.0f;  w.px = w.py = w.pz = 0.0f;
    w.dens = 0.0f;  w.photons = 0.0f;  w.tau = 0.0f;  w.free_path = 0.0f;  w.scat = 0;

    // surface element of this work item (kernel_ASOC.c:109-138)
    int   SIDE = 0;
    float X0 = 0.0f, Y0 = 0.0f, Z0 = 0.0f, DX = 1.0f, DY = 1.0f, DZ = 1.0f;
    if (S.SOURCE == 1) {
        int e = id % AREA;
        if (e < NY * NZ) {
            SIDE = 0;  X0 = SOC_PEPS;  Y0 = e % NY;  Z0 = e / NY;  DX = 0.0f;
        } else {
            e -= NY * NZ;
            if (e < NY * NZ) {
                SIDE = 1;  X0 = NX - SOC_PEPS;  Y0 = e % NY;  Z0 = e / NY;  DX = 0.0f;
            } else {
                e -= NY * NZ;
                if (e < NX * NZ) {
                    SIDE = 2;  Y0 = SOC_PEPS;  X0 = e % NX;  Z0 = e / NX;  DY = 0.0f;
                } else {
                    e -= NX * NZ;
                    if (e < NX * NZ) {
                        SIDE = 3;  Y0 = NY - SOC_PEPS;  X0 = e % NX;  Z0 = e / NX;  DY = 0.0f;
                    } else {
                        e -= NX * NZ;
                        if (e < NX * NY) {
                            SIDE = 4;  Z0 = SOC_PEPS;  X0 = e % NX;  Y0 = e / NX;  DZ = 0.0f;
                        } else {
                            e -= NX * NY;
                            SIDE = 5;  Z0 = NZ - SOC_PEPS;  X0 = e % NX;  Y0 = e / NX;  DZ = 0.0f;
                        }
                    }
                }
            }
        }
    }

    int III = 0;
    int mode = SOC_M_CREATE;
    while (true) {
        const bool nobody_steps = (__ballot(mode == SOC_M_STEP) == 0ull);
        if (soc_service_now(mode == SOC_M_CREATE, nobody_steps)) {
            if (mode == SOC_M_CREATE) {
                if (III >= S.BATCH) {
                    mode = SOC_M_DONE;
                } else {
                // ---------------- create packet III ----------------
                if (S.SOURCE == 1) {
                    w.px = soc_clampf(X0 + DX * soc_rand(&w.rng), SOC_PEPS, NX - SOC_PEPS);
                    w.py = soc_clampf(Y0 + DY * soc_rand(&w.rng), SOC_PEPS, NY - SOC_PEPS);
                    w.pz = soc_clampf(Z0 + DZ * soc_rand(&w.rng), SOC_PEPS, NZ - SOC_PEPS);
                    float cos_theta = soc_sqrtf(soc_rand(&w.rng));
                    float phi       = SOC_TWOPI * soc_rand(&w.rng);
                    float sin_theta = soc_sqrtf(1.0f - cos_theta * cos_theta);
                    float sp, cp;
                    soc_sincosf(phi, &sp, &cp);
                    float v1 = sin_theta * cp, v2 = sin_theta * sp;
                    switch (SIDE) {
                    case 0: w.ux =  cos_theta; w.uy = v1; w.uz = v2; break;
                    case 1: w.ux = -cos_theta; w.uy = v1; w.uz = v2; break;
                    case 2: w.uy =  cos_theta; w.ux = v1; w.uz = v2; break;
                    case 3: w.uy = -cos_theta; w.ux = v1; w.uz = v2; break;
                    case 4: w.uz =  cos_theta; w.ux = v1; w.uy = v2; break;
                    default: w.uz = -cos_theta; w.ux = v1; w.uy = v2; break;
                    }
                    w.photons = S.BG;
                    soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                } else {
                    // point sources, kernel_ASOC.c:202-434
                    float phi       = SOC_TWOPI * soc_rand(&w.rng);
                    float cos_theta = 0.999997f - 1.999995f * soc_rand(&w.rng);
                    float sin_theta = soc_sqrtf(1.0f - cos_theta * cos_theta);
                    float sp, cp;
                    soc_sincosf(phi, &sp, &cp);
                    w.ux = sin_theta * cp;
                    w.uy = sin_theta * sp;
                    w.uz = cos_theta;
                    const int isrc = III % S.NO_PS;
                    w.photons = S.PS[isrc];
                    const float4 src = S.PSPOS[isrc];
                    w.px = src.x;  w.py = src.y;  w.pz = src.z;
                    soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                    if ((w.ind < 0) || (w.ind >= G.CELLS)) {
                        const int method = S.PS_METHOD;
                        if (method == 0) {
                            soc_surface(G, w.px, w.py, w.pz, w.ux, w.uy, w.uz);
                            soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                        } else if (method == 1) {
                            if (src.z > NZ) {
                                if (w.uz > 0.0f) w.uz = -w.uz;
                            } else if (src.z < 0.0f) {
                                if (w.uz < 0.0f) w.uz = -w.uz;
                            } else if (src.x > NX) {
                                if (w.ux > 0.0f) w.ux = -w.ux;
                            } else if (src.x < 0.0f) {
                                if (w.ux < 0.0f) w.ux = -w.ux;
                            } else if (src.y > NY) {
                                if (w.uy > 0.0f) w.uy = -w.uy;
                            } else if (src.y < 0.0f) {
                                if (w.uy < 0.0f) w.uy = -w.uy;
                            }
                            soc_surface(G, w.px, w.py, w.pz, w.ux, w.uy, w.uz);
                            w.photons *= 0.5f;
                            soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                        } else if (method == 2) {
                            int k = (int)soc_floorf(soc_rand(&w.rng) * S.XPS_NSIDE[isrc] * 0.999999f);
                            w.photons /= S.XPS_AREA[3 * isrc + k];
                            const int side = S.XPS_SIDE[3 * isrc + k];
                            float a = soc_rand(&w.rng), b = soc_rand(&w.rng);
                            if (side == 0) { w.px = NX - SOC_PEPS;  w.py = a * NY;  w.pz = b * NZ;  b = NY * NZ; }
                            if (side == 1) { w.px = SOC_PEPS;       w.py = a * NY;  w.pz = b * NZ;  b = NY * NZ; }
                            if (side == 2) { w.py = NY - SOC_PEPS;  w.px = a * NX;  w.pz = b * NZ;  b = NX * NZ; }
                            if (side == 3) { w.py = SOC_PEPS;       w.px = a * NX;  w.pz = b * NZ;  b = NX * NZ; }
                            if (side == 4) { w.pz = NZ - SOC_PEPS;  w.px = a * NX;  w.py = b * NY;  b = NX * NY; }
                            if (side == 5) { w.pz = SOC_PEPS;       w.px = a * NX;  w.py = b * NY;  b = NX * NY; }
                            w.ux = w.px - src.x;  w.uy = w.py - src.y;  w.uz = w.pz - src.z;
                            float v1 = soc_sqrtf(w.ux * w.ux + w.uy * w.uy + w.uz * w.uz);
                            soc_normalize(w.ux, w.uy, w.uz);
                            float v2 = (side < 2) ? soc_fabsf(w.ux) : ((side < 4) ? soc_fabsf(w.uy) : soc_fabsf(w.uz));
                            w.photons *= v2 * b / (4.0f * SOC_PI * v1 * v1);
                            soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                        } else if (method == 4) {
                            float v1 = src.z - NZ;
                            float ct = v1 / soc_sqrtf(v1 * v1 + 0.25f * NX * NX + 0.25f * NY * NY);
                            w.photons *= 0.5f * (1.0f - ct);
                            ct = 1.0f - soc_rand(&w.rng) * (1.0f - ct);
                            v1 = SOC_TWOPI * soc_rand(&w.rng);
                            float s1, c1;
                            soc_sincosf(v1, &s1, &c1);
                            w.ux = soc_sqrtf(1.0f - ct * ct) * c1;
                            w.uy = soc_sqrtf(1.0f - ct * ct) * s1;
                            w.uz = -ct;
                            soc_surface(G, w.px, w.py, w.pz, w.ux, w.uy, w.uz);
                            soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                        } else if (method == 5) {
                            float ct = S.XPS_AREA[3 * isrc];
                            w.photons *= 0.5f * (1.0f - ct);
                            ct = 1.0f - soc_rand(&w.rng) * (1.0f - ct);
                            float v1 = SOC_TWOPI * soc_rand(&w.rng);
                            const int side = S.XPS_SIDE[3 * isrc];
                            float s1, c1;
                            soc_sincosf(v1, &s1, &c1);
                            float a = soc_sqrtf(1.0f - ct * ct) * c1;
                            float b = soc_sqrtf(1.0f - ct * ct) * s1;
                            if (side < 2)      { w.uy = a;  w.uz = b;  w.ux = (side == 0) ? -ct : +ct; }
                            else if (side < 4) { w.ux = a;  w.uz = b;  w.uy = (side == 2) ? -ct : +ct; }
                            else               { w.ux = a;  w.uy = b;  w.uz = (side == 4) ? -ct : +ct; }
                            soc_surface(G, w.px, w.py, w.pz, w.ux, w.uy, w.uz);
                            soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
                        }
                    }
                }
                    III++;
                    w.begin();
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

    long long ICELL = (long long)id - S.GLOBAL;            // kernel_ASOC.c:1283-1290
    int   IRAY = 0, batch = -1;
    float PWEI = 1.0f;
    unsigned int n_pkt = 0;
    int   mode = SOC_M_CREATE;

    while (true) {
        const bool nobody_steps = (__ballot(mode == SOC_M_STEP) == 0ull);
        if (soc_service_now(mode == SOC_M_CREATE, nobody_steps)) {
            if (mode == SOC_M_CREATE) {
                if (IRAY >= batch) {                       // next emitting cell (kernel_ASOC.c:1318-1355)
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
                    w.begin();
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
