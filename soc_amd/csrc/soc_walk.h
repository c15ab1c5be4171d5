// soc_walk.h -- device-side building blocks of the packet path, shared by the direct
// kernels (soc_kernels.hip) and the brick engine (soc_brick.hip): traversal
// (IndexG/Index/GetStep), scattering, the per-lane walker, packet creation for SimRAM_PB.
// Arithmetic contract: reference operand order, -ffp-contract=off, transcendentals from
// soc_math.h (see soc_kernels.hip header).
#ifndef SOC_WALK_H
#define SOC_WALK_H

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

// GetStep with the reference's results from fewer instructions (each identity is tested in tests/test_math.py):
// fmod(p,1) of a non-negative p is v_fract; n/u from the cached, correctly rounded r = 1/u by Markstein's
// correction (soc_div_by_rcp).  For rays whose direction is fixed over many steps (the brick walk, the
// scattered-light kernels).
template <bool OCT, bool DBL>
__device__ __forceinline__ float soc_getstep_rcp(const SocGrid &G, const int *sOFF, float &px, float &py, float &pz,
                                                 float ux, float uy, float uz, float rux, float ruy, float ruz,
                                                 int &level, int &ind, float &dens)
{
    float fx, fy, fz;
    if (__ballot(__builtin_fminf(px, __builtin_fminf(py, pz)) < 0.0f) == 0ull) {
        fx = __builtin_amdgcn_fractf(px);  fy = __builtin_amdgcn_fractf(py);  fz = __builtin_amdgcn_fractf(pz);
    } else {
        fx = soc_fmod1f(px);  fy = soc_fmod1f(py);  fz = soc_fmod1f(pz);
    }
    const float ax = soc_div_by_rcp(((ux > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fx, ux, rux);
    const float ay = soc_div_by_rcp(((uy > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fy, uy, ruy);
    const float az = soc_div_by_rcp(((uz > 0.0f) ? (1.0f + SOC_PEPS) : -SOC_PEPS) - fz, uz, ruz);
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

// The first and every later free path (kernel_ASOC.c:516-535, :752-763 and the same lines of SimRAM_HP / SimRAM_CL):
// -D STEP_WEIGHT=1 samples exp(-SW_A*t) instead of exp(-t), =2 the mixture SW_B*exp(-SW_A*t) + (1-SW_B)*exp(-2*SW_A*t);
// the packet's weight carries the ratio of the two densities.
__device__ __forceinline__ float soc_draw_free_path(const SocSim &S, soc_rng_t *rng, float &photons)
{
    if (S.STEP_WEIGHT <= 0) return -soc_logf(soc_rand(rng));
    const float A = S.SW_A, B = S.SW_B;
    float fp;
    if (S.STEP_WEIGHT == 1) {
        fp = -soc_logf(soc_rand(rng)) / A;
        photons *= soc_expf(A * fp - fp) / A;
        return fp;
    }
    fp = -soc_logf((-B + soc_sqrtf(B * B + 4.0f * soc_rand(rng) * (1.0f - B))) / (2.0f - 2.0f * B)) / A;
    photons *= 1.0f / (A * B * soc_expf((1.0f - A) * fp) + 2.0f * A * (1.0f - B) * soc_expf((1.0f - 2.0f * A) * fp));
    return fp;
}

// -D WITH_MSF: the species that scatters in cell oind, drawn with probabilities ABU*SCA/OPT.sca (kernel_ASOC.c:780-791;
// kernel_ASOC_sca.c:340-347, :427-431).  kernel_ASOC.c limits the index to NDUST-1 when rounding leaves ds > 0 after the
// last species; the sca kernels would read past their tables there -- limited in both here.
__device__ __forceinline__ int soc_msf_dust(const SocSim &S, soc_rng_t *rng, const int oind)
{
    const float dx = S.OPT[oind].y;
    float ds = 0.99999f * soc_rand(rng);
    int idust = 0;
    for (; idust < S.NDUST; idust++) {
        ds -= S.ABU[idust + (long)S.NDUST * oind] * S.MSF_SCA[idust] / dx;
        if (ds <= 0.0f) break;
    }
    return (idust >= S.NDUST) ? S.NDUST - 1 : idust;
}

// New direction after a scattering in cell oind.  With -D WITH_MSF the species' own cumulative scattering function is
// used (kernel_ASOC.c:777-795); SimRAM_CL reuses `free_path` as scratch for OPT.sca there (:1662), so the next free path
// of such a packet IS that number.
template <bool CL_ORDER>
__device__ __forceinline__ void soc_new_direction(const SocSim &S, const float *sCSC, const int oind, float &ux, float &uy, float &uz,
                                                  float &free_path, soc_rng_t *rng)
{
    if (S.NDUST > 1) {
        if (CL_ORDER) free_path = S.OPT[oind].y;
        const int idust = soc_msf_dust(S, rng, oind);
        soc_scatter(ux, uy, uz, S.CSC + (long)idust * S.BINS, S.BINS, rng);
    } else {
        soc_scatter(ux, uy, uz, sCSC, S.BINS, rng);
    }
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

// Mirror (kernel_ASOC_aux.c:1050-1083), for a packet that has just left the grid (ind < 0).
// As written in the reference the direction flip sits outside the if-statement: every enabled
// face flips its component each time the function runs.  Kept.
#define SOC_EPS_MIRROR 5.0e-4f
template <bool OCT>
__device__ __forceinline__ void soc_mirror(const SocGrid &G, const int *sOFF, int MIRROR, float &px, float &py, float &pz,
                                           float &ux, float &uy, float &uz, int &level, int &ind, float &dens)
{
    if (MIRROR & 1)  { if (px < 0.0f) px = SOC_EPS_MIRROR;         ux = -ux;  soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens); }
    if (MIRROR & 2)  { if (px > G.NX) px = G.NX - SOC_EPS_MIRROR;  ux = -ux;  soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens); }
    if (MIRROR & 4)  { if (py < 0.0f) py = SOC_EPS_MIRROR;         uy = -uy;  soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens); }
    if (MIRROR & 8)  { if (py > G.NY) py = G.NY - SOC_EPS_MIRROR;  uy = -uy;  soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens); }
    if (MIRROR & 16) { if (pz < 0.0f) pz = SOC_EPS_MIRROR;         uz = -uz;  soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens); }
    if (MIRROR & 32) { if (pz > G.NZ) pz = G.NZ - SOC_EPS_MIRROR;  uz = -uz;  soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens); }
}

__device__ __forceinline__ void soc_rootpos(const SocGrid &G, const int *sOFF, float &px, float &py, float &pz, int level, int ind);
__device__ __forceinline__ int  soc_angles2pixel_ring(const int nside, float phi, const float theta);

// InRoi (kernel_ASOC_aux.c:1031-1048): root index of the root cell above (level, ind) if it lies in ROI, else -1.
// A packet that has left the model is outside (the reference's arithmetic on ind = -1 gives -1 % NX < ROI[0]).
__device__ __forceinline__ int soc_inroi(const SocGrid &G, const int *sOFF, const SocRoi &R, int level, int ind)
{
    if (ind < 0) return -1;
    while (level > 0) { ind = G.PAR[sOFF[level] + ind - G.NXYZ];  level--; }
    const int i = ind % G.NX, j = (ind / G.NX) % G.NY, k = ind / (G.NX * G.NY);
    return ((i >= R.ROI[0]) && (i <= R.ROI[1]) && (j >= R.ROI[2]) && (j <= R.ROI[3]) && (k >= R.ROI[4]) && (k <= R.ROI[5])) ? ind : -1;
}

// A packet has stepped into ROI (kernel_ASOC.c:618-642, :1510-1535): surface element from the root position,
// Healpix pixel from the direction, PHOTONS added to the record.  `ii` is uninitialised in the reference when
// no border test matches (one always does: the packet is within PEPS of the face it came through); 0 here.
__device__ __forceinline__ void soc_roi_save(const SocGrid &G, const int *sOFF, const SocRoi &R, float px, float py, float pz,
                                             float ux, float uy, float uz, int level, int ind, float photons)
{
    const int NX = (R.ROI[1] - R.ROI[0] + 1) * R.STEP, NY = (R.ROI[3] - R.ROI[2] + 1) * R.STEP, NZ = (R.ROI[5] - R.ROI[4] + 1) * R.STEP;
    soc_rootpos(G, sOFF, px, py, pz, level, ind);
    int ii = 0, jj;
    auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    if ((px < ((float)R.ROI[0] + 1.0e-3f)) || (px > ((float)R.ROI[1] + 0.999f))) {
        ii = cl((int)soc_floorf((py - (float)R.ROI[2]) * (float)R.STEP), 0, NY - 1);
        jj = cl((int)soc_floorf((pz - (float)R.ROI[4]) * (float)R.STEP), 0, NZ - 1);
        ii = ii + NY * jj;
    }
    if ((py < ((float)R.ROI[2] + 1.0e-3f)) || (py > ((float)R.ROI[3] + 0.999f))) {
        ii = cl((int)soc_floorf((px - (float)R.ROI[0]) * (float)R.STEP), 0, NX - 1);
        jj = cl((int)soc_floorf((pz - (float)R.ROI[4]) * (float)R.STEP), 0, NZ - 1);
        ii = NY * NZ + ii + NX * jj;
    }
    if ((pz < ((float)R.ROI[4] + 1.0e-3f)) || (pz > ((float)R.ROI[5] + 0.999f))) {
        ii = cl((int)soc_floorf((px - (float)R.ROI[0]) * (float)R.STEP), 0, NX - 1);
        jj = cl((int)soc_floorf((py - (float)R.ROI[2]) * (float)R.STEP), 0, NY - 1);
        ii = NY * NZ + NX * NZ + ii + NX * jj;
    }
    jj = soc_angles2pixel_ring(R.NSIDE, soc_atan2f(uy, ux), soc_acosf(uz));
    ii = cl(ii, 0, NX * NY + NY * NZ + NZ * NX - 1);
    jj = cl(jj, 0, 12 * R.NSIDE * R.NSIDE - 1);
    soc_tally(R.SAVE, ii * 12 * R.NSIDE * R.NSIDE + jj, photons);
}

template <bool OCT, bool DBL, bool ABU, bool WINT>
struct SocWalker {
    float px, py, pz, ux, uy, uz;
    float photons, free_path, tau, dens;
    int   level, ind, scat;
    int   e_index = -1;            // WITH_ALI: global index of the emitting cell (kernel_ASOC.c:1394-1396)
    bool  roi_on = false;          // SimRAM_PB / SimRAM_CL with WITH_ROI_SAVE (set by those kernels only)
    int   roi = -1;                // root index of the current cell if inside ROI (kernel_ASOC.c:550,562,617)
    soc_rng_t rng;
    unsigned int n_tally, n_scat;

    // absorbed energy of one step: TABS, or XAB for what the emitting cell absorbs of its own
    // emission (WITH_ALI, kernel_ASOC.c:1486-1491)
    __device__ __forceinline__ void deposit(const SocSim &S, int oind, float delta)
    {
        if (S.XAB && (oind == e_index)) soc_tally(S.XAB, oind, delta * S.TW);
        else                            soc_tally(S.TABS, oind, delta * S.TW);
    }

    // -D SAVE_INTENSITY=2: the net flux through the cell, delta * DIR (kernel_ASOC.c:604-612, :724-732)
    __device__ __forceinline__ void intensity_vector(const SocSim &S, int oind, float delta)
    {
        soc_tally(S.INTV, oind, delta * ux);
        soc_tally(S.INTV + S.CELLS, oind, delta * uy);
        soc_tally(S.INTV + 2 * (long)S.CELLS, oind, delta * uz);
    }

    // after creation: kernel_ASOC.c:508-519
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

    // SimRAM_HP conditions the direction itself, before it picks the entry point
    // (kernel_ASOC.c:919-922); only the counters and the first free path remain (:950-955)
    __device__ __forceinline__ void begin_conditioned(const SocSim &S)
    {
        scat = 0;
        tau  = 0.0f;
        free_path = soc_draw_free_path(S, &rng, photons);
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
        const int   oroi = roi;
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
        deposit(S, oind, delta);
        if (WINT) {
            soc_tally(S.INT, oind, delta);
            if (S.INTV) intensity_vector(S, oind, delta);
        }
        n_tally++;
        photons *= e;
        tau += dtau;
        if (roi_on) {                                        // only at the end of a full step (kernel_ASOC.c:615-642)
            roi = soc_inroi(G, sOFF, *S.ROI, level, ind);
            if ((roi >= 0) && (oroi < 0)) soc_roi_save(G, sOFF, *S.ROI, px, py, pz, ux, uy, uz, level, ind, photons);
        }
        if (!CL_ORDER) {
            if ((level == level0) && (ind == ind0)) {       // failed step: nudge (kernel_ASOC.c:649-653)
                px += SOC_PEPS * ux;
                py += SOC_PEPS * uy;
                pz += SOC_PEPS * uz;
            }
        }
        if ((S.MIRROR > 0) && (ind < 0))                     // kernel_ASOC.c:686-688, :1064, :1540
            soc_mirror<OCT>(G, sOFF, S.MIRROR, px, py, pz, ux, uy, uz, level, ind, dens);
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
        deposit(S, oind, delta);
        if (WINT) {
            soc_tally(S.INT, oind, delta);
            if (S.INTV) intensity_vector(S, oind, delta);
        }
        n_tally++;
        n_scat++;
        dx = soc_scale_up(dx, level);
        dx = __builtin_fmaxf(0.0f, dx - 2.0f * SOC_PEPS);
        px = px + dx * ux;
        py = py + dx * uy;
        pz = pz + dx * uz;
        photons *= e;
        free_path = soc_draw_free_path(S, &rng, photons);
        soc_new_direction<CL_ORDER>(S, sCSC, oind, ux, uy, uz, free_path, &rng);
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

// surface element of a background work item (kernel_ASOC.c:109-138)
struct SocSurfElem {
    int   SIDE;
    float X0, Y0, Z0, DX, DY, DZ;
};

// SRC >= 0: the kind of source is known at compile time (the other one's code and registers are not carried)
template <int SRC = -1>
__device__ __forceinline__ SocSurfElem soc_surface_element(const SocGrid &G, const SocSim &S, int id)
{
    const int NX = G.NX, NY = G.NY, NZ = G.NZ;
    const int AREA = 2 * (NX * NY + NY * NZ + NZ * NX);
    int   SIDE = 0;
    float X0 = 0.0f, Y0 = 0.0f, Z0 = 0.0f, DX = 1.0f, DY = 1.0f, DZ = 1.0f;
    if (((SRC >= 0) ? SRC : S.SOURCE) == 1) {
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

    SocSurfElem E;
    E.SIDE = SIDE;  E.X0 = X0;  E.Y0 = Y0;  E.Z0 = Z0;  E.DX = DX;  E.DY = DY;  E.DZ = DZ;
    return E;
}

// Creation of packet III of a SimRAM_PB work item: background (kernel_ASOC.c:439-464) or
// point source (kernel_ASOC.c:202-434).  The caller then runs w.begin().
template <bool OCT, typename W, int SRC = -1>
__device__ __forceinline__ void soc_pb_create(const SocGrid &G, const SocSim &S, const int *sOFF,
                                              const SocSurfElem &E, int III, W &w)
{
    const int   NX = G.NX, NY = G.NY, NZ = G.NZ;
    const int   SIDE = E.SIDE;
    const float X0 = E.X0, Y0 = E.Y0, Z0 = E.Z0, DX = E.DX, DY = E.DY, DZ = E.DZ;
    if (((SRC >= 0) ? SRC : S.SOURCE) == 1) {
        w.px = soc_clampf(X0 + DX * soc_rand(&w.rng), SOC_PEPS, NX - SOC_PEPS);
        w.py = soc_clampf(Y0 + DY * soc_rand(&w.rng), SOC_PEPS, NY - SOC_PEPS);
        w.pz = soc_clampf(Z0 + DZ * soc_rand(&w.rng), SOC_PEPS, NZ - SOC_PEPS);
        float cos_theta = soc_sqrtf(soc_rand(&w.rng));
        float phi       = SOC_TWOPI * soc_rand(&w.rng);
        float sin_theta = soc_sqrtf(1.0f - cos_theta * cos_theta);
        float sp, cp;
        soc_sincosf(phi, &sp, &cp);
        float v1 = sin_theta * cp, v2 = sin_theta * sp;
        // kernel_ASOC.c:449-456: sides 0/1 (c,v1,v2), 2/3 (v1,c,v2), 4/5 (v1,v2,c) with c = -+cos_theta on the
        // odd side.  Written as selects (a switch on SIDE becomes a private array in scratch memory).
        {
            const float c = (SIDE & 1) ? -cos_theta : cos_theta;
            const int   axis = SIDE >> 1;
            w.ux = (axis == 0) ? c : v1;
            w.uy = (axis == 0) ? v1 : ((axis == 1) ? c : v2);
            w.uz = (axis >= 2) ? c : v2;
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
}

// Pixel2AnglesRing (kernel_ASOC_aux.c:987-1026): Healpix RING pixel -> (phi, theta)
__device__ __forceinline__ void soc_pixel2angles_ring(const int nside, const int ipix, float &phi, float &theta)
{
    const int   npix = 12 * nside * nside, ipix1 = ipix + 1, nl2 = 2 * nside, nl4 = 4 * nside;
    const int   ncap = 2 * nside * (nside - 1);
    const float fact1 = 1.5f * nside, fact2 = 3.0f * nside * nside;
    if (ipix1 <= ncap) {
        const float hip = ipix1 / 2.0f, fihip = (float)(int)hip;
        const int   iring = (int)soc_sqrtf(hip - soc_sqrtf(fihip)) + 1;
        const int   iphi = ipix1 - 2 * iring * (iring - 1);
        theta = soc_acosf(1.0f - (float)(iring * iring) / fact2);
        phi   = ((float)iphi - 0.5f) * SOC_PI / (2.0f * (float)iring);
    } else if (ipix1 <= nl2 * (5 * nside + 1)) {
        const int   ip = ipix1 - ncap - 1;
        const int   iring = (ip / nl4) + nside, iphi = (ip % nl4) + 1;
        const float fodd = 0.5f * (float)(1 + (iring + nside) % 2);
        theta = soc_acosf((float)(nl2 - iring) / fact1);
        phi   = ((float)iphi - fodd) * SOC_PI / (2.0f * (float)nside);
    } else {
        const int   ip = npix - ipix1 + 1;
        const float hip = ip / 2.0f, fihip = (float)(int)hip;
        const int   iring = (int)soc_sqrtf(hip - soc_sqrtf(fihip)) + 1;
        const int   iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
        theta = soc_acosf(-1.0f + (float)(iring * iring) / fact2);
        phi   = ((float)iphi - 0.5f) * SOC_PI / (2.0f * (float)iring);
    }
}

// Angles2PixelRing (kernel_ASOC_aux.c:945-984): (phi, theta) -> Healpix RING pixel, -1 outside [0, pi]
__device__ __forceinline__ int soc_angles2pixel_ring(const int nside, float phi, const float theta)
{
    if ((theta < 0.0f) || (theta > SOC_PI)) return -1;
    const float z = soc_cosf(theta), za = soc_fabsf(z);
    if (phi >= SOC_TWOPI) phi -= SOC_TWOPI;
    if (phi < 0.0f)       phi += SOC_TWOPI;
    const float tt = phi / 1.5707963268f;
    const int nl2 = 2 * nside, nl4 = 4 * nside, ncap = nl2 * (nside - 1), npix = 12 * nside * nside;
    int ipix1;
    if (za <= 0.6666666667f) {
        const int jp = (int)(nside * (0.5f + tt - z * 0.75f));
        const int jm = (int)(nside * (0.5f + tt + z * 0.75f));
        const int ir = nside + 1 + jp - jm;
        const int kshift = (ir % 2 == 0) ? 1 : 0;
        int ip = (int)((jp + jm - nside + kshift + 1) / 2) + 1;
        if (ip > nl4) ip -= nl4;
        ipix1 = ncap + nl4 * (ir - 1) + ip;
    } else {
        const float tp = tt - (float)(int)tt;
        const float tmp = soc_sqrtf(3.0f * (1.0f - za));
        const int jp = (int)(nside * tp * tmp);
        const int jm = (int)(nside * (1.0f - tp) * tmp);
        const int ir = jp + jm + 1;
        int ip = (int)(tt * ir) + 1;
        if (ip > (4 * ir)) ip -= 4 * ir;
        ipix1 = 2 * ir * (ir - 1) + ip;
        if (z <= 0.0f) ipix1 = npix - 2 * ir * (ir + 1) + ip;
    }
    return ipix1 - 1;
}

// RootPos (kernel_ASOC_aux.c:169-190): local position of cell (level, ind) -> root-grid coordinates
__device__ __forceinline__ void soc_rootpos(const SocGrid &G, const int *sOFF, float &px, float &py, float &pz, int level, int ind)
{
    while (level > 0) {
        ind = G.PAR[sOFF[level] + ind - G.NXYZ];
        level--;
        px *= 0.5f;  py *= 0.5f;  pz *= 0.5f;
        if (level == 0) {
            px += (float)(ind % G.NX);
            py += (float)((ind / G.NX) % G.NY);
            pz += (float)(ind / (G.NX * G.NY));
        } else {
            const int sid = ind % 8;
            px += (float)(sid % 2);  py += (float)((sid / 2) % 2);  pz += (float)(sid / 4);
        }
    }
}

// Healpix pixel of the next background packet (kernel_ASOC.c:881-902): uniform, or bisection
// (n_bisect halvings + linear scan) on the cumulative probability HPBGP
__device__ __forceinline__ int soc_hp_select_pixel(const SocSim &S, soc_rng_t *rng, int n_bisect)
{
    int ind;
    if (S.HPBG_WEIGHTED < 1) {
        ind = (int)soc_floorf(soc_rand(rng) * 49152);
        ind = ind < 0 ? 0 : (ind > 49151 ? 49151 : ind);
    } else {
        const float x = soc_rand(rng);
        int ind0 = 0, level0 = 49151;
        for (int i = 0; i < n_bisect; i++) {
            ind = (ind0 + level0) / 2;
            if (S.HPBGP[ind] > x) level0 = ind;
            else                  ind0 = ind;
        }
        for (ind = ind0; ind <= level0; ind++) {
            if (S.HPBGP[ind] >= x) break;
        }
        ind = ind > 49151 ? 49151 : ind;          // the host guarantees HPBGP[49151] > 1; never read past the map
    }
    return ind;
}

// Creation of a SimRAM_HP packet (kernel_ASOC.c:878-947): direction from the sky pixel, entry
// face chosen with probability proportional to |DIR_i|, uniform position on that face.
template <bool OCT, typename W>
__device__ __forceinline__ void soc_hp_create(const SocGrid &G, const SocSim &S, const int *sOFF, W &w)
{
    const int NX = G.NX, NY = G.NY, NZ = G.NZ;
    const int pix = soc_hp_select_pixel(S, &w.rng, 10);
    w.photons = S.HPBG[pix];
    float phi, theta, st, ct, sp, cp;
    soc_pixel2angles_ring(64, pix, phi, theta);
    soc_sincosf(theta, &st, &ct);
    soc_sincosf(phi, &sp, &cp);
    w.ux = +st * cp;
    w.uy = +st * sp;
    w.uz = -ct;
    if (soc_fabsf(w.ux) < SOC_DEPS) w.ux = SOC_DEPS;
    if (soc_fabsf(w.uy) < SOC_DEPS) w.uy = SOC_DEPS;
    if (soc_fabsf(w.uz) < SOC_DEPS) w.uz = SOC_DEPS;
    soc_normalize(w.ux, w.uy, w.uz);
    float x = soc_fabsf(w.ux), y = soc_fabsf(w.uy), z = soc_fabsf(w.uz);
    float ds = x + y + z;
    x /= ds;  y /= ds;  z /= ds;
    ds = soc_rand(&w.rng);
    const float v1 = soc_rand(&w.rng), v2 = soc_rand(&w.rng);
    if (ds < x) {
        w.py = v1 * NY;  w.pz = v2 * NZ;
        w.px = (w.ux > 0.0f) ? SOC_PEPS : (NX - SOC_PEPS);
    } else if (ds < (x + y)) {
        w.px = v1 * NX;  w.pz = v2 * NZ;
        w.py = (w.uy > 0.0f) ? SOC_PEPS : (NY - SOC_PEPS);
    } else {
        w.px = v1 * NX;  w.py = v2 * NY;
        w.pz = (w.uz > 0.0f) ? SOC_PEPS : (NZ - SOC_PEPS);
    }
    (void)z;
    soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
}

// Creation of a packet of the scattered-light SimRAM_HP (kernel_ASOC_sca.c:107-222): direction from the
// sky pixel (12 bisections when weighted), start on a disc of radius Rout across the direction on the
// upstream side of the cloud, entry through Surface(); w.ind < 0 when the packet misses the cloud.
template <bool OCT, typename W>
__device__ __forceinline__ void soc_hp_sca_create(const SocGrid &G, const SocSim &S, const int *sOFF, W &w)
{
    const int NX = G.NX, NY = G.NY, NZ = G.NZ;
    const float Rout = 0.5f * soc_sqrtf(1.0f * NX * NX + NY * NY + NZ * NZ);
    const int pix = soc_hp_select_pixel(S, &w.rng, 12);
    w.photons = S.HPBG[pix];
    float phi, theta, st, ct, sp, cp;
    soc_pixel2angles_ring(64, pix, phi, theta);
    soc_sincosf(theta, &st, &ct);
    soc_sincosf(phi, &sp, &cp);
    w.ux = +st * cp;
    w.uy = +st * sp;
    w.uz = -ct;
    if (soc_fabsf(w.ux) < SOC_DEPS) w.ux = SOC_DEPS;
    if (soc_fabsf(w.uy) < SOC_DEPS) w.uy = SOC_DEPS;
    if (soc_fabsf(w.uz) < SOC_DEPS) w.uz = SOC_DEPS;
    soc_normalize(w.ux, w.uy, w.uz);
    const float ds = 2.0f * SOC_PI * soc_rand(&w.rng);
    const float dx = soc_sqrtf(soc_rand(&w.rng));
    float sd, cd;
    soc_sincosf(ds, &sd, &cd);
    float x = dx * cd, y = dx * sd, z = soc_sqrtf(1.001f - dx * dx);
    const float x0 = x * ct + z * st, y0 = y, z0 = -x * st + z * ct;
    float sq, cq;
    soc_sincosf(SOC_PI - phi, &sq, &cq);
    x = x0 * cq + y0 * sq;
    y = -x0 * sq + y0 * cq;
    z = z0;
    w.px = 0.5f * NX + Rout * x;
    w.py = 0.5f * NY + Rout * y;
    w.pz = 0.5f * NZ + Rout * z;
    soc_surface(G, w.px, w.py, w.pz, w.ux, w.uy, w.uz);
    soc_indexg<OCT>(G, sOFF, w.px, w.py, w.pz, w.level, w.ind, w.dens);
}

#endif  // SOC_WALK_H
