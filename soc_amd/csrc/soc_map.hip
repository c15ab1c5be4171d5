// soc_map.hip -- map making: Mapping and HealpixMapping of kernel_ASOC_map.c (:496-888, :890-970) for
// gfx950.  One lane per map pixel integrates emission x extinction along its line
// of sight through the (hierarchical) grid.
//
// The reference's map file has its own copies of the traversal helpers, and they are not the ones of the
// simulation kernels: PEPS = 5e-4 and EPS = 2.5e-4 (:10-11), Index() in double whenever NX > 100 (:297),
// and a climb that stops below the root grid only when the local z is exactly 0 (the test at :345 reads
// "POS.z<=0.0") -- so after every step out of an octet the position is rebuilt from the root.  That
// changes the last bits of positions and step lengths, hence of the maps; it is restated here as written
// (soc_map_index) and pinned bit-exactly by the x86 build of the reference (oracle/_ref/refmap_*.so).
// -D MAP_INTERPOLATION, ROI_MAP and LEVEL_THRESHOLD are launch arguments here; the polarisation kernels are not covered.
#include "soc_walk.h"

#define SOC_MAP_PEPS 5.0e-4f
#define SOC_MAP_EPS  2.5e-4f


template <bool OCT, typename T>
__device__ __forceinline__ void soc_map_index(const SocGrid &G, const int *sOFF, float &px, float &py, float &pz,
                                              int &level, int &ind, float &dens)
{
    const int NX = G.NX, NY = G.NY, NZ = G.NZ;
    if (!OCT || (level == 0)) {
        if ((px <= 0.0f) || (px >= NX) || (py <= 0.0f) || (py >= NY) || (pz <= 0.0f) || (pz >= NZ)) { ind = -1;  return; }
        ind  = (int)soc_floorf(pz) * NX * NY + (int)soc_floorf(py) * NX + (int)soc_floorf(px);
        dens = G.DENS[ind];
        if (!OCT) return;
        if (dens > 0.0f) return;
    }
    if (OCT) {
        T PX = px, PY = py, PZ = pz;
        const T HALF = (T)0.5, TWO = (T)2.0, ZERO = (T)0.0;
        while (level > 0) {
            ind = G.PAR[sOFF[level] + ind - G.NXYZ];
            level--;
            PX *= HALF;  PY *= HALF;  PZ *= HALF;
            if (level == 0) {
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
                if (dens > 0.0f) { px = (float)PX;  py = (float)PY;  pz = (float)PZ;  return; }
                break;
            } else {
                const int sid = ind % 8;
                PX += sid % 2;  PY += (sid / 2) % 2;  PZ += sid / 4;
                // kernel_ASOC_map.c:345, as written: "... &&(POS.z>=0.0)&&(POS.z<=0.0)"
                if ((PX >= ZERO) && (PX <= TWO) && (PY >= ZERO) && (PY <= TWO) && (PZ >= ZERO) && (PZ <= ZERO)) {
                    dens = G.DENS[sOFF[level] + ind];
                    break;
                }
            }
        }
        while (!(dens > 0.0f)) {
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

template <bool OCT, bool DBL>
__device__ __forceinline__ float soc_map_getstep(const SocGrid &G, const int *sOFF, float &px, float &py, float &pz,
                                                 float ux, float uy, float uz, int &level, int &ind, float &dens)
{
    const float ax = (ux > 0.0f) ? (((1.0f + SOC_MAP_PEPS) - soc_fmod1f(px)) / ux) : ((-SOC_MAP_PEPS - soc_fmod1f(px)) / ux);
    const float ay = (uy > 0.0f) ? (((1.0f + SOC_MAP_PEPS) - soc_fmod1f(py)) / uy) : ((-SOC_MAP_PEPS - soc_fmod1f(py)) / uy);
    const float az = (uz > 0.0f) ? (((1.0f + SOC_MAP_PEPS) - soc_fmod1f(pz)) / uz) : ((-SOC_MAP_PEPS - soc_fmod1f(pz)) / uz);
    float s = __builtin_fminf(ax, __builtin_fminf(ay, az));
    px += s * ux;
    py += s * uy;
    pz += s * uz;
    s = soc_scale_down(s, level);
    if (DBL) soc_map_index<OCT, double>(G, sOFF, px, py, pz, level, ind, dens);
    else     soc_map_index<OCT, float>(G, sOFF, px, py, pz, level, ind, dens);
    return s;
}

__device__ __forceinline__ bool soc_map_outside(const SocGrid &G, float x, float y, float z)
{
    return (x < 0.0f) || (x > G.NX) || (y < 0.0f) || (y > G.NY) || (z < 0.0f) || (z > G.NZ);
}

// InRoi (kernel_ASOC_map.c:37-56): is the root cell above cell (level, ind) inside ROI?
template <bool OCT>
__device__ __forceinline__ bool soc_map_inroi(const SocGrid &G, const int *sOFF, const int *ROI, int level, int ind)
{
    if (OCT) { while (level > 0) { ind = G.PAR[sOFF[level] + ind - G.NXYZ];  level--; } }
    const int k = ind / (G.NX * G.NY), j = (ind / G.NX) % G.NY, i = ind % G.NX;
    return (i >= ROI[0]) && (i <= ROI[1]) && (j >= ROI[2]) && (j <= ROI[3]) && (k >= ROI[4]) && (k <= ROI[5]);
}

// One neighbour of the MAP_INTERPOLATION block (kernel_ASOC_map.c:716-731, :771-788): from the middle of the step the
// distance (cell units) to the next cell along +V, else along -V (V stays flipped), else "none" (0.5, nothing to blend)
template <bool OCT, bool DBL>
__device__ __forceinline__ void soc_map_neighbour(const SocGrid &G, const int *sOFF, const float *EMIT, float p0x, float p0y, float p0z,
                                                  float tx, float ty, float tz, float w, int level0, int ind0, float K,
                                                  float &vx, float &vy, float &vz, float lim, bool second_try_unscaled,
                                                  float &dist, float &ndens, float &nemit)
{
    for (int attempt = 0; attempt < 2; attempt++) {
        int   slevel = level0, sind = ind0;
        float nd = 0.0f;
        if (attempt) { vx = -vx;  vy = -vy;  vz = -vz; }
        float mx = p0x + w * tx, my = p0y + w * ty, mz = p0z + w * tz;
        float a = soc_map_getstep<OCT, DBL>(G, sOFF, mx, my, mz, vx, vy, vz, slevel, sind, nd);
        if (!(attempt && second_try_unscaled)) a = a / K;                 // (:736 has no "b /= K" in the MAP_INTERPOLATION==2 block)
        if ((a <= lim) && (sind >= 0)) { dist = a;  ndens = nd;  nemit = EMIT[sOFF[slevel] + sind];  return; }
    }
    dist = 0.5f;  ndens = 0.0f;  nemit = 0.0f;
}

template <bool OCT, bool DBL, bool ABU>
__global__ __launch_bounds__(256) void soc_map_kernel(const SocGrid G, const SocMapArgs A)
{
    __shared__ int sOFF[SOC_MAXL];
    if (threadIdx.x < SOC_MAXL) sOFF[threadIdx.x] = G.OFF[threadIdx.x];
    __syncthreads();
    const int npix = A.mode ? 12 * A.NPIX_X * A.NPIX_X : A.NPIX_X * A.NPIX_Y;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= npix) return;
    const int NX = G.NX, NY = G.NY, NZ = G.NZ;
    float TAU = 0.0f, PHOTONS = 0.0f, colden = 0.0f;
    float px, py, pz, tx, ty, tz;
    if (A.mode) {
        // HealpixMapping: all-sky map seen from INTOBS (:913-925)
        float phi, theta, st, ct, sp, cp;
        soc_pixel2angles_ring(A.NPIX_X, id, phi, theta);
        soc_sincosf(theta, &st, &ct);
        soc_sincosf(phi, &sp, &cp);
        tx = -st * cp;  ty = -st * sp;  tz = +ct;
        if (soc_fabsf(tx) < 1.0e-5f) tx = 1.0e-5f;
        if (soc_fabsf(ty) < 1.0e-5f) ty = 1.0e-5f;
        if (soc_fabsf(tz) < 1.0e-5f) tz = 1.0e-5f;
        px = A.INTOBS[0];  py = A.INTOBS[1];  pz = A.INTOBS[2];
        if ((soc_fmod1f(px) < 1.0e-5f) || (soc_fmod1f(px) < 0.99999f)) px += 2.0e-5f;       // as written
        if ((soc_fmod1f(py) < 1.0e-5f) || (soc_fmod1f(py) < 0.99999f)) py += 2.0e-5f;
        if ((soc_fmod1f(pz) < 1.0e-5f) || (soc_fmod1f(pz) < 0.99999f)) pz += 2.0e-5f;
    } else {
        const int i = id % A.NPIX_X, j = id / A.NPIX_X;
        if (A.INTOBS[0] > -1e10f) {
            // longitude x latitude image seen from inside the model (:534-556)
            float phi = SOC_TWOPI * i / (float)(A.NPIX_X);
            phi += SOC_PI;
            const float pix = SOC_TWOPI / A.NPIX_X;
            const float theta = pix * (j - (A.NPIX_Y - 1) / 2);
            float st, ct, sp, cp;
            soc_sincosf(theta, &st, &ct);
            soc_sincosf(phi, &sp, &cp);
            px = A.INTOBS[0];  py = A.INTOBS[1];  pz = A.INTOBS[2];
            tx = ct * cp;  ty = ct * sp;  tz = st;
            if (soc_fabsf(tx) < 1.0e-5f) tx = 1.0e-5f;
            if (soc_fabsf(ty) < 1.0e-5f) ty = 1.0e-5f;
            if (soc_fabsf(tz) < 1.0e-5f) tz = 1.0e-5f;
            if (soc_fmod1f(px) < 1.0e-5f) px += 2.0e-5f;
            if (soc_fmod1f(py) < 1.0e-5f) py += 2.0e-5f;
            if (soc_fmod1f(pz) < 1.0e-5f) pz += 2.0e-5f;
        } else {
            // orthographic map: start behind the cloud as seen by the observer, enter through the far faces (:557-640)
            const float dx = A.DIR[0], dy = A.DIR[1], dz = A.DIR[2];
            px = A.CENTRE[0] + (i - 0.5f * (A.NPIX_X - 1)) * A.MAP_DX * A.RA[0] + (j - 0.5f * (A.NPIX_Y - 1)) * A.MAP_DX * A.DE[0];
            py = A.CENTRE[1] + (i - 0.5f * (A.NPIX_X - 1)) * A.MAP_DX * A.RA[1] + (j - 0.5f * (A.NPIX_Y - 1)) * A.MAP_DX * A.DE[1];
            pz = A.CENTRE[2] + (i - 0.5f * (A.NPIX_X - 1)) * A.MAP_DX * A.RA[2] + (j - 0.5f * (A.NPIX_Y - 1)) * A.MAP_DX * A.DE[2];
            px += (NX + NY + NZ) * dx;  py += (NX + NY + NZ) * dy;  pz += (NX + NY + NZ) * dz;
            float sx, sy, sz;
            if (NX < 200) {
                if (dx >= 0.0f) sx = (NX - px) / (-dx) + SOC_MAP_EPS;  else sx = (0.0f - px) / (-dx) + SOC_MAP_EPS;
                if (dy >= 0.0f) sy = (NY - py) / (-dy) + SOC_MAP_EPS;  else sy = (0.0f - py) / (-dy) + SOC_MAP_EPS;
                if (dz >= 0.0f) sz = (NZ - pz) / (-dz) + SOC_MAP_EPS;  else sz = (0.0f - pz) / (-dz) + SOC_MAP_EPS;
                if (soc_map_outside(G, px - sx * dx, py - sx * dy, pz - sx * dz)) sx = 1e10f;
                if (soc_map_outside(G, px - sy * dx, py - sy * dy, pz - sy * dz)) sy = 1e10f;
                if (soc_map_outside(G, px - sz * dx, py - sz * dy, pz - sz * dz)) sz = 1e10f;
                sx = __builtin_fminf(sx, __builtin_fminf(sy, sz));
                px = px - sx * dx;  py = py - sx * dy;  pz = pz - sx * dz;
            } else {
                const float ex = (dx > 0.0f) ? (-SOC_MAP_EPS) : (+SOC_MAP_EPS), ey = (dy > 0.0f) ? (-SOC_MAP_EPS) : (+SOC_MAP_EPS),
                            ez = (dz > 0.0f) ? (-SOC_MAP_EPS) : (+SOC_MAP_EPS);
                if (dx >= 0.0f) sx = (NX - px) / (-dx);  else sx = (0.0f - px) / (-dx);
                if (dy >= 0.0f) sy = (NY - py) / (-dy);  else sy = (0.0f - py) / (-dy);
                if (dz >= 0.0f) sz = (NZ - pz) / (-dz);  else sz = (0.0f - pz) / (-dz);
                if (soc_map_outside(G, (px - sx * dx) + ex, (py - sx * dy) + ey, (pz - sx * dz) + ez)) sx = 1e10f;
                if (soc_map_outside(G, (px - sy * dx) + ex, (py - sy * dy) + ey, (pz - sy * dz) + ez)) sy = 1e10f;
                if (soc_map_outside(G, (px - sz * dx) + ex, (py - sz * dy) + ey, (pz - sz * dz) + ez)) sz = 1e10f;
                sx = __builtin_fminf(sx, __builtin_fminf(sy, sz));
                px = px - sx * dx;  py = py - sx * dy;  pz = pz - sx * dz;
                px += ex;  py += ey;  pz += ez;
            }
            tx = -dx;  ty = -dy;  tz = -dz;
            if (soc_fabsf(tx) < 1.0e-5f) tx = 1.0e-5f;
            if (soc_fabsf(ty) < 1.0e-5f) ty = 1.0e-5f;
            if (soc_fabsf(tz) < 1.0e-5f) tz = 1.0e-5f;
        }
    }
    int   level = 0, ind = -1;
    float dens = 0.0f;
    soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens);
    const int MI = A.MAPINT;
    float adx = 0.0f, ady = 0.0f, adz = 0.0f, bdx = 0.0f, bdy = 0.0f, bdz = 0.0f;
    if (MI > 0) {                                                         // two directions across the ray (:664-682)
        if (soc_fabsf(tx) > soc_fabsf(ty)) {
            if (soc_fabsf(tz) > soc_fabsf(tx)) { adx = 0.0005f;  ady = 1.0f;  adz = -ty / tz; }
            else                               { adx = -tz / tx;  ady = 0.0005f;  adz = 1.0f; }
        } else {
            if (soc_fabsf(tz) > soc_fabsf(ty)) { adx = 0.0005f;  ady = 1.0f;  adz = -ty / tz; }
            else                               { adx = 1.0f;  ady = -tx / ty;  adz = 0.0005f; }
        }
        soc_normalize(adx, ady, adz);
        bdx = ty * adz - tz * ady;
        bdy = tz * adx - tx * adz;
        bdz = tx * ady - ty * adx;
        soc_normalize(bdx, bdy, bdz);
    }
    while (ind >= 0) {
        const int   oind = sOFF[level] + ind;
        const int   olevel = level;
        const float p0x = px, p0y = py, p0z = pz;
        const int   ind0 = ind;
        float d0 = dens;
        float sx = soc_map_getstep<OCT, DBL>(G, sOFF, px, py, pz, tx, ty, tz, level, ind, dens);
        float emit = A.EMIT[oind];
        if (MI > 0) {
            const float K = soc_scale_down(1.0f, olevel);                 // local -> root-grid length
            float a, b, Ad, Bd, Ae, Be;
            if (MI == 2) {                                                // steps of at most 0.22 cells (:709-715)
                a = 0.22f * K;
                if (sx > a) {
                    sx = a;
                    px = p0x + 0.22f * tx;  py = p0y + 0.22f * ty;  pz = p0z + 0.22f * tz;
                    ind = ind0;  level = olevel;
                    if (DBL) soc_map_index<OCT, double>(G, sOFF, px, py, pz, level, ind, dens);
                    else     soc_map_index<OCT, float>(G, sOFF, px, py, pz, level, ind, dens);
                }
            }
            const float w = 0.5f * sx / K;
            const float lim = (MI == 2) ? 0.52f : 0.502f;
            soc_map_neighbour<OCT, DBL>(G, sOFF, A.EMIT, p0x, p0y, p0z, tx, ty, tz, w, olevel, ind0, K, adx, ady, adz, lim, false, a, Ad, Ae);
            soc_map_neighbour<OCT, DBL>(G, sOFF, A.EMIT, p0x, p0y, p0z, tx, ty, tz, w, olevel, ind0, K, bdx, bdy, bdz, lim, MI == 2, b, Bd, Be);
            if (MI == 2) {                                                // :746-751
                a = soc_clampf(a, 0.0f, 0.51f);
                b = soc_clampf(b, 0.0f, 0.51f);
                const float c0 = 0.5f - a, c1 = 0.5f - b, c2 = a + b;
                emit = c0 * Ae + c1 * Be + c2 * emit;
                d0   = c0 * Ad + c1 * Bd + c2 * d0;
            } else {                                                      // :806-808
                a = 0.5f - a;  b = 0.5f - b;
                const float c2 = 1.0f - a - b;
                emit = c2 * emit + a * Ae + b * Be;
                d0   = c2 * d0 + a * Ad + b * Bd;
            }
        }
        float DTAU;
        if (ABU) { const float2 o = A.OPT[oind];  DTAU = sx * d0 * (o.x + o.y); }
        else     DTAU = sx * d0 * (A.SCA + A.ABS);
        if (A.ROI_MAP && !soc_map_inroi<OCT>(G, sOFF, A.ROI, olevel, oind - sOFF[olevel])) { }   // `roimap`: cells outside ROI do not emit
        else if (!A.mode && (olevel < A.LEVEL_THRESHOLD)) { }             // `threshold`: coarse levels do not emit (they still absorb); Mapping only
        else if (DTAU < 1.0e-3f) PHOTONS += soc_expf(-TAU) * (1.0f - 0.5f * DTAU) * sx * emit * d0;
        else                     PHOTONS += soc_expf(-TAU) * ((1.0f - soc_expf(-DTAU)) / DTAU) * sx * emit * d0;
        TAU += DTAU;
        if (A.mode || (A.SAVE_COLDEN > 0)) colden += sx * d0;
    }
    A.MAP[id] = PHOTONS;
    A.SAVETAU[id] = A.SAVE_COLDEN ? (colden * A.LENGTH) : TAU;
}

// PSTau (kernel_ASOC_map.c:1545-1584): column density and optical depth from every point source towards the observer
template <bool OCT, bool DBL, bool ABU>
__global__ __launch_bounds__(64) void soc_pstau_kernel(const SocGrid G, const int no, const float4 *PSPOS, const float ux, const float uy, const float uz,
                                                       const float ABS, const float SCA, const float2 *OPT, const float LENGTH, float *pscolden, float *pstau)
{
    __shared__ int sOFF[SOC_MAXL];
    if (threadIdx.x < SOC_MAXL) sOFF[threadIdx.x] = G.OFF[threadIdx.x];
    __syncthreads();
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= no) return;
    float px = PSPOS[id].x, py = PSPOS[id].y, pz = PSPOS[id].z, TAU = 0.0f, colden = 0.0f, dens = 0.0f;
    int   level = 0, ind = -1;
    soc_indexg<OCT>(G, sOFF, px, py, pz, level, ind, dens);
    while (ind >= 0) {
        const int   oind = sOFF[level] + ind;
        const float d0 = dens;
        const float sx = soc_map_getstep<OCT, DBL>(G, sOFF, px, py, pz, ux, uy, uz, level, ind, dens);
        float DTAU;
        if (ABU) { const float2 o = OPT[oind];  DTAU = sx * d0 * (o.x + o.y); }
        else     DTAU = sx * d0 * (SCA + ABS);
        TAU += DTAU;
        colden += sx * d0;
    }
    pscolden[id] = colden * LENGTH;
    pstau[id]    = TAU;
}

hipError_t soc_launch_pstau(const SocGrid &G, int no, const float4 *PSPOS, const float *DIR, float ABS, float SCA, const float2 *OPT, float LENGTH,
                            float *pscolden, float *pstau, hipStream_t st)
{
    if (no <= 0) return hipSuccess;
    const dim3 grid((no + 63) / 64), block(64);
    const bool oct = G.LEVELS > 1, dbl = oct && (G.NX > 100), abu = OPT != nullptr;
#define SOC_PT(O, D, A) soc_pstau_kernel<O, D, A><<<grid, block, 0, st>>>(G, no, PSPOS, DIR[0], DIR[1], DIR[2], ABS, SCA, OPT, LENGTH, pscolden, pstau)
    if (!oct)      { if (abu) SOC_PT(false, false, true); else SOC_PT(false, false, false); }
    else if (!dbl) { if (abu) SOC_PT(true, false, true);  else SOC_PT(true, false, false); }
    else           { if (abu) SOC_PT(true, true, true);   else SOC_PT(true, true, false); }
#undef SOC_PT
    return hipGetLastError();
}

hipError_t soc_launch_map(const SocGrid &G, const SocMapArgs &A, bool abu, hipStream_t st)
{
    const int npix = A.mode ? 12 * A.NPIX_X * A.NPIX_X : A.NPIX_X * A.NPIX_Y;
    if (npix <= 0) return hipSuccess;
    const dim3 grid((npix + 255) / 256), block(256);
    const bool oct = G.LEVELS > 1, dbl = oct && (G.NX > 100);            // kernel_ASOC_map.c:297
    if (!oct)      { if (abu) soc_map_kernel<false, false, true><<<grid, block, 0, st>>>(G, A); else soc_map_kernel<false, false, false><<<grid, block, 0, st>>>(G, A); }
    else if (!dbl) { if (abu) soc_map_kernel<true, false, true><<<grid, block, 0, st>>>(G, A);  else soc_map_kernel<true, false, false><<<grid, block, 0, st>>>(G, A); }
    else           { if (abu) soc_map_kernel<true, true, true><<<grid, block, 0, st>>>(G, A);   else soc_map_kernel<true, true, false><<<grid, block, 0, st>>>(G, A); }
    return hipGetLastError();
}
