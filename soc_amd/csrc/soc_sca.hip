// soc_sca.hip -- scattered-light images (peel-off) for gfx950: SimRAM_PS, SimRAM_PB and
// SimRAM_CL of kernel_ASOC_sca.c (:1462-1938, :471-1094, :1098-1450).
//
// What is computed: packets from point sources / the isotropic background / the cells are
// walked through the grid; at every scattering the packet is "peeled off" towards each of
// NDIR observers (optical depth to the surface along ODIR, weight from the discrete
// scattering function) and added to the pixel of an orthographic image.  With forced first
// scattering (FFS) a look-ahead ray first integrates the optical depth to the surface.
//
// How: a work item of the reference alternates between four loops that all do the same
// thing -- GetStep along a ray and add ds*density*kappa to an optical depth: the FFS
// look-ahead, the free walk, and one loop per observer.  Nested, they leave a 64-wide wave
// at the utilisation of its slowest lane in every loop.  Here they are ONE stepping loop over
// a "current ray" whose meaning is a per-lane mode; the short event blocks in between
// (create, end of look-ahead, start of scattering, end of one peel-off ray) are serviced when
// a ballot shows enough lanes waiting (soc_service_now), as in soc_kernels.hip.  A lane's own
// sequence of operations and RNG draws is exactly the reference's.
//
// Reference behaviours kept on purpose (each pinned bit-exactly by oracle/_ref builds):
//  * ldexp(ds, level) of the scattering offset uses the level AFTER the step (:1008, :1303, :1797);
//  * SimRAM_CL takes the "#ifdef HG_TEST" branch, because kernel_ASOC_aux.c:1 defines HG_TEST
//    (as 0): analytic Henyey-Greenstein g=0.65 and the factor (1-exp(-tau)), DSC unused (:1387-1392);
//  * SimRAM_PS: W=-expm1(-tau) and an fp32 logarithm for the forced free path (:1737-1742); PB/CL
//    evaluate log(1.0-W*u) in fp64 (:906, :1256);
//  * the XPS_* type mismatch is resolved on the host (soc_capi.hip: upload_sources).
#include "soc_walk.h"

enum { SCA_M_FFS = 0, SCA_M_MAIN = 1, SCA_M_PEEL = 2,                 // stepping modes
       SCA_M_CREATE = 3, SCA_M_FFS_END = 4, SCA_M_SCAT = 5, SCA_M_PEEL_END = 6, SCA_M_DONE = 7 };

#define SCA_MAX_SCATTERINGS 30                                        /* kernel_ASOC_sca.c:5 */

// what soc_pb_create fills in
struct ScaRay {
    float px, py, pz, ux, uy, uz, photons, dens;
    int   level, ind;
    soc_rng_t rng;
};

template <bool OCT, bool DBL, bool ABU, int KIND>
__global__ __launch_bounds__(256) void soc_sca_kernel(const SocGrid G, const SocSim S, const SocSca V)
{
    extern __shared__ float lds[];
    float *sCSC = lds;
    float *sDSC = lds + S.BINS;
    int   *sOFF = (int *)(sDSC + S.BINS);
    int   *sLC  = sOFF + SOC_MAXL;
    for (int i = threadIdx.x; i < S.BINS; i += blockDim.x) sDSC[i] = V.DSC[i];
    soc_stage_lds(G, S, sCSC, sOFF, sLC);

    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S.gid_count) return;
    const int id = (int)(S.gid0 + t);
    const int NX = G.NX, NY = G.NY;
    const int AREA = 2 * (G.NX * G.NY + G.NY * G.NZ + G.NZ * G.NX);
    if (KIND == SOC_SCA_PB) { if ((S.SOURCE == 1) && (id >= 8 * AREA)) return; }
    if (KIND == SOC_SCA_CL) { if (id >= G.CELLS) return; }
    // SimRAM_CL and SimRAM_HP share the walk's details: +-0.9999 clamp, HG_TEST weight, no draw on an empty line of sight
    constexpr bool CLW = (KIND == SOC_SCA_CL) || (KIND == SOC_SCA_HP);
    const bool HPX = V.NDIR < 0;                         // Healpix map seen from the position ODIRS[0]
    const int  NDIRS = HPX ? 1 : V.NDIR;

    ScaRay w;                                  // the current ray
    w.rng = soc_seed_stream(S.seed_mul, S.seed_tab, (uint32_t)id);
    w.ind = -1;  w.level = 0;  w.dens = 0.0f;  w.photons = 0.0f;
    w.ux = w.uy = w.uz = 0.0f;  w.px = w.py = w.pz = 0.0f;
    // the packet itself while the current ray is a look-ahead or a peel-off ray
    float mx = 0.0f, my = 0.0f, mz = 0.0f, dx_ = 0.0f, dy_ = 0.0f, dz_ = 0.0f, mdens = 0.0f;
    int   mlevel = 0, mind = -1, lvl_post = 0;
    float free_path = 0.0f, tau = 0.0f, taup = 0.0f;
    float dxrem = 0.0f, invd2 = 0.0f;                    // Healpix: distance left to the observer, 1/d^2
    float rux = 1.0f, ruy = 1.0f, ruz = 1.0f;            // correctly rounded reciprocals of the current ray's direction
    int   scat = 0, idir = 0;
    unsigned int n_add = 0, n_pkt = 0, n_scat = 0;

    SocSurfElem E;
    if (KIND == SOC_SCA_PB || KIND == SOC_SCA_PS) E = soc_surface_element(G, S, id);
    int   III = 0;
    long long ICELL = (long long)id - S.GLOBAL;
    int   IRAY = 0, batch = -1;
    float PWEI = 1.0f;

    int mode = SCA_M_CREATE;
    while (true) {
        const bool stepping = (mode <= SCA_M_PEEL);
        const bool nobody_steps = (__ballot(stepping) == 0ull);
        if (soc_service_now(!stepping && (mode != SCA_M_DONE), nobody_steps)) {
            // ---- start of a scattering event (:1000-1015 PB, :1295-1310 CL, :1789-1805 PS)
            if (mode == SCA_M_SCAT) {
                // w = state at the beginning of the step (restored by the stepping arm)
                const int oind = sOFF[w.level] + w.ind;
                float kabs, ksca;
                if (ABU) { float2 o = S.OPT[oind];  kabs = o.x;  ksca = o.y; }
                else     { kabs = S.ABS;  ksca = S.SCA; }
                scat++;
                n_scat++;
                float dx = (free_path - tau) / (ksca * w.dens);
                dx = soc_scale_up(dx, lvl_post);
                w.px = w.px + dx * w.ux;
                w.py = w.py + dx * w.uy;
                w.pz = w.pz + dx * w.uz;
                w.photons *= soc_expf(-free_path * kabs / ksca);
                // park the packet, aim the current ray at the first observer
                mx = w.px;  my = w.py;  mz = w.pz;  dx_ = w.ux;  dy_ = w.uy;  dz_ = w.uz;
                mlevel = w.level;  mind = w.ind;  mdens = w.dens;
                idir = 0;
                if (HPX) {
                    // direction and distance to the observer from the root position of the scattering (:319-327)
                    float rx = w.px, ry = w.py, rz = w.pz;
                    if (OCT) soc_rootpos(G, sOFF, rx, ry, rz, w.level, w.ind);
                    const float4 o = V.ODIRS[0];
                    w.ux = o.x - rx;  w.uy = o.y - ry;  w.uz = o.z - rz;
                    dxrem = soc_sqrtf(w.ux * w.ux + w.uy * w.uy + w.uz * w.uz);
                    invd2 = 1.0f / (dxrem * dxrem);
                    soc_normalize(w.ux, w.uy, w.uz);
                    taup = 0.0f;
                    mode = (dxrem > 0.0f) ? SCA_M_PEEL : SCA_M_PEEL_END;
                } else if (V.NDIR > 0) {
                    const float4 o = V.ODIRS[0];
                    w.ux = o.x;  w.uy = o.y;  w.uz = o.z;
                    taup = 0.0f;
                    mode = SCA_M_PEEL;
                } else {
                    mode = SCA_M_PEEL_END;                 // no observers: straight to the deflection
                    idir = -1;
                }
            }
            // ---- a peel-off ray has reached the surface: image contribution, next observer or deflection
            if (mode == SCA_M_PEEL_END) {
                if (idir >= 0) {
                    const float CL = CLW ? 0.9999f : 0.999f;
                    const float cos_theta = soc_clampf(dx_ * w.ux + dy_ * w.uy + dz_ * w.uz, -CL, +CL);
                    // -D WITH_MSF: one species per peel-off, its DSC row (kernel_ASOC_sca.c:339-347, :382-390; SimRAM_CL draws too)
                    const int idust = (S.NDUST > 1) ? soc_msf_dust(S, &w.rng, sOFF[mlevel] + mind) : 0;
                    float delta;
                    if (CLW) {
                        const float g = 0.65f;
                        const float fraction = (1.0f / (4.0f * SOC_PI)) * (1.0f - g * g) / soc_pow15f(1.0f + g * g - 2.0f * g * cos_theta);
                        delta = w.photons * fraction * ((taup > SOC_TAULIM) ? (1.0f - soc_expf(-taup)) : (taup * (1.0f - 0.5f * taup)));
                    } else {
                        int b = (int)(S.BINS * (1.0f + cos_theta) * 0.5f);
                        b = b < 0 ? 0 : (b > S.BINS - 1 ? S.BINS - 1 : b);
                        delta = w.photons * soc_expf(-taup) * ((S.NDUST > 1) ? V.DSC[(long)idust * S.BINS + b] : sDSC[b]);
                    }
                    if (HPX) {
                        // 1/d^2 and the pixel of the direction towards the observer (:352-360)
                        delta = invd2 * delta;
                        const float theta = soc_acosf(-w.uz);
                        const float phi   = soc_atan2f(w.uy, w.ux);
                        const int pix = soc_angles2pixel_ring(-V.NDIR, phi, theta);
                        if (pix >= 0) soc_tally(V.OUT, pix, delta);
                        n_add++;
                    } else {
                        const float qx = w.px - V.CX, qy = w.py - V.CY, qz = w.pz - V.CZ;
                        const float4 ra = V.ORA[idir], de = V.ODE[idir];
                        int i = (int)((0.5f * V.NPIX_X - 0.00005f) + (qx * ra.x + qy * ra.y + qz * ra.z) / V.MAP_DX);
                        int j = (int)((0.5f * V.NPIX_Y - 0.00005f) + (qx * de.x + qy * de.y + qz * de.z) / V.MAP_DX);
                        if ((i >= 0) && (j >= 0) && (i < V.NPIX_X) && (j < V.NPIX_Y)) {
                            i += idir * V.NPIX_X * V.NPIX_Y + j * V.NPIX_X;
                            soc_tally(V.OUT, i, delta);
                            n_add++;
                        }
                    }
                    idir++;
                }
                if ((idir >= 0) && (idir < NDIRS)) {
                    const float4 o = V.ODIRS[idir];
                    w.px = mx;  w.py = my;  w.pz = mz;  w.level = mlevel;  w.ind = mind;  w.dens = mdens;
                    w.ux = o.x;  w.uy = o.y;  w.uz = o.z;
                    taup = 0.0f;
                    mode = SCA_M_PEEL;
                } else {
                    // back to the packet at the scattering position; new direction, new free path
                    w.px = mx;  w.py = my;  w.pz = mz;  w.level = mlevel;  w.ind = mind;  w.dens = mdens;
                    w.ux = dx_;  w.uy = dy_;  w.uz = dz_;
                    if (S.NDUST > 1) {                                              // :424-432
                        const int idust = soc_msf_dust(S, &w.rng, sOFF[mlevel] + mind);
                        soc_scatter(w.ux, w.uy, w.uz, S.CSC + (long)idust * S.BINS, S.BINS, &w.rng);
                    } else {
                        soc_scatter(w.ux, w.uy, w.uz, sCSC, S.BINS, &w.rng);
                    }
                    free_path = -soc_logf(soc_rand(&w.rng));
                    tau = 0.0f;
                    mode = (scat == SCA_MAX_SCATTERINGS) ? SCA_M_CREATE : SCA_M_MAIN;
                }
            }
            // ---- the FFS look-ahead has left the cloud (:899-909 PB, :1249-1258 CL, :1733-1745 PS)
            if (mode == SCA_M_FFS_END) {
                // tau = optical depth of scattering along the whole line of sight
                w.px = mx;  w.py = my;  w.pz = mz;  w.level = mlevel;  w.ind = mind;  w.dens = mdens;
                bool alive = true;
                if (tau < 1.0e-22f) {
                    w.ind = -1;
                    if (CLW) alive = false;                            // no random number drawn
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
                mode = (w.ind >= 0) ? SCA_M_MAIN : SCA_M_CREATE;
            }
            // ---- next packet of this work item
            if (mode == SCA_M_CREATE) {
                bool have = false;
                if (KIND == SOC_SCA_CL) {
                    // :1158-1222
                    bool more = true;
                    if (IRAY >= batch) {
                        IRAY = 0;
                        PWEI = 1.0f;
                        while (true) {
                            ICELL += S.GLOBAL;
                            if (ICELL >= G.CELLS) { more = false; break; }
                            if (S.USE_EMWEIGHT > 0) {
                                PWEI = S.EMWEI[ICELL];
                                if ((PWEI < 1e-10f) || (G.DENS[ICELL] <= 0.0f)) continue;
                                batch = (int)soc_floorf(PWEI);
                                if (batch < 1) { batch = 1;  PWEI = (float)(1.0 / (double)(PWEI + 1.0e-30f)); }
                                else           { PWEI = (float)(1.0 / (double)((float)batch + 1.0e-9f)); }
                            } else {
                                batch = S.BATCH;
                                PWEI  = 1.0f / ((float)batch + 1.0e-9f);
                            }
                            break;
                        }
                    }
                    if (!more) {
                        mode = SCA_M_DONE;
                    } else {
                        int ind = (int)ICELL, level;
                        IRAY += 1;
                        for (level = 0; level < G.LEVELS - 1; level++) {
                            ind -= sLC[level];
                            if (ind < 0) { ind += sLC[level]; break; }
                        }
                        float X0, Y0, Z0;
                        if (level == 0) {
                            X0 = (float)(ind % NX);  Y0 = (float)((ind / NX) % NY);  Z0 = (float)(ind / (NX * NY));
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
                } else if (KIND == SOC_SCA_HP) {
                    if (III >= S.BATCH) {
                        mode = SCA_M_DONE;
                    } else {
                        soc_hp_sca_create<OCT>(G, S, sOFF, w);
                        III++;
                        n_pkt++;
                        have = (w.ind >= 0);                   // a packet that misses the cloud is skipped (:222)
                    }
                } else {
                    if (III >= S.BATCH) {
                        mode = SCA_M_DONE;
                    } else {
                        soc_pb_create<OCT>(G, S, sOFF, E, III, w);
                        III++;
                        have = true;
                    }
                }
                if (have) {
                    if (KIND != SOC_SCA_HP) {                  // SimRAM_HP has conditioned the direction itself
                        n_pkt++;
                        if (soc_fabsf(w.ux) < SOC_DEPS) w.ux = SOC_DEPS;
                        if (soc_fabsf(w.uy) < SOC_DEPS) w.uy = SOC_DEPS;
                        if (soc_fabsf(w.uz) < SOC_DEPS) w.uz = SOC_DEPS;
                        soc_normalize(w.ux, w.uy, w.uz);
                    }
                    tau  = 0.0f;
                    scat = 0;
                    if (V.FFS > 0) {
                        mx = w.px;  my = w.py;  mz = w.pz;  mlevel = w.level;  mind = w.ind;  mdens = w.dens;
                        mode = (w.ind >= 0) ? SCA_M_FFS : SCA_M_FFS_END;
                    } else {
                        free_path = -soc_logf(soc_rand(&w.rng));
                        mode = (w.ind >= 0) ? SCA_M_MAIN : SCA_M_CREATE;
                    }
                }
            }
            // every change of the current ray's direction happens in this arm: its reciprocals for GetStep
            if (!stepping) { rux = 1.0f / w.ux;  ruy = 1.0f / w.uy;  ruz = 1.0f / w.uz; }
        }
        if (__ballot(mode != SCA_M_DONE) == 0ull) break;

        // ---- one cell step of the current ray, whatever it is
        if (mode <= SCA_M_PEEL) {
            const int   oind = sOFF[w.level] + w.ind;
            const int   ind0 = w.ind, level0 = w.level;
            const float p0x = w.px, p0y = w.py, p0z = w.pz, d0 = w.dens;
            float kabs, ksca;
            if (ABU) { float2 o = S.OPT[oind];  kabs = o.x;  ksca = o.y; }
            else     { kabs = S.ABS;  ksca = S.SCA; }
            const float ds = soc_getstep_rcp<OCT, DBL>(G, sOFF, w.px, w.py, w.pz, w.ux, w.uy, w.uz, rux, ruy, ruz, w.level, w.ind, w.dens);
            if (mode == SCA_M_PEEL) {
                if (HPX) {
                    // only as far as the observer (:329-335); SimRAM_PB adds its 1.0e-6 in double (:982)
                    float d = (dxrem < ds) ? dxrem : ds;
                    d = (KIND == SOC_SCA_PB) ? (float)((double)d + 1.0e-6) : (d + 1.0e-6f);
                    dxrem -= d;
                    taup += d * d0 * (kabs + ksca);
                    if (!(dxrem > 0.0f) || (w.ind < 0)) mode = SCA_M_PEEL_END;
                } else {
                    taup += ds * d0 * (kabs + ksca);
                    if (w.ind < 0) mode = SCA_M_PEEL_END;
                }
            } else {
                const float dtau = ds * d0 * ksca;
                if (mode == SCA_M_FFS) {
                    tau += dtau;
                    if (w.ind < 0) mode = SCA_M_FFS_END;
                } else if (free_path < (tau + dtau)) {
                    lvl_post = w.level;
                    w.px = p0x;  w.py = p0y;  w.pz = p0z;  w.ind = ind0;  w.level = level0;  w.dens = d0;
                    mode = SCA_M_SCAT;
                } else {
                    tau += dtau;
                    if ((S.MIRROR > 0) && (w.ind < 0))       // kernel_ASOC_sca.c:983, :1283, :1781
                    {
                        soc_mirror<OCT>(G, sOFF, S.MIRROR, w.px, w.py, w.pz, w.ux, w.uy, w.uz, w.level, w.ind, w.dens);
                        rux = 1.0f / w.ux;  ruy = 1.0f / w.uy;  ruz = 1.0f / w.uz;          // a reflected component changed sign
                    }
                    if (w.ind < 0) mode = SCA_M_CREATE;
                }
            }
        }
    }
    if (S.stats) {
        atomicAdd(S.stats + 0, (unsigned long long)n_add);
        atomicAdd(S.stats + 1, (unsigned long long)n_pkt);
        atomicAdd(S.stats + 2, (unsigned long long)n_scat);
    }
}

template <int KIND>
static hipError_t sca_dispatch(const SocGrid &G, const SocSim &S, const SocSca &V, const SocVariant &X, hipStream_t st)
{
    if (S.gid_count == 0) return hipSuccess;
    const dim3   block(256), grid((S.gid_count + 255) / 256);
    const size_t lds = (size_t)(2 * S.BINS) * sizeof(float) + 2 * SOC_MAXL * sizeof(int);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
#define SCA_GO(O, D, A) soc_sca_kernel<O, D, A, KIND><<<grid, block, lds, st>>>(G, S, V)
    if (X.octree) {
        if (X.dbl) { if (X.abu) SCA_GO(true, true, true);  else SCA_GO(true, true, false); }
        else       { if (X.abu) SCA_GO(true, false, true); else SCA_GO(true, false, false); }
    } else {
        if (X.dbl) { if (X.abu) SCA_GO(false, true, true);  else SCA_GO(false, true, false); }
        else       { if (X.abu) SCA_GO(false, false, true); else SCA_GO(false, false, false); }
    }
#undef SCA_GO
    return hipGetLastError();
}

hipError_t soc_launch_sca(const SocGrid &G, const SocSim &S, const SocSca &V, const SocVariant &X, hipStream_t st)
{
    switch (V.kind) {
    case SOC_SCA_PB: return sca_dispatch<SOC_SCA_PB>(G, S, V, X, st);
    case SOC_SCA_CL: return sca_dispatch<SOC_SCA_CL>(G, S, V, X, st);
    case SOC_SCA_PS: return sca_dispatch<SOC_SCA_PS>(G, S, V, X, st);
    case SOC_SCA_HP: return sca_dispatch<SOC_SCA_HP>(G, S, V, X, st);
    default: return hipErrorInvalidValue;
    }
}
