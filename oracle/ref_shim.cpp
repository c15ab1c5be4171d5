// ref_shim.cpp -- glue that lets the reference's OpenCL C kernels, compiled UNMODIFIED from
// /root/reference for x86-64 (oracle/build_ref.py), run on the host.  TEST INFRASTRUCTURE ONLY.
//
// Two parts, both ours (nothing here is taken from the reference):
//   1. the OpenCL built-in functions the kernel object leaves unresolved, defined 1:1 on top
//      of glibc libm / compiler builtins under their Itanium-mangled OpenCL names;
//   2. a driver that executes work items one after another (or id ranges on threads), and
//      thin wrappers around the reference's helper functions for known-answer tests.
//
// Device-defined choices (the OpenCL spec leaves these to the device, so the reference has
// no single answer): normalize(v) = v * (1/sqrt(x*x+y*y+z*z)); sincos = (sinf, cosf);
// float atomics = CAS loop exactly as kernel_ASOC_aux.c:77-93 writes it.  The oracle's libm
// build (oracle/soc_oracle.c, -DSOC_ORACLE_LIBM) makes the same choices, which is what lets
// tests/test_oracle_vs_ref.py demand bit-identical results.
//
// Must be compiled with the same clang that compiles the kernels (ext_vector_type ABI).
#include "ref_builtins.inc"

// ---- 2. reference symbols (kernel_ASOC.c, kernel_ASOC_aux.c, mwc64x_rng.cl) --------------
struct mwc_state { uint x, c; };
extern "C" {
void MWC64X_SeedStreams(mwc_state *s, unsigned long baseOffset, unsigned long perStreamOffset);
uint MWC64X_NextUint(mwc_state *s);
void IndexG(float3 *pos, int *level, int *ind, float *DENS, int *OFF);
float GetStep(float3 *POS, const float3 *DIR, int *level, int *ind, float *DENS, int *OFF, int *PAR);
void Deflect(float3 *DIR, const float COS_THETA, const float phi);
void Scatter(float3 *DIR, float *CSC, mwc_state *rng);
void Parents(float *DENS, int *LCELLS, int *OFF, int *PAR);
void ZeroAMC(int tag, float *TABS, float *XAB, float *INT, float *INTX, float *INTY, float *INTZ);
void SimRAM_PB(int SOURCE, int PACKETS, int BATCH, float SEED, float *ABS, float *SCA, float BG,
               float3 *PSPOS, float *PS, float TW, int *LCELLS, int *OFF, int *PAR, float *DENS,
               float *EMIT, float *TABS, float *DSC, float *CSC, float *XAB, float *EMWEI,
               float *INT, float *INTX, float *INTY, float *INTZ, float *OPT, float *ABU,
               int *XPS_NSIDE, int *XPS_SIDE, float *XPS_AREA
#if REF_ROI_LOAD
               , int *ROI_DIM, float *ROI_LOAD
#endif
#if REF_ROI_SAVE
               , int *ROI, float *ROI_SAVE
#endif
               );
void EqTemperature(int level, float adhoc, float kE, float Emin, int NE, int *OFF, int *LCELLS, float *TTT, float *DENS,
                   float *EMIT, float *TNEW);
void Emission2(int c0, int c1, int nfreq, float *FREQ, float *FABS, float *DENS, float *T, float *EMIT);
void SimRAM_HP(int PACKETS, int BATCH, float SEED, float *ABS, float *SCA, float TW, int *LCELLS, int *OFF, int *PAR,
               float *DENS, float *EMIT, float *TABS, float *DSC, float *CSC, float *XAB, float *INT, float *INTX,
               float *INTY, float *INTZ, float *OPT, float *BG, float *HPBGP, float *ABU);
void SimRAM_CL(int SOURCE, int PACKETS, int BATCH, float SEED, float *ABS, float *SCA, float TW,
               int *LCELLS, int *OFF, int *PAR, float *DENS, float *EMIT, float *TABS, float *DSC,
               float *CSC, float *XAB, float *EMWEI, float *INT, float *INTX, float *INTY,
               float *INTZ, int *EMINDEX, float *OPT, float *ABU
#if REF_ROI_SAVE
               , int *ROI, float *ROI_SAVE
#endif
               );
}

struct ref_args {
    int   SOURCE, PACKETS, BATCH, GLOBAL;
    float SEED, BG, TW;
    float *ABS, *SCA;
    float *PSPOS;            // 4 floats per source
    float *PS;
    int   *LCELLS, *OFF, *PAR;
    float *DENS, *EMIT, *TABS, *DSC, *CSC, *XAB, *EMWEI, *INT, *INTX, *INTY, *INTZ, *OPT, *ABU;
    int   *XPS_NSIDE, *XPS_SIDE;
    float *XPS_AREA;
    int   *EMINDEX;
    float *HPBG, *HPBGP;     // Healpix sky of the current frequency, cumulative pixel probability
    int   *ROI_DIM;  float *ROI_LOAD;   // WITH_ROI_LOAD builds
    int   *ROI;      float *ROI_SAVE;   // WITH_ROI_SAVE builds
};

static void run_range(const ref_args *a, int kind, int gid0, int gid1, int stride = 1)
{
    g_gsize = (size_t)a->GLOBAL;
    for (int id = gid0; id < gid1; id += stride) {
        g_gid = (size_t)id;
        if (kind == 0)
            SimRAM_PB(a->SOURCE, a->PACKETS, a->BATCH, a->SEED, a->ABS, a->SCA, a->BG,
                      (float3 *)a->PSPOS, a->PS, a->TW, a->LCELLS, a->OFF, a->PAR, a->DENS, a->EMIT,
                      a->TABS, a->DSC, a->CSC, a->XAB, a->EMWEI, a->INT, a->INTX, a->INTY, a->INTZ,
                      a->OPT, a->ABU, a->XPS_NSIDE, a->XPS_SIDE, a->XPS_AREA
#if REF_ROI_LOAD
                      , a->ROI_DIM, a->ROI_LOAD
#endif
#if REF_ROI_SAVE
                      , a->ROI, a->ROI_SAVE
#endif
                      );
        else if (kind == 2)
            SimRAM_HP(a->PACKETS, a->BATCH, a->SEED, a->ABS, a->SCA, a->TW, a->LCELLS, a->OFF, a->PAR, a->DENS, a->EMIT,
                      a->TABS, a->DSC, a->CSC, a->XAB, a->INT, a->INTX, a->INTY, a->INTZ, a->OPT, a->HPBG, a->HPBGP, a->ABU);
        else
            SimRAM_CL(a->SOURCE, a->PACKETS, a->BATCH, a->SEED, a->ABS, a->SCA, a->TW, a->LCELLS,
                      a->OFF, a->PAR, a->DENS, a->EMIT, a->TABS, a->DSC, a->CSC, a->XAB, a->EMWEI,
                      a->INT, a->INTX, a->INTY, a->INTZ, a->EMINDEX, a->OPT, a->ABU
#if REF_ROI_SAVE
                      , a->ROI, a->ROI_SAVE
#endif
                      );
    }
}

extern "C" {

// Execute work items gid0, gid0+stride, ... < gid1 of SimRAM_PB (kind 0) / SimRAM_CL (kind 1) / SimRAM_HP (kind 2).
// stride > 1 samples a launch evenly (bench.py's bounded CPU baseline).
void ref_sim(const ref_args *a, int kind, int gid0, int gid1, int stride, int nthreads)
{
    if (stride < 1) stride = 1;
    if (nthreads <= 1) {
        run_range(a, kind, gid0, gid1, stride);
        return;
    }
    std::vector<std::thread> th;
    // interleaved blocks of 64 sampled ids keep the threads balanced
    const long blk = 64L * stride;
    for (int t = 0; t < nthreads; t++) {
        th.emplace_back([=]() {
            for (long b = gid0 + blk * t; b < gid1; b += blk * nthreads)
                run_range(a, kind, (int)b, (int)((b + blk < gid1) ? (b + blk) : gid1), stride);
        });
    }
    for (auto &x : th) x.join();
}

// EqTemperature for every level, Emission2 for cells [c0, c1): one work item walks the whole grid-stride loop
void ref_eqtemp(int LEVELS, float adhoc, float kE, float Emin, int NE, int *OFF, int *LCELLS, float *TTT, float *DENS,
                float *EABS, float *TNEW)
{
    g_gid = 0;  g_gsize = 1;
    for (int l = 0; l < LEVELS; l++) EqTemperature(l, adhoc, kE, Emin, NE, OFF, LCELLS, TTT, DENS, EABS, TNEW);
}

void ref_emission2(int c0, int c1, int nfreq, float *FREQ, float *FABS, float *DENS, float *T, float *EMIT)
{
    g_gid = 0;  g_gsize = 1;
    Emission2(c0, c1, nfreq, FREQ, FABS, DENS, T, EMIT);
}

void ref_parents(float *DENS, int *LCELLS, int *OFF, int *PAR)
{
    g_gid = 0;  g_gsize = 1;
    Parents(DENS, LCELLS, OFF, PAR);
}

void ref_seed(float SEED, unsigned long gid, uint *x, uint *c)
{
    // kernel_ASOC.c:74-77 -- the expression is restated here because it is inline in the kernel
    g_gid = gid;  g_gsize = gid + 1;
    mwc_state s;
    MWC64X_SeedStreams(&s, (unsigned long)(fmodf(SEED * 7.0f * 3.1415926535897f, 1.0f) * 4294967296L),
                       274877906944UL);
    *x = s.x;  *c = s.c;
}

void ref_draws(uint *x, uint *c, int n, uint *out)
{
    mwc_state s = { *x, *c };
    for (int i = 0; i < n; i++) out[i] = MWC64X_NextUint(&s);
    *x = s.x;  *c = s.c;
}

void ref_indexg(float *pos, int *level, int *ind, float *DENS, int *OFF)
{
    float3 p;  p.x = pos[0];  p.y = pos[1];  p.z = pos[2];
    IndexG(&p, level, ind, DENS, OFF);
    pos[0] = p.x;  pos[1] = p.y;  pos[2] = p.z;
}

int ref_trace(const float *pos, const float *dir, int maxsteps, float *DENS, int *OFF, int *PAR,
              int *levels, int *inds, float *dss, float *endpos)
{
    float3 P, D;
    P.x = pos[0];  P.y = pos[1];  P.z = pos[2];
    D.x = dir[0];  D.y = dir[1];  D.z = dir[2];
    int level = 0, ind = -1, n = 0;
    IndexG(&P, &level, &ind, DENS, OFF);
    while ((ind >= 0) && (n < maxsteps)) {
        levels[n] = level;
        inds[n]   = ind;
        dss[n]    = GetStep(&P, &D, &level, &ind, DENS, OFF, PAR);
        n++;
    }
    endpos[0] = P.x;  endpos[1] = P.y;  endpos[2] = P.z;
    return n;
}

void ref_scatter(float *dir, float *CSC, uint *x, uint *c)
{
    float3 D;  D.x = dir[0];  D.y = dir[1];  D.z = dir[2];
    mwc_state s = { *x, *c };
    Scatter(&D, CSC, &s);
    dir[0] = D.x;  dir[1] = D.y;  dir[2] = D.z;
    *x = s.x;  *c = s.c;
}

void ref_deflect(float *dir, float cos_theta, float phi)
{
    float3 D;  D.x = dir[0];  D.y = dir[1];  D.z = dir[2];
    Deflect(&D, cos_theta, phi);
    dir[0] = D.x;  dir[1] = D.y;  dir[2] = D.z;
}

}  // extern "C"
