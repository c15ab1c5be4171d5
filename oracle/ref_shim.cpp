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
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

typedef float float3 __attribute__((ext_vector_type(3)));
typedef unsigned int uint;

static thread_local size_t g_gid = 0, g_gsize = 1;

// ---- 1. OpenCL built-ins ---------------------------------------------------------------
#define CLNAME(n) asm(n)
size_t cl_get_global_id(uint) CLNAME("_Z13get_global_idj");
size_t cl_get_global_id(uint) { return g_gid; }
size_t cl_get_global_size(uint) CLNAME("_Z15get_global_sizej");
size_t cl_get_global_size(uint) { return g_gsize; }
size_t cl_get_local_id(uint) CLNAME("_Z12get_local_idj");
size_t cl_get_local_id(uint) { return 0; }
size_t cl_get_group_id(uint) CLNAME("_Z12get_group_idj");
size_t cl_get_group_id(uint) { return g_gid; }

uint cl_atomic_cmpxchg(volatile uint *p, uint cmp, uint val) CLNAME("_Z14atomic_cmpxchgPU8CLglobalVjjj");
uint cl_atomic_cmpxchg(volatile uint *p, uint cmp, uint val)
{
    uint expected = cmp;
    __atomic_compare_exchange_n(p, &expected, val, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expected;
}

float cl_cos(float x) CLNAME("_Z3cosf");
float cl_cos(float x) { return cosf(x); }
float cl_sin(float x) CLNAME("_Z3sinf");
float cl_sin(float x) { return sinf(x); }
float cl_exp(float x) CLNAME("_Z3expf");
float cl_exp(float x) { return expf(x); }
float cl_log(float x) CLNAME("_Z3logf");
float cl_log(float x) { return logf(x); }
float cl_log10(float x) CLNAME("_Z5log10f");
float cl_log10(float x) { return log10f(x); }
float cl_acos(float x) CLNAME("_Z4acosf");
float cl_acos(float x) { return acosf(x); }
float cl_fabs(float x) CLNAME("_Z4fabsf");
float cl_fabs(float x) { return fabsf(x); }
float cl_sqrt(float x) CLNAME("_Z4sqrtf");
float cl_sqrt(float x) { return sqrtf(x); }
float cl_floor(float x) CLNAME("_Z5floorf");
float cl_floor(float x) { return floorf(x); }
float cl_fmod(float x, float y) CLNAME("_Z4fmodff");
float cl_fmod(float x, float y) { return fmodf(x, y); }
float cl_max(float a, float b) CLNAME("_Z3maxff");
float cl_max(float a, float b) { return fmaxf(a, b); }
float cl_min(float a, float b) CLNAME("_Z3minff");
float cl_min(float a, float b) { return fminf(a, b); }
float cl_pown(float x, int n) CLNAME("_Z4pownfi");
float cl_pown(float x, int n) { return powf(x, (float)n); }
float cl_ldexp(float x, int n) CLNAME("_Z5ldexpfi");
float cl_ldexp(float x, int n) { return ldexpf(x, n); }
float cl_clampf(float x, float lo, float hi) CLNAME("_Z5clampfff");
float cl_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
int cl_clampi(int x, int lo, int hi) CLNAME("_Z5clampiii");
int cl_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
uint cl_mad_hi(uint a, uint b, uint c) CLNAME("_Z6mad_hijjj");
uint cl_mad_hi(uint a, uint b, uint c) { return (uint)(((uint64_t)a * b) >> 32) + c; }
float cl_sincos(float x, float *c) CLNAME("_Z6sincosfPU9CLprivatef");
float cl_sincos(float x, float *c) { *c = cosf(x); return sinf(x); }
float3 cl_normalize(float3 v) CLNAME("_Z9normalizeDv3_f");
float3 cl_normalize(float3 v)
{
    float s = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    float3 r;
    r.x = v.x * s;  r.y = v.y * s;  r.z = v.z * s;
    return r;
}
float cl_distance(float3 a, float3 b) CLNAME("_Z8distanceDv3_fS_");
float cl_distance(float3 a, float3 b)
{
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}
double cl_floord(double x) CLNAME("_Z5floord");
double cl_floord(double x) { return floor(x); }
double cl_fmodd(double x, double y) CLNAME("_Z4fmoddd");
double cl_fmodd(double x, double y) { return fmod(x, y); }
float cl_atan2(float y, float x) CLNAME("_Z5atan2ff");
float cl_atan2(float y, float x) { return atan2f(y, x); }

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
               int *XPS_NSIDE, int *XPS_SIDE, float *XPS_AREA);
void SimRAM_CL(int SOURCE, int PACKETS, int BATCH, float SEED, float *ABS, float *SCA, float TW,
               int *LCELLS, int *OFF, int *PAR, float *DENS, float *EMIT, float *TABS, float *DSC,
               float *CSC, float *XAB, float *EMWEI, float *INT, float *INTX, float *INTY,
               float *INTZ, int *EMINDEX, float *OPT, float *ABU);
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
                      a->OPT, a->ABU, a->XPS_NSIDE, a->XPS_SIDE, a->XPS_AREA);
        else
            SimRAM_CL(a->SOURCE, a->PACKETS, a->BATCH, a->SEED, a->ABS, a->SCA, a->TW, a->LCELLS,
                      a->OFF, a->PAR, a->DENS, a->EMIT, a->TABS, a->DSC, a->CSC, a->XAB, a->EMWEI,
                      a->INT, a->INTX, a->INTY, a->INTZ, a->EMINDEX, a->OPT, a->ABU);
    }
}

extern "C" {

// Execute work items gid0, gid0+stride, ... < gid1 of SimRAM_PB (kind 0) / SimRAM_CL (kind 1).
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
