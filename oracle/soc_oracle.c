/*
 * soc_oracle.c -- CPU restatement of SOC's photon-packet path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (soc_amd/, libsoc_hip.so) never links, imports or calls it.
 *
 * What it restates (file:line in /root/reference):
 *   MWC64X RNG + stream seeding ........ mwc64x_rng.cl:12-48, skip_mwc.cl:9-76
 *   Rand ............................... kernel_ASOC_aux.c:127
 *   IndexG / Index / GetStep ........... kernel_ASOC_aux.c:131-165, 198-278, 282-315
 *   Deflect / Scatter .................. kernel_ASOC_aux.c:499-561
 *   Surface ............................ kernel_ASOC_aux.c:912-940
 *   Parents ............................ kernel_ASOC_aux.c:688-718
 *   SimRAM_PB (SOURCE 0/1, PS_METHOD 0,1,2,4,5) ... kernel_ASOC.c:15-824
 *   SimRAM_CL (USE_EMWEIGHT 0/1, no ALI) ........... kernel_ASOC.c:1223-1689
 * Geometry and feature switches that the reference bakes in with -D macros
 * (ASOC.py:344-362) are run-time fields of orc_model here.
 *
 * Arithmetic: every floating-point expression keeps the reference's operand order and
 * is compiled with -ffp-contract=off.  Two builds of this one source exist:
 *   -DSOC_ORACLE_LIBM  : transcendentals from glibc libm.  Pinned bit-for-bit against the
 *                        reference kernels compiled for x86 (oracle/_ref, same libm).
 *   (default, "soc")   : transcendentals from soc_amd/csrc/soc_math.h, the header the HIP
 *                        kernels use, so that CPU and gfx950 follow identical trajectories.
 * The two builds differ in nothing but those function bodies.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef SOC_ORACLE_LIBM
#  define M_EXP(x)   expf(x)
#  define M_LOG(x)   logf(x)
#  define M_LOGD(x)  log(x)
#  define M_SIN(x)   sinf(x)
#  define M_COS(x)   cosf(x)
#  define M_ACOS(x)  acosf(x)
#  define M_SQRT(x)  sqrtf(x)
#  define M_FMOD1(x) fmodf((x), 1.0f)
#  define M_FMOD1D(x) fmod((x), 1.0)
#  define M_LDEXP_DN(x, l) ldexpf((x), -(l))
#  define M_LDEXP_UP(x, l) ldexpf((x), (l))
#  define M_FLOOR(x) floorf(x)
#  define M_ATAN2(y, x) atan2f((y), (x))
#  define M_LOG10(x) log10f(x)
#  define M_POWN(x, n) powf((x), (float)(n))          /* as the reference shim resolves OpenCL pown */
#  define M_EXPM1(x) expm1f(x)
#  define M_POW15(x) powf((x), 1.5f)
static inline void M_SINCOS(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
#else
#  include "../soc_amd/csrc/soc_math.h"
#  define M_EXP(x)   soc_expf(x)
#  define M_LOG(x)   soc_logf(x)
#  define M_LOGD(x)  soc_logd(x)
#  define M_SIN(x)   soc_sinf(x)
#  define M_COS(x)   soc_cosf(x)
#  define M_ACOS(x)  soc_acosf(x)
#  define M_SQRT(x)  soc_sqrtf(x)
#  define M_FMOD1(x) soc_fmod1f(x)
#  define M_FMOD1D(x) soc_fmod1d(x)
#  define M_LDEXP_DN(x, l) soc_scale_down((x), (l))
#  define M_LDEXP_UP(x, l) soc_scale_up((x), (l))
#  define M_FLOOR(x) soc_floorf(x)
#  define M_ATAN2(y, x) soc_atan2f((y), (x))
#  define M_LOG10(x) soc_log10f(x)
#  define M_POWN(x, n) soc_pownf((x), (n))
#  define M_EXPM1(x) soc_expm1f(x)
#  define M_POW15(x) soc_pow15f(x)
static inline void M_SINCOS(float x, float *s, float *c) { soc_sincosf(x, s, c); }
#endif

/* constants: kernel_ASOC_aux.c:5-9, 99-114 */
#define TWOPI    6.28318531f
#define TAULIM   5.0e-4f
#define PI_F     3.1415926535897f
static const float PEPS = 1.0e-4f;
static const float DEPS = 5.0e-5f;

typedef struct { float x, y, z; } f3;
typedef struct { uint32_t x, c; } rng_t;

/* ---- model / launch description (mirrors the -D list of ASOC.py:344-362 + kernel args) ---- */
typedef struct {
    int NX, NY, NZ, LEVELS, CELLS;
    int BINS, PS_METHOD, NO_PS;       /* NO_PS as passed with -D: max(1, number of sources) */
    int WITH_ABU;                     /* OPT[CELLS,2] instead of scalar ABS/SCA           */
    int WITH_INT;                     /* SAVE_INTENSITY in (1,2) or NOABSORBED==0         */
    int USE_EMWEIGHT;                 /* 0 or 1 (SimRAM_CL)                               */
    int DOUBLE_INDEX;                 /* NX > DIMLIM, kernel_ASOC_aux.c:25-37,207-211     */
    const int   *LCELLS, *OFF, *PAR;
    const float *DENS;
    const float *CSC;                 /* [BINS] row of the current frequency              */
    const float *OPT;                 /* [2*CELLS] or NULL                                */
    /* kernel arguments */
    int   SOURCE, PACKETS, BATCH, GLOBAL;
    float SEED, ABS, SCA, BG, TW;
    const float *PSPOS;               /* 4 floats per source (OpenCL float3 = 16 bytes)   */
    const float *PS;
    const int   *XPS_NSIDE, *XPS_SIDE;
    const float *XPS_AREA;
    const float *EMIT, *EMWEI;
    float *TABS, *INT;
    int   threaded;                   /* !=0: tallies use atomic adds (OpenMP build)      */
    /* scattered-light kernels (kernel_ASOC_sca.c): observers and image buffer */
    int   NDIR, NPIX_X, NPIX_Y, FFS;
    float MAP_DX, CX, CY, CZ;         /* CENTRE                                           */
    const float *ODIRS, *ORA, *ODE;   /* 4 floats per direction (OpenCL float3)           */
    const float *DSC;                 /* [BINS] discrete scattering function              */
    float *OUT;                       /* [NDIR*NPIX_Y*NPIX_X]                             */
    int   XPS_AS_FLOAT;               /* sca kernels declare XPS_NSIDE/XPS_SIDE as float* */
    /* Healpix background (SimRAM_HP): sky map of the current frequency, NSIDE=64 RING */
    int   HPBG_WEIGHTED;
    const float *HPBG;                /* [49152] photons per package                      */
    const float *HPBGP;               /* [49152] cumulative pixel probability (weighted)  */
    int   MIRROR;                     /* bit mask x,X,y,Y,z,Z = 1,2,4,8,16,32 (ASOC.py:319-321) */
    int   WITH_ALI;                   /* SimRAM_CL: absorptions in the emitting cell go to XAB           */
    float *XAB;                       /* [CELLS]                                                          */
    const int *EMINDEX;               /* [CELLS] USE_EMWEIGHT==2: cells to emit from, terminated by -1    */
    /* region of interest (nested runs): packets entering ROI = [x0,x1,y0,y1,z0,z1] (root cells, inclusive) are
     * recorded per surface element and Healpix direction (WITH_ROI_SAVE); SOURCE == 3 emits such a record from
     * the model surface (WITH_ROI_LOAD) */
    int   WITH_ROI_SAVE, ROI[6], ROI_STEP, ROI_NSIDE;
    float *ROI_SAVE;                  /* [elements * 12*ROI_NSIDE^2]                                      */
    int   WITH_ROI_LOAD, ROI_DIM[3];
    const float *ROI_LOAD;            /* [elements * 12*ROI_NSIDE^2] photons, already scaled by the host  */
    /* weighted free paths (-D STEP_WEIGHT, SW_A, SW_B) and per-dust scattering functions (-D WITH_MSF,
     * NDUST): MSF_NDUST > 1 -> CSC holds NDUST tables, MSF_SCA[NDUST] the scattering cross sections, ABU[CELLS*NDUST] */
    int   STEP_WEIGHT;  float SW_A, SW_B;
    int   MSF_NDUST;
    const float *MSF_SCA, *ABU;
    float *INTV;                      /* -D SAVE_INTENSITY=2: INTX | INTY | INTZ, CELLS floats each (kernel_ASOC.c:604-612) */
    int   LEVEL_THRESHOLD;            /* -D LEVEL_THRESHOLD: Mapping ignores the emission of coarser levels (kernel_ASOC_map.c:825-834) */
    int   ROI_MAP;                    /* -D ROI_MAP: the map kernels count the emission of cells inside ROI only (kernel_ASOC_map.c:821-823,947-949) */
    float CR_HEATING_RATE;            /* -D CR_HEATING=1 -D CR_HEATING_RATE: EqTemperature adds 1e-27*FACTOR*rate (kernel_ASOC_aux.c:769-773); 0 = off */
    int   MAP_INTERPOLATION;          /* -D MAP_INTERPOLATION=1|2 (ini key mapint): Mapping blends density and emission with two neighbours (kernel_ASOC_map.c:656-810) */
} orc_model;

/* kernel_ASOC_sca.c:495-497,1486-1488 declare XPS_NSIDE and XPS_SIDE "__global float *" while
 * the host uploads int32 arrays (ASOCS.py:267-268, AnalyseExternalPointSources): the kernel
 * sees the integer bit patterns as (denormal) floats.  Restated as compiled. */
static inline float xps_value(const orc_model *M, const int *a, int i)
{
    if (M->XPS_AS_FLOAT) { union { int i; float f; } v;  v.i = a[i];  return v.f; }
    return (float)a[i];
}

/* ================================ RNG ================================================== */

#define MWC64X_A 4294883355U
#define MWC64X_M 18446383549859758079UL
#define MWC_BASEID 4077358422479273989UL

static uint64_t AddMod64(uint64_t a, uint64_t b, uint64_t M)
{
    uint64_t v = a + b;
    if ((v >= M) || (v < a)) v = v - M;
    return v;
}
static uint64_t MulMod64(uint64_t a, uint64_t b, uint64_t M)
{
    uint64_t r = 0;
    while (a != 0) {
        if (a & 1) r = AddMod64(r, b, M);
        b = AddMod64(b, b, M);
        a = a >> 1;
    }
    return r;
}
static uint64_t PowMod64(uint64_t a, uint64_t e, uint64_t M)
{
    uint64_t sqr = a, acc = 1;
    while (e != 0) {
        if (e & 1) acc = MulMod64(acc, sqr, M);
        sqr = MulMod64(sqr, sqr, M);
        e = e >> 1;
    }
    return acc;
}
/* MWC64X_SeedStreams with vecSize=1, vecOffset=0 (mwc64x_rng.cl:35-40, skip_mwc.cl:64-76) */
static void SeedStreams(rng_t *s, uint64_t gid, uint64_t baseOffset, uint64_t perStreamOffset)
{
    uint64_t dist = baseOffset + gid * perStreamOffset;
    uint64_t m = PowMod64(MWC64X_A, dist, MWC64X_M);
    uint64_t x = MulMod64(MWC_BASEID, m, MWC64X_M);
    s->x = (uint32_t)(x / MWC64X_A);
    s->c = (uint32_t)(x % MWC64X_A);
}
static uint32_t NextUint(rng_t *s)
{
    uint32_t res = s->x ^ s->c;
    uint32_t X = s->x, C = s->c;
    uint32_t Xn = MWC64X_A * X + C;
    uint32_t carry = (uint32_t)(Xn < C);
    uint32_t Cn = (uint32_t)(((uint64_t)MWC64X_A * X) >> 32) + carry;   /* mad_hi(A, X, carry) */
    s->x = Xn;
    s->c = Cn;
    return res;
}
static inline float Rand(rng_t *s) { return NextUint(s) / 4294967295.0f; }

/* kernel_ASOC.c:74-77 */
static uint64_t seed_base(float SEED)
{
    return (unsigned long)(fmodf(SEED * 7.0f * PI_F, 1.0f) * 4294967296L);
}
static void seed_workitem(rng_t *s, float SEED, uint64_t gid)
{
    SeedStreams(s, gid, seed_base(SEED), 274877906944UL);
}

/* ================================ traversal ============================================ */

static void IndexG(const orc_model *M, f3 *pos, int *level, int *ind)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    const float *DENS = M->DENS;
    const int *OFF = M->OFF;
    *ind = -1;
    if ((pos->x <= 0.0f) || (pos->y <= 0.0f) || (pos->z <= 0.0f)) return;
    if ((pos->x >= NX) || (pos->y >= NY) || (pos->z >= NZ)) return;
    *level = 0;
    *ind = (int)M_FLOOR(pos->z) * NX * NY + (int)M_FLOOR(pos->y) * NX + (int)M_FLOOR(pos->x);
    if (DENS[*ind] > 0.0f) return;
    while (1) {
        pos->x = 2.0f * M_FMOD1(pos->x);
        pos->y = 2.0f * M_FMOD1(pos->y);
        pos->z = 2.0f * M_FMOD1(pos->z);
        float link = -DENS[OFF[*level] + (*ind)];
        int   li;
        memcpy(&li, &link, 4);
        *ind = li;
        (*level)++;
        *ind += 4 * (int)M_FLOOR(pos->z) + 2 * (int)M_FLOOR(pos->y) + (int)M_FLOOR(pos->x);
        if (DENS[OFF[*level] + (*ind)] > 0.0f) return;
    }
}

/* Index(): float and double position variants, kernel_ASOC_aux.c:198-278 */
#define REAL float
#define RFLOOR(x) M_FLOOR(x)
#define RFMOD1(x) M_FMOD1(x)
#define INDEX_NAME Index_f
#include "soc_oracle_index.inc"
#undef REAL
#undef RFLOOR
#undef RFMOD1
#undef INDEX_NAME

#define REAL double
#define RFLOOR(x) floor(x)
#define RFMOD1(x) M_FMOD1D(x)
#define INDEX_NAME Index_d
#include "soc_oracle_index.inc"
#undef REAL
#undef RFLOOR
#undef RFMOD1
#undef INDEX_NAME

static inline void Index(const orc_model *M, f3 *pos, int *level, int *ind)
{
    if (M->DOUBLE_INDEX) Index_d(M, pos, level, ind);
    else                 Index_f(M, pos, level, ind);
}

/* kernel_ASOC_aux.c:282-315, float branch (NX <= 9999) */
static float GetStep(const orc_model *M, f3 *POS, const f3 *DIR, int *level, int *ind)
{
    float dx, dy, dz;
    dx = (DIR->x > 0.0f) ? ((1.0f + PEPS - M_FMOD1(POS->x)) / DIR->x) : ((-PEPS - M_FMOD1(POS->x)) / DIR->x);
    dy = (DIR->y > 0.0f) ? ((1.0f + PEPS - M_FMOD1(POS->y)) / DIR->y) : ((-PEPS - M_FMOD1(POS->y)) / DIR->y);
    dz = (DIR->z > 0.0f) ? ((1.0f + PEPS - M_FMOD1(POS->z)) / DIR->z) : ((-PEPS - M_FMOD1(POS->z)) / DIR->z);
    dx = fminf(dx, fminf(dy, dz));
    POS->x += dx * DIR->x;
    POS->y += dx * DIR->y;
    POS->z += dx * DIR->z;
    dx = M_LDEXP_DN(dx, *level);
    Index(M, POS, level, ind);
    return dx;
}

static inline void normalize3(f3 *v)
{
    /* OpenCL normalize(): device-defined; fixed here (and in oracle/ref_shim.cpp, and in the
       HIP kernel) as v * (1/sqrt(x*x+y*y+z*z)) */
    float s = 1.0f / M_SQRT(v->x * v->x + v->y * v->y + v->z * v->z);
    v->x = v->x * s;
    v->y = v->y * s;
    v->z = v->z * s;
}

/* kernel_ASOC_aux.c:499-533 */
static void Deflect(f3 *DIR, const float COS_THETA, const float phi)
{
    float cx, cy, cz, ox, oy, oz, theta0, phi0, cos_theta, sin_theta, sin_phi, cos_phi;
    cx = DIR->x;  cy = DIR->y;  cz = DIR->z;
    sin_theta = M_SQRT(1.0f - COS_THETA * COS_THETA);
    M_SINCOS(phi, &sin_phi, &cos_phi);
    ox = sin_theta * cos_phi;
    oy = sin_theta * sin_phi;
    oz = COS_THETA;
    theta0 = M_ACOS(cz / M_SQRT(cx * cx + cy * cy + cz * cz + DEPS));
    phi0   = M_ACOS(cx / M_SQRT(cx * cx + cy * cy + DEPS));
    if (DIR->y < 0.0f) phi0 = (TWOPI - phi0);
    theta0 = -theta0;
    phi0   = -phi0;
    M_SINCOS(theta0, &sin_theta, &cos_theta);
    M_SINCOS(phi0, &sin_phi, &cos_phi);
    DIR->x = +ox * cos_theta * cos_phi + oy * sin_phi - oz * sin_theta * cos_phi;
    DIR->y = -ox * cos_theta * sin_phi + oy * cos_phi + oz * sin_theta * sin_phi;
    DIR->z = +ox * sin_theta + oz * cos_theta;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

/* kernel_ASOC_aux.c:540-561 (HG_TEST==0) */
static void Scatter(f3 *DIR, const float *CSC, int BINS, rng_t *rng)
{
    float cos_theta = CSC[clampi((int)M_FLOOR(Rand(rng) * BINS), 0, BINS - 1)];
    float phi = TWOPI * Rand(rng);
    Deflect(DIR, cos_theta, phi);
    if (fabsf(DIR->x) < DEPS) DIR->x = DEPS;
    if (fabsf(DIR->y) < DEPS) DIR->y = DEPS;
    if (fabsf(DIR->z) < DEPS) DIR->z = DEPS;
    normalize3(DIR);
}

/* the free path of a new flight, kernel_ASOC.c:516-535 (creation) = :736-755 (after a scattering): unweighted, or drawn
 * from a stretched exponential (STEP_WEIGHT 1) / a sum of two (2) with the packet weight corrected */
static float DrawFreePath(const orc_model *M, rng_t *rng, float *PHOTONS)
{
    float free_path;
    const float SW_A = M->SW_A, SW_B = M->SW_B;
    if (M->STEP_WEIGHT <= 0) return -M_LOG(Rand(rng));
    if (M->STEP_WEIGHT == 1) {
        free_path = -M_LOG(Rand(rng)) / SW_A;
        *PHOTONS *= M_EXP(SW_A * free_path - free_path) / SW_A;
        return free_path;
    }
    free_path = -M_LOG((-SW_B + M_SQRT(SW_B * SW_B + 4.0f * Rand(rng) * (1.0f - SW_B))) / (2.0f - 2.0f * SW_B)) / SW_A;
    *PHOTONS *= 1.0f / (SW_A * SW_B * M_EXP((1.0f - SW_A) * free_path) + 2.0f * SW_A * (1.0f - SW_B) * M_EXP((1.0f - 2.0f * SW_A) * free_path));
    return free_path;
}

/* the new direction after a scattering in cell oind, kernel_ASOC.c:768-799 (SimRAM_PB), :1146-1175 (HP), :1644-1675 (CL):
 * the scattering function of a dust species picked by its share of the cell's scattering cross section (WITH_MSF), or the
 * one table.  cl: SimRAM_CL keeps the cell's OPT value in free_path (:1662).  (-D DIR_WEIGHT > 0 does not compile in
 * the reference -- :770-775 use undeclared pweight, pind -- and is not restated.) */
/* -D WITH_MSF: the species that scatters in cell oind, drawn with probabilities ABU*SCA/OPT.sca (kernel_ASOC.c:780-791;
 * kernel_ASOC_sca.c:340-347, :427-431 and the same lines of the other kernels).  kernel_ASOC.c limits the index to
 * NDUST-1 when rounding leaves ds > 0 after the last species; the sca kernels do not (they would read past DSC/CSC):
 * the restatement limits it in both. */
static int MsfDust(const orc_model *M, rng_t *rng, int oind)
{
    const int NDUST = M->MSF_NDUST;
    const float dx = M->OPT[2 * (long)oind + 1];
    float ds = 0.99999f * Rand(rng);
    int   idust;
    for (idust = 0; idust < NDUST; idust++) {
        ds -= M->ABU[idust + NDUST * ((long)oind)] * M->MSF_SCA[idust] / dx;
        if (ds <= 0.0) break;
    }
    if (idust >= NDUST) idust = NDUST - 1;
    return idust;
}
static void NewDirection(const orc_model *M, f3 *DIR, rng_t *rng, int oind, float *PHOTONS, float *free_path, int cl)
{
    if (M->MSF_NDUST > 1) {
        if (cl) *free_path = M->OPT[2 * (long)oind + 1];                  /* SimRAM_CL: `free_path` is the scratch (:1662) */
        const int idust = MsfDust(M, rng, oind);
        Scatter(DIR, M->CSC + (long)idust * M->BINS, M->BINS, rng);
    } else {
        Scatter(DIR, M->CSC, M->BINS, rng);
    }
}

/* kernel_ASOC_aux.c:912-940 */
static void Surface(const orc_model *M, f3 *POS, f3 *DIR)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    float dx, dy, dz;
    if (DIR->x > 0.0f) {
        if (POS->x < 0.0f) dx = (PEPS - POS->x) / DIR->x;
        else               dx = -1.0e10f;
    } else {
        if (POS->x > NX)   dx = (NX - PEPS - POS->x) / DIR->x;
        else               dx = -1.0e10f;
    }
    if (DIR->y > 0.0f) {
        if (POS->y < 0.0f) dy = (PEPS - POS->y) / DIR->y;
        else               dy = -1.0e10f;
    } else {
        if (POS->y > NY)   dy = (NY - PEPS - POS->y) / DIR->y;
        else               dy = -1.0e10f;
    }
    if (DIR->z > 0.0f) {
        if (POS->z < 0.0f) dz = (PEPS - POS->z) / DIR->z;
        else               dz = -1.0e10f;
    } else {
        if (POS->z > NZ)   dz = (NZ - PEPS - POS->z) / DIR->z;
        else               dz = -1.0e10f;
    }
    dx = fmaxf(dx, fmaxf(dy, dz));
    POS->x += dx * DIR->x;
    POS->y += dx * DIR->y;
    POS->z += dx * DIR->z;
}

static inline void tally(const orc_model *M, float *buf, int oind, float v)
{
    if (M->threaded) {
#pragma omp atomic
        buf[oind] += v;
    } else {
        buf[oind] += v;
    }
}

/* The part of SimRAM_PB / SimRAM_CL that follows packet creation: kernel_ASOC.c:508-820
 * (PB) and :1409-1676 (CL).  cl_order selects the CL kernel's placement of the
 * scatterings>20 test (before the scatter tally, kernel_ASOC.c:1545-1551) instead of the
 * PB placement (after Scatter(), :802-804).  Returns the number of tally events. */
#define EPS_MIRROR 5.0e-4f      /* EPS, kernel_ASOC_aux.c:111 */

/* Mirror (kernel_ASOC_aux.c:1050-1083), called with ind<0.  As written in the reference the
 * direction flip is outside the if-statement (no braces): every enabled face flips its
 * component whenever the function runs, whichever face the packet left through. */
static void Mirror(const orc_model *M, f3 *pos, f3 *dir, int *level, int *ind)
{
    const int MIRROR = M->MIRROR;
    if (MIRROR & 1)  { if (pos->x < 0.0f)  pos->x = EPS_MIRROR;          dir->x = -dir->x;  IndexG(M, pos, level, ind); }
    if (MIRROR & 2)  { if (pos->x > M->NX) pos->x = M->NX - EPS_MIRROR;  dir->x = -dir->x;  IndexG(M, pos, level, ind); }
    if (MIRROR & 4)  { if (pos->y < 0.0f)  pos->y = EPS_MIRROR;          dir->y = -dir->y;  IndexG(M, pos, level, ind); }
    if (MIRROR & 8)  { if (pos->y > M->NY) pos->y = M->NY - EPS_MIRROR;  dir->y = -dir->y;  IndexG(M, pos, level, ind); }
    if (MIRROR & 16) { if (pos->z < 0.0f)  pos->z = EPS_MIRROR;          dir->z = -dir->z;  IndexG(M, pos, level, ind); }
    if (MIRROR & 32) { if (pos->z > M->NZ) pos->z = M->NZ - EPS_MIRROR;  dir->z = -dir->z;  IndexG(M, pos, level, ind); }
}

/* e_index: global index of the emitting cell (WITH_ALI, kernel_ASOC.c:1394-1396) or -1 */
static void RootPos(const orc_model *M, f3 *POS, const int ilevel, const int iind);
static int  angles2pixel_ring(const int nside, float phi, float theta);
static void pixel2angles_ring(const int nside, const int ipix, float *phi, float *theta);

/* InRoi (kernel_ASOC_aux.c:1031-1048): root-grid index of the root cell above (level, ind) if that cell lies in
 * ROI, else -1.  A packet that has left the model (ind < 0, level 0) is outside: the reference's arithmetic on
 * i = -1 gives i % NX = -1 < ROI[0]. */
static int InRoi(const orc_model *M, int level, int ind)
{
    const int NX = M->NX, NY = M->NY;
    int i = ind, j, k = level;
    if (ind < 0) return -1;
    while (k > 0) { i = M->PAR[M->OFF[k] + i - NX * NY * M->NZ];  k--; }
    k = i / (NX * NY);
    j = (i / NX) % NY;
    if (((i % NX) >= M->ROI[0]) && ((i % NX) <= M->ROI[1]) && (j >= M->ROI[2]) && (j <= M->ROI[3]) && (k >= M->ROI[4]) && (k <= M->ROI[5]))
        return i;
    return -1;
}

/* a packet has stepped into ROI: kernel_ASOC.c:618-642 (SimRAM_PB), :1510-1535 (SimRAM_CL).  The element index
 * `ii` is uninitialised in the reference when no border test matches (the position after the step is within PEPS
 * of the face it came through, so one always does); 0 here, as in the zero-initialised reference builds. */
static void roi_save(const orc_model *M, f3 POS, f3 DIR, int level, int ind, float PHOTONS)
{
    const int *ROI = M->ROI, ROI_STEP = M->ROI_STEP, ROI_NSIDE = M->ROI_NSIDE;
    const int ROI_NX = (ROI[1] - ROI[0] + 1) * ROI_STEP, ROI_NY = (ROI[3] - ROI[2] + 1) * ROI_STEP, ROI_NZ = (ROI[5] - ROI[4] + 1) * ROI_STEP;
    int ii = 0, jj;
    f3  RPOS = POS;
    RootPos(M, &RPOS, level, ind);
    if ((RPOS.x < (ROI[0] + 1.0e-3f)) || (RPOS.x > (ROI[1] + 0.999f))) {
        ii = clampi((int)M_FLOOR((RPOS.y - ROI[2]) * ROI_STEP), 0, ROI_NY - 1);
        jj = clampi((int)M_FLOOR((RPOS.z - ROI[4]) * ROI_STEP), 0, ROI_NZ - 1);
        ii = ii + ROI_NY * jj;
    }
    if ((RPOS.y < (ROI[2] + 1.0e-3f)) || (RPOS.y > (ROI[3] + 0.999f))) {
        ii = clampi((int)M_FLOOR((RPOS.x - ROI[0]) * ROI_STEP), 0, ROI_NX - 1);
        jj = clampi((int)M_FLOOR((RPOS.z - ROI[4]) * ROI_STEP), 0, ROI_NZ - 1);
        ii = ROI_NY * ROI_NZ + ii + ROI_NX * jj;
    }
    if ((RPOS.z < (ROI[4] + 1.0e-3f)) || (RPOS.z > (ROI[5] + 0.999f))) {
        ii = clampi((int)M_FLOOR((RPOS.x - ROI[0]) * ROI_STEP), 0, ROI_NX - 1);
        jj = clampi((int)M_FLOOR((RPOS.y - ROI[2]) * ROI_STEP), 0, ROI_NY - 1);
        ii = ROI_NY * ROI_NZ + ROI_NX * ROI_NZ + ii + ROI_NX * jj;
    }
    {
        const float theta = M_ACOS(DIR.z);
        const float phi   = M_ATAN2(DIR.y, DIR.x);
        jj = angles2pixel_ring(ROI_NSIDE, phi, theta);
    }
    ii = clampi(ii, 0, ROI_NX * ROI_NY + ROI_NY * ROI_NZ + ROI_NZ * ROI_NX - 1);
    jj = clampi(jj, 0, 12 * ROI_NSIDE * ROI_NSIDE - 1);
    tally(M, M->ROI_SAVE, ii * 12 * ROI_NSIDE * ROI_NSIDE + jj, PHOTONS);
}

static long walk_packet(const orc_model *M, rng_t *rng, f3 POS, f3 DIR, float PHOTONS,
                        int level, int ind, int cl_order, int e_index)
{
    const float *DENS = M->DENS;
    const int *OFF = M->OFF;
    int   oind = 0, ind0 = -1, level0 = 0, scatterings, steps;
    float ds, free_path, tau, dtau, delta, tauA, dx;
    f3    POS0 = POS;
    long  nt = 0;

    if (!(cl_order & 2)) {              /* bit 1: direction conditioned by the caller (SimRAM_HP) */
        if (fabsf(DIR.x) < DEPS) DIR.x = DEPS;
        if (fabsf(DIR.y) < DEPS) DIR.y = DEPS;
        if (fabsf(DIR.z) < DEPS) DIR.z = DEPS;
        normalize3(&DIR);
    }
    cl_order &= 1;
    scatterings = 0;
    tau = 0.0f;
    free_path = DrawFreePath(M, rng, &PHOTONS);
    steps = 0;
    int roi = -1, oroi = -1;
    if (M->WITH_ROI_SAVE) roi = oroi = InRoi(M, level, ind);            /* kernel_ASOC.c:550, :1439 */

    while (ind >= 0) {
        tau = 0.0f;
        while (ind >= 0) {
            oroi   = roi;
            oind   = OFF[level] + ind;
            ind0   = ind;
            level0 = level;
            POS0   = POS;
            ds     = GetStep(M, &POS, &DIR, &level, &ind);
            if (M->WITH_ABU) {
                tauA = ds * DENS[oind] * M->OPT[2 * (long)oind];
                dtau = ds * DENS[oind] * M->OPT[2 * (long)oind + 1];
            } else {
                tauA = ds * DENS[oind] * M->ABS;
                dtau = ds * DENS[oind] * M->SCA;
            }
            if (free_path < (tau + dtau)) {
                ind = ind0;
                break;
            }
            delta = (tauA > TAULIM) ? (PHOTONS * (1.0f - M_EXP(-tauA))) : (PHOTONS * tauA * (1.0f - 0.5f * tauA));
            if ((M->WITH_ALI == 1) && (oind == e_index)) tally(M, M->XAB, oind, delta * M->TW);   /* :1486-1491, :1589-1594 */
            else tally(M, M->TABS, oind, delta * M->TW * 1.0f);
            if (M->WITH_INT) tally(M, M->INT, oind, delta);
            if (M->INTV) {                                               /* net flux vector, :604-612 */
                tally(M, M->INTV, oind, delta * DIR.x);
                tally(M, M->INTV + M->CELLS, oind, delta * DIR.y);
                tally(M, M->INTV + 2 * (long)M->CELLS, oind, delta * DIR.z);
            }
            nt++;
            PHOTONS *= M_EXP(-tauA);
            tau += dtau;
            if (M->WITH_ROI_SAVE) {                                      /* only at the end of a full step */
                roi = InRoi(M, level, ind);
                if ((roi >= 0) && (oroi < 0)) roi_save(M, POS, DIR, level, ind, PHOTONS);
            }
            if (!cl_order) {
                /* failed-step guard exists only in SimRAM_PB (kernel_ASOC.c:649-683) */
                if ((level == level0) && (ind == ind0)) {
                    POS.x += PEPS * DIR.x;
                    POS.y += PEPS * DIR.y;
                    POS.z += PEPS * DIR.z;
                    steps += 1;
                }
            }
            if ((M->MIRROR > 0) && (ind < 0)) Mirror(M, &POS, &DIR, &level, &ind);   /* :686-688, :1064, :1540 */
        }
        if (ind < 0) break;
        /* scatter */
        scatterings++;
        if (cl_order && (scatterings > 20)) { ind = -1; continue; }
        dtau = free_path - tau;
        if (M->WITH_ABU) {
            dx   = dtau / (M->OPT[2 * (long)oind + 1] * DENS[oind]);
            tauA = dx * DENS[oind] * M->OPT[2 * (long)oind];
        } else {
            dx   = dtau / (M->SCA * DENS[oind]);
            tauA = dx * DENS[oind] * M->ABS;
        }
        delta = (tauA > TAULIM) ? (PHOTONS * (1.0f - M_EXP(-tauA))) : (PHOTONS * tauA * (1.0f - 0.5f * tauA));
        if ((M->WITH_ALI == 1) && (oind == e_index)) tally(M, M->XAB, oind, delta * M->TW);   /* :1486-1491, :1589-1594 */
            else tally(M, M->TABS, oind, delta * M->TW * 1.0f);
        if (M->WITH_INT) tally(M, M->INT, oind, delta);
        if (M->INTV) {                                                   /* :724-732 */
            tally(M, M->INTV, oind, delta * DIR.x);
            tally(M, M->INTV + M->CELLS, oind, delta * DIR.y);
            tally(M, M->INTV + 2 * (long)M->CELLS, oind, delta * DIR.z);
        }
        nt++;
        dx = M_LDEXP_UP(dx, level0);
        dx = fmaxf(0.0f, dx - 2.0f * PEPS);
        POS.x = POS0.x + dx * DIR.x;
        POS.y = POS0.y + dx * DIR.y;
        POS.z = POS0.z + dx * DIR.z;
        PHOTONS *= M_EXP(-tauA);
        free_path = DrawFreePath(M, rng, &PHOTONS);
        ind   = ind0;
        level = level0;
        NewDirection(M, &DIR, rng, oind, &PHOTONS, &free_path, cl_order);
        if (!cl_order && (scatterings > 20)) { ind = -1; continue; }
    }
    (void)steps;
    return nt;
}

/* ================================ SimRAM_PB ============================================ */

/* surface element of a background work item (kernel_ASOC.c:109-138) */
typedef struct { int SIDE; float X0, Y0, Z0, DX, DY, DZ; } surf_t;

static void pb_surface_element(const orc_model *M, int id, surf_t *E)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    const int AREA = 2 * (NX * NY + NY * NZ + NZ * NX);
    const int SOURCE = M->SOURCE;
    int   ind, SIDE = 0;
    float X0 = 0, Y0 = 0, Z0 = 0, DX = 0, DY = 0, DZ = 0;
    if (SOURCE == 1) {
        ind = id % AREA;
        DX = 1.0f; DY = 1.0f; DZ = 1.0f;
        if (ind < (NY * NZ)) {
            SIDE = 0;  X0 = PEPS;       Y0 = ind % NY;  Z0 = ind / NY;  DX = 0.0f;
        } else {
            ind -= NY * NZ;
            if (ind < (NY * NZ)) {
                SIDE = 1;  X0 = NX - PEPS;  Y0 = ind % NY;  Z0 = ind / NY;  DX = 0.0f;
            } else {
                ind -= NY * NZ;
                if (ind < (NX * NZ)) {
                    SIDE = 2;  Y0 = PEPS;  X0 = ind % NX;  Z0 = ind / NX;  DY = 0.0f;
                } else {
                    ind -= NX * NZ;
                    if (ind < (NX * NZ)) {
                        SIDE = 3;  Y0 = NY - PEPS;  X0 = ind % NX;  Z0 = ind / NX;  DY = 0.0f;
                    } else {
                        ind -= NX * NZ;
                        if (ind < (NX * NY)) {
                            SIDE = 4;  Z0 = PEPS;  X0 = ind % NX;  Y0 = ind / NX;  DZ = 0.0f;
                        } else {
                            ind -= NX * NY;
                            SIDE = 5;  Z0 = NZ - PEPS;  X0 = ind % NX;  Y0 = ind / NX;  DZ = 0.0f;
                        }
                    }
                }
            }
        }
    }

    E->SIDE = SIDE;  E->X0 = X0;  E->Y0 = Y0;  E->Z0 = Z0;  E->DX = DX;  E->DY = DY;  E->DZ = DZ;
}

/* creation of packet III of a SimRAM_PB work item: point sources (kernel_ASOC.c:202-434) or
 * background (kernel_ASOC.c:439-464); identical in kernel_ASOC_sca.c:640-836 */
static void pb_create(const orc_model *M, const surf_t *E, int III, rng_t *rng, f3 *pPOS, f3 *pDIR, float *pPHOTONS,
                      int *plevel, int *pind)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    const int SOURCE = M->SOURCE, NO_PS = M->NO_PS;
    const int SIDE = E->SIDE;
    const float X0 = E->X0, Y0 = E->Y0, Z0 = E->Z0, DX = E->DX, DY = E->DY, DZ = E->DZ;
    f3    POS = *pPOS, DIR = *pDIR;
    float PHOTONS = *pPHOTONS, phi, cos_theta, sin_theta, v1, v2;
    int   level = *plevel, ind = *pind, oind = 0, level0;
    if (SOURCE == 0) {
        phi       = TWOPI * Rand(rng);
        cos_theta = 0.999997f - 1.999995f * Rand(rng);
        sin_theta = M_SQRT(1.0f - cos_theta * cos_theta);
        DIR.x = sin_theta * M_COS(phi);
        DIR.y = sin_theta * M_SIN(phi);
        DIR.z = cos_theta;
        level0  = III % NO_PS;
        PHOTONS = M->PS[level0];
        f3 SRC = { M->PSPOS[4 * level0], M->PSPOS[4 * level0 + 1], M->PSPOS[4 * level0 + 2] };
        POS = SRC;
        IndexG(M, &POS, &level, &ind);
        if ((ind < 0) || (ind >= M->CELLS)) {
            if (M->PS_METHOD == 0) {
                Surface(M, &POS, &DIR);
                IndexG(M, &POS, &level, &ind);
            }
            if (M->PS_METHOD == 1) {
                POS = SRC;
                if (POS.z > NZ) {
                    if (DIR.z > 0.0f) DIR.z = -DIR.z;
                } else {
                    if (POS.z < 0.0f) {
                        if (DIR.z < 0.0f) DIR.z = -DIR.z;
                    } else {
                        if (POS.x > NX) {
                            if (DIR.x > 0.0f) DIR.x = -DIR.x;
                        } else {
                            if (POS.x < 0.0f) {
                                if (DIR.x < 0.0f) DIR.x = -DIR.x;
                            } else {
                                if (POS.y > NY) {
                                    if (DIR.y > 0.0f) DIR.y = -DIR.y;
                                } else {
                                    if (POS.y < 0.0f) {
                                        if (DIR.y < 0.0f) DIR.y = -DIR.y;
                                    }
                                }
                            }
                        }
                    }
                }
                Surface(M, &POS, &DIR);
                PHOTONS *= 0.5f;
                IndexG(M, &POS, &level, &ind);
            }
            if (M->PS_METHOD == 2) {
                POS = SRC;
                ind = M_FLOOR(Rand(rng) * xps_value(M, M->XPS_NSIDE, level0) * 0.999999f);
                PHOTONS /= M->XPS_AREA[3 * level0 + ind];
                ind = (int)xps_value(M, M->XPS_SIDE, 3 * level0 + ind);
                float a = Rand(rng), b = Rand(rng);
                if (ind == 0) { POS.x = NX - PEPS;  POS.y = a * NY;  POS.z = b * NZ;  b = NY * NZ; }
                if (ind == 1) { POS.x = PEPS;       POS.y = a * NY;  POS.z = b * NZ;  b = NY * NZ; }
                if (ind == 2) { POS.y = NY - PEPS;  POS.x = a * NX;  POS.z = b * NZ;  b = NX * NZ; }
                if (ind == 3) { POS.y = PEPS;       POS.x = a * NX;  POS.z = b * NZ;  b = NX * NZ; }
                if (ind == 4) { POS.z = NZ - PEPS;  POS.x = a * NX;  POS.y = b * NY;  b = NX * NY; }
                if (ind == 5) { POS.z = PEPS;       POS.x = a * NX;  POS.y = b * NY;  b = NX * NY; }
                DIR.x = POS.x - SRC.x;  DIR.y = POS.y - SRC.y;  DIR.z = POS.z - SRC.z;
                v1 = M_SQRT(DIR.x * DIR.x + DIR.y * DIR.y + DIR.z * DIR.z);   /* distance() */
                normalize3(&DIR);
                v2 = (ind < 2) ? (fabsf(DIR.x)) : ((ind < 4) ? (fabsf(DIR.y)) : (fabsf(DIR.z)));
                PHOTONS *= v2 * b / (4.0f * PI_F * v1 * v1);
                IndexG(M, &POS, &level, &ind);
            }
            if (M->PS_METHOD == 4) {
                v1 = SRC.z - NZ;
                cos_theta = v1 / M_SQRT(v1 * v1 + 0.25f * NX * NX + 0.25f * NY * NY);
                PHOTONS *= 0.5f * (1.0f - cos_theta);
                cos_theta = 1.0f - Rand(rng) * (1.0f - cos_theta);
                v1 = TWOPI * Rand(rng);
                DIR.x = M_SQRT(1.0f - cos_theta * cos_theta) * M_COS(v1);
                DIR.y = M_SQRT(1.0f - cos_theta * cos_theta) * M_SIN(v1);
                DIR.z = -cos_theta;
                Surface(M, &POS, &DIR);
                IndexG(M, &POS, &level, &ind);
            }
            if (M->PS_METHOD == 5) {
                cos_theta = M->XPS_AREA[3 * level0];
                PHOTONS *= 0.5f * (1.0f - cos_theta);
                cos_theta = 1.0f - Rand(rng) * (1.0f - cos_theta);
                v1   = TWOPI * Rand(rng);
                oind = (int)xps_value(M, M->XPS_SIDE, 3 * level0);
                if (oind < 2) {
                    DIR.y = M_SQRT(1.0f - cos_theta * cos_theta) * M_COS(v1);
                    DIR.z = M_SQRT(1.0f - cos_theta * cos_theta) * M_SIN(v1);
                    if (oind == 0) DIR.x = -cos_theta;
                    else           DIR.x = +cos_theta;
                } else {
                    if (oind < 4) {
                        DIR.x = M_SQRT(1.0f - cos_theta * cos_theta) * M_COS(v1);
                        DIR.z = M_SQRT(1.0f - cos_theta * cos_theta) * M_SIN(v1);
                        if (oind == 2) DIR.y = -cos_theta;
                        else           DIR.y = +cos_theta;
                    } else {
                        DIR.x = M_SQRT(1.0f - cos_theta * cos_theta) * M_COS(v1);
                        DIR.y = M_SQRT(1.0f - cos_theta * cos_theta) * M_SIN(v1);
                        if (oind == 4) DIR.z = -cos_theta;
                        else           DIR.z = +cos_theta;
                    }
                }
                Surface(M, &POS, &DIR);
                IndexG(M, &POS, &level, &ind);
            }
        }
    }
    if (SOURCE == 1) {
        POS.x = clampf(X0 + DX * Rand(rng), PEPS, NX - PEPS);
        POS.y = clampf(Y0 + DY * Rand(rng), PEPS, NY - PEPS);
        POS.z = clampf(Z0 + DZ * Rand(rng), PEPS, NZ - PEPS);
        cos_theta = M_SQRT(Rand(rng));
        phi       = TWOPI * Rand(rng);
        sin_theta = M_SQRT(1.0f - cos_theta * cos_theta);
        v1 = sin_theta * M_COS(phi);
        v2 = sin_theta * M_SIN(phi);
        switch (SIDE) {
        case 0: DIR.x =  cos_theta; DIR.y = v1; DIR.z = v2; break;
        case 1: DIR.x = -cos_theta; DIR.y = v1; DIR.z = v2; break;
        case 2: DIR.y =  cos_theta; DIR.x = v1; DIR.z = v2; break;
        case 3: DIR.y = -cos_theta; DIR.x = v1; DIR.z = v2; break;
        case 4: DIR.z =  cos_theta; DIR.x = v1; DIR.y = v2; break;
        case 5: DIR.z = -cos_theta; DIR.x = v1; DIR.y = v2; break;
        }
        PHOTONS = M->BG;
        IndexG(M, &POS, &level, &ind);
    }
    *pPOS = POS;  *pDIR = DIR;  *pPHOTONS = PHOTONS;  *plevel = level;  *pind = ind;
}

/* One work item of SimRAM_PB (kernel_ASOC.c:15-824).  Returns tally events. */
static long sim_pb_workitem(const orc_model *M, int id)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    const int AREA = 2 * (NX * NY + NY * NZ + NZ * NX);
    const int SOURCE = M->SOURCE, BATCH = M->BATCH;
    int   level = 0, ind = -1;
    f3    DIR = {0.0f, 0.0f, 0.0f}, POS = {0.0f, 0.0f, 0.0f};
    float PHOTONS = 0.0f;
    rng_t rng;
    surf_t E;
    long  nt = 0;

    seed_workitem(&rng, M->SEED, (uint64_t)id);
    if ((SOURCE == 1) && (id >= (8 * AREA))) return 0;
    if (SOURCE == 3) {
        /* packets recorded by an enclosing run, sent in from the model surface (kernel_ASOC.c:97-105,141-179,
         * 469-501): 100 work items per surface element of the file's discretisation (PACKETS = number of
         * elements), BATCH = a multiple of the Healpix pixel count, a pixel per packet in turn */
        if (!M->WITH_ROI_LOAD) return 0;
        if (id >= (100 * M->PACKETS)) return 0;
        const int *RD = M->ROI_DIM, NS = M->ROI_NSIDE;
        const int ielem = id % M->PACKETS;
        int   iside = ielem;
        float DX = 0.0f, DY = 0.0f;
        const float rd = NX / ((float)RD[0]);
        if (iside < (RD[1] * RD[2])) {
            DX = ((iside % RD[1]) + 0.5f) * rd;  DY = ((iside / RD[1]) + 0.5f) * rd;  iside = 0;
        } else {
            iside -= (RD[1] * RD[2]);
            if (iside < (RD[0] * RD[2])) {
                DX = ((iside % RD[0]) + 0.5f) * rd;  DY = ((iside / RD[0]) + 0.5f) * rd;  iside = 1;
            } else {
                iside -= (RD[0] * RD[2]);
                if (iside < (RD[0] * RD[1])) {
                    DX = ((iside % RD[0]) + 0.5f) * rd;  DY = ((iside / RD[0]) + 0.5f) * rd;  iside = 2;
                }
            }
        }
        const float X0 = (float)(NS * NS * 12.0 / (100.0 * BATCH));
        for (int III = 0; III < BATCH; III++) {
            float v1, v2;
            ind = III % (12 * NS * NS);
            PHOTONS = X0 * M->ROI_LOAD[(long)ielem * 12 * NS * NS + ind];
            if (PHOTONS <= 0.0f) continue;
            pixel2angles_ring(NS, ind, &v1, &v2);
            v1 += (Rand(&rng) - 0.5f) * 0.05f;
            v2 += (Rand(&rng) - 0.5f) * 0.05f;
            DIR.x = M_SIN(v2) * M_COS(v1);
            DIR.y = M_SIN(v2) * M_SIN(v1);
            DIR.z = M_COS(v2);
            if (iside == 0) {
                POS.y = DX + (-0.49f + 0.98f * Rand(&rng)) * rd;  POS.z = DY + (-0.49f + 0.98f * Rand(&rng)) * rd;
                POS.x = (DIR.x > 0.0f) ? (PEPS) : (NX - PEPS);
            }
            if (iside == 1) {
                POS.x = DX + (-0.49f + 0.98f * Rand(&rng)) * rd;  POS.z = DY + (-0.49f + 0.98f * Rand(&rng)) * rd;
                POS.y = (DIR.y > 0.0f) ? (PEPS) : (NY - PEPS);
            }
            if (iside == 2) {
                POS.x = DX + (-0.49f + 0.98f * Rand(&rng)) * rd;  POS.y = DY + (-0.49f + 0.98f * Rand(&rng)) * rd;
                POS.z = (DIR.z > 0.0f) ? (PEPS) : (NZ - PEPS);
            }
            IndexG(M, &POS, &level, &ind);
            nt += walk_packet(M, &rng, POS, DIR, PHOTONS, level, ind, 0, -1);
            ind = -1;
        }
        return nt;
    }
    pb_surface_element(M, id, &E);
    for (int III = 0; III < BATCH; III++) {
        pb_create(M, &E, III, &rng, &POS, &DIR, &PHOTONS, &level, &ind);
        nt += walk_packet(M, &rng, POS, DIR, PHOTONS, level, ind, 0, -1);
        ind = -1;
    }
    return nt;
}

/* One work item of the USE_EMWEIGHT==2 SimRAM_CL (kernel_ASOC.c:1693-2105): the host lists the emitting
 * cells in EMINDEX (a cell index 0 is skipped by the "> 0" test, -1 ends the list) and the packet weight of
 * each cell in EMWEI; 100 packets (EMWEI2_STEP, ASOC.py:79) per listed cell and launch. */
static long sim_cl2_workitem(const orc_model *M, int id)
{
    const int NX = M->NX, NY = M->NY, CELLS = M->CELLS, GLOBAL = M->GLOBAL, LEVELS = M->LEVELS;
    const int *LCELLS = M->LCELLS, *OFF = M->OFF;
    int   level = 0, ind, IND = id - GLOBAL, ICELL = -1;
    float phi, cos_theta, sin_theta, PHOTONS, X0, Y0, Z0, PWEI = 1.0f;
    f3    DIR, POS;
    rng_t rng;
    long  nt = 0;
    if (id >= CELLS) return 0;
    seed_workitem(&rng, M->SEED, (uint64_t)id);
    while (1) {
        while (1) {
            IND += GLOBAL;
            if (IND >= CELLS) return nt;
            if (M->EMINDEX[IND] < 0) return nt;
            if (M->EMINDEX[IND] > 0) {
                ICELL = M->EMINDEX[IND];
                PWEI  = M->EMWEI[ICELL];
                break;
            }
        }
        for (int iter = 0; iter < 100; iter++) {
            ind = ICELL;
            for (level = 0; level < LEVELS - 1; level++) {
                ind -= LCELLS[level];
                if (ind < 0) { ind += LCELLS[level]; break; }
            }
            if (level == 0) {
                X0 = (ind % NX);  Y0 = ((ind / NX) % NY);  Z0 = (ind / (NX * NY));
            } else {
                int sid = ind % 8;
                X0 = (sid % 2);  Y0 = ((sid % 4) > 1) ? 1.0f : 0.0f;  Z0 = (sid / 4);
            }
            PHOTONS = M->EMIT[OFF[level] + ind] * PWEI;
            POS.x = X0 + Rand(&rng);  POS.y = Y0 + Rand(&rng);  POS.z = Z0 + Rand(&rng);
            phi       = TWOPI * Rand(&rng);
            cos_theta = 0.999997f - 1.999995f * Rand(&rng);
            sin_theta = M_SQRT(1.0f - cos_theta * cos_theta);
            DIR.x = sin_theta * M_COS(phi);
            DIR.y = sin_theta * M_SIN(phi);
            DIR.z = cos_theta;
            nt += walk_packet(M, &rng, POS, DIR, PHOTONS, level, ind, 1, (M->WITH_ALI > 0) ? (OFF[level] + ind) : -1);
        }
    }
}

/* ================================ SimRAM_HP ============================================ */

/* Pixel2AnglesRing (kernel_ASOC_aux.c:987-1026): Healpix RING pixel -> (phi, theta) */
static void pixel2angles_ring(const int nside, const int ipix, float *phi, float *theta)
{
    int   nl2, nl4, npix, ncap, iring, iphi, ip, ipix1;
    float fact1, fact2, fodd, hip, fihip;
    npix  = 12 * nside * nside;
    ipix1 = ipix + 1;
    nl2   = 2 * nside;
    nl4   = 4 * nside;
    ncap  = 2 * nside * (nside - 1);
    fact1 = 1.5f * nside;
    fact2 = 3.0f * nside * nside;
    if (ipix1 <= ncap) {
        hip    = ipix1 / 2.0f;
        fihip  = (int)(hip);
        iring  = (int)(M_SQRT(hip - M_SQRT(fihip))) + 1;
        iphi   = ipix1 - 2 * iring * (iring - 1);
        *theta = M_ACOS(1.0f - iring * iring / fact2);
        *phi   = (iphi - 0.5f) * PI_F / (2.0f * iring);
    } else {
        if (ipix1 <= nl2 * (5 * nside + 1)) {
            ip     = ipix1 - ncap - 1;
            iring  = (int)(ip / nl4) + nside;
            iphi   = (ip % nl4) + 1;
            fodd   = 0.5f * (1 + (iring + nside) % 2);
            *theta = M_ACOS((nl2 - iring) / fact1);
            *phi   = (iphi - fodd) * PI_F / (2.0f * nside);
        } else {
            ip     = npix - ipix1 + 1;
            hip    = ip / 2.0f;
            fihip  = (int)(hip);
            iring  = (int)(M_SQRT(hip - M_SQRT(fihip))) + 1;
            iphi   = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
            *theta = M_ACOS(-1.0f + iring * iring / fact2);
            *phi   = (iphi - 0.5f) * PI_F / (2.0f * iring);
        }
    }
}

/* Healpix pixel of the next background packet: uniform (kernel_ASOC.c:881-884) or by
 * bisection on the cumulative probability, n_bisect steps + linear scan (:885-902;
 * the sca kernel bisects 12 times, kernel_ASOC_sca.c:118-131) */
static int hp_select_pixel(const orc_model *M, rng_t *rng, int n_bisect)
{
    int ind, ind0, level0;
    if (M->HPBG_WEIGHTED < 1) {
        ind = clampi((int)(M_FLOOR(Rand(rng) * 49152)), 0, 49151);
    } else {
        const float x = Rand(rng);
        ind0 = 0;
        level0 = 49151;
        for (int i = 0; i < n_bisect; i++) {
            ind = (ind0 + level0) / 2;
            if (M->HPBGP[ind] > x) level0 = ind;
            else                   ind0 = ind;
        }
        for (ind = ind0; ind <= level0; ind++) {
            if (M->HPBGP[ind] >= x) break;
        }
    }
    return ind;
}

/* One work item of SimRAM_HP (kernel_ASOC.c:826-1207): Healpix background, entry face chosen
 * with probability proportional to |DIR_i| (:929-945); the walk is SimRAM_PB's. */
static long sim_hp_workitem(const orc_model *M, int id)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    const int AREA = 2 * (NX * NY + NY * NZ + NZ * NX);
    int   level = 0, ind = -1;
    float phi, theta, x, y, z, ds, v1, v2, PHOTONS;
    f3    DIR, POS = {0.0f, 0.0f, 0.0f};
    rng_t rng;
    long  nt = 0;
    seed_workitem(&rng, M->SEED, (uint64_t)id);
    if (id >= (8 * AREA)) return 0;
    for (int III = 0; III < M->BATCH; III++) {
        ind     = hp_select_pixel(M, &rng, 10);
        PHOTONS = M->HPBG[ind];
        pixel2angles_ring(64, ind, &phi, &theta);
        DIR.x = +M_SIN(theta) * M_COS(phi);
        DIR.y = +M_SIN(theta) * M_SIN(phi);
        DIR.z = -M_COS(theta);
        if (fabsf(DIR.x) < DEPS) DIR.x = DEPS;
        if (fabsf(DIR.y) < DEPS) DIR.y = DEPS;
        if (fabsf(DIR.z) < DEPS) DIR.z = DEPS;
        normalize3(&DIR);
        x = fabsf(DIR.x);  y = fabsf(DIR.y);  z = fabsf(DIR.z);
        ds = x + y + z;  x /= ds;  y /= ds;  z /= ds;
        ds = Rand(&rng);  v1 = Rand(&rng);  v2 = Rand(&rng);
        if (ds < x) {
            POS.y = v1 * NY;  POS.z = v2 * NZ;
            POS.x = (DIR.x > 0.0f) ? (PEPS) : (NX - PEPS);
        } else {
            if (ds < (x + y)) {
                POS.x = v1 * NX;  POS.z = v2 * NZ;
                POS.y = (DIR.y > 0.0f) ? (PEPS) : (NY - PEPS);
            } else {
                POS.x = v1 * NX;  POS.y = v2 * NY;
                POS.z = (DIR.z > 0.0f) ? (PEPS) : (NZ - PEPS);
            }
        }
        IndexG(M, &POS, &level, &ind);
        nt += walk_packet(M, &rng, POS, DIR, PHOTONS, level, ind, 2, -1);
        ind = -1;
    }
    return nt;
}

/* ================================ SimRAM_CL ============================================ */

/* One work item of SimRAM_CL, USE_EMWEIGHT 0/1, WITH_ALI 0 (kernel_ASOC.c:1223-1689). */
static long sim_cl_workitem(const orc_model *M, int id)
{
    const int NX = M->NX, NY = M->NY, CELLS = M->CELLS, GLOBAL = M->GLOBAL, LEVELS = M->LEVELS;
    const int *LCELLS = M->LCELLS, *OFF = M->OFF;
    int   level = 0, batch;
    float phi, cos_theta, sin_theta;
    f3    DIR, POS;
    float PHOTONS, X0, Y0, Z0, PWEI = 1.0f;
    rng_t rng;
    long  nt = 0;
    if (id >= CELLS) return 0;
    seed_workitem(&rng, M->SEED, (uint64_t)id);
    int ICELL = id - GLOBAL;
    int IRAY = 0;
    int ind = -1;
    batch = -1;
    while (1) {
        if (IRAY >= batch) {
            IRAY = 0;
            PWEI = 1.0f;
            while (1) {
                ICELL += GLOBAL;
                if (ICELL >= CELLS) return nt;
                if (M->USE_EMWEIGHT > 0) {
                    PWEI = M->EMWEI[ICELL];
                    if ((PWEI < 1e-10f) || (M->DENS[ICELL] <= 0.0f)) continue;
                    batch = (int)M_FLOOR(PWEI);
                    if (batch < 1) {
                        batch = 1;  PWEI = 1.0 / (PWEI + 1.0e-30f);
                    } else {
                        PWEI = 1.0 / (batch + 1.0e-9f);
                    }
                } else {
                    batch = M->BATCH;
                    PWEI = 1.0f / (batch + 1.0e-9f);
                }
                break;
            }
        }
        ind = ICELL;
        IRAY += 1;
        for (level = 0; level < LEVELS - 1; level++) {
            ind -= LCELLS[level];
            if (ind < 0) {
                ind += LCELLS[level];
                break;
            }
        }
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
        PHOTONS = M->EMIT[OFF[level] + ind] * PWEI;
        POS.x = X0 + Rand(&rng);  POS.y = Y0 + Rand(&rng);  POS.z = Z0 + Rand(&rng);
        phi       = TWOPI * Rand(&rng);
        cos_theta = 0.999997f - 1.999995f * Rand(&rng);
        sin_theta = M_SQRT(1.0f - cos_theta * cos_theta);
        DIR.x = sin_theta * M_COS(phi);
        DIR.y = sin_theta * M_SIN(phi);
        DIR.z = cos_theta;
        nt += walk_packet(M, &rng, POS, DIR, PHOTONS, level, ind, 1, (M->WITH_ALI > 0) ? (OFF[level] + ind) : -1);
    }
}

/* ================================ scattered light: kernel_ASOC_sca.c ==================== */

#define MAX_SCATTERINGS 30            /* kernel_ASOC_sca.c:5 */

static inline void out_add(const orc_model *M, int i, float v)
{
    if (M->threaded) {
#pragma omp atomic
        M->OUT[i] += v;
    } else {
        M->OUT[i] += v;
    }
}

/* Everything after packet creation in the sca kernels: forced first scattering
 * (kernel_ASOC_sca.c:888-910 / 1232-1258), the tau-only walk (:922-946), the scattering
 * block with peel-off towards NDIR observers (:951-1047) and the new direction (:1056-1080).
 * variant 0 = SimRAM_PB; 1 = SimRAM_CL: no random draw when nothing lies along the line of
 * sight (:1249-1252) and a +-0.9999 clamp of cos(theta) (:1349); 2 = SimRAM_PS: the forced
 * first free path is evaluated in fp32, -log(1.0f - W*u) (:1742), where PB/CL promote to
 * double through the literal 1.0 (:906, :1256).
 * Returns the number of peel-off contributions added to OUT. */
/* RootPos (kernel_ASOC_aux.c:169-190): local position of cell (level, ind) -> root-grid coordinates */
static void RootPos(const orc_model *M, f3 *POS, const int ilevel, const int iind)
{
    int level = ilevel, ind = iind, sid;
    if (level == 0) return;
    while (level > 0) {
        ind = M->PAR[M->OFF[level] + ind - M->NX * M->NY * M->NZ];
        level--;
        if (level == 0) {
            POS->x *= 0.5f;  POS->y *= 0.5f;  POS->z *= 0.5f;
            POS->x += ind % M->NX;
            POS->y += (ind / M->NX) % M->NY;
            POS->z += ind / (M->NX * M->NY);
            return;
        } else {
            sid = ind % 8;
            POS->x *= 0.5f;  POS->y *= 0.5f;  POS->z *= 0.5f;
            POS->x += sid % 2;  POS->y += (sid / 2) % 2;  POS->z += sid / 4;
        }
    }
}

/* Angles2PixelRing (kernel_ASOC_aux.c:945-984): (phi, theta) -> Healpix RING pixel */
static int angles2pixel_ring(const int nside, float phi, float theta)
{
    int   nl2, nl4, ncap, npix, jp, jm, ipix1, ir, ip, kshift;
    float z, za, tt, tp, tmp;
    if ((theta < 0.0f) || (theta > PI_F)) return -1;
    z  = M_COS(theta);
    za = fabsf(z);
    if (phi >= TWOPI) phi -= TWOPI;
    if (phi < 0.0f)   phi += TWOPI;
    tt   = phi / 1.5707963268f;
    nl2  = 2 * nside;
    nl4  = 4 * nside;
    ncap = nl2 * (nside - 1);
    npix = 12 * nside * nside;
    if (za <= 0.6666666667f) {
        jp = (int)(nside * (0.5f + tt - z * 0.75f));
        jm = (int)(nside * (0.5f + tt + z * 0.75f));
        ir = nside + 1 + jp - jm;
        kshift = 0;
        if (ir % 2 == 0) kshift = 1;
        ip = (int)((jp + jm - nside + kshift + 1) / 2) + 1;
        if (ip > nl4) ip -= nl4;
        ipix1 = ncap + nl4 * (ir - 1) + ip;
    } else {
        tp  = tt - (int)(tt);
        tmp = M_SQRT(3.0f * (1.0f - za));
        jp  = (int)(nside * tp * tmp);
        jm  = (int)(nside * (1.0f - tp) * tmp);
        ir  = jp + jm + 1;
        ip  = (int)(tt * ir) + 1;
        if (ip > (4 * ir)) ip -= 4 * ir;
        ipix1 = 2 * ir * (ir - 1) + ip;
        if (z <= 0.0f) ipix1 = npix - 2 * ir * (ir + 1) + ip;
    }
    return (ipix1 - 1);
}

static long walk_packet_sca(const orc_model *M, rng_t *rng, f3 POS, f3 DIR, float PHOTONS, int level, int ind, int variant)
{
    const int conditioned = variant & 4;        /* direction already clamped + normalised by the caller (SimRAM_HP) */
    variant &= 3;
    const int is_cl = (variant == 1);
    const float *DENS = M->DENS;
    const int *OFF = M->OFF;
    const float ABS = M->ABS, SCA = M->SCA;
    const float CLAMP = is_cl ? 0.9999f : 0.999f;
    int   oind = 0, ind0 = -1, level0 = 0, scatterings, i, j, idust = 0;
    float ds, free_path, tau, dtau, delta, dx, cos_theta, W;
    f3    POS0, ODIR;
    long  nadd = 0;

    if (!conditioned) {
        if (fabsf(DIR.x) < DEPS) DIR.x = DEPS;
        if (fabsf(DIR.y) < DEPS) DIR.y = DEPS;
        if (fabsf(DIR.z) < DEPS) DIR.z = DEPS;
        normalize3(&DIR);
    }
    if (M->FFS > 0) {
        POS0 = POS;  ind0 = ind;  level0 = level;
        tau = 0.0f;
        while (ind0 >= 0) {
            oind = OFF[level0] + ind0;
            ds   = GetStep(M, &POS0, &DIR, &level0, &ind0);
            if (M->WITH_ABU) tau += ds * DENS[oind] * M->OPT[2 * (long)oind + 1];
            else             tau += ds * DENS[oind] * SCA;
        }
        if (tau < 1.0e-22f) {
            ind = -1;
            if (is_cl) return 0;
        }
        if (variant == 2) {
            W = -M_EXPM1(-tau);
            free_path = -M_LOG(1.0f - W * Rand(rng));
        } else {
            W = 1.0f - M_EXP(-tau);
            free_path = -M_LOGD(1.0 - W * Rand(rng));
        }
        PHOTONS *= W;
    } else {
        free_path = -M_LOG(Rand(rng));
    }
    scatterings = 0;
    while (ind >= 0) {
        tau = 0.0f;
        while (ind >= 0) {
            ind0 = ind;  level0 = level;  POS0 = POS;
            oind = OFF[level0] + ind0;
            ds   = GetStep(M, &POS, &DIR, &level, &ind);
            if (M->WITH_ABU) dtau = ds * DENS[oind] * M->OPT[2 * (long)oind + 1];
            else             dtau = ds * DENS[oind] * SCA;
            if (free_path < (tau + dtau)) {
                ind = ind0;                       /* level keeps its post-step value: reference quirk */
                break;
            }
            tau += dtau;
            if ((M->MIRROR > 0) && (ind < 0)) Mirror(M, &POS, &DIR, &level, &ind);   /* kernel_ASOC_sca.c:983, :1283, :1781 */
        }
        if (ind < 0) break;
        scatterings++;
        dtau = free_path - tau;
        if (M->WITH_ABU) dx = dtau / (M->OPT[2 * (long)oind + 1] * DENS[oind]);
        else             dx = dtau / (SCA * DENS[oind]);
        dx = M_LDEXP_UP(dx, level);
        POS0.x = POS0.x + dx * DIR.x;
        POS0.y = POS0.y + dx * DIR.y;
        POS0.z = POS0.z + dx * DIR.z;
        if (M->WITH_ABU) PHOTONS *= M_EXP(-free_path * M->OPT[2 * (long)oind] / M->OPT[2 * (long)oind + 1]);
        else             PHOTONS *= M_EXP(-free_path * ABS / SCA);
        if (M->NDIR < 0) {
            /* Healpix map seen by an observer at ODIRS[0] (kernel_ASOC_sca.c:319-361, :1021-1061, :1312-1358,
             * :1806-1846): direction and distance from the root position of the scattering, optical depth along
             * at most that distance, 1/d^2, pixel from the direction */
            f3 RP;
            POS = POS0;  ind = ind0;  level = level0;
            RP = POS;
            RootPos(M, &RP, level, ind);
            ODIR.x = M->ODIRS[0] - RP.x;  ODIR.y = M->ODIRS[1] - RP.y;  ODIR.z = M->ODIRS[2] - RP.z;
            dx = M_SQRT(ODIR.x * ODIR.x + ODIR.y * ODIR.y + ODIR.z * ODIR.z);
            delta = 1.0f / (dx * dx);
            normalize3(&ODIR);
            tau = 0.0f;
            while ((dx > 0) && (ind >= 0)) {
                oind = OFF[level] + ind;
                ds   = GetStep(M, &POS, &ODIR, &level, &ind);
                ds   = (dx < ds) ? dx : ds;
                if (variant == 0) ds = (float)((double)ds + 1.0e-6);      /* SimRAM_PB writes "+ 1.0e-6" (:982): double add */
                else              ds = ds + 1.0e-6f;                      /* :329, :1323, :1821 */
                dx  -= ds;
                if (M->WITH_ABU) tau += ds * DENS[oind] * (M->OPT[2 * (long)oind] + M->OPT[2 * (long)oind + 1]);
                else             tau += ds * DENS[oind] * (ABS + SCA);
            }
            cos_theta = clampf(DIR.x * ODIR.x + DIR.y * ODIR.y + DIR.z * ODIR.z, -CLAMP, +CLAMP);
            idust = (M->MSF_NDUST > 1) ? MsfDust(M, rng, OFF[level0] + ind0) : 0;     /* one draw per peel-off (:339-347) */
            if (is_cl) {
                const float G = 0.65f;
                const float fraction = (1.0f / (4.0f * PI_F)) * (1.0f - G * G) / M_POW15(1.0f + G * G - 2.0f * G * cos_theta);
                delta *= PHOTONS * fraction * ((tau > TAULIM) ? (1.0f - M_EXP(-tau)) : (tau * (1.0f - 0.5f * tau)));
            } else {
                delta *= PHOTONS * M_EXP(-tau) * M->DSC[(long)idust * M->BINS + clampi((int)(M->BINS * (1.0f + cos_theta) * 0.5f), 0, M->BINS - 1)];
            }
            {
                const float theta = M_ACOS(-ODIR.z);
                const float phi   = M_ATAN2(+ODIR.y, +ODIR.x);
                i = angles2pixel_ring(-M->NDIR, phi, theta);
                out_add(M, i, delta);
                nadd++;
            }
        }
        for (int idir = 0; idir < M->NDIR; idir++) {
            POS = POS0;  ind = ind0;  level = level0;
            tau = 0.0f;
            ODIR.x = M->ODIRS[4 * idir];  ODIR.y = M->ODIRS[4 * idir + 1];  ODIR.z = M->ODIRS[4 * idir + 2];
            while (ind >= 0) {
                oind = OFF[level] + ind;
                ds   = GetStep(M, &POS, &ODIR, &level, &ind);
                if (M->WITH_ABU) tau += ds * DENS[oind] * (M->OPT[2 * (long)oind] + M->OPT[2 * (long)oind + 1]);
                else             tau += ds * DENS[oind] * (ABS + SCA);
            }
            cos_theta = clampf(DIR.x * ODIR.x + DIR.y * ODIR.y + DIR.z * ODIR.z, -CLAMP, +CLAMP);
            idust = (M->MSF_NDUST > 1) ? MsfDust(M, rng, OFF[level0] + ind0) : 0;     /* one draw per observer (:382-390) */
            if (is_cl) {
                /* kernel_ASOC_aux.c:1 has "#define HG_TEST 0" and SimRAM_CL tests "#ifdef HG_TEST"
                 * (:1387): the reference as shipped takes the analytic branch -- Henyey-Greenstein
                 * with g=0.65 and the factor (1-exp(-tau)) [tau>TAULIM] or tau*(1-tau/2) -- and
                 * never reads DSC.  Restated as compiled. */
                const float G = 0.65f;
                const float fraction = (1.0f / (4.0f * PI_F)) * (1.0f - G * G) / M_POW15(1.0f + G * G - 2.0f * G * cos_theta);
                delta = PHOTONS * fraction * ((tau > TAULIM) ? (1.0f - M_EXP(-tau)) : (tau * (1.0f - 0.5f * tau)));
            } else {
                delta = PHOTONS * M_EXP(-tau) * M->DSC[(long)idust * M->BINS + clampi((int)(M->BINS * (1.0f + cos_theta) * 0.5f), 0, M->BINS - 1)];
            }
            POS.x -= M->CX;  POS.y -= M->CY;  POS.z -= M->CZ;
            i = (0.5f * M->NPIX_X - 0.00005f) + (POS.x * M->ORA[4 * idir] + POS.y * M->ORA[4 * idir + 1] + POS.z * M->ORA[4 * idir + 2]) / M->MAP_DX;
            j = (0.5f * M->NPIX_Y - 0.00005f) + (POS.x * M->ODE[4 * idir] + POS.y * M->ODE[4 * idir + 1] + POS.z * M->ODE[4 * idir + 2]) / M->MAP_DX;
            if ((i >= 0) && (j >= 0) && (i < M->NPIX_X) && (j < M->NPIX_Y)) {
                i += idir * M->NPIX_X * M->NPIX_Y + j * M->NPIX_X;
                out_add(M, i, delta);
                nadd++;
            }
        }
        POS = POS0;  ind = ind0;  level = level0;
        idust = (M->MSF_NDUST > 1) ? MsfDust(M, rng, OFF[level0] + ind0) : 0;         /* :424-432 */
        Scatter(&DIR, M->CSC + (long)idust * M->BINS, M->BINS, rng);
        free_path = -M_LOG(Rand(rng));
        if (scatterings == MAX_SCATTERINGS) { ind = -1; continue; }
    }
    return nadd;
}

/* One work item of the sca SimRAM_PB (kernel_ASOC_sca.c:471-1094); SimRAM_PS (:1462-1938) is
 * its SOURCE==0 path with the launch arguments of ASOCS.py:655-665. */
static long sim_sca_pb_workitem(const orc_model *M, int id, int variant)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    const int AREA = 2 * (NX * NY + NY * NZ + NZ * NX);
    int   level = 0, ind = -1;
    f3    DIR = {0.0f, 0.0f, 0.0f}, POS = {0.0f, 0.0f, 0.0f};
    float PHOTONS = 0.0f;
    rng_t rng;
    surf_t E;
    long  n = 0;
    seed_workitem(&rng, M->SEED, (uint64_t)id);
    if ((M->SOURCE == 1) && (id >= (8 * AREA))) return 0;
    pb_surface_element(M, id, &E);
    for (int III = 0; III < M->BATCH; III++) {
        pb_create(M, &E, III, &rng, &POS, &DIR, &PHOTONS, &level, &ind);
        n += walk_packet_sca(M, &rng, POS, DIR, PHOTONS, level, ind, variant);
        ind = -1;
    }
    return n;
}

/* One work item of the sca SimRAM_CL (kernel_ASOC_sca.c:1098-1461) */
static long sim_sca_cl_workitem(const orc_model *M, int id)
{
    const int NX = M->NX, NY = M->NY, CELLS = M->CELLS, GLOBAL = M->GLOBAL, LEVELS = M->LEVELS;
    const int *LCELLS = M->LCELLS, *OFF = M->OFF;
    int   level = 0, batch = -1, ICELL = id - GLOBAL, IRAY = 0, ind;
    float phi, cos_theta, sin_theta, PHOTONS, X0, Y0, Z0, PWEI = 1.0f;
    f3    DIR, POS;
    rng_t rng;
    long  n = 0;
    if (id >= CELLS) return 0;
    seed_workitem(&rng, M->SEED, (uint64_t)id);
    while (1) {
        if (IRAY >= batch) {
            IRAY = 0;
            PWEI = 1.0f;
            while (1) {
                ICELL += GLOBAL;
                if (ICELL >= CELLS) return n;
                if (M->USE_EMWEIGHT > 0) {
                    PWEI = M->EMWEI[ICELL];
                    if ((PWEI < 1e-10f) || (M->DENS[ICELL] <= 0.0f)) continue;
                    batch = (int)M_FLOOR(PWEI);
                    if (batch < 1) { batch = 1;  PWEI = 1.0 / (PWEI + 1.0e-30f); }
                    else           { PWEI = 1.0 / (batch + 1.0e-9f); }
                } else {
                    batch = M->BATCH;
                    PWEI = 1.0f / (batch + 1.0e-9f);
                }
                break;
            }
        }
        ind = ICELL;
        IRAY += 1;
        for (level = 0; level < LEVELS - 1; level++) {
            ind -= LCELLS[level];
            if (ind < 0) { ind += LCELLS[level]; break; }
        }
        if (level == 0) {
            X0 = (ind % NX);  Y0 = ((ind / NX) % NY);  Z0 = (ind / (NX * NY));
        } else {
            int sid = ind % 8;
            X0 = (sid % 2);  Y0 = ((sid % 4) > 1) ? 1.0f : 0.0f;  Z0 = (sid / 4);
        }
        PHOTONS = M->EMIT[OFF[level] + ind] * PWEI;
        POS.x = X0 + Rand(&rng);  POS.y = Y0 + Rand(&rng);  POS.z = Z0 + Rand(&rng);
        phi       = TWOPI * Rand(&rng);
        cos_theta = 0.999997f - 1.999995f * Rand(&rng);
        sin_theta = M_SQRT(1.0f - cos_theta * cos_theta);
        DIR.x = sin_theta * M_COS(phi);
        DIR.y = sin_theta * M_SIN(phi);
        DIR.z = cos_theta;
        n += walk_packet_sca(M, &rng, POS, DIR, PHOTONS, level, ind, 1);
    }
}

/* One work item of the sca SimRAM_HP (kernel_ASOC_sca.c:40-470): Healpix background; the packet is
 * aimed at a disc of radius Rout perpendicular to its direction, on the upstream side of the cloud, and
 * enters through Surface() -- packets that miss are skipped.  The walk is SimRAM_CL's (same clamp,
 * HG_TEST weight, no draw when the line of sight is empty). */
static long sim_sca_hp_workitem(const orc_model *M, int id)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    const float Rout = 0.5f * M_SQRT(1.0f * NX * NX + NY * NY + NZ * NZ);
    int   level = 0, ind = -1;
    float phi, theta, ds, dx, PHOTONS;
    f3    DIR, POS, POS0;
    rng_t rng;
    long  n = 0;
    seed_workitem(&rng, M->SEED, (uint64_t)id);
    for (int III = 0; III < M->BATCH; III++) {
        ind     = hp_select_pixel(M, &rng, 12);
        PHOTONS = M->HPBG[ind];
        pixel2angles_ring(64, ind, &phi, &theta);
        DIR.x = +M_SIN(theta) * M_COS(phi);
        DIR.y = +M_SIN(theta) * M_SIN(phi);
        DIR.z = -M_COS(theta);
        if (fabsf(DIR.x) < DEPS) DIR.x = DEPS;
        if (fabsf(DIR.y) < DEPS) DIR.y = DEPS;
        if (fabsf(DIR.z) < DEPS) DIR.z = DEPS;
        normalize3(&DIR);
        ds = 2.0f * PI_F * Rand(&rng);
        dx = M_SQRT(Rand(&rng));
        POS.x = dx * M_COS(ds);
        POS.y = dx * M_SIN(ds);
        POS.z = M_SQRT(1.001f - dx * dx);
        POS0.x = POS.x * M_COS(theta) + POS.z * M_SIN(theta);
        POS0.y = POS.y;
        POS0.z = -POS.x * M_SIN(theta) + POS.z * M_COS(theta);
        POS.x = POS0.x * M_COS(PI_F - phi) + POS0.y * M_SIN(PI_F - phi);
        POS.y = -POS0.x * M_SIN(PI_F - phi) + POS0.y * M_COS(PI_F - phi);
        POS.z = POS0.z;
        POS.x = 0.5f * NX + Rout * POS.x;
        POS.y = 0.5f * NY + Rout * POS.y;
        POS.z = 0.5f * NZ + Rout * POS.z;
        Surface(M, &POS, &DIR);
        IndexG(M, &POS, &level, &ind);
        if (ind < 0) continue;
        n += walk_packet_sca(M, &rng, POS, DIR, PHOTONS, level, ind, 1 | 4);
    }
    return n;
}

/* ================================ map making (kernel_ASOC_map.c) ======================== */
/* The map file carries its own copies of the traversal: PEPS = 5e-4, EPS = 2.5e-4 (:10-11), Index() in
 * double for NX > 100 whatever LEVELS (:297-301) and with the climb test of :345. */
#define PEPS_MAP 5.0e-4f
#define EPS_MAP  2.5e-4f

#define INDEX_MAP_VARIANT
#define REAL float
#define RFLOOR(x) M_FLOOR(x)
#define RFMOD1(x) M_FMOD1(x)
#define INDEX_NAME IndexMap_f
#include "soc_oracle_index.inc"
#undef REAL
#undef RFLOOR
#undef RFMOD1
#undef INDEX_NAME
#define REAL double
#define RFLOOR(x) floor(x)
#define RFMOD1(x) M_FMOD1D(x)
#define INDEX_NAME IndexMap_d
#include "soc_oracle_index.inc"
#undef REAL
#undef RFLOOR
#undef RFMOD1
#undef INDEX_NAME
#undef INDEX_MAP_VARIANT

static float GetStepMap(const orc_model *M, f3 *POS, const f3 *DIR, int *level, int *ind)
{
    float dx, dy, dz;
    dx = (DIR->x > 0.0f) ? ((1.0f + PEPS_MAP - M_FMOD1(POS->x)) / DIR->x) : ((-PEPS_MAP - M_FMOD1(POS->x)) / DIR->x);
    dy = (DIR->y > 0.0f) ? ((1.0f + PEPS_MAP - M_FMOD1(POS->y)) / DIR->y) : ((-PEPS_MAP - M_FMOD1(POS->y)) / DIR->y);
    dz = (DIR->z > 0.0f) ? ((1.0f + PEPS_MAP - M_FMOD1(POS->z)) / DIR->z) : ((-PEPS_MAP - M_FMOD1(POS->z)) / DIR->z);
    dx = fminf(dx, fminf(dy, dz));
    POS->x += dx * DIR->x;
    POS->y += dx * DIR->y;
    POS->z += dx * DIR->z;
    dx = M_LDEXP_DN(dx, *level);
    if (M->NX > 100) IndexMap_d(M, POS, level, ind);
    else             IndexMap_f(M, POS, level, ind);
    return dx;
}

static int map_outside(const orc_model *M, f3 T)
{
    return (T.x < 0.0f) || (T.x > M->NX) || (T.y < 0.0f) || (T.y > M->NY) || (T.z < 0.0f) || (T.z > M->NZ);
}

/* One neighbour of the MAP_INTERPOLATION block (kernel_ASOC_map.c:716-731, :771-788): from the middle of the step, the
 * distance (in cell units) to the next cell along +V, else along -V (V is flipped for good), else "no neighbour". */
static void map_neighbour(const orc_model *M, const float *EMIT, f3 POS0, f3 TMP, float w, int level0, int ind0, float K,
                          f3 *V, float lim, int second_try_unscaled, float *dist, float *ndens, float *nemit)
{
    for (int attempt = 0; attempt < 2; attempt++) {
        int   slevel = level0, sind = ind0;
        f3    MPOS;
        float a;
        if (attempt) { V->x *= -1.0f;  V->y *= -1.0f;  V->z *= -1.0f; }
        MPOS.x = POS0.x + w * TMP.x;  MPOS.y = POS0.y + w * TMP.y;  MPOS.z = POS0.z + w * TMP.z;
        a = GetStepMap(M, &MPOS, V, &slevel, &sind);
        if (!(attempt && second_try_unscaled)) a /= K;        /* (:736 has no "b /= K" in the MAP_INTERPOLATION==2 block) */
        if ((a <= lim) && (sind >= 0)) {
            *dist = a;  *ndens = M->DENS[M->OFF[slevel] + sind];  *nemit = EMIT[M->OFF[slevel] + sind];
            return;
        }
    }
    *dist = 0.5f;  *ndens = 0.0f;  *nemit = 0.0f;
}

/* Mapping (kernel_ASOC_map.c:496-888, with -D MAP_INTERPOLATION, ROI_MAP and LEVEL_THRESHOLD as model fields): line-of-sight
 * integral of the emission with extinction for the pixels of an orthographic map, or of an all-sky
 * (longitude x latitude) image seen from INTOBS when INTOBS[0] > -1e10.  One call = all pixels.
 * mode 1: HealpixMapping (:890-970), NSIDE = NPIX_X, seen from INTOBS. */
__attribute__((visibility("default"))) void orc_mapping(const orc_model *M, int mode, float MAP_DX, int NPIX_X, int NPIX_Y, const float *EMIT, const float *DIRv,
                        const float *RAv, const float *DEv, const float *CENTREv, const float *INTOBSv, int SAVE_COLDEN,
                        float LENGTH, float *MAP, float *SAVETAU)
{
    const int NX = M->NX, NY = M->NY, NZ = M->NZ;
    const f3 DIR = { DIRv[0], DIRv[1], DIRv[2] }, RA = { RAv[0], RAv[1], RAv[2] }, DE = { DEv[0], DEv[1], DEv[2] };
    const f3 CENTRE = { CENTREv[0], CENTREv[1], CENTREv[2] }, INTOBS = { INTOBSv[0], INTOBSv[1], INTOBSv[2] };
    const int npix = mode ? 12 * NPIX_X * NPIX_X : NPIX_X * NPIX_Y;
    for (int id = 0; id < npix; id++) {
        float DTAU, TAU = 0.0f, PHOTONS = 0.0f, colden = 0.0f, sx, sy, sz, dens, emit;
        f3    POS, TMP;
        int   ind, level = 0, oind;
        const int i = mode ? 0 : id % NPIX_X, j = mode ? 0 : id / NPIX_X;
        if (mode) {
            float phi, theta;
            pixel2angles_ring(NPIX_X, id, &phi, &theta);
            TMP.x = -M_SIN(theta) * M_COS(phi);
            TMP.y = -M_SIN(theta) * M_SIN(phi);
            TMP.z = +M_COS(theta);
            if (fabsf(TMP.x) < 1.0e-5f) TMP.x = 1.0e-5f;
            if (fabsf(TMP.y) < 1.0e-5f) TMP.y = 1.0e-5f;
            if (fabsf(TMP.z) < 1.0e-5f) TMP.z = 1.0e-5f;
            POS = INTOBS;
            if ((M_FMOD1(POS.x) < 1.0e-5f) || (M_FMOD1(POS.x) < 0.99999f)) POS.x += 2.0e-5f;      /* :923-925, as written */
            if ((M_FMOD1(POS.y) < 1.0e-5f) || (M_FMOD1(POS.y) < 0.99999f)) POS.y += 2.0e-5f;
            if ((M_FMOD1(POS.z) < 1.0e-5f) || (M_FMOD1(POS.z) < 0.99999f)) POS.z += 2.0e-5f;
        } else if (INTOBS.x > -1e10f) {
            float phi = TWOPI * i / (float)(NPIX_X);
            phi += PI_F;
            const float pix = TWOPI / NPIX_X;
            const float theta = pix * (j - (NPIX_Y - 1) / 2);
            POS = INTOBS;
            TMP.x = M_COS(theta) * M_COS(phi);
            TMP.y = M_COS(theta) * M_SIN(phi);
            TMP.z = M_SIN(theta);
            if (fabsf(TMP.x) < 1.0e-5f) TMP.x = 1.0e-5f;
            if (fabsf(TMP.y) < 1.0e-5f) TMP.y = 1.0e-5f;
            if (fabsf(TMP.z) < 1.0e-5f) TMP.z = 1.0e-5f;
            if (M_FMOD1(POS.x) < 1.0e-5f) POS.x += 2.0e-5f;
            if (M_FMOD1(POS.y) < 1.0e-5f) POS.y += 2.0e-5f;
            if (M_FMOD1(POS.z) < 1.0e-5f) POS.z += 2.0e-5f;
        } else {
            POS.x = CENTRE.x + (i - 0.5f * (NPIX_X - 1)) * MAP_DX * RA.x + (j - 0.5f * (NPIX_Y - 1)) * MAP_DX * DE.x;
            POS.y = CENTRE.y + (i - 0.5f * (NPIX_X - 1)) * MAP_DX * RA.y + (j - 0.5f * (NPIX_Y - 1)) * MAP_DX * DE.y;
            POS.z = CENTRE.z + (i - 0.5f * (NPIX_X - 1)) * MAP_DX * RA.z + (j - 0.5f * (NPIX_Y - 1)) * MAP_DX * DE.z;
            POS.x += (NX + NY + NZ) * DIR.x;  POS.y += (NX + NY + NZ) * DIR.y;  POS.z += (NX + NY + NZ) * DIR.z;
            if (NX < 200) {
                if (DIR.x >= 0.0f) sx = (NX - POS.x) / (-DIR.x) + EPS_MAP;  else sx = (0.0f - POS.x) / (-DIR.x) + EPS_MAP;
                if (DIR.y >= 0.0f) sy = (NY - POS.y) / (-DIR.y) + EPS_MAP;  else sy = (0.0f - POS.y) / (-DIR.y) + EPS_MAP;
                if (DIR.z >= 0.0f) sz = (NZ - POS.z) / (-DIR.z) + EPS_MAP;  else sz = (0.0f - POS.z) / (-DIR.z) + EPS_MAP;
                TMP.x = POS.x - sx * DIR.x;  TMP.y = POS.y - sx * DIR.y;  TMP.z = POS.z - sx * DIR.z;
                if (map_outside(M, TMP)) sx = 1e10f;
                TMP.x = POS.x - sy * DIR.x;  TMP.y = POS.y - sy * DIR.y;  TMP.z = POS.z - sy * DIR.z;
                if (map_outside(M, TMP)) sy = 1e10f;
                TMP.x = POS.x - sz * DIR.x;  TMP.y = POS.y - sz * DIR.y;  TMP.z = POS.z - sz * DIR.z;
                if (map_outside(M, TMP)) sz = 1e10f;
                sx = fminf(sx, fminf(sy, sz));
                POS.x = POS.x - sx * DIR.x;  POS.y = POS.y - sx * DIR.y;  POS.z = POS.z - sx * DIR.z;
            } else {
                const float ex = (DIR.x > 0.0f) ? (-EPS_MAP) : (+EPS_MAP), ey = (DIR.y > 0.0f) ? (-EPS_MAP) : (+EPS_MAP),
                            ez = (DIR.z > 0.0f) ? (-EPS_MAP) : (+EPS_MAP);
                if (DIR.x >= 0.0f) sx = (NX - POS.x) / (-DIR.x);  else sx = (0.0f - POS.x) / (-DIR.x);
                if (DIR.y >= 0.0f) sy = (NY - POS.y) / (-DIR.y);  else sy = (0.0f - POS.y) / (-DIR.y);
                if (DIR.z >= 0.0f) sz = (NZ - POS.z) / (-DIR.z);  else sz = (0.0f - POS.z) / (-DIR.z);
                TMP.x = POS.x - sx * DIR.x;  TMP.y = POS.y - sx * DIR.y;  TMP.z = POS.z - sx * DIR.z;
                TMP.x += ex;  TMP.y += ey;  TMP.z += ez;
                if (map_outside(M, TMP)) sx = 1e10f;
                TMP.x = POS.x - sy * DIR.x;  TMP.y = POS.y - sy * DIR.y;  TMP.z = POS.z - sy * DIR.z;
                TMP.x += ex;  TMP.y += ey;  TMP.z += ez;
                if (map_outside(M, TMP)) sy = 1e10f;
                TMP.x = POS.x - sz * DIR.x;  TMP.y = POS.y - sz * DIR.y;  TMP.z = POS.z - sz * DIR.z;
                TMP.x += ex;  TMP.y += ey;  TMP.z += ez;
                if (map_outside(M, TMP)) sz = 1e10f;
                sx = fminf(sx, fminf(sy, sz));
                POS.x = POS.x - sx * DIR.x;  POS.y = POS.y - sx * DIR.y;  POS.z = POS.z - sx * DIR.z;
                POS.x += ex;  POS.y += ey;  POS.z += ez;
            }
            TMP.x = -DIR.x;  TMP.y = -DIR.y;  TMP.z = -DIR.z;
            if (fabsf(TMP.x) < 1.0e-5f) TMP.x = 1.0e-5f;
            if (fabsf(TMP.y) < 1.0e-5f) TMP.y = 1.0e-5f;
            if (fabsf(TMP.z) < 1.0e-5f) TMP.z = 1.0e-5f;
        }
        IndexG(M, &POS, &level, &ind);
        const int MI = mode ? 0 : M->MAP_INTERPOLATION;      /* (Mapping only: HealpixMapping has no such block) */
        f3 ADIR = { 0.0f, 0.0f, 0.0f }, BDIR = ADIR;
        if (MI > 0) {                                        /* two directions perpendicular to the ray (:664-682) */
            if (fabsf(TMP.x) > fabsf(TMP.y)) {
                if (fabsf(TMP.z) > fabsf(TMP.x)) { ADIR.x = 0.0005f;  ADIR.y = 1.0f;  ADIR.z = -TMP.y / TMP.z; }
                else                             { ADIR.x = -TMP.z / TMP.x;  ADIR.y = 0.0005f;  ADIR.z = 1.0f; }
            } else {
                if (fabsf(TMP.z) > fabsf(TMP.y)) { ADIR.x = 0.0005f;  ADIR.y = 1.0f;  ADIR.z = -TMP.y / TMP.z; }
                else                             { ADIR.x = 1.0f;  ADIR.y = -TMP.x / TMP.y;  ADIR.z = 0.0005f; }
            }
            normalize3(&ADIR);
            BDIR.x = TMP.y * ADIR.z - TMP.z * ADIR.y;
            BDIR.y = TMP.z * ADIR.x - TMP.x * ADIR.z;
            BDIR.z = TMP.x * ADIR.y - TMP.y * ADIR.x;
            normalize3(&BDIR);
        }
        while (ind >= 0) {
            const int olevel = level;
            const f3  POS0 = POS;
            const int ind0 = ind, level0 = level;
            oind = M->OFF[level] + ind;
            sx   = GetStepMap(M, &POS, &TMP, &level, &ind);
            dens = M->DENS[oind];
            emit = EMIT[oind];
            if (MI > 0) {
                const float K = ldexpf(1.0f, -level0);       /* local -> root-grid length */
                float a, b, Adens, Bdens, Aemit, Bemit;
                if (MI == 2) {                               /* steps of at most 0.22 cells (:709-715) */
                    a = 0.22f * K;
                    if (sx > a) {
                        sx = a;
                        POS.x = POS0.x + 0.22f * TMP.x;  POS.y = POS0.y + 0.22f * TMP.y;  POS.z = POS0.z + 0.22f * TMP.z;
                        ind = ind0;  level = level0;
                        if (M->NX > 100) IndexMap_d(M, &POS, &level, &ind);
                        else             IndexMap_f(M, &POS, &level, &ind);
                    }
                }
                const float w = 0.5f * sx / K;               /* to the middle of the step, local units */
                map_neighbour(M, EMIT, POS0, TMP, w, level0, ind0, K, &ADIR, (MI == 2) ? 0.52f : 0.502f, 0, &a, &Adens, &Aemit);
                map_neighbour(M, EMIT, POS0, TMP, w, level0, ind0, K, &BDIR, (MI == 2) ? 0.52f : 0.502f, MI == 2, &b, &Bdens, &Bemit);
                if (MI == 2) {                               /* :746-751 */
                    a = clampf(a, 0.0f, 0.51f);
                    b = clampf(b, 0.0f, 0.51f);
                    dens = (0.5f - a) * Adens + (0.5f - b) * Bdens + (a + b) * dens;
                    emit = (0.5f - a) * Aemit + (0.5f - b) * Bemit + (a + b) * emit;
                } else {                                     /* :806-808 */
                    a = 0.5f - a;  b = 0.5f - b;
                    dens = (1.0f - a - b) * dens + a * Adens + b * Bdens;
                    emit = (1.0f - a - b) * emit + a * Aemit + b * Bemit;
                }
            }
            if (M->WITH_ABU) DTAU = sx * dens * (M->OPT[2 * (long)oind] + M->OPT[2 * (long)oind + 1]);
            else             DTAU = sx * dens * (M->SCA + M->ABS);
            if (M->ROI_MAP && (InRoi(M, olevel, oind - M->OFF[olevel]) < 0)) {
                /* -D ROI_MAP: emission from ROI only (Mapping and HealpixMapping alike) */
            } else
            if ((mode == 0) && (M->LEVEL_THRESHOLD > 0) && (olevel < M->LEVEL_THRESHOLD)) {
                /* Mapping with -D LEVEL_THRESHOLD: no emission from coarser levels, extinction as usual (:825-834); HealpixMapping has no such test */
            } else
            if (DTAU < 1.0e-3f) PHOTONS += M_EXP(-TAU) * (1.0f - 0.5f * DTAU) * sx * emit * dens;
            else                PHOTONS += M_EXP(-TAU) * ((1.0f - M_EXP(-DTAU)) / DTAU) * sx * emit * dens;
            TAU += DTAU;
            if (mode || (SAVE_COLDEN > 0)) colden += sx * dens;
        }
        MAP[id] = PHOTONS;
        if (SAVE_COLDEN) SAVETAU[id] = colden * LENGTH;
        else             SAVETAU[id] = TAU;
    }
}

/* ================================ exported API ========================================= */

#define EXPORT __attribute__((visibility("default")))

EXPORT int orc_math_mode(void)
{
#ifdef SOC_ORACLE_LIBM
    return 0;
#else
    return 1;
#endif
}

EXPORT uint64_t orc_seed_base(float SEED) { return seed_base(SEED); }

EXPORT void orc_seed(float SEED, uint64_t gid, uint32_t *x, uint32_t *c)
{
    rng_t s;
    seed_workitem(&s, SEED, gid);
    *x = s.x;
    *c = s.c;
}

EXPORT void orc_draws(uint32_t *x, uint32_t *c, int n, uint32_t *out_uint, float *out_rand)
{
    rng_t s = { *x, *c };
    for (int i = 0; i < n; i++) {
        rng_t t = s;
        uint32_t u = NextUint(&s);
        if (out_uint) out_uint[i] = u;
        if (out_rand) out_rand[i] = Rand(&t);
    }
    *x = s.x;
    *c = s.c;
}

/* Parents kernel, kernel_ASOC_aux.c:688-718.  PAR has CELLS-NX*NY*NZ entries. */
EXPORT void orc_parents(const orc_model *M, int *PAR)
{
    const int NXYZ = M->NX * M->NY * M->NZ;
    for (int level = 0; level < (M->LEVELS - 1); level++) {
        for (int ipar = 0; ipar < M->LCELLS[level]; ipar++) {
            float link = M->DENS[M->OFF[level] + ipar];
            if (link < 1.0e-10f) {
                link = -link;
                int ind;
                memcpy(&ind, &link, 4);
                for (int i = 0; i < 8; i++) PAR[M->OFF[level + 1] - NXYZ + ind + i] = ipar;
            }
        }
    }
}

EXPORT void orc_indexg(const orc_model *M, float *pos, int *level, int *ind)
{
    f3 p = { pos[0], pos[1], pos[2] };
    IndexG(M, &p, level, ind);
    pos[0] = p.x; pos[1] = p.y; pos[2] = p.z;
}

/* Follow one ray from a global position until it leaves the grid (no scattering):
 * records (level, ind, ds) of each step.  Returns the number of steps taken. */
EXPORT int orc_trace(const orc_model *M, const float *pos, const float *dir, int maxsteps,
                     int *levels, int *inds, float *dss, float *endpos)
{
    f3 POS = { pos[0], pos[1], pos[2] }, DIR = { dir[0], dir[1], dir[2] };
    int level = 0, ind = -1, n = 0;
    IndexG(M, &POS, &level, &ind);
    while ((ind >= 0) && (n < maxsteps)) {
        levels[n] = level;
        inds[n]   = ind;
        dss[n]    = GetStep(M, &POS, &DIR, &level, &ind);
        n++;
    }
    endpos[0] = POS.x; endpos[1] = POS.y; endpos[2] = POS.z;
    return n;
}

EXPORT void orc_scatter(float *dir, const float *CSC, int BINS, uint32_t *x, uint32_t *c)
{
    f3 D = { dir[0], dir[1], dir[2] };
    rng_t s = { *x, *c };
    Scatter(&D, CSC, BINS, &s);
    dir[0] = D.x; dir[1] = D.y; dir[2] = D.z;
    *x = s.x; *c = s.c;
}

EXPORT void orc_deflect(float *dir, float cos_theta, float phi)
{
    f3 D = { dir[0], dir[1], dir[2] };
    Deflect(&D, cos_theta, phi);
    dir[0] = D.x; dir[1] = D.y; dir[2] = D.z;
}

/* Run work items gid0, gid0+stride, ... < gid1 of SimRAM_PB (kind 0) or SimRAM_CL (kind 1).
 * nthreads <= 1: sequential, deterministic summation order (= the reference run one
 * work item after another).  Returns the number of tally events (TABS updates). */
EXPORT long orc_sim(orc_model *M, int kind, int gid0, int gid1, int stride, int nthreads)
{
    long total = 0;
    if (stride < 1) stride = 1;
    if (nthreads <= 1) {
        M->threaded = 0;
        for (int id = gid0; id < gid1; id += stride)
            total += (kind == 2) ? sim_hp_workitem(M, id) : (kind ? ((M->USE_EMWEIGHT == 2) ? sim_cl2_workitem(M, id) : sim_cl_workitem(M, id)) : sim_pb_workitem(M, id));
    } else {
        M->threaded = 1;
        const long n = ((long)gid1 - gid0 + stride - 1) / stride;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total) num_threads(nthreads)
        for (long k = 0; k < n; k++) {
            int id = (int)(gid0 + k * stride);
            total += (kind == 2) ? sim_hp_workitem(M, id) : (kind ? ((M->USE_EMWEIGHT == 2) ? sim_cl2_workitem(M, id) : sim_cl_workitem(M, id)) : sim_pb_workitem(M, id));
        }
    }
    return total;
}

/* scattered-light kernels: work items gid0, gid0+stride, ... of the sca SimRAM_PB (kind 0),
 * SimRAM_CL (kind 1) or SimRAM_PS (kind 2, needs SOURCE == 0); returns the number of image contributions */
EXPORT long orc_sim_sca(orc_model *M, int kind, int gid0, int gid1, int stride, int nthreads)
{
    long total = 0;
    if (stride < 1) stride = 1;
    if (nthreads <= 1) {
        M->threaded = 0;
        for (int id = gid0; id < gid1; id += stride)
            total += (kind == 3) ? sim_sca_hp_workitem(M, id) : ((kind == 1) ? sim_sca_cl_workitem(M, id) : sim_sca_pb_workitem(M, id, kind));
    } else {
        M->threaded = 1;
        const long n = ((long)gid1 - gid0 + stride - 1) / stride;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total) num_threads(nthreads)
        for (long k = 0; k < n; k++) {
            int id = (int)(gid0 + k * stride);
            total += (kind == 3) ? sim_sca_hp_workitem(M, id) : ((kind == 1) ? sim_sca_cl_workitem(M, id) : sim_sca_pb_workitem(M, id, kind));
        }
    }
    return total;
}

/* EqTemperature (kernel_ASOC_aux.c:745-790) for all levels; FACTOR and LENGTH are the -D literals */
EXPORT void orc_eqtemp(const orc_model *M, float adhoc, float kE, float Emin, int NE, float FACTOR, float LENGTH,
                       const float *TTT, const float *EABS, float *TNEW)
{
    const float scale  = (6.62607e-27f * FACTOR) / LENGTH;
    const float oplgkE = 1.0f / M_LOG10(kE);
    const float beta = 1.0f;
    for (int level = 0; level < M->LEVELS; level++) {
        for (int i = 0; i < M->LCELLS[level]; i++) {
            const int   ind = M->OFF[level] + i;
            float Ein = (scale / adhoc) * EABS[ind] * M_POWN(8.0f, level) / M->DENS[ind];
            if (M->CR_HEATING_RATE > 0.0f) Ein += 1.0e-27f * FACTOR * M->CR_HEATING_RATE;       /* kernel_ASOC_aux.c:769-773 */
            const float a   = M_FLOOR(oplgkE * M_LOG10((Ein / beta) / Emin));
            int   iE = (a != a) ? 0 : ((a < 0.0f) ? 0 : ((a > (float)(NE - 2)) ? NE - 2 : (int)a));
            const float wi  = (Emin * M_POWN(kE, iE + 1) - (Ein / beta)) / (Emin * M_POWN(kE, iE) * (kE - 1.0f));
            TNEW[ind] = (M->DENS[ind] > 1.0e-7f) ? clampf(wi * TTT[iE] + (1.0f - wi) * TTT[iE + 1], 3.0f, 1600.0f) : (10.0f);
        }
    }
}

/* PSTau (kernel_ASOC_map.c:1545-1584): column density and optical depth from every point source towards the observer,
 * with the map kernels' GetStep.  PSPOS: 4 floats per source (cl float3). */
EXPORT void orc_pstau(const orc_model *M, int no, const float *PSPOS, const float *DIRv, float LENGTH, float *pscolden, float *pstau)
{
    const f3 DIR = { DIRv[0], DIRv[1], DIRv[2] };
    for (int id = 0; id < no; id++) {
        float DTAU, TAU = 0.0f, colden = 0.0f, sx;
        f3    POS = { PSPOS[4 * id], PSPOS[4 * id + 1], PSPOS[4 * id + 2] };
        int   ind, level = 0, oind;
        IndexG(M, &POS, &level, &ind);
        while (ind >= 0) {
            oind = M->OFF[level] + ind;
            sx   = GetStepMap(M, &POS, &DIR, &level, &ind);
            if (M->WITH_ABU) DTAU = sx * M->DENS[oind] * (M->OPT[2 * (long)oind] + M->OPT[2 * (long)oind + 1]);
            else             DTAU = sx * M->DENS[oind] * (M->SCA + M->ABS);
            TAU += DTAU;
            colden += sx * M->DENS[oind];
        }
        pscolden[id] = colden * LENGTH;
        pstau[id]    = TAU;
    }
}

/* Emission2 (kernel_ASOC_aux.c:862-888): EMIT[(icell-c0)*nfreq+ifreq] */
EXPORT void orc_emission(int c0, int c1, int nfreq, float FACTOR, float LENGTH, const float *FREQ, const float *FABS,
                         const float *T, float *EMIT)
{
    for (int icell = c0; icell < c1; icell++) {
        const float t = T[icell];
        for (int ifreq = 0; ifreq < nfreq; ifreq++) {
            const float freq = FREQ[ifreq];
            EMIT[(long)(icell - c0) * nfreq + ifreq] =
                (2.79639459e-20f * FACTOR) * FABS[ifreq] * (freq * freq / (M_EXP(4.7995074e-11f * freq / t) - 1.0f)) / LENGTH;
        }
    }
}

/* math probes for tests/test_math.py */
EXPORT void orc_math_eval(int fn, const float *x, float *y, long n)
{
    for (long i = 0; i < n; i++) {
        switch (fn) {
        case 0: y[i] = M_EXP(x[i]); break;
        case 1: y[i] = M_LOG(x[i]); break;
        case 2: y[i] = M_SIN(x[i]); break;
        case 3: y[i] = M_COS(x[i]); break;
        case 4: y[i] = M_ACOS(x[i]); break;
        case 5: y[i] = M_SQRT(x[i]); break;
        case 6: y[i] = M_FMOD1(x[i]); break;
        case 8: y[i] = M_EXPM1(x[i]); break;
        case 9: y[i] = M_POW15(x[i]); break;
        case 10: y[i] = (float)M_LOGD((double)x[i]); break;
        default: y[i] = 0.0f;
        }
    }
}
