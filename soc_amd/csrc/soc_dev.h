// soc_dev.h -- structures shared by the C-ABI host code (soc_capi.hip) and the kernels.
#ifndef SOC_DEV_H
#define SOC_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SOC_MAXL 16            /* hierarchy levels supported (reference models use <= 8) */

// Model geometry.  The reference bakes these into the kernel with -D NX= ... -D CELLS=
// (ASOC.py:344-362); here they are run-time kernel arguments.
struct SocGrid {
    int NX, NY, NZ, LEVELS, CELLS, NXYZ;
    int OFF[SOC_MAXL];         /* first cell of each level (ASOC_aux.py:772)            */
    int LCELLS[SOC_MAXL];      /* cells per level                                        */
    const float *DENS;         /* [CELLS] density (>0) or child link (<=0)               */
    const int   *PAR;          /* [CELLS-NXYZ] parent cell index within its level        */
};

// Region of interest of nested runs (kernel_ASOC.c:44-51, 1250-1254; -D ROI_STEP, ROI_NSIDE, WITH_ROI_SAVE,
// WITH_ROI_LOAD): packets entering ROI are recorded per surface element and Healpix direction; SOURCE == 3 sends
// such a record in from the model surface.  Lives in device memory, SocSim points at it.
struct SocRoi {
    int   save, load;          /* WITH_ROI_SAVE, WITH_ROI_LOAD                                          */
    int   ROI[6];              /* x0, x1, y0, y1, z0, z1: root cells, inclusive                         */
    int   STEP, NSIDE;         /* surface elements per root-cell edge (save); Healpix NSIDE of the records */
    int   DIM[3], NELEM;       /* discretisation of the record to load; its number of surface elements  */
    float *SAVE;               /* [elements * 12 * NSIDE^2]                                             */
    const float *LOAD;         /* [NELEM * 12 * NSIDE^2] photons                                        */
};

// One launch of SimRAM_PB / SimRAM_CL (argument lists: kernel_ASOC.c:15-52, 1223-1256).
struct SocSim {
    int   SOURCE, BATCH, GLOBAL, PS_METHOD, NO_PS, BINS, USE_EMWEIGHT;
    int   MIRROR;              /* reflecting faces x,X,y,Y,z,Z = 1,2,4,8,16,32 (ASOC.py:319-321) */
    uint32_t gid0, gid_count;  /* this device runs logical work items [gid0, gid0+gid_count) */
    uint64_t seed_mul;         /* BASEID * A^base mod M for this SEED                     */
    const uint64_t *seed_tab;  /* 4 x 256 table of G^(b*256^k), see soc_rng.h             */
    float ABS, SCA, BG, TW;
    const float  *CSC;         /* [BINS] cumulative scattering function, current frequency */
    const float2 *OPT;         /* [CELLS] (abs, sca) per cell when WITH_ABU               */
    const float4 *PSPOS;       /* point-source positions (cl float3 = 16 bytes)           */
    const float  *PS;
    const int    *XPS_NSIDE, *XPS_SIDE;
    const float  *XPS_AREA;
    const float  *EMIT, *EMWEI;
    const int    *EMINDEX;     /* USE_EMWEIGHT == 2: cells to emit from, -1 terminated        */
    float        *XAB;         /* WITH_ALI: absorptions in the emitting cell (else NULL)      */
    int    HPBG_WEIGHTED;      /* SimRAM_HP: pixel chosen by cumulative probability       */
    const float  *HPBG, *HPBGP; /* [49152] sky (photons per package), cumulative probability */
    float *TABS, *INT;
    unsigned long long *stats; /* [0] tally events  [1] packets  [2] scatterings          */
    const SocRoi *ROI;         /* NULL without roisave / roiload                          */
    int    ROILOAD;            /* SOURCE == 3: surface elements of the loaded record (host copy of ROI->NELEM; 0: none loaded) */
    int    ROISAVE;            /* the record of packets entering ROI is kept (host copy of ROI->save: the sweep's queues depend on it) */
    int    STEP_WEIGHT;        /* -D STEP_WEIGHT: 0 none, 1 | 2 weighted free paths (kernel_ASOC.c:516-535) */
    float  SW_A, SW_B;
    int    NDUST;              /* > 1: -D WITH_MSF, CSC holds [NDUST][BINS] (kernel_ASOC.c:777-795)          */
    const float  *MSF_SCA;     /* [NDUST] scattering cross sections of the species, current frequency       */
    const float  *ABU;         /* [CELLS][NDUST] abundances                                                  */
    float  *INTV;              /* -D SAVE_INTENSITY=2: INTX | INTY | INTZ (CELLS each), else NULL; direct kernels only */
    int     CELLS;             /* stride of INTV                                                             */
    /* a launch of the scattered-light kernels run as a sweep of rays (soc_brick.hip: soc_sca_events): which kernel, its
     * discrete scattering function and the image it adds to (launches of one sweep may belong to several frequencies) */
    int          SCAKIND;      /* SOC_SCA_* + 1; 0: an absorption launch                                     */
    const float *DSC;          /* [BINS]                                                                     */
    float       *OUT;          /* [NDIR*NPIX_Y*NPIX_X]                                                       */
};

#define SOC_SOURCE_HP 4        /* brick sweep only: the launch is a SimRAM_HP one (Healpix sky instead of BG) */
#define SOC_SOURCE_CL 5        /* brick sweep only: a SimRAM_CL launch (cell emission)                        */

// Several launches of SimRAM_PB executed in one brick sweep (soc_brick.hip): launch l owns the
// sweep's work items [first[l], first[l+1]); geometry and tallies are shared.  Lives in device memory
// (16 launches exceed the 4 KB of kernel arguments).  128: the two source blocks of a 50-frequency run (ASOC.py:1028-1545) fit one sweep.
#define SOC_MAXLAUNCH 128
struct SocSimPack {
    int      n;
    // launches that tally into one INT array form a group (the source blocks of one frequency): with several groups in a sweep the
    // brick queues are per group -- a workgroup's LDS tallies then belong to one INT array -- queue = group * NB + brick
    int      grp[SOC_MAXLAUNCH];         // group of every launch
    int      gfirst[SOC_MAXLAUNCH];      // a launch of every group (for its INT / INTV pointers)
    uint32_t first[SOC_MAXLAUNCH + 1];
    SocSim   S[SOC_MAXLAUNCH];
};

// feature switches that the reference selects with #if; compiled ahead of time here
struct SocVariant {
    int octree;                /* LEVELS > 1                                              */
    int dbl;                   /* Index() in double: NX > DIMLIM (kernel_ASOC_aux.c:25-37) */
    int abu;                   /* WITH_ABU                                                */
    int wint;                  /* INT tally: SAVE_INTENSITY in (1,2) or NOABSORBED==0     */
};

hipError_t soc_launch_sim_pb(const SocGrid &G, const SocSim &S, const SocVariant &V, hipStream_t st);
hipError_t soc_launch_sim_cl(const SocGrid &G, const SocSim &S, const SocVariant &V, hipStream_t st);
hipError_t soc_launch_sim_hp(const SocGrid &G, const SocSim &S, const SocVariant &V, hipStream_t st);
hipError_t soc_launch_parents(const SocGrid &G, int *PAR, hipStream_t st);
hipError_t soc_launch_seed_probe(uint64_t seed_mul, const uint64_t *tab, uint32_t gid0, uint32_t n,
                                 int ndraw, uint32_t *out_state, uint32_t *out_draws, hipStream_t st);
hipError_t soc_launch_math_probe(int fn, const float *x, float *y, long n, hipStream_t st);
hipError_t soc_launch_trace(const SocGrid &G, const SocVariant &V, const float *pos, const float *dir,
                            int maxsteps, int *levels, int *inds, float *dss, float *endpos, int *nsteps,
                            hipStream_t st);


// scattered-light images (soc_sca.hip): observers and image of one launch of the
// kernel_ASOC_sca.c kernels (argument lists :471-501, :1098-1122, :1462-1489)
enum { SOC_SCA_PB = 0, SOC_SCA_CL = 1, SOC_SCA_PS = 2, SOC_SCA_HP = 3 };
struct SocSca {
    int   kind, NDIR, NPIX_X, NPIX_Y, FFS;   /* NDIR < 0: Healpix map of NSIDE = -NDIR seen from ODIRS[0] (a position) */
    float MAP_DX, CX, CY, CZ;
    const float4 *ODIRS, *ORA, *ODE;   /* [NDIR] cl float3 = 16 bytes                      */
    const float  *DSC;                 /* [BINS] discrete scattering function              */
    float *OUT;                        /* [NDIR*NPIX_Y*NPIX_X]                             */
};
hipError_t soc_launch_sca(const SocGrid &G, const SocSim &S, const SocSca &V, const SocVariant &X, hipStream_t st);

// equilibrium temperature and thermal emission (soc_emit.hip)
hipError_t soc_launch_eqtemp(const SocGrid &G, float adhoc, float kE, float Emin, int NE, float FACTOR, float LENGTH, float cr_rate,
                             const float *TTT, const float *EABS, float *TNEW, hipStream_t st);
hipError_t soc_launch_emission(int c0, int c1, int nfreq, float FACTOR, float LENGTH, const float *FREQ, const float *FABS,
                               const float *T, float *EMIT, hipStream_t st);

// solver-file preprocessing (soc_a2e_pre.hip): integration weights and cooling rates of one grain size
hipError_t soc_launch_a2e_pre(int NFREQ, int NE, float FACTOR, const float *FREQ, const float *Ef, const float *SKABS, const float *E, const float *T,
                              int *L1, int *L2, float *IW, float *wrk, int *noIw, float *Tdown, hipStream_t st);

// OPT from abundances on the device (soc_emit.hip)
hipError_t soc_launch_opt(int cells, int ndust, int single, const float *ABU, const float *AF, float2 *OPT, hipStream_t st);
hipError_t soc_launch_opt_half(int cells, float2 *OPT, hipStream_t st);

// map making (soc_map.hip): one launch of Mapping / HealpixMapping (kernel_ASOC_map.c:496-516, 890-910)
struct SocMapArgs {
    int   mode;                    // 0 Mapping, 1 HealpixMapping (NSIDE = NPIX_X)
    int   NPIX_X, NPIX_Y, SAVE_COLDEN;
    int   ROI_MAP, ROI[6];         // -D ROI_MAP: only the emission of cells inside ROI = [x0,x1,y0,y1,z0,z1] (root cells, inclusive)
    int   LEVEL_THRESHOLD;         // Mapping: no emission from levels below it (-D LEVEL_THRESHOLD, kernel_ASOC_map.c:825-834)
    int   MAPINT;                  // Mapping: -D MAP_INTERPOLATION 0 | 1 | 2 (kernel_ASOC_map.c:656-810)
    float MAP_DX, ABS, SCA, LENGTH;
    float DIR[3], RA[3], DE[3], CENTRE[3], INTOBS[3];
    const float  *EMIT;
    const float2 *OPT;
    float *MAP, *SAVETAU;
};
hipError_t soc_launch_map(const SocGrid &G, const SocMapArgs &A, bool abu, hipStream_t st);
hipError_t soc_launch_pstau(const SocGrid &G, int no, const float4 *PSPOS, const float *DIR, float ABS, float SCA, const float2 *OPT, float LENGTH,
                            float *pscolden, float *pstau, hipStream_t st);

// stochastic-heating solver (soc_a2e.hip)
struct SocA2EArgs {
    int NE, NFREQ, npair, batch;
    const float *Iw;                 // integration weights, in (l,u,i) loop order
    const int   *pair_first;         // [npair] L1[l*NE+u]
    const int   *pair_last;          // [npair] L2[l*NE+u]
    const int   *pair_iw;            // [npair] offset of the pair's first weight in Iw
    const int   *pair_dst;           // [npair] (u*u-u)/2 + l
    const float *Tdown;              // [NE]
    const float *EA;                 // [NE*NFREQ]: transposed by soc_a2e_set_size (bin-major)
    const int   *Ibeg;               // [NFREQ]
    const float *AF;                 // [NFREQ]
    const float *AABS;               // [batch*NFREQ]
    float       *AEMIT;              // [batch*NFREQ]
    int          accumulate;         // 1: AEMIT += the emission of this size (the sum over the sizes stays on the device: soc_a2e_resident_*)
};

struct SocEqTArgs {
    int batch, icell, CELLS, NFREQ, NIP;
    float FACTOR, kE, oplgkE, Emin;
    const float *FREQ, *KABS, *TTT, *ABS;
    float *T, *EMIT;
};

hipError_t soc_launch_a2e_dosolve(const SocA2EArgs &A, hipStream_t st);
hipError_t soc_launch_a2e_eqtemp(const SocEqTArgs &A, hipStream_t st);
hipError_t soc_launch_eqsolver(const SocEqTArgs &A, hipStream_t st);

// Shape of the brick sweep; 0 = the built-in choice for the grid (measured).  Set per context with
// soc_set_tuning (include/soc_hip.h); the parity tests use small CAP / HS values to exercise brick boundaries.
struct SocBrickTune {
    int T, P, KCAP, FTH, CTH, CAP, TAIL, POP, HS;
    int global_tree;           // hierarchies: the walk that reads the hierarchy from global memory, also where brick-local ones apply
    int park;                  // brick-local hierarchies: brick queues shorter than this (and than the mean queue) wait for more packets (0 = built-in 4096, 1 = never)
    int slow_every;            // brick-local hierarchies, test knob: every n-th step below the root grid goes through the slow-step queue
    int nolean;                // keep the general SimRAM_PB kernel for background-only sweeps
    int oversub;               // experiment: background work items beyond 8*AREA are not clipped
    int verbose;
};

// brick-sweep execution (soc_brick.hip): LDS-resident tallies, packets sorted by brick
// population: packets in flight (0 = all work items at once, -1 = chosen from the number of bricks): the other work
// items are admitted as earlier ones finish
hipError_t soc_brick_run_pb(int device, const SocGrid &G, const SocSim *S, int nlaunch, const SocVariant &V, int LB,
                            int population, const SocBrickTune &tune, hipStream_t st, int *passes_out, int *form_out,
                            const struct SocSca *sca = nullptr);      // sca: the launch is one of the scattered-light kernels (rays, soc_sca_events)
void soc_brick_release(int device);
void soc_brick_invalidate(int device);      // the grid changed: bricks of a hierarchy are rebuilt at the next sweep

#endif
