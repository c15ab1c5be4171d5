/*
 * soc_hip.h -- C ABI of libsoc_hip.so, the MI355X (gfx950) engine for SOC's photon-packet path.
 *
 * The reference has no FFI layer: its host scripts drive OpenCL kernels through pyopencl
 * (cl.Buffer / enqueue_copy / kernel(queue,[GLOBAL],[LOCAL],args...)).  Each entry point
 * below replaces one group of those calls; file:line citations are into /root/reference.
 * Conventions:
 *   - every function returns 0 on success, a negative code on error; the message is
 *     available from soc_last_error().  Nothing throws or exits.
 *   - host pointers are only read/written during the call; the library owns all device
 *     memory behind the handle (NULL is allowed for unused optional arrays).
 *   - scalars have the types pyopencl passes (set_scalar_arg_dtypes, ASOC.py:846-858):
 *     int32 / float32.  SEED, BG, TW are float32 at the boundary.
 *   - a handle is bound to one GPU and is not thread-safe; launches are asynchronous on the
 *     handle's stream; soc_read_tally() and soc_sync() synchronise.
 *   - the scratch of the brick sweep (packet queues, the brick tables of the current grid) is kept per GPU, shared by
 *     the handles of that GPU and freed when the last of them is destroyed: calls on DIFFERENT handles of one GPU must
 *     not overlap in time either (one process drives one GPU from one thread -- the layout of soc_amd/dist.py).
 */
#ifndef SOC_HIP_H
#define SOC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct soc_ctx soc_ctx;

#define SOC_OK            0
#define SOC_ERR_ARG      -1   /* invalid argument / inconsistent model        */
#define SOC_ERR_STATE    -2   /* call order: grid / tables not set yet         */
#define SOC_ERR_HIP      -3   /* HIP runtime error (message has the details)   */

/* tallies selectable in soc_zero / soc_read_tally / soc_tally_ptr */
#define SOC_TALLY_TABS    0   /* absorbed energy integrated over frequency (TABS, kernel arg 15) */
#define SOC_TALLY_INT     1   /* per-frequency absorptions (INT, kernel arg 20)                  */
#define SOC_TALLY_XAB     2   /* WITH_ALI: absorptions inside the emitting cell (XAB, kernel arg 19) */
#define SOC_TALLY_INTX    3   /* SAVE_INTENSITY==2: sum of delta*DIR.x per cell (INTX, kernel arg 21); soc_zero(1) clears  */
#define SOC_TALLY_INTY    4   /*   INT and these three, as ZeroAMC tag 1 does (kernel_ASOC_aux.c:676-681)                   */
#define SOC_TALLY_INTZ    5

/* replaces ASOC_aux.py:1188-1256 opencl_init(): create a context on GPU `device` */
int  soc_create(int device, soc_ctx **out);
void soc_destroy(soc_ctx *ctx);
const char *soc_last_error(const soc_ctx *ctx);   /* ctx may be NULL: last creation error */
const char *soc_version(void);

/* run on an externally owned HIP stream (e.g. the stream of the caller's framework); NULL = own stream.
 * NOTE: a framework's "default stream" is the null handle too -- pass a stream the framework created
 * (torch.cuda.Stream().cuda_stream) and make it the current one, or the collectives of the framework
 * are not ordered with the kernels (soc_amd/dist.py does that). */
int soc_set_stream(soc_ctx *ctx, void *hip_stream);

/* replaces the -D NX,NY,NZ,LEVELS,CELLS macros (ASOC.py:344-362), the LCELLS/OFF/DENS
 * uploads (ASOC.py:524-539) and the Parents kernel launch (ASOC.py:585,
 * kernel_ASOC_aux.c:688-718).  DENS is the concatenated hierarchy as read_cloud() returns
 * it (ASOC_aux.py:716-803): value > 0 leaf density, value <= 0 link to 8 children.
 * The hierarchy is validated (links in range, octets aligned) before anything is uploaded. */
int soc_set_grid(soc_ctx *ctx, int NX, int NY, int NZ, int LEVELS, const int32_t *LCELLS, const float *DENS);

/* feature switches the reference compiles in with -D (ASOC.py:344-362):
 *   with_int     : 1 = SAVE_INTENSITY==1 or NOABSORBED==0 -> INT tally is updated; 2 = SAVE_INTENSITY==2 -> INT and the
 *                  vector sums INTX, INTY, INTZ (kernel_ASOC.c:604-612, :724-732; needs soc_set_grid first; such launches
 *                  run on the direct kernels)
 *   ps_method    : PS_METHOD 0,1,2,4,5 (3 does not compile in the reference)
 *   use_emweight : USE_EMWEIGHT 0, 1, or 2 = cells listed by soc_set_emindex (SimRAM_CL)     */
int soc_set_features(soc_ctx *ctx, int with_int, int ps_method, int use_emweight);

/* replaces -D MIRROR=%d (ASOC.py:319-321,352; ASOCS.py:119-122): reflecting model faces, bits
 * x,X,y,Y,z,Z = 1,2,4,8,16,32 (lower/upper face per axis); 0 = none (default) */
int soc_set_mirror(soc_ctx *ctx, int mask);

/* how launches are executed (no counterpart in the reference; results are the same packets):
 *   mode 0  direct: one lane per work item, one global float atomic per tally event
 *   mode 1  brick sweep (SimRAM_PB): packets sorted by brick, tallies accumulated in LDS and flushed per
 *           brick; a brick is 2^brick_log2 root cells per edge on Cartesian grids, a set of <= 8192
 *           neighbouring leaves on hierarchies (built at the first sweep after soc_set_grid)
 *   mode -1 automatic (default): brick sweep where it pays -- Cartesian grids from 65536 work items on,
 *           hierarchies for launches deferred by soc_batch_begin (two or more per sweep) -- direct
 *           otherwise (reflecting faces, region-of-interest records, SOURCE 3)                     */
int soc_set_exec(soc_ctx *ctx, int mode, int brick_log2);

/* Shape of the brick sweep (no counterpart in the reference; results are the same packets whatever the values).
 * value 0 = the built-in choice for the grid.  Names: "threads" (workgroup size of the walk, 64..512; ..1024 on brick-local
 * hierarchies), "chunk" (packets per workgroup, <= 4096; <= 32768 on brick-local hierarchies), "steps_per_visit" (cell steps
 * before a packet goes back to its queue), "swap_lanes" (lanes of a wave that must wait before the packet swap runs),
 * "climb_lanes" (the same for the deferred Index() of the global-tree form), "brick_cells" (cells per brick on hierarchies,
 * <= 36864), "tail_lanes", "park_below" (hierarchies: brick queues shorter than this and than the mean wait a pass; 1 = never), "population" (packets in flight), "hash_slots" (per-workgroup
 * arrival table, power of two), "general_kernel" (1: no background-only kernel), "global_tree" (1: hierarchies are walked in
 * global memory also where the brick-local form applies), "slow_every" (test knob of that form),
 * "oversubscribe", "verbose". */
int soc_set_tuning(soc_ctx *ctx, const char *name, int value);

/* replaces the per-frequency uploads of ABS, SCA (ASOC.py:1171-1175); ndust must be 1
 * (the host sums the species, ASOC.py:1166-1170) */
int soc_set_optical(soc_ctx *ctx, const float *ABS, const float *SCA, int ndust);

/* replaces the OPT upload for abundance runs (WITH_ABU, ASOC.py:1146-1160):
 * OPT[CELLS][2] = (abs, sca) per cell; NULL switches back to scalar ABS/SCA */
int soc_set_opt(soc_ctx *ctx, const float *OPT);

/* The same OPT without the 8*CELLS-byte upload per frequency (ASOC.py:1146-1160, "0.43 s / 2.5 s" :1177): the
 * abundances go to the device once -- ABU[CELLS][NDUST], or with single != 0 ABU[CELLS] for two species with
 * abundances ABU and 1-ABU (USER.SINGLE_ABU, :1148-1153); ABU = NULL forgets them -- and for every frequency
 * soc_set_optical_abu(AFABS[NDUST], AFSCA[NDUST]) computes OPT = sum ABU * AF on the device, in the order and
 * precision of the numpy expressions (bit-identical to the host's OPT).  soc_read_opt copies OPT back. */
int soc_set_abundances(soc_ctx *ctx, int NDUST, int single, const float *ABU);
int soc_set_optical_abu(soc_ctx *ctx, const float *AFABS, const float *AFSCA, int ndust);
int soc_read_opt(soc_ctx *ctx, float *OPT);

/* -D OPT_IS_HALF (ini key `optishalf`; kernel_ASOC_aux.c:12-18, ASOC.py:1158-1159): the reference stores OPT as fp16.
 * on != 0: every later soc_set_opt / soc_set_optical_abu rounds OPT to fp16 (nearest even) -- the kernels then compute
 * with exactly the values vload_half would give them. */
int soc_set_opt_half(soc_ctx *ctx, int on);

/* replaces the DSC/CSC row uploads (ASOC.py:1234-1243, ASOCS.py:625-626); DSC may be NULL
 * (unused by the absorption kernels, required by soc_sca_sim_ps/pb); BINS = USER.DSC_BINS */
int soc_set_scatter_table(soc_ctx *ctx, const float *DSC, const float *CSC, int BINS);

/* -D WITH_MSF (ASOC.py:132-138, :1239-1243; kernel_ASOC.c:777-795, :1654-1668): one scattering function per dust species,
 * DSC[NDUST][BINS] and CSC[NDUST][BINS]; NDUST == 1 is soc_set_scatter_table.  With NDUST > 1 a launch needs the
 * abundances of the same NDUST species (soc_set_abundances, not the one-abundance form) and this frequency's
 * soc_set_optical_abu -- whose AFSCA are the per-species SCA the kernels draw the scatterer with -- and is not deferred. */
int soc_set_scatter_tables(soc_ctx *ctx, int NDUST, const float *DSC, const float *CSC, int BINS);

/* -D STEP_WEIGHT / SW_A / SW_B (ASOC.py:348,357; kernel_ASOC.c:516-535, :752-763, :941-955, :1444-1462, :1625-1640):
 * free paths from p(t) = A*exp(-A*t) (mode 1) or B*A*exp(-A*t) + (1-B)*2A*exp(-2A*t) (mode 2) with the packet weight
 * corrected; mode <= 0 switches the weighting off.  The arguments are the VALUES OF THE -D MACROS; the reference's host
 * fills them from the ini key `stepweight a b c` as STEP_WEIGHT=int(c), SW_A=int(a), SW_B=b (ASOC.py:357). */
int soc_set_step_weight(soc_ctx *ctx, int mode, float SW_A, float SW_B);

/* replaces the EMIT / EMWEI uploads (ASOC.py:1276, 1291); arrays of CELLS floats */
int soc_set_emission(soc_ctx *ctx, const float *EMIT, const float *EMWEI);

/* replaces the EMINDEX upload of the USE_EMWEIGHT==2 loop (ASOC.py:1809-1840): EMINDEX[CELLS], the cells to
 * emit 100 packets from in the next soc_sim_cl launch, terminated by -1 */
int soc_set_emindex(soc_ctx *ctx, const int32_t *EMINDEX);

/* replaces -D WITH_ALI=1 (ASOC.py:344): what a cell absorbs of its own emission is tallied in XAB
 * (SOC_TALLY_XAB) instead of TABS (kernel_ASOC.c:1486-1491); soc_zero(ctx, 0) clears both */
int soc_set_ali(soc_ctx *ctx, int with_ali);

/* replaces ZeroAMC (kernel_ASOC_aux.c:657-683; ASOC.py:1115,1183): tag 0 clears TABS,
 * tag 1 clears INT */
int soc_zero(soc_ctx *ctx, int tag);

/* replaces the kernel_ram_pb launch (ASOC.py:1360-1372 -> SimRAM_PB, kernel_ASOC.c:15-52).
 * SOURCE 0 = point sources, 1 = isotropic background.  PSPOS holds 4 floats per source
 * (cl float3).  GLOBAL is the logical launch size; this call executes the logical work
 * items [gid_first, gid_first+gid_count) so that several GPUs can share one launch with
 * exactly the streams a single device would use (get_global_id -> logical id). */
int soc_sim_pb(soc_ctx *ctx, int SOURCE, int PACKETS, int BATCH, float SEED, float BG, float TW,
               const float *PSPOS, const float *PS, int NO_PS,
               const int32_t *XPS_NSIDE, const int32_t *XPS_SIDE, const float *XPS_AREA,
               int GLOBAL, int gid_first, int gid_count);

/* replaces the kernel_ram_cl launch (ASOC.py:1308-1316, 1847 -> SimRAM_CL,
 * kernel_ASOC.c:1223-1256); uses EMIT/EMWEI from soc_set_emission().  Executed by the direct kernel, or -- with at
 * least 262144 work items that own a cell (GLOBAL ~ CELLS), USE_EMWEIGHT 0/1, no ALI, no roisave -- by the brick sweep;
 * deferred inside soc_batch_begin/end with its own copy of EMIT and EMWEI */
int soc_sim_cl(soc_ctx *ctx, int SOURCE, int PACKETS, int BATCH, float SEED, float TW,
               int GLOBAL, int gid_first, int gid_count);

/* Deferred execution of consecutive soc_sim_pb launches (no counterpart in the reference, which
 * runs one kernel per frequency and waits for it, ASOC.py:1360-1461).  Between soc_batch_begin and
 * soc_batch_end a launch that qualifies for the brick sweep and runs without the per-frequency INT
 * tally is recorded with a snapshot of its inputs (ABS, SCA or the per-cell OPT, scattering table,
 * BG, TW, seed, sources) and executed together with up to max_launches-1 others (0 = default = at most: 16;
 * on Cartesian grids 2.7e6 packets are in flight at a time and the later launches' work items are admitted as earlier
 * ones finish): the same packets, the same per-launch RNG streams, the same tallies -- more packets in
 * flight per pass.  Any other call that reads or changes engine state executes what is pending
 * first; launches that do not qualify run immediately as always. */
int soc_batch_begin(soc_ctx *ctx, int max_launches);
int soc_batch_end(soc_ctx *ctx);
/* The same for runs that keep the per-frequency INT tally (the absorbed file, ASOC.py:1482-1498): every deferred
 * launch gets its own, zeroed INT tally instead of the shared one (TABS stays shared).  At most max_launches (<= 16)
 * launches of one kind per batch; after soc_batch_end, soc_batch_read_int(k) copies the INT tally of the k-th launch of
 * the batch (n = CELLS).  Replaces K x [kernel launch + enqueue_copy(INT)] by K launches + K copies. */
int soc_batch_begin_int(soc_ctx *ctx, int max_launches);
/* The launches of ONE frequency that keep the INT tally (the source blocks of ASOC.py:1028-1545 at one IFREQ: point sources,
 * background, diffuse emission): deferred until soc_batch_end like soc_batch_begin's, all tallying into the handle's INT buffer
 * (soc_zero(ctx, 1) before, soc_read_tally(ctx, 1) after), so that they share brick sweeps -- and, where the walk keeps tallies
 * in LDS per workgroup, the same brick queues.  Replaces the per-launch enqueue + finish() + enqueue_copy(INT) of ASOC.py:1360-1372,
 * :1461, :1482-1498 for the launches of one frequency. */
int soc_batch_begin_shared_int(soc_ctx *ctx, int max_launches);
/* Several frequencies in one sweep, each with its own INT tally: as soc_batch_begin_int, but the launches between two
 * soc_batch_next_int calls -- the source blocks of ONE frequency -- share a tally (soc_batch_read_int(k) reads the k-th group's).
 * On brick-local hierarchies the brick queues are per group, so a workgroup's LDS tallies belong to one frequency; point-source,
 * background and cell-emission launches mix freely.  At most max_groups groups (0: 128) until soc_batch_end. */
int soc_batch_begin_int_groups(soc_ctx *ctx, int max_groups);
int soc_batch_next_int(soc_ctx *ctx);
int soc_batch_read_int(soc_ctx *ctx, int k, float *out, long n);

/* ---- region of interest of nested runs (ini keys roi, roisave, roiload, roipac, roinside) ---- */

/* replaces -D WITH_ROI_SAVE -D ROI_STEP -D ROI_NSIDE and ROI_buf / ROI_SAVE_buf (ASOC.py:346,927-944): from now on
 * soc_sim_pb and soc_sim_cl add every packet that steps into ROI = [x0,x1,y0,y1,z0,z1] (root cells, inclusive) to a
 * record [surface element, Healpix pixel of its direction] (kernel_ASOC.c:547-562,615-642, 1436-1535; InRoi
 * kernel_ASOC_aux.c:1031-1048), ROI_STEP elements per root-cell edge, NSIDE = ROI_NSIDE, RING order.
 * The record is zeroed here; ROI = NULL turns recording off.  Direct kernel only (the brick sweep stands aside). */
int soc_set_roi_save(soc_ctx *ctx, const int32_t *ROI, int ROI_STEP, int ROI_NSIDE);
/* replaces enqueue_copy(ROI_SAVE_buf, zeros) per frequency (ASOC.py:1301-1302) */
int soc_roi_zero(soc_ctx *ctx);
/* replaces enqueue_copy(tmp, ROI_SAVE_buf) (ASOC.py:1468-1471); n = (nx*ny + ny*nz + nz*nx) * 12 * ROI_NSIDE^2 with
 * n? = (ROI[2?+1] - ROI[2?] + 1) * ROI_STEP */
int soc_roi_read(soc_ctx *ctx, float *out, long n);
/* replaces -D WITH_ROI_LOAD and ROI_DIM_buf / ROI_LOAD_buf (ASOC.py:909-925,1419-1421): the record of one frequency,
 * LOAD[nelem, 12*ROI_NSIDE^2] photons (already scaled by the host) with nelem = DIM[0]*DIM[1] + DIM[1]*DIM[2] +
 * DIM[2]*DIM[0] surface elements, sent by soc_sim_pb(SOURCE = 3, PACKETS = nelem, BATCH = k * 12*ROI_NSIDE^2,
 * GLOBAL >= 100 * nelem) (kernel_ASOC.c:97-105,141-179,469-501).  LOAD = NULL turns it off. */
int soc_set_roi_load(soc_ctx *ctx, const int32_t *DIM, int ROI_NSIDE, const float *LOAD);

/* replaces the HPBG_buf / HPBGP_buf uploads (ASOC.py:1196-1214): the Healpix sky of the current
 * frequency in photons per package, 49152 floats (NSIDE 64, RING order); HPBGP = cumulative
 * pixel probability for `hpbg ... weighted` runs (-D HPBG_WEIGHTED=1) or NULL */
int soc_set_hpbg(soc_ctx *ctx, const float *BG, const float *HPBGP);

/* replaces the kernel_ram_hp launch (ASOC.py:1349-1354 -> SimRAM_HP, kernel_ASOC.c:826-850); executed like
 * soc_sim_pb (direct kernel or brick sweep, deferred inside soc_batch_begin/end with its own copy of the sky) */
int soc_sim_hp(soc_ctx *ctx, int PACKETS, int BATCH, float SEED, float TW, int GLOBAL, int gid_first, int gid_count);

/* replaces queue.finish() (ASOC.py:1461) */
int soc_sync(soc_ctx *ctx);

/* replaces cl.enqueue_copy(host, TABS_buf / INT_buf) (ASOC.py:1482, 1533); n = CELLS */
int soc_read_tally(soc_ctx *ctx, int which, float *out, int64_t n);
/* the inverse (e.g. `cload` restart files, ASOC.py:1013-1018) */
int soc_write_tally(soc_ctx *ctx, int which, const float *in, int64_t n);

/* device address of a tally (for an RCCL all-reduce by the caller), or bind caller-owned
 * device memory (n = CELLS floats, checked) as the tally so a framework tensor can be reduced in place;
 * device_ptr = NULL gives the tally back to memory of the library.  A grid with another cell count cannot
 * be set while a caller-owned tally is bound. */
void *soc_tally_ptr(soc_ctx *ctx, int which);
int   soc_bind_tally(soc_ctx *ctx, int which, void *device_ptr, int64_t n);

/* PAR table computed by soc_set_grid (CELLS - NX*NY*NZ entries), for verification */
int soc_read_par(soc_ctx *ctx, int32_t *out, int64_t n);

/* counters accumulated by the kernels since the last reset:
 * out[0] tally events, out[1] packets created, out[2] scattering events */
int soc_stats(soc_ctx *ctx, uint64_t out[3], int reset);
/* cell steps of all rays (look-ahead, packet, peel-off) of the scattered-light launches that ran as sweeps of rays on brick-local
 * hierarchies, as of the last soc_stats call (what the read-only roofline of SURVEY 8(d) counts: 4 B per step); -1: no handle */
int64_t soc_sca_ray_steps(soc_ctx *ctx);

/* number of brick-sweep passes of the last launch (0 if it ran in direct mode) */
int soc_last_passes(soc_ctx *ctx);
/* how the last launch was executed: 0 direct kernel, 1 brick sweep on a Cartesian grid, 2 on a hierarchy read from global
 * memory, 3 on brick-local hierarchies (soc_ltree.h: hierarchies whose Index() the reference evaluates in double) */
int soc_last_form(soc_ctx *ctx);

/* HIP-event timing on the handle's stream: bracket launches, then read elapsed ms */
int soc_timer_start(soc_ctx *ctx);
int soc_timer_stop(soc_ctx *ctx, float *elapsed_ms);

/* ---- scattered-light images: ASOCS.py / kernel_ASOC_sca.c ---- */

/* replaces the ODIR/RA/DE buffers and the NDIR, NPIX, MAP_DX, MAPCENTRE scalars of every
 * kernel_ASOC_sca.c launch (ASOCS.py:247-262, 655-708) and the -D FFS= build option
 * (ASOCS.py:139).  ODIR, RA, DE hold 4 floats per direction (cl float3), as returned by
 * set_observer_directions (ASOC_aux.py:1129-1183); CENTRE = 3 floats.  Allocates the image
 * OUT[NDIR][NPIX_Y][NPIX_X] on the device (ASOCS.py:246).  Healpix output (NDIR<0) is not
 * supported. */
int soc_sca_set_view(soc_ctx *ctx, int NDIR, const float *ODIR, const float *RA, const float *DE,
                     int NPIX_X, int NPIX_Y, float MAP_DX, const float *CENTRE, int FFS);

/* the other form of the view: one Healpix map (RING, NSIDE = USER.OUT_NSIDE) seen by an observer at a
 * position in root-grid units (`perspective x y z`; ASOCS.py:44-48: NDIR = -NSIDE, ODIR[0] = the
 * position).  The image has 12*NSIDE^2 pixels; each contribution carries 1/d^2. */
int soc_sca_set_healpix(soc_ctx *ctx, int NSIDE, const float *OBSERVER, int FFS);

/* replaces zero_out (kernel_ASOC_sca.c:14-35; ASOCS.py:515, 781) */
int soc_sca_zero(soc_ctx *ctx);

/* replaces the kernel_PS launch (ASOCS.py:665-671 -> SimRAM_PS, kernel_ASOC_sca.c:1462-1489):
 * point sources.  XPS_* are the int32/float32 arrays of AnalyseExternalPointSources exactly as
 * ASOCS.py uploads them; the reference kernel declares the two integer arrays as float and
 * that reading is reproduced.  Work-item range as in soc_sim_pb. */
int soc_sca_sim_ps(soc_ctx *ctx, int PACKETS, int BATCH, float SEED, float BG, const float *PSPOS, const float *PS,
                   int NO_PS, const int32_t *XPS_NSIDE, const int32_t *XPS_SIDE, const float *XPS_AREA,
                   int GLOBAL, int gid_first, int gid_count);

/* replaces the kernel_PB launch (ASOCS.py:681-688 -> SimRAM_PB, kernel_ASOC_sca.c:471-501):
 * SOURCE 1 = isotropic background (what ASOCS.py uses it for), 0 = point sources */
int soc_sca_sim_pb(soc_ctx *ctx, int SOURCE, int PACKETS, int BATCH, float SEED, float BG, const float *PSPOS,
                   const float *PS, int NO_PS, const int32_t *XPS_NSIDE, const int32_t *XPS_SIDE,
                   const float *XPS_AREA, int GLOBAL, int gid_first, int gid_count);

/* replaces the kernel_CL launches (ASOCS.py:692-698, 862-868 -> SimRAM_CL,
 * kernel_ASOC_sca.c:1098-1122); uses EMIT/EMWEI from soc_set_emission() */
int soc_sca_sim_cl(soc_ctx *ctx, int SOURCE, int PACKETS, int BATCH, float SEED, int GLOBAL, int gid_first, int gid_count);

/* replaces the kernel_HP launch (ASOCS.py:673-679 -> SimRAM_HP, kernel_ASOC_sca.c:40-63): background from
 * the Healpix sky given to soc_set_hpbg (49152 pixels, photons per package) */
int soc_sca_sim_hp(soc_ctx *ctx, int PACKETS, int BATCH, float SEED, int GLOBAL, int gid_first, int gid_count);

/* replaces cl.enqueue_copy(OUT, OUT_buf) (ASOCS.py:715, 874); n = NDIR*NPIX_Y*NPIX_X, or 12*NSIDE^2 */
int soc_sca_read_out(soc_ctx *ctx, float *out, int64_t n);

/* device address of the image, or bind caller-owned device memory as the image (for an RCCL
 * all-reduce over the GPUs that shared a launch); NULL gives the image back to memory of the library */
void *soc_sca_out_ptr(soc_ctx *ctx);

/* Scattered-light launches in batches.  The reference runs one kernel after the other per frequency and source (ASOCS.py:655-708)
 * and reads the image after each frequency (:710-716).  Between soc_batch_begin and soc_batch_end (above) the soc_sca_sim_ps / _pb /
 * _cl launches that can run as rays on brick-local hierarchies (flat image, scalar opacities, one scattering function, a hierarchy
 * whose Index() the reference evaluates in double) are deferred, each with a snapshot of its inputs, and run together in one sweep
 * -- more rays per brick and pass than any single launch has; every other launch runs at once, as without the batch.
 * soc_sca_batch_images(n) gives the batch n zeroed images (n = 0: the one image of soc_sca_set_view again); the launches that
 * follow soc_sca_batch_select(k) add to image k -- one image per frequency -- and soc_sca_batch_read(k, ...) replaces the
 * enqueue_copy of that frequency's image after soc_batch_end. */
int soc_sca_batch_images(soc_ctx *ctx, int n);
int soc_sca_batch_select(soc_ctx *ctx, int k);
int soc_sca_batch_read(soc_ctx *ctx, int k, float *out, int64_t n);
int   soc_sca_bind_out(soc_ctx *ctx, void *device_ptr);

/* ---- equilibrium dust temperature and emission (ASOC.py `CLT`/`CLE` paths) ---- */

/* -D CR_HEATING=1 -D CR_HEATING_RATE=<rate> (ini key CR_HEATING; ASOC.py:352,362; kernel_ASOC_aux.c:769-773): soc_solve_temperature
 * adds 1e-27 * FACTOR * rate to the energy a cell absorbs.  rate = 0 switches it off. */
int soc_set_cr_heating(soc_ctx *ctx, float rate);

/* -D LEVEL_THRESHOLD=<level> (ini key threshold; kernel_ASOC_map.c:825-834): soc_map (flat maps; HealpixMapping has no such
 * test) leaves out the emission of cells on levels below `level`; they still absorb.  0 switches it off. */
int soc_set_map_threshold(soc_ctx *ctx, int level);

/* -D MAP_INTERPOLATION=<mode> (ini key mapint; ASOC.py:352,362; kernel_ASOC_map.c:656-810): soc_map (flat maps and the
 * longitude x latitude image; HealpixMapping has no such block) blends the density and the emission of every cell on the
 * ray with those of two neighbours across the ray; mode 2 also limits a step to 0.22 cells.  0 switches it off. */
int soc_set_map_interpolation(soc_ctx *ctx, int mode);

/* -D ROI_MAP=1 and the ROI argument of Mapping / HealpixMapping (ini key roimap with roi; ASOC.py:2941-2956,3126-3133;
 * kernel_ASOC_map.c:37-56,821-823,947-949): soc_map counts the emission of cells whose root cell lies inside
 * ROI = [x0,x1,y0,y1,z0,z1] (inclusive) only; extinction as usual.  NULL switches it off. */
int soc_set_map_roi(soc_ctx *ctx, const int32_t *ROI);

/* replaces the EqTemperature launches per level (ASOC.py:2024-2040 -> kernel_ASOC_aux.c:745-790):
 * EABS[CELLS] = integrated absorbed energy per cell (the array the reference calls EMIT at this
 * point: TABS of the dust-emission iteration + CTABS), TTT[NE] the host's E->T table with
 * E[i] = Emin*kE^i (ASOC.py:643-689); FACTOR and LENGTH = GL*PARSEC are the -D FACTOR / -D LENGTH
 * literals (ASOC.py:345,348: %.4e and %.5e).  Temperatures stay on the device for soc_emission and
 * are copied to TNEW[CELLS] unless NULL. */
int soc_solve_temperature(soc_ctx *ctx, float adhoc, float kE, float Emin, int NE, const float *TTT, float FACTOR,
                          float LENGTH, const float *EABS, float *TNEW);

/* temperatures from elsewhere (`loadtemp`, ASOC.py:744-760) */
int soc_set_temperature(soc_ctx *ctx, const float *T);

/* replaces the Emission / Emission2 launches (ASOC.py:2154-2197 -> kernel_ASOC_aux.c:795-808, 862-888):
 * EMITTED[CELLS][nfreq] = FACTOR x photons / Hz / cm3 of the modified black body at the device
 * temperatures, for the nfreq frequencies FREQ with absorption cross sections FABS */
int soc_emission(soc_ctx *ctx, int nfreq, const float *FREQ, const float *FABS, float FACTOR, float LENGTH, float *EMITTED);

/* ---- map making ---- */

/* replaces the kernel_map launch + copies (ASOC.py:3113-3128 -> Mapping / HealpixMapping, kernel_ASOC_map.c:496-516,
 * 890-910): line-of-sight integral of EMIT[CELLS] (x density, with extinction ABS+SCA or the per-cell OPT of soc_set_opt)
 * for one map.  healpix = 0: orthographic map of NPIX_X x NPIX_Y pixels of MAP_DX root cells towards DIR with image axes
 * RA (right), DE (up) through CENTRE -- or, with INTOBS given (INTOBS[0] > -1e10), the longitude x latitude image seen
 * from that position; healpix = 1: Healpix map of NSIDE = NPIX_X seen from INTOBS.  MAP gets the surface brightness
 * integral, SAVETAU the optical depth or (save_colden) column density x LENGTH.  -D MAP_INTERPOLATION, ROI_MAP and
 * LEVEL_THRESHOLD: soc_set_map_interpolation, soc_set_map_roi, soc_set_map_threshold.  Polarisation maps are not covered. */
int soc_map(soc_ctx *ctx, int healpix, int NPIX_X, int NPIX_Y, float MAP_DX, const float *EMIT, const float *DIR,
            const float *RA, const float *DE, const float *CENTRE, const float *INTOBS, float ABS, float SCA,
            int save_colden, float LENGTH, float *MAP, float *SAVETAU);

/* ---- stochastically heated grains: A2E.py / kernel_A2E.c ---- */

/* replaces the PSTau launch of ASOC.py:3576-3645 (ini key pssavetau; kernel_ASOC_map.c:1545-1584): for every point source
 * the column density (x LENGTH) and the optical depth (ABS + SCA, or the per-cell OPT of soc_set_opt) along the ray from the
 * source towards the observer direction DIR[3].  PSPOS: 4 floats per source (cl float3). */
int soc_ps_tau(soc_ctx *ctx, int NO_PS, const float *PSPOS, const float *DIR, float ABS, float SCA, float LENGTH,
               float *pscolden, float *pstau);

/* replaces the per-size uploads of A2E.py:338-371 (AF, Iw, L1, L2, Tdown, EA, Ibeg) and the
 * -D NE -D NFREQ build of A2E.py:283-304.  L1/L2 are [NE*NE] indexed l*NE+u, Iw holds noIw
 * weights in (l, u, i) loop order, EA is [NFREQ*NE]. */
int soc_a2e_set_size(soc_ctx *ctx, int NE, int NFREQ, int noIw, const float *Iw, const int32_t *L1,
                     const int32_t *L2, const float *Tdown, const float *EA, const int32_t *Ibeg, const float *AF);

/* replaces enqueue_copy(ABS_buf) + DoSolve(...) + enqueue_copy(emit, EMIT_buf) for one batch of
 * cells (A2E.py:387-412 -> kernel_A2E.c:2-104): AABS, AEMIT are [batch*NFREQ] host arrays */
int soc_a2e_solve(soc_ctx *ctx, int batch, const float *AABS, float *AEMIT);
/* the same in three steps with the batch resident on the device (timing the kernel alone) */
int soc_a2e_upload(soc_ctx *ctx, int batch, const float *AABS);
int soc_a2e_run(soc_ctx *ctx, int batch);
int soc_a2e_download(soc_ctx *ctx, int batch, float *AEMIT);

/* The same with the cells RESIDENT in device memory.  The reference uploads every batch of cells once per grain size and adds the
 * sizes' emission up on the host (A2E.py:520-600: NSIZE x (absorptions in + emission out) over PCIe); a model's absorptions are
 * 4*NFREQ bytes per cell (config 3: 9.9 GB) and fit the device many times over.  soc_a2e_resident_begin(cells, NFREQ) allocates them
 * and a zeroed emission sum; _upload(c0, n, ABS) fills rows [c0, c0+n) (any chunking, e.g. from a memory-mapped absorbed file);
 * after every soc_a2e_set_size, _solve() runs DoSolve over all cells and ADDS the emission of that size to the sum -- the same fp32
 * additions in the same order as the host's EMITTED += emit; _download(c0, n, EMIT) reads rows of the sum; _end() frees both. */
int soc_a2e_resident_begin(soc_ctx *ctx, int64_t cells, int NFREQ);
int soc_a2e_resident_upload(soc_ctx *ctx, int64_t c0, int64_t n, const float *AABS);
int soc_a2e_resident_solve(soc_ctx *ctx);
int soc_a2e_resident_download(soc_ctx *ctx, int64_t c0, int64_t n, float *AEMIT);
int soc_a2e_resident_end(soc_ctx *ctx);

/* replaces kernel_T(...) = EqTemperature for one batch (A2E.py:511-530 -> kernel_A2E.c:110-154);
 * TTT holds NIP temperatures, ABS is [batch*NFREQ] (already multiplied by AF on the host),
 * outputs T[batch] and EMIT[batch*NFREQ] */
int soc_a2e_eqtemp(soc_ctx *ctx, int batch, int icell, int CELLS, int NFREQ, int NIP, float FACTOR, float kE,
                   float oplgkE, float Emin, const float *FREQ, const float *KABS, const float *TTT,
                   const float *ABS, float *T, float *EMIT);

/* What A2E_pre.py computes per grain size for a <dust>.solver file (A2E_pre.py:233-256; kernel_A2E_pre.c:580-736
 * PrepareIntegrationWeightsTrapezoid, :123-212 PrepareTdown; -D FACTOR of A2E_pre.py:134 is an argument).
 * In:  FREQ[NFREQ], Ef[NFREQ] = PLANCK*FREQ, SKABS[NFREQ] = pi a^2 Q_abs of ONE grain of this size, the enthalpy grid E[NE+1]
 *      with its temperatures T[NE+1].
 * Out: L1, L2[NE*NE] (first and last frequency feeding the transition l -> u at [l*NE+u]; -1, -2 = none; entries with
 *      u <= l are 0 -- the caller sets [0] = -2 as A2E_pre.py:246,249 does), Iw[NE*NE*NFREQ] (the weights of lower bin l
 *      start at l*NE*NFREQ, noIw[l] of them: the file holds them back to back), noIw[NE-1], Tdown[NE]. */
int soc_a2e_pre(soc_ctx *ctx, int NFREQ, int NE, float FACTOR, const float *FREQ, const float *Ef, const float *SKABS,
                const float *E, const float *T, int32_t *L1, int32_t *L2, float *Iw, int32_t *noIw, float *Tdown);

/* ---- equilibrium dust components of a multi-dust run: A2E_MABU.py / kernel_eqsolver.c ---- */

/* replaces kernel_T(...) + the per-frequency kernel_emission(...) launches of SolveEquilibriumDust for one batch of cells
 * (A2E_MABU.py:520-560,608 -> kernel_eqsolver.c EqTemperature :5-62, Emission :66-79): ABS is [batch*NFREQ], the share of
 * the absorptions taken by this dust component (split_absorbed, kernel_A2E_MABU_aux.c:3-23, is done on the host side of
 * soc_amd/driver.py); TTT holds NE temperatures; outputs T[batch] and EMIT[batch*NFREQ] per unit density and abundance */
int soc_eqsolver(soc_ctx *ctx, int batch, int icell, int CELLS, int NFREQ, int NE, float FACTOR, float kE,
                 float oplgkE, float Emin, const float *FREQ, const float *KABS, const float *TTT,
                 const float *ABS, float *T, float *EMIT);

/* ---- verification probes (used by the parity tests only) ---- */
/* RNG stream states and first draws of logical work items [gid_first, gid_first+n)
 * (MWC64X_SeedStreams + MWC64X_NextUint, mwc64x_rng.cl:35-48) */
int soc_probe_rng(soc_ctx *ctx, float SEED, uint32_t gid_first, uint32_t n, int ndraw,
                  uint32_t *state_xc, uint32_t *draws);
/* device math header: fn 0 exp, 1 log, 2 sin, 3 cos, 4 acos, 5 sqrt, 6 fmod(x,1), 7 1/x */
int soc_probe_math(soc_ctx *ctx, int fn, const float *x, float *y, int64_t n);
/* follow one ray (IndexG + GetStep until exit); returns the number of steps in *nsteps */
int soc_probe_trace(soc_ctx *ctx, const float pos[3], const float dir[3], int maxsteps,
                    int32_t *levels, int32_t *inds, float *ds, float endpos[3], int32_t *nsteps);

#ifdef __cplusplus
}
#endif
#endif /* SOC_HIP_H */
