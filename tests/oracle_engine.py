"""An object with the method names of soc_amd.lib.Engine, backed by the CPU oracle.
Lives in tests/: it lets the host driver (soc_amd.asoc.AbsorptionRun) and the multi-process
sharding logic be exercised without a GPU.  Never imported by the product."""
import numpy as np

from oracle.pyoracle import Job, Oracle, ScaView, oracle_sim_sca, oracle_mapping, NO_INTOBS


class OracleEngine:
    def __init__(self, mode="soc"):
        self.orc = Oracle(mode)
        self.cloud = None
        self.feat = dict(with_int=0, ps_method=0, use_emweight=0)
        self.OPT = None
        self.ABS = self.SCA = 0.0
        self.CSC = self.DSC = None
        self.EMIT = self.EMWEI = None
        self.T = [None, None]
        self.events = 0
        self.ali = 0
        self.EMINDEX = None

    def set_cloud(self, cloud):
        self.cloud = cloud
        self.CELLS = cloud.CELLS
        self.T = [np.zeros(cloud.CELLS, np.float32), np.zeros(cloud.CELLS, np.float32), np.zeros(cloud.CELLS, np.float32)]

    def set_features(self, with_int=0, ps_method=0, use_emweight=0):
        self.feat = dict(with_int=with_int, ps_method=ps_method, use_emweight=use_emweight)
        self.INTV = np.zeros((3, self.cloud.CELLS), np.float32) if with_int == 2 else None

    def set_mirror(self, mask=0):
        self.mirror = int(mask)

    def set_optical(self, ABS, SCA):
        self.ABS, self.SCA = ABS, SCA

    def set_opt(self, OPT):
        self.OPT = OPT

    def set_abundances(self, ABU, single=False):
        self.ABU, self.abu_single = (None if ABU is None else np.asarray(ABU, np.float32)), bool(single)

    def set_optical_abu(self, AFABS, AFSCA):
        """the numpy expressions of ASOC.py:1146-1160"""
        OPT = np.zeros((self.cloud.CELLS, 2), np.float32)
        if self.abu_single:
            OPT[:, 0] += self.ABU * np.float32(AFABS[0]) + (1.0 - self.ABU) * np.float32(AFABS[1])
            OPT[:, 1] += self.ABU * np.float32(AFSCA[0]) + (1.0 - self.ABU) * np.float32(AFSCA[1])
        else:
            ABU = self.ABU.reshape(self.cloud.CELLS, -1)
            for d in range(ABU.shape[1]):
                OPT[:, 0] += ABU[:, d] * np.float32(AFABS[d])
                OPT[:, 1] += ABU[:, d] * np.float32(AFSCA[d])
        if getattr(self, "opt_half", False):
            OPT = np.asarray(np.asarray(OPT, np.float16), np.float32)
        self.OPT = OPT
        self.af = (np.asarray(AFABS, np.float32).copy(), np.asarray(AFSCA, np.float32).copy())

    def set_opt_half(self, on=True):
        self.opt_half = bool(on)

    def read_opt(self):
        return self.OPT.copy()

    def set_scatter_table(self, DSC, CSC):
        self.DSC, self.CSC = DSC, CSC
        self.msf_csc = None

    def set_scatter_tables(self, DSC, CSC):
        CSC = np.asarray(CSC, np.float32)
        self.DSC, self.CSC, self.msf_csc = (None if DSC is None else np.asarray(DSC, np.float32).copy()), CSC[0], CSC.copy()

    def set_step_weight(self, mode, SW_A=0.0, SW_B=0.0):
        self.step_weight = None if mode <= 0 else (int(mode), float(SW_A), float(SW_B))

    def set_emission(self, EMIT, EMWEI=None):
        self.EMIT, self.EMWEI = EMIT, EMWEI

    def zero(self, tag):
        self.T[tag][:] = 0
        if tag == 0:
            self.T[2][:] = 0
        if tag == 1 and getattr(self, "INTV", None) is not None:
            self.INTV[:] = 0

    def set_ali(self, with_ali=1):
        self.ali = int(with_ali)

    def set_emindex(self, EMINDEX):
        self.EMINDEX = np.asarray(EMINDEX, np.int32).copy()

    # region of interest
    def set_roi_save(self, ROI, ROI_STEP=1, ROI_NSIDE=16):
        self.roi = None if ROI is None else (np.asarray(ROI, np.int32).copy(), int(ROI_STEP), int(ROI_NSIDE))
        self.roi_rec = None
        if ROI is not None:
            n = [(int(ROI[2 * i + 1]) - int(ROI[2 * i]) + 1) * int(ROI_STEP) for i in range(3)]
            self.roi_rec = np.zeros((n[0] * n[1] + n[1] * n[2] + n[2] * n[0]) * 12 * int(ROI_NSIDE) ** 2, np.float32)
            return self.roi_rec.size
        return 0

    def roi_zero(self):
        self.roi_rec[:] = 0

    def roi_read(self):
        return self.roi_rec.copy()

    def set_roi_load(self, DIM, ROI_NSIDE, LOAD):
        self.roi_load = None if LOAD is None else (np.asarray(DIM, np.int32).copy(), int(ROI_NSIDE), np.asarray(LOAD, np.float32).copy())

    def _roi(self, job, source):
        if getattr(self, "roi", None) is not None:
            job.ROI, job.ROI_STEP, job.ROI_NSIDE = self.roi
            job.ROI_SAVE = self.roi_rec
        if source == 3 and getattr(self, "roi_load", None) is not None:
            job.ROI_DIM, job.ROI_NSIDE, job.ROI_LOAD = self.roi_load
        return job

    # batches (the CPU stand-in executes every launch at once)
    def batch_begin(self, max_launches=0):
        self._int_batch = None

    def batch_begin_shared_int(self, max_launches=0):
        self._int_batch = None                     # the launches tally into the shared INT array, as immediate ones do

    def batch_begin_int(self, max_launches=0):
        self._int_batch = []
        self._int_max = max_launches or 16
        self._int_groups = False

    def batch_begin_int_groups(self, max_groups=0):
        self.batch_begin_int(max_groups or 128)
        self._int_groups, self._int_open = True, False

    def batch_next_int(self):
        self._int_open = False

    def batch_end(self):
        if getattr(self, "_int_batch", None) is not None:
            self._int_done, self._int_batch = self._int_batch, None     # launches after the batch tally into the shared INT again

    def batch_read_int(self, k):
        return self._int_done[k].copy()

    def _int_target(self):
        """the INT array of the next launch: its own inside batch_begin_int, else the shared one"""
        b = getattr(self, "_int_batch", None)
        if b is None or not self.feat["with_int"]:
            return self.T[1]
        if getattr(self, "_int_groups", False) and self._int_open:
            return b[-1]                                   # a further launch of the current group
        if len(b) >= self._int_max:
            raise RuntimeError("soc_batch_end and soc_batch_read_int first")
        b.append(np.zeros(self.cloud.CELLS, np.float32))
        self._int_open = True
        return b[-1]

    def bind_tally(self, which, ptr, n=None):
        raise NotImplementedError

    def set_stream(self, s):
        pass

    def _job(self, SOURCE, PACKETS, BATCH, SEED, BG, TW, GLOBAL, PSPOS=None, PS=None, XPS=None):
        msf = None
        if getattr(self, "msf_csc", None) is not None:
            msf = (self.af[0], self.af[1], self.msf_csc, self.ABU.reshape(self.cloud.CELLS, -1))
        job = Job(self.cloud, self.CSC, ABS=self.ABS, SCA=self.SCA, SOURCE=SOURCE, BATCH=BATCH, SEED=SEED, BG=BG,
                   TW=TW, GLOBAL=GLOBAL, PACKETS=PACKETS, PSPOS=PSPOS if SOURCE == 0 else None,
                   PS=PS if SOURCE == 0 else None, PS_METHOD=self.feat["ps_method"], XPS=XPS if SOURCE == 0 else None,
                   OPT=self.OPT, EMIT=self.EMIT, EMWEI=self.EMWEI, USE_EMWEIGHT=self.feat["use_emweight"],
                   WITH_INT=self.feat["with_int"], DSC=self.DSC, MIRROR=getattr(self, "mirror", 0),
                   STEP_WEIGHT=getattr(self, "step_weight", None), MSF=msf)
        if job.INTV is not None:
            job.INTV = self.INTV
        return job

    def sim_pb(self, SOURCE, PACKETS, BATCH, SEED, BG, TW, PSPOS=None, PS=None, XPS=None, GLOBAL=None,
               gid_first=0, gid_count=None):
        job = self._roi(self._job(SOURCE, PACKETS, BATCH, SEED, BG, TW, GLOBAL, PSPOS, PS, XPS), SOURCE)
        gid_count = GLOBAL - gid_first if gid_count is None else gid_count
        _, _, n = self.orc.sim(job, 0, gid_first, gid_first + gid_count, TABS=self.T[0], INT=self._int_target(), nthreads=getattr(self, 'threads', 1))
        self.events += n

    def set_hpbg(self, BG, HPBGP=None):
        self.HPBG, self.HPBGP = np.asarray(BG, np.float32).copy(), None if HPBGP is None else np.asarray(HPBGP, np.float32).copy()

    def sim_hp(self, PACKETS, BATCH, SEED, TW, GLOBAL, gid_first=0, gid_count=None):
        job = self._job(1, PACKETS, BATCH, SEED, 0.0, TW, GLOBAL)
        job.HPBG, job.HPBGP = self.HPBG, self.HPBGP
        gid_count = GLOBAL - gid_first if gid_count is None else gid_count
        _, _, n = self.orc.sim(job, 2, gid_first, gid_first + gid_count, TABS=self.T[0], INT=self._int_target(), nthreads=getattr(self, 'threads', 1))
        self.events += n

    def sim_cl(self, SOURCE, PACKETS, BATCH, SEED, TW, GLOBAL, gid_first=0, gid_count=None):
        job = self._job(SOURCE, PACKETS, BATCH, SEED, 0.0, TW, GLOBAL)
        job.WITH_ALI, job.XAB, job.EMINDEX = self.ali, self.T[2], self.EMINDEX
        self._roi(job, SOURCE)
        gid_count = GLOBAL - gid_first if gid_count is None else gid_count
        _, _, n = self.orc.sim(job, 1, gid_first, gid_first + gid_count, TABS=self.T[0], INT=self._int_target(), nthreads=getattr(self, 'threads', 1))
        self.events += n

    # ---- temperature and emission ----
    def set_cr_heating(self, rate):
        self.cr_rate = float(rate)

    def set_map_roi(self, ROI):
        self.map_roi = None if ROI is None else [int(v) for v in ROI]

    def set_map_threshold(self, level):
        self.map_threshold = int(level)

    def set_map_interpolation(self, mode):
        self.map_interpolation = int(mode)

    def solve_temperature(self, adhoc, kE, Emin, TTT, FACTOR, LENGTH, EABS):
        job = Job(self.cloud, np.linspace(1, -1, 8))
        job.CR_HEATING_RATE = getattr(self, "cr_rate", 0.0)
        self.Tdust = self.orc.eqtemp(job, adhoc, kE, Emin, TTT, FACTOR, LENGTH, EABS)
        return self.Tdust.copy()

    def set_temperature(self, T):
        self.Tdust = np.asarray(T, np.float32).copy()

    def emission(self, FREQ, FABS, FACTOR, LENGTH):
        return self.orc.emission(FREQ, FABS, FACTOR, LENGTH, self.Tdust)

    # ---- map making ----
    def map(self, EMIT, DIR, RA, DE, NPIX, MAP_DX, CENTRE, ABS, SCA, INTOBS=None, save_colden=0, LENGTH=1.0, healpix=0):
        job = Job(self.cloud, np.linspace(1, -1, 8), ABS=ABS, SCA=SCA, OPT=self.OPT)
        job.LEVEL_THRESHOLD = getattr(self, "map_threshold", 0)
        job.MAP_INTERPOLATION = 0 if healpix else getattr(self, "map_interpolation", 0)
        job.ROI_MAP = getattr(self, "map_roi", None)
        io = NO_INTOBS if (INTOBS is None or INTOBS[0] < -1e10) else INTOBS
        m, t = oracle_mapping(self.orc, job, EMIT, DIR, RA, DE, NPIX, MAP_DX, CENTRE, io, save_colden, LENGTH, healpix)
        shape = (m.size,) if healpix else (int(NPIX[1]), int(NPIX[0]))
        return m.reshape(shape), t.reshape(shape)

    def ps_tau(self, PSPOS, DIR, ABS, SCA, LENGTH=1.0):
        from oracle.pyoracle import oracle_pstau
        job = Job(self.cloud, np.linspace(1, -1, 8), ABS=ABS, SCA=SCA, OPT=self.OPT)
        return oracle_pstau(self.orc, job, PSPOS, DIR, LENGTH)

    # ---- scattered-light images ----
    def sca_set_view(self, ODIR, RA, DE, NPIX, MAP_DX, CENTRE, FFS=1):
        self.view = ScaView(ODIR, RA, DE, NPIX=NPIX, MAP_DX=MAP_DX, CENTRE=CENTRE, FFS=FFS)
        self.OUT = np.zeros(self.view.out_size(), np.float32)
        self.sca_shape = (self.view.NDIR, int(NPIX[1]), int(NPIX[0]))

    def sca_set_healpix(self, NSIDE, OBSERVER, FFS=1):
        z = np.zeros((1, 4), np.float32)
        o = np.zeros((1, 4), np.float32)
        o[0, :3] = OBSERVER
        self.view = ScaView(o, z, z, NPIX=(1, 1), MAP_DX=1.0, CENTRE=(0, 0, 0), FFS=FFS, nside=NSIDE)
        self.OUT = np.zeros(self.view.out_size(), np.float32)
        self.sca_shape = (self.view.out_size(),)

    def sca_sim_hp(self, PACKETS, BATCH, SEED, GLOBAL, gid_first=0, gid_count=None):
        job = self._job(1, PACKETS, BATCH, SEED, 0.0, 0.0, GLOBAL)
        job.HPBG, job.HPBGP = self.HPBG, self.HPBGP
        self._sca(3, job, GLOBAL, gid_first, gid_count)

    def sca_zero(self):
        self.OUT[:] = 0

    def _sca(self, kind, job, GLOBAL, gid_first, gid_count):
        gid_count = GLOBAL - gid_first if gid_count is None else gid_count
        _, n = oracle_sim_sca(self.orc, job, self.view, kind, gid_first, gid_first + gid_count, OUT=self.OUT)
        self.events += n

    def sca_sim_ps(self, PACKETS, BATCH, SEED, BG, PSPOS, PS, XPS=None, GLOBAL=None, gid_first=0, gid_count=None):
        self._sca(2, self._job(0, PACKETS, BATCH, SEED, BG, 0.0, GLOBAL, PSPOS, PS, XPS), GLOBAL, gid_first, gid_count)

    def sca_sim_pb(self, SOURCE, PACKETS, BATCH, SEED, BG, PSPOS=None, PS=None, XPS=None, GLOBAL=None, gid_first=0,
                   gid_count=None):
        self._sca(0, self._job(SOURCE, PACKETS, BATCH, SEED, BG, 0.0, GLOBAL, PSPOS, PS, XPS), GLOBAL, gid_first, gid_count)

    def sca_sim_cl(self, SOURCE, PACKETS, BATCH, SEED, GLOBAL, gid_first=0, gid_count=None):
        self._sca(1, self._job(2, PACKETS, BATCH, SEED, 0.0, 0.0, GLOBAL), GLOBAL, gid_first, gid_count)

    def sca_read_out(self):
        return self.OUT.reshape(self.sca_shape).copy()

    # images of a batch (soc_sca_batch_images / _select / _read): the oracle runs every launch at once, into the selected image
    def sca_batch_images(self, n):
        if getattr(self, "_one", None) is None:
            self._one = self.OUT
        self._imgs = [np.zeros_like(self._one) for _ in range(n)]
        self.OUT = self._imgs[0] if n else self._one
        if not n:
            self._one = None

    def sca_batch_select(self, k):
        self.OUT = self._imgs[k]

    def sca_batch_read(self, k):
        return self._imgs[k].reshape(self.sca_shape).copy()

    def sca_bind_out(self, ptr):
        raise NotImplementedError

    def sync(self):
        pass

    def read_tally(self, which=0):
        if which >= 3:
            return self.INTV[which - 3].copy()
        return self.T[which].copy()

    def write_tally(self, which, values):
        self.T[which][:] = values

    def close(self):
        pass


class _A2EMethods:
    """the A2E / equilibrium-dust entry points of soc_amd.lib.Engine on the CPU oracle"""

    def a2e_set_size(self, NE, NFREQ, size, AF):
        self.NE, self.NFREQ, self.size, self.AF = NE, NFREQ, size, np.asarray(AF, np.float32)

    def a2e_solve(self, AABS):
        from oracle.pyoracle import a2e_oracle_dosolve
        return a2e_oracle_dosolve(self.orc, self.NE, self.NFREQ, self.size, self.AF, np.ascontiguousarray(AABS, np.float32))

    def a2e_eqtemp(self, icell, CELLS, NIP, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS):
        from oracle.pyoracle import a2e_oracle_eqtemp
        return a2e_oracle_eqtemp(self.orc, icell, CELLS, NIP, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS)

    def eqsolver(self, icell, CELLS, NE, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS):
        from oracle.pyoracle import oracle_eqsolver
        return oracle_eqsolver(self.orc, icell, CELLS, NE, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS)


class OracleA2E(_A2EMethods):
    """multi-process tests of soc_amd.a2e"""

    def __init__(self, mode="soc"):
        self.orc = Oracle(mode)


class OraclePipelineEngine(OracleEngine, _A2EMethods):
    """everything soc_amd.driver.Pipeline calls: the packet engine, the map kernel and the emission solvers"""
