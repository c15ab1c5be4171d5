"""ctypes access to the CPU oracle (oracle/soc_oracle.c) and to the x86 builds of the
reference kernels (oracle/_ref).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by soc_amd/.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int)
_U = C.POINTER(C.c_uint32)


def _fp(a):
    return None if a is None else a.ctypes.data_as(_F)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_I)


class OrcModel(C.Structure):
    _fields_ = [
        ("NX", C.c_int), ("NY", C.c_int), ("NZ", C.c_int), ("LEVELS", C.c_int), ("CELLS", C.c_int),
        ("BINS", C.c_int), ("PS_METHOD", C.c_int), ("NO_PS", C.c_int),
        ("WITH_ABU", C.c_int), ("WITH_INT", C.c_int), ("USE_EMWEIGHT", C.c_int), ("DOUBLE_INDEX", C.c_int),
        ("LCELLS", _I), ("OFF", _I), ("PAR", _I),
        ("DENS", _F), ("CSC", _F), ("OPT", _F),
        ("SOURCE", C.c_int), ("PACKETS", C.c_int), ("BATCH", C.c_int), ("GLOBAL", C.c_int),
        ("SEED", C.c_float), ("ABS", C.c_float), ("SCA", C.c_float), ("BG", C.c_float), ("TW", C.c_float),
        ("PSPOS", _F), ("PS", _F), ("XPS_NSIDE", _I), ("XPS_SIDE", _I), ("XPS_AREA", _F),
        ("EMIT", _F), ("EMWEI", _F), ("TABS", _F), ("INT", _F),
        ("threaded", C.c_int),
        ("NDIR", C.c_int), ("NPIX_X", C.c_int), ("NPIX_Y", C.c_int), ("FFS", C.c_int),
        ("MAP_DX", C.c_float), ("CX", C.c_float), ("CY", C.c_float), ("CZ", C.c_float),
        ("ODIRS", _F), ("ORA", _F), ("ODE", _F), ("DSC", _F), ("OUT", _F), ("XPS_AS_FLOAT", C.c_int),
        ("HPBG_WEIGHTED", C.c_int), ("HPBG", _F), ("HPBGP", _F), ("MIRROR", C.c_int),
        ("WITH_ALI", C.c_int), ("XAB", _F), ("EMINDEX", _I),
        ("WITH_ROI_SAVE", C.c_int), ("ROI", C.c_int * 6), ("ROI_STEP", C.c_int), ("ROI_NSIDE", C.c_int), ("ROI_SAVE", _F),
        ("WITH_ROI_LOAD", C.c_int), ("ROI_DIM", C.c_int * 3), ("ROI_LOAD", _F),
        ("STEP_WEIGHT", C.c_int), ("SW_A", C.c_float), ("SW_B", C.c_float),
        ("MSF_NDUST", C.c_int), ("MSF_SCA", _F), ("ABU", _F),
        ("INTV", _F),
        ("LEVEL_THRESHOLD", C.c_int), ("ROI_MAP", C.c_int), ("CR_HEATING_RATE", C.c_float),
        ("MAP_INTERPOLATION", C.c_int),
    ]


class RefArgs(C.Structure):
    _fields_ = [
        ("SOURCE", C.c_int), ("PACKETS", C.c_int), ("BATCH", C.c_int), ("GLOBAL", C.c_int),
        ("SEED", C.c_float), ("BG", C.c_float), ("TW", C.c_float),
        ("ABS", _F), ("SCA", _F), ("PSPOS", _F), ("PS", _F),
        ("LCELLS", _I), ("OFF", _I), ("PAR", _I),
        ("DENS", _F), ("EMIT", _F), ("TABS", _F), ("DSC", _F), ("CSC", _F), ("XAB", _F), ("EMWEI", _F),
        ("INT", _F), ("INTX", _F), ("INTY", _F), ("INTZ", _F), ("OPT", _F), ("ABU", _F),
        ("XPS_NSIDE", _I), ("XPS_SIDE", _I), ("XPS_AREA", _F), ("EMINDEX", _I), ("HPBG", _F), ("HPBGP", _F),
        ("ROI_DIM", _I), ("ROI_LOAD", _F), ("ROI", _I), ("ROI_SAVE", _F),
    ]


def double_index(NX, LEVELS):
    """kernel_ASOC_aux.c:25-37: Index() switches to double when NX > DIMLIM."""
    return int(NX > (399 if LEVELS < 3 else 100))


class Job:
    """Everything one kernel launch needs, as numpy arrays (shared by Oracle and Ref)."""

    def __init__(self, cloud, CSC, ABS=0.0, SCA=0.0, SOURCE=1, BATCH=1, SEED=0.5, BG=1.0, TW=1.0,
                 GLOBAL=None, PACKETS=0, PSPOS=None, PS=None, PS_METHOD=0, XPS=None, OPT=None,
                 EMIT=None, EMWEI=None, USE_EMWEIGHT=0, WITH_INT=0, DSC=None, HPBG=None, HPBGP=None, MIRROR=0,
                 WITH_ALI=0, EMINDEX=None, ROI=None, ROI_STEP=1, ROI_NSIDE=2, ROI_LOAD=None, ROI_DIM=None,
                 STEP_WEIGHT=None, MSF=None):
        self.cloud = cloud
        # weighted free paths: STEP_WEIGHT = (mode 1|2, SW_A, SW_B) as the -D values (-D DIR_WEIGHT > 0 does not compile);
        # MSF = (ABS[NDUST], SCA[NDUST], CSC[NDUST, BINS], ABU[CELLS, NDUST]): per-dust scattering functions (WITH_MSF),
        # OPT must be the matching sum(ABU * cross sections) (WITH_ABU)
        self.STEP_WEIGHT = (0, 0.0, 0.0) if STEP_WEIGHT is None else (int(STEP_WEIGHT[0]), float(STEP_WEIGHT[1]), float(STEP_WEIGHT[2]))
        self.MSF = None
        if MSF is not None:
            self.MSF = (np.ascontiguousarray(MSF[0], np.float32), np.ascontiguousarray(MSF[1], np.float32),
                        np.ascontiguousarray(MSF[2], np.float32), np.ascontiguousarray(MSF[3], np.float32))
            CSC = self.MSF[2].reshape(len(self.MSF[1]), -1)[0]
        # region of interest: ROI = [x0,x1,y0,y1,z0,z1] turns on WITH_ROI_SAVE; ROI_LOAD [elements, 12*NSIDE^2]
        # with ROI_DIM = (nx, ny, nz) of its surface discretisation is the SOURCE == 3 input (WITH_ROI_LOAD)
        self.ROI = None if ROI is None else np.ascontiguousarray(ROI, np.int32)
        self.ROI_STEP, self.ROI_NSIDE = int(ROI_STEP), int(ROI_NSIDE)
        self.ROI_LOAD = None if ROI_LOAD is None else np.ascontiguousarray(ROI_LOAD, np.float32)
        self.ROI_DIM = None if ROI_DIM is None else np.ascontiguousarray(ROI_DIM, np.int32)
        self.ROI_SAVE = None
        if self.ROI is not None:
            n = [(self.ROI[2 * i + 1] - self.ROI[2 * i] + 1) * self.ROI_STEP for i in range(3)]
            self.ROI_SAVE = np.zeros((n[0] * n[1] + n[1] * n[2] + n[2] * n[0]) * 12 * self.ROI_NSIDE ** 2, np.float32)
        self.WITH_ALI = int(WITH_ALI)
        self.EMINDEX = None if EMINDEX is None else np.ascontiguousarray(EMINDEX, np.int32)
        self.XAB = np.zeros(cloud.CELLS, np.float32)
        self.MIRROR = int(MIRROR)
        # Healpix sky (NSIDE 64, RING) in photons per package; HPBGP given = weighted pixel selection
        self.HPBG = None if HPBG is None else np.ascontiguousarray(HPBG, np.float32)
        self.HPBGP = None if HPBGP is None else np.ascontiguousarray(HPBGP, np.float32)
        self.CSC = np.ascontiguousarray(CSC, np.float32)
        # discrete scattering function of the peel-off (sca kernels); with MSF one row per species, DSC[NDUST, BINS]
        self.DSC = np.ascontiguousarray(DSC if DSC is not None else np.ones_like(self.CSC if self.MSF is None else self.MSF[2]), np.float32)
        self.BINS = len(self.CSC)
        self.ABS, self.SCA = np.float32(ABS), np.float32(SCA)
        self.SOURCE, self.BATCH = int(SOURCE), int(BATCH)
        self.SEED, self.BG, self.TW = np.float32(SEED), np.float32(BG), np.float32(TW)
        self.GLOBAL = int(GLOBAL if GLOBAL is not None else 8 * cloud.AREA)
        self.PACKETS = int(PACKETS)
        self.PS_METHOD = int(PS_METHOD)
        if PSPOS is None:
            PSPOS = np.zeros((1, 3), np.float32)
            PS = np.zeros(1, np.float32)
        PSPOS = np.asarray(PSPOS, np.float32).reshape(-1, 3)
        self.NO_PS = len(PSPOS)
        self.PSPOS = np.zeros((self.NO_PS, 4), np.float32)     # cl float3 = 16 bytes
        self.PSPOS[:, :3] = PSPOS
        self.PS = np.ascontiguousarray(PS, np.float32)
        if XPS is None:
            XPS = (np.zeros(self.NO_PS, np.int32), np.zeros(3 * self.NO_PS, np.int32),
                   np.ones(3 * self.NO_PS, np.float32))
        self.XPS_NSIDE, self.XPS_SIDE, self.XPS_AREA = (np.ascontiguousarray(XPS[0], np.int32),
                                                        np.ascontiguousarray(XPS[1], np.int32),
                                                        np.ascontiguousarray(XPS[2], np.float32))
        self.OPT = None if OPT is None else np.ascontiguousarray(OPT, np.float32)
        self.EMIT = np.ascontiguousarray(EMIT if EMIT is not None else np.zeros(cloud.CELLS), np.float32)
        self.EMWEI = np.ascontiguousarray(EMWEI if EMWEI is not None else np.ones(cloud.CELLS), np.float32)
        self.USE_EMWEIGHT = int(USE_EMWEIGHT)
        self.WITH_INT = int(WITH_INT)             # 2 = -D SAVE_INTENSITY=2: INT and the vector sums INTV[3, CELLS] = INTX, INTY, INTZ
        self.INTV = np.zeros((3, cloud.CELLS), np.float32) if self.WITH_INT == 2 else None
        self.LCELLS = np.ascontiguousarray(cloud.LCELLS, np.int32)
        self.OFF = np.ascontiguousarray(cloud.OFF, np.int32)
        self.DENS = np.ascontiguousarray(cloud.DENS, np.float32)
        self.PAR = None


class Oracle:
    """The CPU restatement.  mode = 'soc' (shared math header) or 'libm'."""

    def __init__(self, mode="soc"):
        libs = _build.build_oracle()
        self.mode = mode
        self.lib = C.CDLL(libs[mode])
        L = self.lib
        L.orc_seed_base.restype = C.c_uint64
        L.orc_seed_base.argtypes = [C.c_float]
        L.orc_seed.argtypes = [C.c_float, C.c_uint64, _U, _U]
        L.orc_draws.argtypes = [_U, _U, C.c_int, _U, _F]
        L.orc_parents.argtypes = [C.POINTER(OrcModel), _I]
        L.orc_trace.restype = C.c_int
        L.orc_trace.argtypes = [C.POINTER(OrcModel), _F, _F, C.c_int, _I, _I, _F, _F]
        L.orc_scatter.argtypes = [_F, _F, C.c_int, _U, _U]
        L.orc_deflect.argtypes = [_F, C.c_float, C.c_float]
        L.orc_sim.restype = C.c_long
        L.orc_sim.argtypes = [C.POINTER(OrcModel), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_math_eval.argtypes = [C.c_int, _F, _F, C.c_long]
        L.orc_indexg.argtypes = [C.POINTER(OrcModel), _F, _I, _I]
        assert L.orc_math_mode() == (1 if mode == "soc" else 0)

    # ---- RNG ----
    def seed(self, SEED, gid):
        x, c = C.c_uint32(), C.c_uint32()
        self.lib.orc_seed(np.float32(SEED), int(gid), C.byref(x), C.byref(c))
        return x.value, c.value

    def seed_base(self, SEED):
        return int(self.lib.orc_seed_base(np.float32(SEED)))

    def draws(self, x, c, n):
        xx, cc = C.c_uint32(x), C.c_uint32(c)
        u = np.zeros(n, np.uint32)
        r = np.zeros(n, np.float32)
        self.lib.orc_draws(C.byref(xx), C.byref(cc), n, u.ctypes.data_as(_U), _fp(r))
        return u, r, (xx.value, cc.value)

    # ---- model plumbing ----
    def _model(self, job):
        cl = job.cloud
        m = OrcModel()
        m.NX, m.NY, m.NZ, m.LEVELS, m.CELLS = cl.NX, cl.NY, cl.NZ, cl.LEVELS, cl.CELLS
        m.BINS, m.PS_METHOD, m.NO_PS = job.BINS, job.PS_METHOD, max(1, job.NO_PS)
        m.WITH_ABU = int(job.OPT is not None)
        m.WITH_INT, m.USE_EMWEIGHT = int(job.WITH_INT > 0), job.USE_EMWEIGHT
        m.INTV = _fp(job.INTV) if job.INTV is not None else None
        m.LEVEL_THRESHOLD = int(getattr(job, "LEVEL_THRESHOLD", 0))
        m.CR_HEATING_RATE = float(getattr(job, "CR_HEATING_RATE", 0.0))
        m.MAP_INTERPOLATION = int(getattr(job, "MAP_INTERPOLATION", 0))
        m.ROI_MAP = 0
        if getattr(job, "ROI_MAP", None) is not None:        # maps of the emission inside ROI = [x0,x1,y0,y1,z0,z1] only
            m.ROI_MAP = 1
            for k in range(6):
                m.ROI[k] = int(job.ROI_MAP[k])
        m.DOUBLE_INDEX = double_index(cl.NX, cl.LEVELS)
        m.LCELLS, m.OFF, m.DENS = _ip(job.LCELLS), _ip(job.OFF), _fp(job.DENS)
        if job.PAR is None:
            job.PAR = np.zeros(max(1, cl.CELLS - cl.NX * cl.NY * cl.NZ), np.int32)
            m.PAR = _ip(job.PAR)
            self.lib.orc_parents(C.byref(m), _ip(job.PAR))
        m.PAR = _ip(job.PAR)
        m.CSC, m.OPT = _fp(job.CSC), _fp(job.OPT)
        m.SOURCE, m.PACKETS, m.BATCH, m.GLOBAL = job.SOURCE, job.PACKETS, job.BATCH, job.GLOBAL
        m.SEED, m.ABS, m.SCA, m.BG, m.TW = job.SEED, job.ABS, job.SCA, job.BG, job.TW
        m.PSPOS, m.PS = _fp(job.PSPOS), _fp(job.PS)
        m.XPS_NSIDE, m.XPS_SIDE, m.XPS_AREA = _ip(job.XPS_NSIDE), _ip(job.XPS_SIDE), _fp(job.XPS_AREA)
        m.EMIT, m.EMWEI = _fp(job.EMIT), _fp(job.EMWEI)
        m.MIRROR = job.MIRROR
        m.WITH_ALI, m.XAB, m.EMINDEX = job.WITH_ALI, _fp(job.XAB), _ip(job.EMINDEX)
        m.HPBG_WEIGHTED = int(job.HPBGP is not None)
        m.HPBG, m.HPBGP = _fp(job.HPBG), _fp(job.HPBGP)
        m.ROI_STEP, m.ROI_NSIDE = job.ROI_STEP, job.ROI_NSIDE
        m.WITH_ROI_SAVE = int(job.ROI is not None)
        if job.ROI is not None:
            m.ROI = (C.c_int * 6)(*[int(v) for v in job.ROI])
            m.ROI_SAVE = _fp(job.ROI_SAVE)
        m.WITH_ROI_LOAD = int(job.ROI_LOAD is not None)
        if job.ROI_LOAD is not None:
            m.ROI_DIM = (C.c_int * 3)(*[int(v) for v in job.ROI_DIM])
            m.ROI_LOAD = _fp(job.ROI_LOAD)
        m.STEP_WEIGHT, m.SW_A, m.SW_B = job.STEP_WEIGHT
        m.MSF_NDUST = 0
        if job.MSF is not None:
            m.MSF_NDUST = len(job.MSF[1])
            m.MSF_SCA, m.CSC, m.ABU = _fp(job.MSF[1]), _fp(job.MSF[2]), _fp(job.MSF[3])
        return m

    def parents(self, job):
        self._model(job)
        return job.PAR

    def trace(self, job, pos, direction, maxsteps=100000):
        m = self._model(job)
        pos = np.ascontiguousarray(pos, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        lev = np.zeros(maxsteps, np.int32)
        ind = np.zeros(maxsteps, np.int32)
        ds = np.zeros(maxsteps, np.float32)
        end = np.zeros(3, np.float32)
        n = self.lib.orc_trace(C.byref(m), _fp(pos), _fp(d), maxsteps, _ip(lev), _ip(ind), _fp(ds), _fp(end))
        return lev[:n].copy(), ind[:n].copy(), ds[:n].copy(), end

    def indexg(self, job, pos):
        m = self._model(job)
        p = np.ascontiguousarray(pos, np.float32).copy()
        lev, ind = C.c_int(0), C.c_int(-1)
        self.lib.orc_indexg(C.byref(m), _fp(p), C.byref(lev), C.byref(ind))
        return p, lev.value, ind.value

    def scatter(self, direction, CSC, x, c):
        d = np.ascontiguousarray(direction, np.float32).copy()
        CSC = np.ascontiguousarray(CSC, np.float32)
        xx, cc = C.c_uint32(x), C.c_uint32(c)
        self.lib.orc_scatter(_fp(d), _fp(CSC), len(CSC), C.byref(xx), C.byref(cc))
        return d, (xx.value, cc.value)

    def deflect(self, direction, cos_theta, phi):
        d = np.ascontiguousarray(direction, np.float32).copy()
        self.lib.orc_deflect(_fp(d), np.float32(cos_theta), np.float32(phi))
        return d

    def sim(self, job, kind=0, gid0=0, gid1=None, nthreads=1, TABS=None, INT=None, stride=1):
        """Run work items gid0, gid0+stride, ... < gid1 of SimRAM_PB (kind 0) / SimRAM_CL (kind 1).
        Returns (TABS, INT, tally_events)."""
        m = self._model(job)
        cells = job.cloud.CELLS
        TABS = np.zeros(cells, np.float32) if TABS is None else TABS
        INT = np.zeros(cells, np.float32) if INT is None else INT
        m.TABS, m.INT = _fp(TABS), _fp(INT)
        gid1 = job.GLOBAL if gid1 is None else gid1
        n = self.lib.orc_sim(C.byref(m), kind, gid0, gid1, stride, nthreads)
        return TABS, INT, int(n)

    def eqtemp(self, job, adhoc, kE, Emin, TTT, FACTOR, LENGTH, EABS):
        """EqTemperature (all levels) -> TNEW[CELLS]"""
        m = self._model(job)
        TTT = np.ascontiguousarray(TTT, np.float32)
        EABS = np.ascontiguousarray(EABS, np.float32)
        T = np.zeros(job.cloud.CELLS, np.float32)
        self.lib.orc_eqtemp.argtypes = [C.POINTER(OrcModel), C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, _F, _F, _F]
        self.lib.orc_eqtemp(C.byref(m), np.float32(adhoc), np.float32(kE), np.float32(Emin), TTT.size, np.float32(FACTOR),
                            np.float32(LENGTH), _fp(TTT), _fp(EABS), _fp(T))
        return T

    def emission(self, FREQ, FABS, FACTOR, LENGTH, T, c0=0, c1=None):
        """Emission2 -> EMIT[c1-c0, nfreq]"""
        FREQ = np.ascontiguousarray(FREQ, np.float32)
        FABS = np.ascontiguousarray(FABS, np.float32)
        T = np.ascontiguousarray(T, np.float32)
        c1 = T.size if c1 is None else c1
        out = np.zeros((c1 - c0, FREQ.size), np.float32)
        self.lib.orc_emission.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _F, _F, _F, _F]
        self.lib.orc_emission(c0, c1, FREQ.size, np.float32(FACTOR), np.float32(LENGTH), _fp(FREQ), _fp(FABS), _fp(T), _fp(out))
        return out

    def math(self, fn, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        code = dict(exp=0, log=1, sin=2, cos=3, acos=4, sqrt=5, fmod1=6, expm1=8, pow15=9, logd=10)[fn]
        self.lib.orc_math_eval(code, _fp(x), _fp(y), x.size)
        return y


class Ref:
    """One x86 build of the reference kernels (geometry baked in): oracle/_ref/ref_<tag>.so."""

    def __init__(self, tag):
        models = _build.ref_models()
        self.tag = tag
        self.model = models[tag]
        path = _build.build_ref(tag, **self.model)
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("reference build %s not available" % tag)
        self.lib = C.CDLL(path)
        L = self.lib
        L.ref_sim.argtypes = [C.POINTER(RefArgs), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_parents.argtypes = [_F, _I, _I, _I]
        L.ref_seed.argtypes = [C.c_float, C.c_ulong, _U, _U]
        L.ref_draws.argtypes = [_U, _U, C.c_int, _U]
        L.ref_trace.restype = C.c_int
        L.ref_trace.argtypes = [_F, _F, C.c_int, _F, _I, _I, _I, _I, _F, _F]
        L.ref_scatter.argtypes = [_F, _F, _U, _U]
        L.ref_deflect.argtypes = [_F, C.c_float, C.c_float]
        L.ref_indexg.argtypes = [_F, _I, _I, _F, _I]

    @staticmethod
    def available(tag):
        try:
            Ref(tag)
            return True
        except (FileNotFoundError, OSError, KeyError):
            return False

    def _check(self, job):
        cl, m = job.cloud, self.model
        assert (cl.NX, cl.NY, cl.NZ, cl.LEVELS, cl.CELLS) == (m["NX"], m["NY"], m["NZ"], m["LEVELS"], m["CELLS"]), \
            "job geometry does not match reference build %s" % self.tag
        assert job.BINS == m.get("BINS", 2500)
        assert job.PS_METHOD == m.get("PS_METHOD", 0)
        assert int(job.OPT is not None) == m.get("WITH_ABU", 0)
        assert job.USE_EMWEIGHT == m.get("USE_EMWEIGHT", 0)
        assert int(job.WITH_INT > 0) == int(m.get("NOABSORBED", 1) == 0 or m.get("SAVE_INTENSITY", 0) in (1, 2))
        assert (job.WITH_INT == 2) == (m.get("SAVE_INTENSITY", 0) == 2)
        assert max(1, job.NO_PS) == max(1, m.get("NO_PS", 1))
        assert int(job.HPBGP is not None) == m.get("HPBG_WEIGHTED", 0)
        assert job.MIRROR == m.get("MIRROR", 0)
        assert job.WITH_ALI == m.get("WITH_ALI", 0)
        assert job.STEP_WEIGHT[0] == max(0, m.get("STEP_WEIGHT", -1))
        if job.STEP_WEIGHT[0] > 0:
            assert (np.float32(job.STEP_WEIGHT[1]), np.float32(job.STEP_WEIGHT[2])) == (np.float32(m["SW_A"]), np.float32(m["SW_B"]))
        assert (0 if job.MSF is None else len(job.MSF[1])) == (m.get("NDUST", 1) if m.get("WITH_MSF", 0) else 0)
        assert int(job.ROI is not None) == m.get("WITH_ROI_SAVE", 0) and int(job.ROI_LOAD is not None) == m.get("WITH_ROI_LOAD", 0)
        if job.ROI is not None:
            assert job.ROI_STEP == m.get("ROI_STEP", 0)
        if job.ROI is not None or job.ROI_LOAD is not None:
            assert job.ROI_NSIDE == m.get("ROI_NSIDE", 16)

    def seed(self, SEED, gid):
        x, c = C.c_uint32(), C.c_uint32()
        self.lib.ref_seed(np.float32(SEED), int(gid), C.byref(x), C.byref(c))
        return x.value, c.value

    def draws(self, x, c, n):
        xx, cc = C.c_uint32(x), C.c_uint32(c)
        u = np.zeros(n, np.uint32)
        self.lib.ref_draws(C.byref(xx), C.byref(cc), n, u.ctypes.data_as(_U))
        return u, (xx.value, cc.value)

    def parents(self, job):
        self._check(job)
        cl = job.cloud
        PAR = np.zeros(max(1, cl.CELLS - cl.NX * cl.NY * cl.NZ), np.int32)
        self.lib.ref_parents(_fp(job.DENS), _ip(job.LCELLS), _ip(job.OFF), _ip(PAR))
        return PAR

    def trace(self, job, pos, direction, maxsteps=100000):
        self._check(job)
        PAR = self.parents(job)
        pos = np.ascontiguousarray(pos, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        lev = np.zeros(maxsteps, np.int32)
        ind = np.zeros(maxsteps, np.int32)
        ds = np.zeros(maxsteps, np.float32)
        end = np.zeros(3, np.float32)
        n = self.lib.ref_trace(_fp(pos), _fp(d), maxsteps, _fp(job.DENS), _ip(job.OFF), _ip(PAR),
                               _ip(lev), _ip(ind), _fp(ds), _fp(end))
        return lev[:n].copy(), ind[:n].copy(), ds[:n].copy(), end

    def indexg(self, job, pos):
        p = np.ascontiguousarray(pos, np.float32).copy()
        lev, ind = C.c_int(0), C.c_int(-1)
        self.lib.ref_indexg(_fp(p), C.byref(lev), C.byref(ind), _fp(job.DENS), _ip(job.OFF))
        return p, lev.value, ind.value

    def scatter(self, direction, CSC, x, c):
        assert len(CSC) == self.model.get("BINS", 2500)
        d = np.ascontiguousarray(direction, np.float32).copy()
        CSC = np.ascontiguousarray(CSC, np.float32)
        xx, cc = C.c_uint32(x), C.c_uint32(c)
        self.lib.ref_scatter(_fp(d), _fp(CSC), C.byref(xx), C.byref(cc))
        return d, (xx.value, cc.value)

    def deflect(self, direction, cos_theta, phi):
        d = np.ascontiguousarray(direction, np.float32).copy()
        self.lib.ref_deflect(_fp(d), np.float32(cos_theta), np.float32(phi))
        return d

    def eqtemp(self, job, adhoc, kE, Emin, TTT, EABS):
        """the build's -D FACTOR / -D LENGTH apply (oracle/build.py: ref_defs)"""
        TTT = np.ascontiguousarray(TTT, np.float32)
        EABS = np.ascontiguousarray(EABS, np.float32)
        T = np.zeros(job.cloud.CELLS, np.float32)
        self.lib.ref_eqtemp.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, _I, _I, _F, _F, _F, _F]
        self.lib.ref_eqtemp(job.cloud.LEVELS, np.float32(adhoc), np.float32(kE), np.float32(Emin), TTT.size, _ip(job.OFF),
                            _ip(job.LCELLS), _fp(TTT), _fp(job.DENS), _fp(EABS), _fp(T))
        return T

    def emission(self, job, FREQ, FABS, T, c0=0, c1=None):
        FREQ = np.ascontiguousarray(FREQ, np.float32)
        FABS = np.ascontiguousarray(FABS, np.float32)
        T = np.ascontiguousarray(T, np.float32)
        c1 = T.size if c1 is None else c1
        out = np.zeros((c1 - c0, FREQ.size), np.float32)
        self.lib.ref_emission2.argtypes = [C.c_int, C.c_int, C.c_int, _F, _F, _F, _F, _F]
        self.lib.ref_emission2(c0, c1, FREQ.size, _fp(FREQ), _fp(FABS), _fp(job.DENS), _fp(T), _fp(out))
        return out

    def sim(self, job, kind=0, gid0=0, gid1=None, nthreads=1, TABS=None, INT=None, stride=1):
        self._check(job)
        cells = job.cloud.CELLS
        PAR = self.parents(job)
        TABS = np.zeros(cells, np.float32) if TABS is None else TABS
        INT = np.zeros(cells, np.float32) if INT is None else INT
        dummy = np.zeros(4, np.float32)
        idummy = np.zeros(4, np.int32)
        ABS = np.asarray([job.ABS], np.float32)
        SCA = np.asarray([job.SCA], np.float32)
        CSC, ABU = job.CSC, dummy
        if job.MSF is not None:                      # -D WITH_MSF: ABS[NDUST], SCA[NDUST], CSC[NDUST*BINS], ABU[CELLS*NDUST]
            ABS, SCA, CSC, ABU = job.MSF
        a = RefArgs()
        a.SOURCE, a.PACKETS, a.BATCH, a.GLOBAL = job.SOURCE, job.PACKETS, job.BATCH, job.GLOBAL
        a.SEED, a.BG, a.TW = job.SEED, job.BG, job.TW
        a.ABS, a.SCA, a.PSPOS, a.PS = _fp(ABS), _fp(SCA), _fp(job.PSPOS), _fp(job.PS)
        a.LCELLS, a.OFF, a.PAR = _ip(job.LCELLS), _ip(job.OFF), _ip(PAR)
        a.DENS, a.EMIT, a.TABS = _fp(job.DENS), _fp(job.EMIT), _fp(TABS)
        a.DSC, a.CSC, a.XAB, a.EMWEI = _fp(job.DSC), _fp(CSC), _fp(job.XAB), _fp(job.EMWEI)
        a.INT, a.INTX, a.INTY, a.INTZ = _fp(INT), _fp(dummy), _fp(dummy), _fp(dummy)
        if job.INTV is not None:
            a.INTX, a.INTY, a.INTZ = (job.INTV[k].ctypes.data_as(_F) for k in range(3))
        a.OPT = _fp(job.OPT) if job.OPT is not None else _fp(dummy)
        a.ABU = _fp(ABU)
        a.XPS_NSIDE, a.XPS_SIDE, a.XPS_AREA = _ip(job.XPS_NSIDE), _ip(job.XPS_SIDE), _fp(job.XPS_AREA)
        a.EMINDEX = _ip(job.EMINDEX) if job.EMINDEX is not None else _ip(idummy)
        a.HPBG, a.HPBGP = _fp(job.HPBG), _fp(job.HPBGP if job.HPBGP is not None else dummy)
        a.ROI_DIM, a.ROI_LOAD = _ip(job.ROI_DIM), _fp(job.ROI_LOAD)
        a.ROI, a.ROI_SAVE = _ip(job.ROI), _fp(job.ROI_SAVE)
        gid1 = job.GLOBAL if gid1 is None else gid1
        self.lib.ref_sim(C.byref(a), kind, gid0, gid1, stride, nthreads)
        return TABS, INT


# ---------------------------------------------------------------------------------------
# A2E (stochastic heating): oracle restatement and x86 builds of kernel_A2E.c
# ---------------------------------------------------------------------------------------

def a2e_oracle_dosolve(orc, NE, NFREQ, size, AF, AABS):
    """size: dict with Iw, L1, L2, Tdown, EA, Ibeg (one grain size of a solver file)."""
    L = orc.lib
    L.orc_a2e_dosolve.restype = C.c_int
    L.orc_a2e_dosolve.argtypes = [C.c_int, C.c_int, C.c_int, _F, _I, _I, _F, _F, _I, _F, _F, _F]
    AABS = np.ascontiguousarray(AABS, np.float32)
    batch = AABS.shape[0]
    out = np.zeros((batch, NFREQ), np.float32)
    arrs = [np.ascontiguousarray(size[k], t) for k, t in (("Iw", np.float32), ("L1", np.int32), ("L2", np.int32),
                                                           ("Tdown", np.float32), ("EA", np.float32), ("Ibeg", np.int32))]
    AF = np.ascontiguousarray(AF, np.float32)
    rc = L.orc_a2e_dosolve(batch, NE, NFREQ, _fp(arrs[0]), _ip(arrs[1]), _ip(arrs[2]), _fp(arrs[3]), _fp(arrs[4]),
                           _ip(arrs[5]), _fp(AF), _fp(AABS), _fp(out))
    assert rc == 0
    return out


def a2e_oracle_eqtemp(orc, icell, CELLS, NIP, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS):
    L = orc.lib
    L.orc_a2e_eqtemp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                 C.c_float, _F, _F, _F, _F, _F, _F]
    ABS = np.ascontiguousarray(ABS, np.float32)
    batch, NFREQ = ABS.shape
    T = np.zeros(batch, np.float32)
    E = np.zeros((batch, NFREQ), np.float32)
    FREQ, KABS, TTT = (np.ascontiguousarray(a, np.float32) for a in (FREQ, KABS, TTT))
    L.orc_a2e_eqtemp(batch, icell, CELLS, NFREQ, NIP, np.float32(FACTOR), np.float32(kE), np.float32(oplgkE),
                     np.float32(Emin), _fp(FREQ), _fp(KABS), _fp(TTT), _fp(ABS), _fp(T), _fp(E))
    return T, E


def oracle_eqsolver(orc, icell, CELLS, NE, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS):
    """kernel_eqsolver.c EqTemperature + Emission (orc_eqsolver): ABS[batch, NFREQ] -> T[batch], EMIT[batch, NFREQ]"""
    L = orc.lib
    L.orc_eqsolver.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                               C.c_float, _F, _F, _F, _F, _F, _F]
    ABS = np.ascontiguousarray(ABS, np.float32)
    batch, NFREQ = ABS.shape
    T = np.zeros(batch, np.float32)
    E = np.zeros((batch, NFREQ), np.float32)
    FREQ, KABS, TTT = (np.ascontiguousarray(a, np.float32) for a in (FREQ, KABS, TTT))
    L.orc_eqsolver(batch, icell, CELLS, NFREQ, NE, np.float32(FACTOR), np.float32(kE), np.float32(oplgkE),
                   np.float32(Emin), _fp(FREQ), _fp(KABS), _fp(TTT), _fp(ABS), _fp(T), _fp(E))
    return T, E


def a2e_oracle_pre(orc, FREQ, Ef, SKABS, E, T, FACTOR):
    """oracle restatement of the two kernels A2E_pre.py runs per grain size; same return value as Engine.a2e_pre"""
    L = orc.lib
    FREQ, Ef, SKABS = (np.ascontiguousarray(a, np.float32) for a in (FREQ, Ef, SKABS))
    E, T = np.ascontiguousarray(E, np.float32), np.ascontiguousarray(T, np.float32)
    NFREQ, NE = FREQ.size, E.size - 1
    L.orc_a2e_pre_weights.argtypes = [C.c_int, C.c_int, C.c_float, _F, _F, _I, _I, _F, _F, _I]
    L.orc_a2e_pre_tdown.argtypes = [C.c_int, _F, _F, _F, C.c_int, _F, _F, _F]
    L.orc_a2e_pre_weights.restype = None
    L.orc_a2e_pre_tdown.restype = None
    L1, L2 = np.zeros(NE * NE, np.int32), np.zeros(NE * NE, np.int32)
    Iw, noIw, Tdown = np.zeros(NE * NE * NFREQ, np.float32), np.zeros(NE - 1, np.int32), np.zeros(NE, np.float32)
    wrk = np.zeros(NE * NFREQ, np.float32)
    L.orc_a2e_pre_weights(NFREQ, NE, np.float32(FACTOR), _fp(Ef), _fp(E), _ip(L1), _ip(L2), _fp(Iw), _fp(wrk), _ip(noIw))
    L.orc_a2e_pre_tdown(NFREQ, _fp(FREQ), _fp(Ef), _fp(SKABS), NE, _fp(E), _fp(T), _fp(Tdown))
    packed = np.concatenate([Iw[l * NE * NFREQ:l * NE * NFREQ + noIw[l]] for l in range(NE - 1)])
    L1[0] = -2
    L2[0] = -2
    return dict(Iw=packed, L1=L1, L2=L2, Tdown=Tdown, noIw=noIw)


class RefA2EPre:
    """x86 build of kernel_A2E_pre.c (FACTOR = 1e20): oracle/_ref/refa2epre.so"""

    def __init__(self):
        path = _build.build_ref_a2e_pre()
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("reference build of kernel_A2E_pre.c not available")
        self.lib = C.CDLL(path)
        self.lib.ref_pre_weights.argtypes = [C.c_int, C.c_int, C.c_int, _F, _F, _I, _I, _F, _F, _I]
        self.lib.ref_pre_tdown.argtypes = [C.c_int, C.c_int, _F, _F, _F, C.c_int, _F, _F, _F]

    def pre(self, FREQ, Ef, SKABS, E, T):
        FREQ, Ef, SKABS = (np.ascontiguousarray(a, np.float32) for a in (FREQ, Ef, SKABS))
        E, T = np.ascontiguousarray(E, np.float32), np.ascontiguousarray(T, np.float32)
        NFREQ, NE = FREQ.size, E.size - 1
        GLOBAL = int((NE / 64) + 1) * 64                                     # A2E_pre.py:47
        L1, L2 = np.zeros(NE * NE, np.int32), np.zeros(NE * NE, np.int32)
        Iw, noIw, Tdown = np.zeros(NE * NE * NFREQ, np.float32), np.zeros(NE - 1, np.int32), np.zeros(NE, np.float32)
        wrk = np.zeros(NE * NFREQ + NE * (NFREQ + 4), np.float32)
        self.lib.ref_pre_weights(GLOBAL, NFREQ, NE, _fp(Ef), _fp(E), _ip(L1), _ip(L2), _fp(Iw), _fp(wrk), _ip(noIw))
        self.lib.ref_pre_tdown(GLOBAL, NFREQ, _fp(FREQ), _fp(Ef), _fp(SKABS), NE, _fp(E), _fp(T), _fp(Tdown))
        packed = np.concatenate([Iw[l * NE * NFREQ:l * NE * NFREQ + noIw[l]] for l in range(NE - 1)])
        L1[0] = -2
        L2[0] = -2
        return dict(Iw=packed, L1=L1, L2=L2, Tdown=Tdown, noIw=noIw)


class RefA2E:
    """x86 build of kernel_A2E.c for one (NE, NFREQ, LOCAL, CELLS, NIP): oracle/_ref/refa2e_<tag>.so"""

    def __init__(self, tag):
        self.model = _build.a2e_ref_models()[tag]
        path = _build.build_ref_a2e(tag, **self.model)
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("reference build a2e %s not available" % tag)
        self.lib = C.CDLL(path)
        self.lib.ref_dosolve.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _F, _I, _I, _F, _F, _I, _F, _F, _F, _F]
        self.lib.ref_eqtemp.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _F, _F, _F, _F, _F, _F]

    @staticmethod
    def available(tag):
        try:
            RefA2E(tag)
            return True
        except (FileNotFoundError, OSError, KeyError):
            return False

    def dosolve(self, size, AF, AABS, GLOBAL=None):
        m = self.model
        NE, NFREQ, LOCAL = m["NE"], m["NFREQ"], m["LOCAL"]
        AABS = np.ascontiguousarray(AABS, np.float32)
        batch = AABS.shape[0]
        GLOBAL = GLOBAL or ((batch + LOCAL - 1) // LOCAL) * LOCAL
        assert AABS.shape[1] == NFREQ and GLOBAL % LOCAL == 0
        absb = np.zeros((GLOBAL, NFREQ), np.float32)
        absb[:batch] = AABS
        out = np.zeros((GLOBAL, NFREQ), np.float32)
        LL = np.zeros(GLOBAL * ((NE * NE - NE) // 2), np.float32)
        arrs = [np.ascontiguousarray(size[k], t) for k, t in (("Iw", np.float32), ("L1", np.int32), ("L2", np.int32),
                                                               ("Tdown", np.float32), ("EA", np.float32), ("Ibeg", np.int32))]
        AF = np.ascontiguousarray(AF, np.float32)
        self.lib.ref_dosolve(GLOBAL, LOCAL, batch, 0, _fp(arrs[0]), _ip(arrs[1]), _ip(arrs[2]), _fp(arrs[3]),
                             _fp(arrs[4]), _ip(arrs[5]), _fp(AF), _fp(absb), _fp(out), _fp(LL))
        return out[:batch]

    def eqtemp(self, icell, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS):
        m = self.model
        ABS = np.ascontiguousarray(ABS, np.float32)
        batch, NFREQ = ABS.shape
        assert NFREQ == m["NFREQ"] and len(TTT) == m["NIP"]
        T = np.zeros(batch, np.float32)
        E = np.zeros((batch, NFREQ), np.float32)
        FREQ, KABS, TTT = (np.ascontiguousarray(a, np.float32) for a in (FREQ, KABS, TTT))
        self.lib.ref_eqtemp(batch, icell, np.float32(kE), np.float32(oplgkE), np.float32(Emin), _fp(FREQ), _fp(KABS),
                            _fp(TTT), _fp(ABS), _fp(T), _fp(E))
        return T, E


# ---------------------------------------------------------------------------------------
# scattered light (kernel_ASOC_sca.c): oracle + x86 builds of the reference
# ---------------------------------------------------------------------------------------

class ScaView:
    """Observer set-up of one scattered-light launch: directions, image size, pixel, centre."""

    def __init__(self, ODIR, RA, DE, NPIX=(16, 16), MAP_DX=1.0, CENTRE=(4.0, 4.0, 4.0), FFS=1, nside=0):
        """nside > 0: Healpix map of that NSIDE seen from the position ODIR[0] (NDIR = -nside in the kernels)"""
        self.ODIR, self.RA, self.DE = (np.ascontiguousarray(a, np.float32).reshape(-1, 4) for a in (ODIR, RA, DE))
        self.NDIR = -int(nside) if nside else len(self.ODIR)
        self.nside = int(nside)
        self.NPIX = (int(NPIX[0]), int(NPIX[1]))
        self.MAP_DX = np.float32(MAP_DX)
        self.CENTRE = tuple(np.float32(c) for c in CENTRE)
        self.FFS = int(FFS)

    def out_size(self):
        return 12 * self.nside * self.nside if self.nside else self.NDIR * self.NPIX[0] * self.NPIX[1]


class SArgs(C.Structure):
    _fields_ = [("SOURCE", C.c_int), ("PACKETS", C.c_int), ("BATCH", C.c_int), ("GLOBAL", C.c_int), ("NDIR", C.c_int),
                ("NPIX_X", C.c_int), ("NPIX_Y", C.c_int),
                ("SEED", C.c_float), ("BG", C.c_float), ("MAP_DX", C.c_float), ("CX", C.c_float), ("CY", C.c_float), ("CZ", C.c_float),
                ("ABS", _F), ("SCA", _F), ("PSPOS", _F), ("PS", _F), ("LCELLS", _I), ("OFF", _I), ("PAR", _I),
                ("DENS", _F), ("EMIT", _F), ("DSC", _F), ("CSC", _F), ("ODIRS", _F), ("ORA", _F), ("ODE", _F), ("OUT", _F),
                ("OPT", _F), ("EMWEI", _F), ("XPS_NSIDE", _I), ("XPS_SIDE", _I), ("XPS_AREA", _F), ("HPBG", _F), ("HPBGP", _F), ("ABU", _F)]


def oracle_sim_sca(orc, job, view, kind=0, gid0=0, gid1=None, nthreads=1, stride=1, OUT=None):
    """kind 0 = SimRAM_PB, 1 = SimRAM_CL, 2 = SimRAM_PS, 3 = SimRAM_HP (sca versions).  Returns (OUT, contributions)."""
    L = orc.lib
    L.orc_sim_sca.restype = C.c_long
    L.orc_sim_sca.argtypes = [C.POINTER(OrcModel), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    m = orc._model(job)
    OUT = np.zeros(view.out_size(), np.float32) if OUT is None else OUT
    m.NDIR, m.NPIX_X, m.NPIX_Y, m.FFS = view.NDIR, view.NPIX[0], view.NPIX[1], view.FFS
    m.MAP_DX, m.CX, m.CY, m.CZ = view.MAP_DX, view.CENTRE[0], view.CENTRE[1], view.CENTRE[2]
    m.ODIRS, m.ORA, m.ODE, m.DSC, m.OUT = _fp(view.ODIR), _fp(view.RA), _fp(view.DE), _fp(job.DSC), _fp(OUT)
    m.XPS_AS_FLOAT = 1
    gid1 = job.GLOBAL if gid1 is None else gid1
    n = L.orc_sim_sca(C.byref(m), kind, gid0, gid1, stride, nthreads)
    return OUT, int(n)


class RefSca:
    """x86 build of kernel_ASOC_sca.c for one model: oracle/_ref/refsca_<tag>.so"""

    def __init__(self, tag):
        self.model = _build.sca_ref_models()[tag]
        path = _build.build_ref_sca(tag, **self.model)
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("reference build sca %s not available" % tag)
        self.lib = C.CDLL(path)
        self.lib.ref_sca_sim.argtypes = [C.POINTER(SArgs), C.c_int, C.c_int, C.c_int, C.c_int]
        self.lib.ref_sca_parents.argtypes = [_F, _I, _I, _I]

    @staticmethod
    def available(tag):
        try:
            RefSca(tag)
            return True
        except (FileNotFoundError, OSError, KeyError):
            return False

    def sim(self, job, view, kind=0, gid0=0, gid1=None, stride=1, OUT=None):
        m, cl = self.model, job.cloud
        assert (cl.NX, cl.NY, cl.NZ, cl.LEVELS, cl.CELLS) == (m["NX"], m["NY"], m["NZ"], m["LEVELS"], m["CELLS"])
        assert job.BINS == m.get("BINS", 2500) and view.FFS == m.get("FFS", 1)
        assert int(job.OPT is not None) == m.get("WITH_ABU", 0) and job.USE_EMWEIGHT == m.get("USE_EMWEIGHT", 0)
        assert max(1, job.NO_PS) == max(1, m.get("NO_PS", 1)) and job.PS_METHOD == m.get("PS_METHOD", 0)
        assert job.MIRROR == m.get("MIRROR", 0)
        assert (0 if job.MSF is None else len(job.MSF[1])) == (m.get("NDUST", 1) if m.get("WITH_MSF", 0) else 0)
        PAR = np.zeros(max(1, cl.CELLS - cl.NX * cl.NY * cl.NZ), np.int32)
        self.lib.ref_sca_parents(_fp(job.DENS), _ip(job.LCELLS), _ip(job.OFF), _ip(PAR))
        OUT = np.zeros(view.out_size(), np.float32) if OUT is None else OUT
        ABS = np.asarray([job.ABS], np.float32)
        SCA = np.asarray([job.SCA], np.float32)
        a = SArgs()
        a.SOURCE, a.PACKETS, a.BATCH, a.GLOBAL, a.NDIR = job.SOURCE, job.PACKETS, job.BATCH, job.GLOBAL, view.NDIR
        a.NPIX_X, a.NPIX_Y = view.NPIX
        a.SEED, a.BG, a.MAP_DX = job.SEED, job.BG, view.MAP_DX
        a.CX, a.CY, a.CZ = view.CENTRE
        a.ABS, a.SCA, a.PSPOS, a.PS = _fp(ABS), _fp(SCA), _fp(job.PSPOS), _fp(job.PS)
        a.LCELLS, a.OFF, a.PAR = _ip(job.LCELLS), _ip(job.OFF), _ip(PAR)
        a.DENS, a.EMIT, a.DSC, a.CSC = _fp(job.DENS), _fp(job.EMIT), _fp(job.DSC), _fp(job.CSC)
        a.ODIRS, a.ORA, a.ODE, a.OUT = _fp(view.ODIR), _fp(view.RA), _fp(view.DE), _fp(OUT)
        a.OPT = _fp(job.OPT) if job.OPT is not None else None
        a.EMWEI = _fp(job.EMWEI)
        a.XPS_NSIDE, a.XPS_SIDE, a.XPS_AREA = _ip(job.XPS_NSIDE), _ip(job.XPS_SIDE), _fp(job.XPS_AREA)
        a.HPBG, a.HPBGP = _fp(job.HPBG), _fp(job.HPBGP)
        assert int(job.HPBGP is not None) == m.get("HPBG_WEIGHTED", 0)
        if job.MSF is not None:                      # -D WITH_MSF: ABS[NDUST], SCA[NDUST], DSC/CSC[NDUST*BINS], ABU[CELLS*NDUST]
            a.ABS, a.SCA, a.CSC, a.ABU = _fp(job.MSF[0]), _fp(job.MSF[1]), _fp(job.MSF[2]), _fp(job.MSF[3])
            assert job.DSC.size == job.MSF[2].size
        gid1 = job.GLOBAL if gid1 is None else gid1
        self.lib.ref_sca_sim(C.byref(a), kind, gid0, gid1, stride)
        return OUT


# ---------------------------------------------------------------------------------------
# map making (kernel_ASOC_map.c)
# ---------------------------------------------------------------------------------------

class MArgs(C.Structure):
    _fields_ = [("NPIX_X", C.c_int), ("NPIX_Y", C.c_int), ("SAVE_COLDEN", C.c_int), ("healpix", C.c_int),
                ("MAP_DX", C.c_float), ("ABS", C.c_float), ("SCA", C.c_float),
                ("DIR", C.c_float * 4), ("RA", C.c_float * 4), ("DE", C.c_float * 4), ("CENTRE", C.c_float * 4),
                ("INTOBS", C.c_float * 4),
                ("LCELLS", _I), ("OFF", _I), ("PAR", _I),
                ("DENS", _F), ("EMIT", _F), ("OPT", _F), ("MAP", _F), ("SAVETAU", _F), ("ROI", _I)]


NO_INTOBS = (-1.0e12, 0.0, 0.0)


def oracle_mapping(orc, job, EMIT, DIR, RA, DE, NPIX, MAP_DX, CENTRE, INTOBS=NO_INTOBS, save_colden=0, LENGTH=1.0, healpix=0):
    """Mapping (healpix=0: NPIX = (x, y)) or HealpixMapping (healpix=NSIDE).  Returns (MAP, SAVETAU)."""
    L = orc.lib
    L.orc_mapping.argtypes = [C.POINTER(OrcModel), C.c_int, C.c_float, C.c_int, C.c_int, _F, _F, _F, _F, _F, _F, C.c_int, C.c_float, _F, _F]
    m = orc._model(job)
    nx, ny = (healpix, 1) if healpix else (int(NPIX[0]), int(NPIX[1]))
    n = 12 * healpix * healpix if healpix else nx * ny
    MAP, TAU = np.zeros(n, np.float32), np.zeros(n, np.float32)
    v = [np.ascontiguousarray(np.asarray(a, np.float32).ravel()[:3]) for a in (DIR, RA, DE, CENTRE, INTOBS)]
    EMIT = np.ascontiguousarray(EMIT, np.float32)
    L.orc_mapping(C.byref(m), int(bool(healpix)), np.float32(MAP_DX), nx, ny, _fp(EMIT), _fp(v[0]), _fp(v[1]), _fp(v[2]), _fp(v[3]),
                  _fp(v[4]), int(save_colden), np.float32(LENGTH), _fp(MAP), _fp(TAU))
    return MAP, TAU


def oracle_pstau(orc, job, PSPOS, DIR, LENGTH=1.0):
    """PSTau: (column density * LENGTH, optical depth) from every point source towards the observer direction DIR"""
    L = orc.lib
    L.orc_pstau.argtypes = [C.POINTER(OrcModel), C.c_int, _F, _F, C.c_float, _F, _F]
    m = orc._model(job)
    P = np.zeros((len(PSPOS), 4), np.float32)
    P[:, :3] = np.asarray(PSPOS, np.float32)[:, :3]
    d = np.ascontiguousarray(np.asarray(DIR, np.float32).ravel()[:3])
    col, tau = np.zeros(len(P), np.float32), np.zeros(len(P), np.float32)
    L.orc_pstau(C.byref(m), len(P), _fp(P), _fp(d), np.float32(LENGTH), _fp(col), _fp(tau))
    return col, tau


class RefMap:
    """x86 build of kernel_ASOC_map.c for one model: oracle/_ref/refmap_<tag>.so"""

    def __init__(self, tag, NSIDE=8):
        self.model = _build.map_ref_models()[tag]
        self.tag = tag if NSIDE == 8 else "%s_n%d" % (tag, NSIDE)
        path = _build.build_ref_map(self.tag, NSIDE=NSIDE, **self.model)
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("reference build map %s not available" % tag)
        self.lib = C.CDLL(path)
        self.lib.ref_map.argtypes = [C.POINTER(MArgs), C.c_int]
        self.NSIDE = NSIDE

    @staticmethod
    def available(tag):
        try:
            RefMap(tag)
            return True
        except (FileNotFoundError, OSError, KeyError):
            return False

    def pstau(self, job, PAR, PSPOS, DIR):
        """the build's -D LENGTH applies to the column density"""
        a = MArgs()
        for name, val in (("DIR", DIR), ("RA", DIR), ("DE", DIR)):
            v = np.zeros(4, np.float32)
            v[:3] = np.asarray(val, np.float32).ravel()[:3]
            setattr(a, name, (C.c_float * 4)(*v))
        a.ABS, a.SCA = np.float32(job.ABS), np.float32(job.SCA)
        a.LCELLS, a.OFF, a.PAR, a.DENS = _ip(job.LCELLS), _ip(job.OFF), _ip(PAR), _fp(job.DENS)
        a.OPT = _fp(job.OPT) if job.OPT is not None else None
        P = np.zeros((len(PSPOS), 4), np.float32)
        P[:, :3] = np.asarray(PSPOS, np.float32)[:, :3]
        col, tau = np.zeros(len(P), np.float32), np.zeros(len(P), np.float32)
        self.lib.ref_pstau.argtypes = [C.POINTER(MArgs), C.c_int, _F, _F, _F]
        self.lib.ref_pstau(C.byref(a), len(P), _fp(P), _fp(col), _fp(tau))
        return col, tau

    def mapping(self, job, PAR, EMIT, DIR, RA, DE, NPIX, MAP_DX, CENTRE, INTOBS=NO_INTOBS, save_colden=0, healpix=0):
        a = MArgs()
        nx, ny = (healpix, 1) if healpix else (int(NPIX[0]), int(NPIX[1]))
        n = 12 * healpix * healpix if healpix else nx * ny
        assert (not healpix) or healpix == self.NSIDE
        MAP, TAU = np.zeros(n, np.float32), np.zeros(n, np.float32)
        EMIT = np.ascontiguousarray(EMIT, np.float32)
        a.NPIX_X, a.NPIX_Y, a.SAVE_COLDEN, a.healpix = nx, ny, int(save_colden), int(bool(healpix))
        a.MAP_DX, a.ABS, a.SCA = np.float32(MAP_DX), job.ABS, job.SCA
        for name, val in (("DIR", DIR), ("RA", RA), ("DE", DE), ("CENTRE", CENTRE), ("INTOBS", INTOBS)):
            arr = getattr(a, name)
            for k in range(3):
                arr[k] = float(np.float32(np.asarray(val).ravel()[k]))
        a.LCELLS, a.OFF, a.PAR = _ip(job.LCELLS), _ip(job.OFF), _ip(PAR)
        a.DENS, a.EMIT, a.MAP, a.SAVETAU = _fp(job.DENS), _fp(EMIT), _fp(MAP), _fp(TAU)
        a.OPT = _fp(job.OPT) if job.OPT is not None else None
        roi = getattr(job, "ROI_MAP", None)
        assert (roi is not None) == bool(self.model.get("ROI_MAP", 0))
        if roi is not None:
            roi = np.ascontiguousarray(roi, np.int32)
            a.ROI = _ip(roi)
        self.lib.ref_map(C.byref(a), n)
        return MAP, TAU
