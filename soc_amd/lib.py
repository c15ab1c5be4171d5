"""ctypes binding of libsoc_hip.so (include/soc_hip.h) and a thin ``Engine`` object.

The engine is the HIP library and nothing else: if the library cannot be loaded or no GPU
is present the constructor raises -- there is no CPU fallback in the product path.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBNAME = os.path.join(HERE, "libsoc_hip.so")

TALLY_TABS = 0
TALLY_INT = 1
TALLY_XAB = 2
TALLY_INTX, TALLY_INTY, TALLY_INTZ = 3, 4, 5      # with_int == 2 (SAVE_INTENSITY 2)

_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)
_U = C.POINTER(C.c_uint32)

# every symbol include/soc_hip.h declares: (restype, argtypes)
API = {
    "soc_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "soc_destroy": (None, [C.c_void_p]),
    "soc_last_error": (C.c_char_p, [C.c_void_p]),
    "soc_version": (C.c_char_p, []),
    "soc_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "soc_set_grid": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _I, _F]),
    "soc_set_features": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "soc_set_exec": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "soc_last_form": (C.c_int, [C.c_void_p]),
    "soc_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "soc_last_passes": (C.c_int, [C.c_void_p]),
    "soc_set_optical": (C.c_int, [C.c_void_p, _F, _F, C.c_int]),
    "soc_set_opt": (C.c_int, [C.c_void_p, _F]),
    "soc_set_scatter_table": (C.c_int, [C.c_void_p, _F, _F, C.c_int]),
    "soc_set_opt_half": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_set_scatter_tables": (C.c_int, [C.c_void_p, C.c_int, _F, _F, C.c_int]),
    "soc_set_step_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float]),
    "soc_set_emission": (C.c_int, [C.c_void_p, _F, _F]),
    "soc_set_emindex": (C.c_int, [C.c_void_p, _I]),
    "soc_set_ali": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_zero": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_sim_pb": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                             _F, _F, C.c_int, _I, _I, _F, C.c_int, C.c_int, C.c_int]),
    "soc_sim_cl": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                             C.c_int, C.c_int, C.c_int]),
    "soc_batch_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_batch_end": (C.c_int, [C.c_void_p]),
    "soc_batch_begin_int": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_batch_begin_shared_int": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_batch_begin_int_groups": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_batch_next_int": (C.c_int, [C.c_void_p]),
    "soc_batch_read_int": (C.c_int, [C.c_void_p, C.c_int, _F, C.c_long]),
    "soc_set_mirror": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_set_hpbg": (C.c_int, [C.c_void_p, _F, _F]),
    "soc_set_abundances": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _F]),
    "soc_set_optical_abu": (C.c_int, [C.c_void_p, _F, _F, C.c_int]),
    "soc_read_opt": (C.c_int, [C.c_void_p, _F]),
    "soc_set_roi_save": (C.c_int, [C.c_void_p, _I, C.c_int, C.c_int]),
    "soc_roi_zero": (C.c_int, [C.c_void_p]),
    "soc_roi_read": (C.c_int, [C.c_void_p, _F, C.c_long]),
    "soc_set_roi_load": (C.c_int, [C.c_void_p, _I, C.c_int, _F]),
    "soc_sim_hp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]),
    "soc_sca_set_view": (C.c_int, [C.c_void_p, C.c_int, _F, _F, _F, C.c_int, C.c_int, C.c_float, _F, C.c_int]),
    "soc_sca_set_healpix": (C.c_int, [C.c_void_p, C.c_int, _F, C.c_int]),
    "soc_sca_sim_hp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]),
    "soc_sca_zero": (C.c_int, [C.c_void_p]),
    "soc_sca_sim_ps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, _F, _F, C.c_int, _I, _I, _F,
                                 C.c_int, C.c_int, C.c_int]),
    "soc_sca_sim_pb": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _F, _F, C.c_int,
                                 _I, _I, _F, C.c_int, C.c_int, C.c_int]),
    "soc_sca_sim_cl": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]),
    "soc_sca_read_out": (C.c_int, [C.c_void_p, _F, C.c_int64]),
    "soc_sca_out_ptr": (C.c_void_p, [C.c_void_p]),
    "soc_sca_batch_images": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_sca_batch_select": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_sca_batch_read": (C.c_int, [C.c_void_p, C.c_int, _F, C.c_int64]),
    "soc_sca_bind_out": (C.c_int, [C.c_void_p, C.c_void_p]),
    "soc_sync": (C.c_int, [C.c_void_p]),
    "soc_read_tally": (C.c_int, [C.c_void_p, C.c_int, _F, C.c_int64]),
    "soc_write_tally": (C.c_int, [C.c_void_p, C.c_int, _F, C.c_int64]),
    "soc_tally_ptr": (C.c_void_p, [C.c_void_p, C.c_int]),
    "soc_bind_tally": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "soc_read_par": (C.c_int, [C.c_void_p, _I, C.c_int64]),
    "soc_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    "soc_sca_ray_steps": (C.c_int64, [C.c_void_p]),
    "soc_timer_start": (C.c_int, [C.c_void_p]),
    "soc_timer_stop": (C.c_int, [C.c_void_p, _F]),
    "soc_solve_temperature": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, _F, C.c_float, C.c_float, _F, _F]),
    "soc_set_cr_heating": (C.c_int, [C.c_void_p, C.c_float]),
    "soc_set_map_threshold": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_set_map_interpolation": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_a2e_pre": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, _F, _F, _F, _F, _F, _I, _I, _F, _I, _F]),
    "soc_set_map_roi": (C.c_int, [C.c_void_p, _I]),
    "soc_set_temperature": (C.c_int, [C.c_void_p, _F]),
    "soc_emission": (C.c_int, [C.c_void_p, C.c_int, _F, _F, C.c_float, C.c_float, _F]),
    "soc_map": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, _F, _F, _F, _F, _F, _F, C.c_float, C.c_float, C.c_int,
                          C.c_float, _F, _F]),
    "soc_ps_tau": (C.c_int, [C.c_void_p, C.c_int, _F, _F, C.c_float, C.c_float, C.c_float, _F, _F]),
    "soc_a2e_set_size": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _F, _I, _I, _F, _F, _I, _F]),
    "soc_a2e_solve": (C.c_int, [C.c_void_p, C.c_int, _F, _F]),
    "soc_a2e_upload": (C.c_int, [C.c_void_p, C.c_int, _F]),
    "soc_a2e_run": (C.c_int, [C.c_void_p, C.c_int]),
    "soc_a2e_download": (C.c_int, [C.c_void_p, C.c_int, _F]),
    "soc_a2e_resident_begin": (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    "soc_a2e_resident_upload": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _F]),
    "soc_a2e_resident_solve": (C.c_int, [C.c_void_p]),
    "soc_a2e_resident_download": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _F]),
    "soc_a2e_resident_end": (C.c_int, [C.c_void_p]),
    "soc_a2e_eqtemp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                 C.c_float, C.c_float, _F, _F, _F, _F, _F, _F]),
    "soc_eqsolver": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                               C.c_float, C.c_float, _F, _F, _F, _F, _F, _F]),
    "soc_probe_rng": (C.c_int, [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_int, _U, _U]),
    "soc_probe_math": (C.c_int, [C.c_void_p, C.c_int, _F, _F, C.c_int64]),
    "soc_probe_trace": (C.c_int, [C.c_void_p, _F, _F, C.c_int, _I, _I, _F, _F, _I]),
}

_lib = None


class SocError(RuntimeError):
    pass


def _torch_runtime_first():
    """torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, without a soname), libsoc_hip.so is linked against the
    system's (libamdhip64.so.7): a process that uses both holds two runtimes, and the one initialised second finds no device.
    With torch's runtime loaded first both work -- so where torch is installed it is imported before libsoc_hip.so is opened,
    whatever the order of the caller's own imports.  (Without torch there is one runtime and nothing to do.)"""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch                                   # noqa: F401
    except Exception:                                      # a broken torch installation must not stop a run that does not need it
        pass


def load_library(path=None):
    """dlopen libsoc_hip.so and declare every prototype.  Raises SocError if it is missing.
    Safe to call before or after `import torch` (see _torch_runtime_first)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    _torch_runtime_first()
    path = path or LIBNAME
    if not os.path.exists(path):
        raise SocError("%s not found: build it with `python -m soc_amd.build` "
                       "(there is no CPU fallback)" % path)
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise SocError("cannot load %s: %s" % (path, e))
    for name, (res, args) in API.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f(a):
    return None if a is None else a.ctypes.data_as(_F)


def _i(a):
    return None if a is None else a.ctypes.data_as(_I)


class Engine:
    """One GPU's packet engine.  Method names follow the C ABI one to one."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.soc_create(int(device), C.byref(h))
        if rc != 0:
            raise SocError("soc_create(device=%d) failed: %s" %
                           (device, self.lib.soc_last_error(None).decode()))
        self.h = h
        self.device = int(device)
        self.CELLS = 0
        self.NPAR = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.soc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SocError("%s (code %d)" % (self.lib.soc_last_error(self.h).decode(), rc))

    # ---- model ----
    def set_stream(self, stream_ptr):
        self._chk(self.lib.soc_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_grid(self, NX, NY, NZ, LEVELS, LCELLS, DENS):
        LCELLS = np.ascontiguousarray(LCELLS, np.int32)
        DENS = np.ascontiguousarray(DENS, np.float32)
        if len(LCELLS) < LEVELS or DENS.size != int(LCELLS[:LEVELS].sum()):
            raise SocError("set_grid: DENS has %d values, LCELLS sums to %d" % (DENS.size, int(LCELLS[:LEVELS].sum())))
        self._chk(self.lib.soc_set_grid(self.h, int(NX), int(NY), int(NZ), int(LEVELS), _i(LCELLS), _f(DENS)))
        self.CELLS = int(DENS.size)
        self.NPAR = self.CELLS - int(NX) * int(NY) * int(NZ)

    def set_cloud(self, cloud):
        self.set_grid(cloud.NX, cloud.NY, cloud.NZ, cloud.LEVELS, cloud.LCELLS, cloud.DENS)

    def set_features(self, with_int=0, ps_method=0, use_emweight=0):
        self._chk(self.lib.soc_set_features(self.h, int(with_int), int(ps_method), int(use_emweight)))

    def batch_begin(self, max_launches=0):
        """Defer the following sim_pb launches and run them together (brick sweep, TABS only)."""
        self._chk(self.lib.soc_batch_begin(self.h, int(max_launches)))

    def batch_begin_int(self, max_launches=0):
        """like batch_begin, for launches that keep the per-frequency INT tally: each deferred launch gets its own;
        read them with batch_read_int(k) after batch_end"""
        self._chk(self.lib.soc_batch_begin_int(self.h, int(max_launches)))

    def batch_begin_shared_int(self, max_launches=0):
        """Launches until batch_end() that keep the INT tally are deferred too and tally into the handle's INT buffer together:
        the source blocks of one frequency (zero(1) before, read_tally(1) after)."""
        self._chk(self.lib.soc_batch_begin_shared_int(self.h, int(max_launches)))

    def batch_begin_int_groups(self, max_groups=0):
        """As batch_begin_int, but the launches between two batch_next_int() calls -- the source blocks of one frequency -- share an
        INT tally; batch_read_int(k) reads the k-th group's after batch_end()."""
        self._chk(self.lib.soc_batch_begin_int_groups(self.h, int(max_groups)))

    def batch_next_int(self):
        self._chk(self.lib.soc_batch_next_int(self.h))

    def batch_read_int(self, k):
        out = np.zeros(self.CELLS, np.float32)
        self._chk(self.lib.soc_batch_read_int(self.h, int(k), _f(out), out.size))
        return out

    def batch_end(self):
        self._chk(self.lib.soc_batch_end(self.h))

    def set_mirror(self, mask=0):
        """reflecting faces, bits x,X,y,Y,z,Z = 1,2,4,8,16,32"""
        self._chk(self.lib.soc_set_mirror(self.h, int(mask)))

    def set_exec(self, mode=-1, brick_log2=4):
        """0 direct kernel, 1 brick sweep (LDS tallies), -1 automatic."""
        self._chk(self.lib.soc_set_exec(self.h, int(mode), int(brick_log2)))

    def last_form(self):
        """0 direct kernel, 1 brick sweep (Cartesian), 2 hierarchy in global memory, 3 brick-local hierarchies"""
        return int(self.lib.soc_last_form(self.h))

    def set_tuning(self, **params):
        """shape of the brick sweep (soc_set_tuning): threads, chunk, steps_per_visit, swap_lanes, climb_lanes,
        brick_cells, tail_lanes, park_below, population, hash_slots, general_kernel, oversubscribe, verbose; 0 = built-in"""
        for k, v in params.items():
            self._chk(self.lib.soc_set_tuning(self.h, k.encode(), int(v)))

    def last_passes(self):
        return int(self.lib.soc_last_passes(self.h))

    def set_optical(self, ABS, SCA):
        a = np.asarray([ABS], np.float32).ravel()
        s = np.asarray([SCA], np.float32).ravel()
        self._chk(self.lib.soc_set_optical(self.h, _f(a), _f(s), 1))

    def set_opt(self, OPT):
        if OPT is None:
            self._chk(self.lib.soc_set_opt(self.h, None))
            return
        OPT = np.ascontiguousarray(OPT, np.float32)
        if OPT.size != 2 * self.CELLS:
            raise SocError("set_opt: OPT must hold 2*CELLS floats")
        self._chk(self.lib.soc_set_opt(self.h, _f(OPT)))

    def set_abundances(self, ABU, single=False):
        """abundances once per run: ABU[CELLS, NDUST], or ABU[CELLS] with single=True (two species, ABU and 1-ABU);
        None forgets them.  set_optical_abu then builds OPT on the device for every frequency."""
        if ABU is None:
            self._chk(self.lib.soc_set_abundances(self.h, 0, 0, None))
            return
        ABU = np.ascontiguousarray(ABU, np.float32)
        ndust = 2 if single else (ABU.shape[1] if ABU.ndim == 2 else 1)
        if ABU.size != self.CELLS * (1 if single else ndust):
            raise SocError("set_abundances: ABU must hold CELLS x NDUST values")
        self._chk(self.lib.soc_set_abundances(self.h, int(ndust), int(bool(single)), _f(ABU)))

    def set_optical_abu(self, AFABS, AFSCA):
        a = np.ascontiguousarray(AFABS, np.float32).ravel()
        s = np.ascontiguousarray(AFSCA, np.float32).ravel()
        self._chk(self.lib.soc_set_optical_abu(self.h, _f(a), _f(s), int(a.size)))

    def set_cr_heating(self, rate):
        """-D CR_HEATING_RATE (with -D CR_HEATING=1): added to the absorbed energy in solve_temperature; 0 = off"""
        self._chk(self.lib.soc_set_cr_heating(self.h, float(rate)))

    def set_map_interpolation(self, mode):
        """-D MAP_INTERPOLATION (ini key mapint): 1 | 2 = flat maps blend every cell on the ray with two neighbours; 0 = off"""
        self._chk(self.lib.soc_set_map_interpolation(self.h, int(mode)))

    def set_map_threshold(self, level):
        """-D LEVEL_THRESHOLD: flat maps leave out the emission of coarser levels; 0 = off"""
        self._chk(self.lib.soc_set_map_threshold(self.h, int(level)))

    def set_map_roi(self, ROI):
        """-D ROI_MAP: maps of the emission inside ROI = [x0,x1,y0,y1,z0,z1] only; None = all cells"""
        r = None if ROI is None else np.ascontiguousarray(ROI, np.int32)
        self._chk(self.lib.soc_set_map_roi(self.h, _i(r)))

    def set_opt_half(self, on=True):
        """-D OPT_IS_HALF: OPT of later set_opt / set_optical_abu calls is rounded to fp16 as the reference stores it"""
        self._chk(self.lib.soc_set_opt_half(self.h, int(bool(on))))

    def read_opt(self):
        out = np.zeros((self.CELLS, 2), np.float32)
        self._chk(self.lib.soc_read_opt(self.h, _f(out)))
        return out

    def set_scatter_table(self, DSC, CSC):
        CSC = np.ascontiguousarray(CSC, np.float32)
        DSC = None if DSC is None else np.ascontiguousarray(DSC, np.float32)
        self._chk(self.lib.soc_set_scatter_table(self.h, _f(DSC), _f(CSC), int(CSC.size)))

    def set_scatter_tables(self, DSC, CSC):
        """-D WITH_MSF: CSC[NDUST, BINS] (and DSC) per dust species; needs set_abundances + set_optical_abu per frequency"""
        CSC = np.ascontiguousarray(CSC, np.float32)
        if CSC.ndim != 2:
            raise SocError("set_scatter_tables: CSC[NDUST, BINS]")
        DSC = None if DSC is None else np.ascontiguousarray(DSC, np.float32)
        if DSC is not None and DSC.shape != CSC.shape:
            raise SocError("set_scatter_tables: DSC and CSC differ in shape")
        self._chk(self.lib.soc_set_scatter_tables(self.h, int(CSC.shape[0]), _f(DSC), _f(CSC), int(CSC.shape[1])))

    def set_step_weight(self, mode, SW_A=0.0, SW_B=0.0):
        """values of -D STEP_WEIGHT, -D SW_A, -D SW_B (kernel_ASOC.c:516-535); mode 0 = off"""
        self._chk(self.lib.soc_set_step_weight(self.h, int(mode), float(SW_A), float(SW_B)))

    def set_emission(self, EMIT, EMWEI=None):
        EMIT = np.ascontiguousarray(EMIT, np.float32)
        EMWEI = None if EMWEI is None else np.ascontiguousarray(EMWEI, np.float32)
        if EMIT.size != self.CELLS or (EMWEI is not None and EMWEI.size != self.CELLS):
            raise SocError("set_emission: arrays must hold CELLS floats")
        self._chk(self.lib.soc_set_emission(self.h, _f(EMIT), _f(EMWEI)))

    def set_emindex(self, EMINDEX):
        EMINDEX = np.ascontiguousarray(EMINDEX, np.int32)
        if EMINDEX.size != self.CELLS:
            raise SocError("set_emindex: EMINDEX must hold CELLS ints")
        self._chk(self.lib.soc_set_emindex(self.h, _i(EMINDEX)))

    def set_ali(self, with_ali=1):
        self._chk(self.lib.soc_set_ali(self.h, int(with_ali)))

    # ---- launches ----
    def zero(self, tag):
        self._chk(self.lib.soc_zero(self.h, int(tag)))

    def sim_pb(self, SOURCE, PACKETS, BATCH, SEED, BG, TW, PSPOS=None, PS=None, XPS=None,
               GLOBAL=None, gid_first=0, gid_count=None):
        NO_PS = 0
        pspos = ps = nside = side = area = None
        if SOURCE == 0:
            p = np.asarray(PSPOS, np.float32)
            if p.ndim == 1:
                p = p.reshape(-1, 3)
            NO_PS = p.shape[0]
            pspos = np.zeros((NO_PS, 4), np.float32)
            pspos[:, :3] = p[:, :3]
            ps = np.ascontiguousarray(PS, np.float32)
            if XPS is not None:
                nside = np.ascontiguousarray(XPS[0], np.int32)
                side = np.ascontiguousarray(XPS[1], np.int32)
                area = np.ascontiguousarray(XPS[2], np.float32)
        gid_count = (GLOBAL - gid_first) if gid_count is None else gid_count
        self._chk(self.lib.soc_sim_pb(self.h, int(SOURCE), int(PACKETS), int(BATCH), np.float32(SEED),
                                      np.float32(BG), np.float32(TW), _f(pspos), _f(ps), NO_PS,
                                      _i(nside), _i(side), _f(area), int(GLOBAL), int(gid_first), int(gid_count)))

    def sim_cl(self, SOURCE, PACKETS, BATCH, SEED, TW, GLOBAL, gid_first=0, gid_count=None):
        gid_count = (GLOBAL - gid_first) if gid_count is None else gid_count
        self._chk(self.lib.soc_sim_cl(self.h, int(SOURCE), int(PACKETS), int(BATCH), np.float32(SEED),
                                      np.float32(TW), int(GLOBAL), int(gid_first), int(gid_count)))

    def set_hpbg(self, BG, HPBGP=None):
        """Healpix sky of the current frequency (49152 pixels, photons per package); HPBGP = cumulative
        pixel probability for weighted sampling."""
        BG = np.ascontiguousarray(BG, np.float32)
        HPBGP = None if HPBGP is None else np.ascontiguousarray(HPBGP, np.float32)
        if BG.size != 49152 or (HPBGP is not None and HPBGP.size != 49152):
            raise SocError("set_hpbg: the sky map must hold 49152 pixels (NSIDE 64)")
        self._chk(self.lib.soc_set_hpbg(self.h, _f(BG), _f(HPBGP)))

    # ---- region of interest (nested runs) ----
    def set_roi_save(self, ROI, ROI_STEP=1, ROI_NSIDE=16):
        """record the packets that step into ROI = [x0,x1,y0,y1,z0,z1] (root cells, inclusive) during sim_pb / sim_cl;
        ROI=None turns recording off.  Returns the number of record entries."""
        if ROI is None:
            self._chk(self.lib.soc_set_roi_save(self.h, None, 0, 0))
            self._roi_n = 0
            return 0
        ROI = np.ascontiguousarray(ROI, np.int32)
        if ROI.size != 6:
            raise SocError("set_roi_save: ROI = [x0, x1, y0, y1, z0, z1]")
        self._chk(self.lib.soc_set_roi_save(self.h, ROI.ctypes.data_as(_I), int(ROI_STEP), int(ROI_NSIDE)))
        n = [(int(ROI[2 * i + 1]) - int(ROI[2 * i]) + 1) * int(ROI_STEP) for i in range(3)]
        self._roi_n = (n[0] * n[1] + n[1] * n[2] + n[2] * n[0]) * 12 * int(ROI_NSIDE) ** 2
        return self._roi_n

    def roi_zero(self):
        self._chk(self.lib.soc_roi_zero(self.h))

    def roi_read(self):
        out = np.zeros(getattr(self, "_roi_n", 0), np.float32)
        self._chk(self.lib.soc_roi_read(self.h, _f(out), out.size))
        return out

    def set_roi_load(self, DIM, ROI_NSIDE, LOAD):
        """the record of one frequency to send in with sim_pb(SOURCE=3): LOAD[nelem, 12*NSIDE^2] photons on the
        (nx, ny, nz) = DIM surface discretisation; LOAD=None turns it off"""
        if LOAD is None:
            self._chk(self.lib.soc_set_roi_load(self.h, None, 0, None))
            return
        DIM = np.ascontiguousarray(DIM, np.int32)
        LOAD = np.ascontiguousarray(LOAD, np.float32)
        nelem = int(DIM[0]) * int(DIM[1]) + int(DIM[1]) * int(DIM[2]) + int(DIM[2]) * int(DIM[0])
        if DIM.size != 3 or LOAD.size != nelem * 12 * int(ROI_NSIDE) ** 2:
            raise SocError("set_roi_load: LOAD must hold %d x %d values" % (nelem, 12 * int(ROI_NSIDE) ** 2))
        self._chk(self.lib.soc_set_roi_load(self.h, DIM.ctypes.data_as(_I), int(ROI_NSIDE), _f(LOAD)))

    def sim_hp(self, PACKETS, BATCH, SEED, TW, GLOBAL, gid_first=0, gid_count=None):
        gid_count = (GLOBAL - gid_first) if gid_count is None else gid_count
        self._chk(self.lib.soc_sim_hp(self.h, int(PACKETS), int(BATCH), np.float32(SEED), np.float32(TW), int(GLOBAL),
                                      int(gid_first), int(gid_count)))

    # ---- equilibrium temperature and emission ----
    def solve_temperature(self, adhoc, kE, Emin, TTT, FACTOR, LENGTH, EABS):
        """EABS[CELLS]: integrated absorbed energy; returns TNEW[CELLS] (kept on the device for emission())"""
        TTT = np.ascontiguousarray(TTT, np.float32)
        EABS = np.ascontiguousarray(EABS, np.float32)
        if EABS.size != self.CELLS:
            raise SocError("solve_temperature: EABS must hold CELLS floats")
        T = np.zeros(self.CELLS, np.float32)
        self._chk(self.lib.soc_solve_temperature(self.h, np.float32(adhoc), np.float32(kE), np.float32(Emin), int(TTT.size), _f(TTT),
                                                 np.float32(FACTOR), np.float32(LENGTH), _f(EABS), _f(T)))
        return T

    def set_temperature(self, T):
        T = np.ascontiguousarray(T, np.float32)
        if T.size != self.CELLS:
            raise SocError("set_temperature: T must hold CELLS floats")
        self._chk(self.lib.soc_set_temperature(self.h, _f(T)))

    def emission(self, FREQ, FABS, FACTOR, LENGTH):
        """EMITTED[CELLS, nfreq] (FACTOR x photons/Hz/cm3) at the device temperatures"""
        FREQ = np.ascontiguousarray(FREQ, np.float32)
        FABS = np.ascontiguousarray(FABS, np.float32)
        out = np.zeros((self.CELLS, FREQ.size), np.float32)
        self._chk(self.lib.soc_emission(self.h, int(FREQ.size), _f(FREQ), _f(FABS), np.float32(FACTOR), np.float32(LENGTH), _f(out)))
        return out

    # ---- map making ----
    def map(self, EMIT, DIR, RA, DE, NPIX, MAP_DX, CENTRE, ABS, SCA, INTOBS=None, save_colden=0, LENGTH=1.0, healpix=0):
        """One map (kernel_ASOC_map.c Mapping / HealpixMapping).  Returns (MAP, SAVETAU): [NPIX.y, NPIX.x] or [12*NSIDE^2]."""
        EMIT = np.ascontiguousarray(EMIT, np.float32)
        if EMIT.size != self.CELLS:
            raise SocError("map: EMIT must hold CELLS floats")
        v = [None if a is None else np.ascontiguousarray(np.asarray(a, np.float32).ravel()[:3]) for a in (DIR, RA, DE, CENTRE, INTOBS)]
        nx, ny = (int(healpix), 1) if healpix else (int(NPIX[0]), int(NPIX[1]))
        shape = (12 * nx * nx,) if healpix else (ny, nx)
        MAP, TAU = np.zeros(shape, np.float32), np.zeros(shape, np.float32)
        self._chk(self.lib.soc_map(self.h, int(bool(healpix)), nx, ny, np.float32(MAP_DX), _f(EMIT), _f(v[0]), _f(v[1]), _f(v[2]),
                                   _f(v[3]), _f(v[4]), np.float32(ABS), np.float32(SCA), int(save_colden), np.float32(LENGTH),
                                   _f(MAP), _f(TAU)))
        return MAP, TAU

    def ps_tau(self, PSPOS, DIR, ABS, SCA, LENGTH=1.0):
        """PSTau: (column density * LENGTH, optical depth) from every point source towards the observer direction DIR"""
        P = np.zeros((len(PSPOS), 4), np.float32)
        P[:, :3] = np.asarray(PSPOS, np.float32)[:, :3]
        d = np.ascontiguousarray(np.asarray(DIR, np.float32).ravel()[:3])
        col, tau = np.zeros(len(P), np.float32), np.zeros(len(P), np.float32)
        self._chk(self.lib.soc_ps_tau(self.h, len(P), _f(P), _f(d), np.float32(ABS), np.float32(SCA), np.float32(LENGTH), _f(col), _f(tau)))
        return col, tau

    # ---- scattered-light images (ASOCS) ----
    @staticmethod
    def _sources(PSPOS, PS, XPS):
        p = np.asarray(PSPOS, np.float32)
        if p.ndim == 1:
            p = p.reshape(-1, p.size // max(1, np.asarray(PS).size))
        pspos = np.zeros((p.shape[0], 4), np.float32)
        pspos[:, :3] = p[:, :3]
        ps = np.ascontiguousarray(PS, np.float32)
        nside = side = area = None
        if XPS is not None:
            nside = np.ascontiguousarray(XPS[0], np.int32)
            side = np.ascontiguousarray(XPS[1], np.int32)
            area = np.ascontiguousarray(XPS[2], np.float32)
        return pspos, ps, nside, side, area

    def sca_set_view(self, ODIR, RA, DE, NPIX, MAP_DX, CENTRE, FFS=1):
        """ODIR, RA, DE: [NDIR,4] float32 (launch.set_observer_directions); NPIX = (x, y)."""
        ODIR, RA, DE = (np.ascontiguousarray(a, np.float32).reshape(-1, 4) for a in (ODIR, RA, DE))
        cen = np.asarray(CENTRE, np.float32).ravel()[:3].copy()
        self.sca_shape = (len(ODIR), int(NPIX[1]), int(NPIX[0]))
        self._chk(self.lib.soc_sca_set_view(self.h, len(ODIR), _f(ODIR), _f(RA), _f(DE), int(NPIX[0]), int(NPIX[1]),
                                            np.float32(MAP_DX), _f(cen), int(FFS)))

    def sca_set_healpix(self, NSIDE, OBSERVER, FFS=1):
        """one Healpix map (RING) of scattered light seen from the position OBSERVER [root-grid units]"""
        obs = np.asarray(OBSERVER, np.float32).ravel()[:3].copy()
        self.sca_shape = (12 * int(NSIDE) * int(NSIDE),)
        self._chk(self.lib.soc_sca_set_healpix(self.h, int(NSIDE), _f(obs), int(FFS)))

    def sca_sim_hp(self, PACKETS, BATCH, SEED, GLOBAL, gid_first=0, gid_count=None):
        gid_count = (GLOBAL - gid_first) if gid_count is None else gid_count
        self._chk(self.lib.soc_sca_sim_hp(self.h, int(PACKETS), int(BATCH), np.float32(SEED), int(GLOBAL), int(gid_first),
                                          int(gid_count)))

    def sca_zero(self):
        self._chk(self.lib.soc_sca_zero(self.h))

    def sca_sim_ps(self, PACKETS, BATCH, SEED, BG, PSPOS, PS, XPS=None, GLOBAL=None, gid_first=0, gid_count=None):
        pspos, ps, nside, side, area = self._sources(PSPOS, PS, XPS)
        gid_count = (GLOBAL - gid_first) if gid_count is None else gid_count
        self._chk(self.lib.soc_sca_sim_ps(self.h, int(PACKETS), int(BATCH), np.float32(SEED), np.float32(BG), _f(pspos), _f(ps),
                                          len(ps), _i(nside), _i(side), _f(area), int(GLOBAL), int(gid_first), int(gid_count)))

    def sca_sim_pb(self, SOURCE, PACKETS, BATCH, SEED, BG, PSPOS=None, PS=None, XPS=None, GLOBAL=None, gid_first=0,
                   gid_count=None):
        pspos = ps = nside = side = area = None
        NO_PS = 0
        if SOURCE == 0:
            pspos, ps, nside, side, area = self._sources(PSPOS, PS, XPS)
            NO_PS = len(ps)
        gid_count = (GLOBAL - gid_first) if gid_count is None else gid_count
        self._chk(self.lib.soc_sca_sim_pb(self.h, int(SOURCE), int(PACKETS), int(BATCH), np.float32(SEED), np.float32(BG),
                                          _f(pspos), _f(ps), NO_PS, _i(nside), _i(side), _f(area), int(GLOBAL),
                                          int(gid_first), int(gid_count)))

    def sca_sim_cl(self, SOURCE, PACKETS, BATCH, SEED, GLOBAL, gid_first=0, gid_count=None):
        gid_count = (GLOBAL - gid_first) if gid_count is None else gid_count
        self._chk(self.lib.soc_sca_sim_cl(self.h, int(SOURCE), int(PACKETS), int(BATCH), np.float32(SEED), int(GLOBAL),
                                          int(gid_first), int(gid_count)))

    def sca_read_out(self):
        out = np.zeros(self.sca_shape, np.float32)
        self._chk(self.lib.soc_sca_read_out(self.h, _f(out), out.size))
        return out

    def sca_out_ptr(self):
        return self.lib.soc_sca_out_ptr(self.h)

    def sca_batch_images(self, n):
        """n zeroed images for the scattered-light launches of a batch (batch_begin ... batch_end), 0 = the one image again"""
        self._chk(self.lib.soc_sca_batch_images(self.h, int(n)))

    def sca_batch_select(self, k):
        self._chk(self.lib.soc_sca_batch_select(self.h, int(k)))

    def sca_batch_read(self, k):
        out = np.zeros(self.sca_shape, np.float32)
        self._chk(self.lib.soc_sca_batch_read(self.h, int(k), _f(out), out.size))
        return out

    def sca_bind_out(self, device_ptr):
        self._chk(self.lib.soc_sca_bind_out(self.h, C.c_void_p(device_ptr) if device_ptr else None))

    def sync(self):
        self._chk(self.lib.soc_sync(self.h))

    # ---- results ----
    def read_tally(self, which=TALLY_TABS):
        out = np.zeros(self.CELLS, np.float32)
        self._chk(self.lib.soc_read_tally(self.h, int(which), _f(out), self.CELLS))
        return out

    def write_tally(self, which, values):
        v = np.ascontiguousarray(values, np.float32)
        self._chk(self.lib.soc_write_tally(self.h, int(which), _f(v), v.size))

    def tally_ptr(self, which=TALLY_TABS):
        return self.lib.soc_tally_ptr(self.h, int(which))

    def bind_tally(self, which, device_ptr, n=None):
        """caller-owned device memory (n floats, default CELLS) as tally `which`; device_ptr None/0 = back to the library's"""
        if not device_ptr:
            self._chk(self.lib.soc_bind_tally(self.h, int(which), None, 0))
        else:
            self._chk(self.lib.soc_bind_tally(self.h, int(which), C.c_void_p(device_ptr), int(self.CELLS if n is None else n)))

    def read_par(self):
        out = np.zeros(max(self.NPAR, 1), np.int32)
        self._chk(self.lib.soc_read_par(self.h, _i(out), self.NPAR))
        return out[:self.NPAR]

    def stats(self, reset=False):
        out = (C.c_uint64 * 3)()
        self._chk(self.lib.soc_stats(self.h, out, int(reset)))
        return dict(tally_events=int(out[0]), packets=int(out[1]), scatterings=int(out[2]))

    def sca_ray_steps(self):
        """cell steps of the rays of the scattered-light sweeps, as of the last stats() call"""
        return int(self.lib.soc_sca_ray_steps(self.h))

    def timer_start(self):
        self._chk(self.lib.soc_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.lib.soc_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    # ---- A2E ----
    def a2e_pre(self, FREQ, Ef, SKABS, E, T, FACTOR):
        """One grain size of a solver file (A2E_pre.py:233-256): integration weights and cooling rates on the enthalpy grid
        E[NE+1] (temperatures T[NE+1]); SKABS = pi a^2 Q_abs per grain.  Returns dict(Iw (packed as in the file), L1, L2, Tdown)."""
        FREQ, Ef, SKABS = (np.ascontiguousarray(a, np.float32) for a in (FREQ, Ef, SKABS))
        E, T = np.ascontiguousarray(E, np.float32), np.ascontiguousarray(T, np.float32)
        NFREQ, NE = FREQ.size, E.size - 1
        if Ef.size != NFREQ or SKABS.size != NFREQ or T.size != NE + 1:
            raise SocError("a2e_pre: Ef, SKABS must hold NFREQ floats, E and T NE+1")
        L1, L2 = np.zeros(NE * NE, np.int32), np.zeros(NE * NE, np.int32)
        Iw, noIw, Tdown = np.zeros(NE * NE * NFREQ, np.float32), np.zeros(NE - 1, np.int32), np.zeros(NE, np.float32)
        self._chk(self.lib.soc_a2e_pre(self.h, int(NFREQ), int(NE), np.float32(FACTOR), _f(FREQ), _f(Ef), _f(SKABS), _f(E), _f(T),
                                       _i(L1), _i(L2), _f(Iw), _i(noIw), _f(Tdown)))
        packed = np.concatenate([Iw[l * NE * NFREQ:l * NE * NFREQ + noIw[l]] for l in range(NE - 1)]) if NE > 1 else Iw[:0]
        L1[0] = -2                                         # A2E_pre.py:246, :249
        L2[0] = -2
        return dict(Iw=packed, L1=L1, L2=L2, Tdown=Tdown, noIw=noIw)

    def a2e_set_size(self, NE, NFREQ, size, AF):
        """size: dict with Iw, L1, L2, Tdown, EA, Ibeg of one grain size (solver file)."""
        a = {k: np.ascontiguousarray(size[k], t) for k, t in (("Iw", np.float32), ("L1", np.int32), ("L2", np.int32),
                                                                ("Tdown", np.float32), ("EA", np.float32), ("Ibeg", np.int32))}
        AF = np.ascontiguousarray(AF, np.float32)
        if a["L1"].size != NE * NE or a["L2"].size != NE * NE or a["Tdown"].size != NE or a["EA"].size != NE * NFREQ \
                or a["Ibeg"].size != NFREQ or AF.size != NFREQ:
            raise SocError("a2e_set_size: array sizes do not match NE=%d NFREQ=%d" % (NE, NFREQ))
        self._a2e_nfreq = NFREQ
        self._chk(self.lib.soc_a2e_set_size(self.h, int(NE), int(NFREQ), int(a["Iw"].size), _f(a["Iw"]), _i(a["L1"]),
                                            _i(a["L2"]), _f(a["Tdown"]), _f(a["EA"]), _i(a["Ibeg"]), _f(AF)))

    def a2e_solve(self, AABS):
        AABS = np.ascontiguousarray(AABS, np.float32)
        out = np.zeros_like(AABS)
        self._chk(self.lib.soc_a2e_solve(self.h, AABS.shape[0], _f(AABS), _f(out)))
        return out

    def a2e_upload(self, AABS):
        AABS = np.ascontiguousarray(AABS, np.float32)
        self._chk(self.lib.soc_a2e_upload(self.h, AABS.shape[0], _f(AABS)))

    # the cells resident in device memory: absorptions up once, the sum over the sizes down once (soc_a2e_resident_*)
    def a2e_resident_begin(self, cells, NFREQ):
        self._chk(self.lib.soc_a2e_resident_begin(self.h, int(cells), int(NFREQ)))
        self._a2e_res = (int(cells), int(NFREQ))

    def a2e_resident_upload(self, c0, AABS):
        AABS = np.ascontiguousarray(AABS, np.float32)
        self._chk(self.lib.soc_a2e_resident_upload(self.h, int(c0), AABS.shape[0], _f(AABS)))

    def a2e_resident_solve(self):
        self._chk(self.lib.soc_a2e_resident_solve(self.h))

    def a2e_resident_download(self, c0, n, out=None):
        out = np.zeros((int(n), self._a2e_res[1]), np.float32) if out is None else out
        self._chk(self.lib.soc_a2e_resident_download(self.h, int(c0), int(n), _f(out)))
        return out

    def a2e_resident_end(self):
        self._chk(self.lib.soc_a2e_resident_end(self.h))

    def a2e_run(self, batch):
        self._chk(self.lib.soc_a2e_run(self.h, int(batch)))

    def a2e_download(self, batch):
        out = np.zeros((batch, self._a2e_nfreq), np.float32)
        self._chk(self.lib.soc_a2e_download(self.h, int(batch), _f(out)))
        return out

    def eqsolver(self, icell, CELLS, NE, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS):
        """SolveEquilibriumDust of A2E_MABU.py for one batch: ABS[batch, NFREQ] -> T[batch], EMIT[batch, NFREQ]"""
        ABS = np.ascontiguousarray(ABS, np.float32)
        batch, NFREQ = ABS.shape
        FREQ, KABS, TTT = (np.ascontiguousarray(a, np.float32) for a in (FREQ, KABS, TTT))
        T = np.zeros(batch, np.float32)
        E = np.zeros((batch, NFREQ), np.float32)
        self._chk(self.lib.soc_eqsolver(self.h, batch, int(icell), int(CELLS), NFREQ, int(NE), np.float32(FACTOR),
                                        np.float32(kE), np.float32(oplgkE), np.float32(Emin), _f(FREQ), _f(KABS),
                                        _f(TTT), _f(ABS), _f(T), _f(E)))
        return T, E

    def a2e_eqtemp(self, icell, CELLS, NIP, FACTOR, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS):
        ABS = np.ascontiguousarray(ABS, np.float32)
        batch, NFREQ = ABS.shape
        FREQ, KABS, TTT = (np.ascontiguousarray(a, np.float32) for a in (FREQ, KABS, TTT))
        T = np.zeros(batch, np.float32)
        E = np.zeros((batch, NFREQ), np.float32)
        self._chk(self.lib.soc_a2e_eqtemp(self.h, batch, int(icell), int(CELLS), NFREQ, int(NIP), np.float32(FACTOR),
                                          np.float32(kE), np.float32(oplgkE), np.float32(Emin), _f(FREQ), _f(KABS),
                                          _f(TTT), _f(ABS), _f(T), _f(E)))
        return T, E

    # ---- probes ----
    def probe_rng(self, SEED, gid_first, n, ndraw):
        st = np.zeros((n, 2), np.uint32)
        dr = np.zeros((n, max(ndraw, 1)), np.uint32)
        self._chk(self.lib.soc_probe_rng(self.h, np.float32(SEED), int(gid_first), int(n), int(ndraw),
                                         st.ctypes.data_as(_U), dr.ctypes.data_as(_U)))
        return st, dr[:, :ndraw]

    def probe_math(self, fn, x):
        code = dict(exp=0, log=1, sin=2, cos=3, acos=4, sqrt=5, fmod1=6, rcp=7, expm1=8, pow15=9, logd=10)[fn]
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        self._chk(self.lib.soc_probe_math(self.h, code, _f(x), _f(y), x.size))
        return y

    def probe_trace(self, pos, direction, maxsteps=100000):
        pos = np.ascontiguousarray(pos, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        lev = np.zeros(maxsteps, np.int32)
        ind = np.zeros(maxsteps, np.int32)
        ds = np.zeros(maxsteps, np.float32)
        end = np.zeros(3, np.float32)
        n = C.c_int32()
        self._chk(self.lib.soc_probe_trace(self.h, _f(pos), _f(d), maxsteps, _i(lev), _i(ind), _f(ds), _f(end), C.byref(n)))
        n = n.value
        return lev[:n].copy(), ind[:n].copy(), ds[:n].copy(), end
