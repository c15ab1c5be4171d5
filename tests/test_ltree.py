"""Brick-local hierarchies (soc_amd/csrc/soc_ltree.h, soc_lbricks.h) on the CPU: the integer / single-fma form of
Index() against the oracle's restatement of kernel_ASOC_aux.c:198-278 (double POS), ray by ray and bit for bit.
The harness tests/ltree_host.cpp compiles the product headers with g++ (-fsanitize=address,undefined)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.pyoracle import Job, Oracle
from soc_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
_F, _I = C.POINTER(C.c_float), C.POINTER(C.c_int)


def _harness(sanitize=False):
    out = os.path.join(REPO, "oracle", "_build", "libltree_host%s.so" % ("_san" if sanitize else ""))
    src = os.path.join(HERE, "ltree_host.cpp")
    deps = [src] + [os.path.join(REPO, "soc_amd", "csrc", f) for f in ("soc_ltree.h", "soc_lbricks.h", "soc_octbricks.h", "soc_math.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cmd = ["g++", "-O1" if sanitize else "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-Wall",
               "-o", out, src]
        if sanitize:
            cmd[1:1] = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
        subprocess.check_call(cmd)
    return out


class LTree:
    def __init__(self, cloud, cap, lib=None):
        self.lib = L = C.CDLL(lib or _harness())
        L.lt_build.restype = C.c_void_p
        L.lt_build.argtypes = [C.c_int] * 4 + [_I, _I, _F, C.c_int]
        L.lt_free.argtypes = [C.c_void_p]
        L.lt_info.argtypes = [C.c_void_p, _I, _I, C.POINTER(C.c_long)]
        L.lt_check.argtypes = [C.c_void_p, _F, C.c_long]
        L.lt_trace_move.restype = C.c_int
        L.lt_trace_move.argtypes = [C.c_void_p, _F, _F, C.c_int, _I, _I, _F, _F, _I]
        self.cloud = cloud
        self.LCELLS = np.ascontiguousarray(cloud.LCELLS, np.int32)
        self.OFF = np.ascontiguousarray(cloud.OFF, np.int32)
        self.DENS = np.ascontiguousarray(cloud.DENS, np.float32)
        self.h = L.lt_build(cloud.NX, cloud.NY, cloud.NZ, cloud.LEVELS, self.LCELLS.ctypes.data_as(_I), self.OFF.ctypes.data_as(_I),
                            self.DENS.ctypes.data_as(_F), cap)

    def info(self):
        nb, ms, ns = C.c_int(), C.c_int(), C.c_long()
        self.lib.lt_info(self.h, C.byref(nb), C.byref(ms), C.byref(ns))
        return nb.value, ms.value, ns.value

    def check(self):
        return self.lib.lt_check(self.h, self.DENS.ctypes.data_as(_F), self.cloud.CELLS)

    def trace(self, pos, d, maxsteps=20000):
        pos = np.ascontiguousarray(pos, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        lev = np.zeros(maxsteps, np.int32)
        cel = np.zeros(maxsteps, np.int32)
        ds = np.zeros(maxsteps, np.float32)
        end = np.zeros(3, np.float32)
        st = C.c_int()
        n = self.lib.lt_trace_move(self.h, pos.ctypes.data_as(_F), d.ctypes.data_as(_F), maxsteps, lev.ctypes.data_as(_I),
                              cel.ctypes.data_as(_I), ds.ctypes.data_as(_F), end.ctypes.data_as(_F), C.byref(st))
        return lev[:n], cel[:n], ds[:n], end, st.value

    def close(self):
        if self.h:
            self.lib.lt_free(self.h)
            self.h = None


def _rays(cloud, n, rng):
    """starting points on the faces and inside, directions with the reference's clamp (|u| >= DEPS) and normalisation"""
    N = np.asarray([cloud.NX, cloud.NY, cloud.NZ], np.float32)
    pos = (rng.random((n, 3)) * (N - 2e-4) + 1e-4).astype(np.float32)
    face = rng.integers(0, 7, n)
    for i in range(n):
        if face[i] < 6:
            a = face[i] // 2
            pos[i, a] = np.float32(1e-4) if face[i] % 2 == 0 else np.float32(N[a] - 1e-4)
    u = rng.standard_normal((n, 3)).astype(np.float32)
    grazing = rng.random(n) < 0.1
    u[grazing, rng.integers(0, 3)] *= np.float32(1e-5)
    u = np.where(np.abs(u) < 5e-5, np.float32(5e-5), u)
    u /= np.sqrt((u.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32))[:, None]
    return pos, u.astype(np.float32)


CLOUDS = {
    # NX > 100 with >= 3 levels: the reference evaluates Index() with double3 POS (kernel_ASOC_aux.c:25-37)
    "oct104_l4": lambda: synth.octree_cloud(104, levels=4, frac=0.08, seed=3),
    "oct128_l5": lambda: synth.octree_cloud(128, levels=5, frac=0.05, seed=5),
}


@pytest.mark.parametrize("name,cap", [("oct104_l4", 6144), ("oct104_l4", 700), ("oct128_l5", 4096)])
def test_local_tree_steps_are_the_oracles_steps(name, cap, oracle_soc):
    """soc_lt_aim / soc_lt_land (one path for every kind of move, the device walk's) against the oracle's Index, step by step"""
    cloud = CLOUDS[name]()
    lt = LTree(cloud, cap)
    assert lt.h, "bricks could not be built"
    nb, ms, ns = lt.info()
    assert ms <= cap and ns == cloud.CELLS and lt.check() == 0
    job = Job(cloud, np.linspace(1, -1, 16).astype(np.float32))
    rng = np.random.default_rng(11)
    pos, u = _rays(cloud, 6000, rng)
    steps = slow = 0
    deeper = set()
    for i in range(len(pos)):
        lev, cel, ds, end, st = lt.trace(pos[i], u[i])
        assert st in (0, 2), "ray %d: status %d" % (i, st)
        olev, oind, ods, oend = oracle_soc.trace(job, pos[i], u[i], maxsteps=20000)
        n = len(lev)
        if st == 2:                      # stopped where the generic Index() has to decide: compare what was walked
            slow += 1
            assert n <= len(olev)
        else:
            assert n == len(olev), "ray %d: %d steps, oracle %d" % (i, n, len(olev))
            assert np.array_equal(end.view(np.uint32), oend.view(np.uint32)), "ray %d: end position" % i
        ocell = cloud.OFF[olev[:n]] + oind[:n]
        assert np.array_equal(lev, olev[:n]) and np.array_equal(cel, ocell), "ray %d: cells differ" % i
        assert np.array_equal(ds.view(np.uint32), ods[:n].view(np.uint32)), "ray %d: step lengths differ" % i
        steps += n
        deeper.update(lev.tolist())
    assert steps > 250000 and deeper == set(range(cloud.LEVELS))
    assert slow < 0.02 * len(pos)
    lt.close()


def test_bricks_partition_the_hierarchy_under_sanitizers():
    """rectangular root grid that is not a multiple of the 16-cell tile, a clustered hierarchy, small caps"""
    lib = _harness(sanitize=True)
    # the sanitizer runtime must come first in the process: a child python with libasan preloaded
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_ltree as T\nT._sanitized_body(%r)\nprint('sanitized ok')\n" % (REPO, HERE, lib))
    asan = subprocess.check_output(["g++", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run(["python3", "-c", code], env=env, cwd=REPO, capture_output=True, text=True)
    assert out.returncode == 0 and "sanitized ok" in out.stdout, out.stderr[-3000:]


def _sanitized_body(lib):
    for cloud, cap in ((synth.octree_cloud(19, levels=4, frac=0.2, seed=2), 600), (synth.octree_cloud(40, levels=3, frac=0.1, seed=4), 5000)):
        lt = LTree(cloud, cap, lib=lib)
        assert lt.h and lt.check() == 0
        nb, ms, ns = lt.info()
        assert ms <= cap and ns == cloud.CELLS
        rng = np.random.default_rng(5)
        pos, u = _rays(cloud, 200, rng)
        for i in range(len(pos)):
            lt.trace(pos[i], u[i])
        lt.close()
    # one root cell with more cells below it than the cap: refused, not mis-built
    deep = synth.octree_cloud(4, levels=4, frac=0.5, seed=1)
    lt = LTree(deep, 64, lib=lib)
    assert not lt.h
    # the older brick builder (hierarchy in global memory: abundance runs, > 8 levels, forced with global_tree): subtrees
    # larger than the cap are split below their refined head cell
    L = C.CDLL(lib)
    L.ob_check.argtypes = [C.c_int] * 4 + [_I, _I, _F, C.c_int, _I, _I]
    for cloud, cap in ((synth.octree_cloud(19, levels=4, frac=0.2, seed=2), 8), (synth.octree_cloud(19, levels=4, frac=0.2, seed=2), 600),
                       (synth.octree_cloud(40, levels=3, frac=0.1, seed=4), 5000), (deep, 16), (synth.kat_octree(), 8)):
        nb, big = C.c_int(0), C.c_int(0)
        lc, off = np.ascontiguousarray(cloud.LCELLS, np.int32), np.ascontiguousarray(cloud.OFF, np.int32)
        dens = np.ascontiguousarray(cloud.DENS, np.float32)
        rc = L.ob_check(cloud.NX, cloud.NY, cloud.NZ, cloud.LEVELS, lc.ctypes.data_as(_I), off.ctypes.data_as(_I), dens.ctypes.data_as(_F), cap,
                        C.byref(nb), C.byref(big))
        assert rc == 0 and nb.value >= cloud.CELLS // cap and 0 < big.value <= cap, (rc, nb.value, big.value)
