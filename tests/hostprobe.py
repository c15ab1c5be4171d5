"""Compile tests/csrc/host_probe.c (host build of the product headers) and bind it."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "host_probe.c")
SO = os.path.join(HERE, "csrc", "libhostprobe.so")
DEPS = [SRC, os.path.join(HERE, "..", "soc_amd", "csrc", "soc_math.h"), os.path.join(HERE, "..", "soc_amd", "csrc", "soc_rng.h")]

_U32 = C.POINTER(C.c_uint32)
_U64 = C.POINTER(C.c_uint64)
_F = C.POINTER(C.c_float)


def load():
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-mfma",
                               "-msse4.1", "-o", SO, SRC, "-lm"])
    lib = C.CDLL(SO)
    lib.hp_seed_base.restype = C.c_uint64
    lib.hp_seed_base.argtypes = [C.c_float]
    lib.hp_seed_mul.restype = C.c_uint64
    lib.hp_seed_mul.argtypes = [C.c_float]
    lib.hp_build_table.argtypes = [_U64]
    lib.hp_mulmod.restype = C.c_uint64
    lib.hp_mulmod.argtypes = [C.c_uint64, C.c_uint64]
    lib.hp_powmod.restype = C.c_uint64
    lib.hp_powmod.argtypes = [C.c_uint64, C.c_uint64]
    lib.hp_seed_stream.argtypes = [C.c_float, _U64, C.c_uint32, _U32, _U32]
    lib.hp_draws.argtypes = [_U32, _U32, C.c_int, _U32, _F]
    lib.hp_math.argtypes = [C.c_int, _F, _F, C.c_long]
    return lib


class HostProbe:
    def __init__(self):
        self.lib = load()
        self.tab = np.zeros(1024, np.uint64)
        self.lib.hp_build_table(self.tab.ctypes.data_as(_U64))

    def seed(self, SEED, gid):
        x, c = C.c_uint32(), C.c_uint32()
        self.lib.hp_seed_stream(np.float32(SEED), self.tab.ctypes.data_as(_U64), int(gid), C.byref(x), C.byref(c))
        return x.value, c.value

    def draws(self, x, c, n):
        xx, cc = C.c_uint32(x), C.c_uint32(c)
        u = np.zeros(n, np.uint32)
        r = np.zeros(n, np.float32)
        self.lib.hp_draws(C.byref(xx), C.byref(cc), n, u.ctypes.data_as(_U32), r.ctypes.data_as(_F))
        return u, r

    def math(self, fn, x):
        code = dict(exp=0, log=1, sin=2, cos=3, acos=4, sqrt=5, fmod1=6, expm1=8, pow15=9, logd=10, exp_small=11)[fn]
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        self.lib.hp_math(code, x.ctypes.data_as(_F), y.ctypes.data_as(_F), x.size)
        return y

    def div_by_rcp(self, n, u):
        n = np.ascontiguousarray(n, np.float32)
        u = np.ascontiguousarray(u, np.float32)
        q = np.zeros_like(n)
        self.lib.hp_div_by_rcp(n.ctypes.data_as(_F), u.ctypes.data_as(_F), q.ctypes.data_as(_F), n.size)
        return q

    def atan2(self, y, x):
        y = np.ascontiguousarray(y, np.float32)
        x = np.ascontiguousarray(x, np.float32)
        r = np.zeros_like(y)
        self.lib.hp_atan2(y.ctypes.data_as(_F), x.ctypes.data_as(_F), r.ctypes.data_as(_F), y.size)
        return r
