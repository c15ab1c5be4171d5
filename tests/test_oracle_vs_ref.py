"""Live comparison of the oracle (libm mode) with the x86 builds of the reference kernels.
Needs oracle/_ref (built where /root/reference exists); skipped elsewhere -- the committed
golden vectors carry the same pins."""
import numpy as np
import pytest

import cases
from oracle.pyoracle import Job, Ref
from soc_amd import synth

pytestmark = pytest.mark.skipif(not Ref.available("c8"), reason="reference builds (oracle/_ref) not present")


@pytest.mark.parametrize("name", ["bg_c8", "bg_oct8", "ps_ext2_c8", "cl_oct8_emw", "bg_c8_sw1", "bg_oct8_sw2", "cl_oct8_sw2", "bg_oct8_msf",
                                  "cl_oct8_msf", "hp_oct8_msf", "bg_c8_int2", "cl_oct8_int2", "ps_in_oct8"])
def test_live_bit_exact(name, oracle_libm):
    ref, kind, mk = cases.CASES[name]
    job = mk()
    T, I, _ = oracle_libm.sim(job, kind)
    V = None if job.INTV is None else job.INTV.copy()
    if V is not None:
        job.INTV[:] = 0
    T2, I2 = Ref(ref).sim(job, kind)
    assert np.array_equal(T.view(np.uint32), T2.view(np.uint32))
    if V is not None:
        assert np.abs(V).sum() > 0 and np.array_equal(V.view(np.uint32), job.INTV.view(np.uint32))


def test_double_index_path_bit_exact(oracle_libm):
    """NX=104 > DIMLIM=100 with LEVELS=3: Index() runs in double (kernel_ASOC_aux.c:25-37,207-211)."""
    o = synth.octree_cloud(104, levels=3, frac=0.002, seed=11)
    _, csc = synth.hg_scattering_table(0.6)
    job = Job(o, csc, ABS=1e-6, SCA=5e-6, SOURCE=1, BATCH=1, SEED=0.41)
    # work items whose surface elements face refined regions are spread over the launch
    T, _, n = oracle_libm.sim(job, 0, gid0=0, gid1=6000)
    T2, _ = Ref("oct104").sim(job, 0, gid0=0, gid1=6000)
    assert n > 0 and np.array_equal(T.view(np.uint32), T2.view(np.uint32))
    rays = np.random.default_rng(1)
    for _ in range(20):
        pos = [1e-4, rays.uniform(1, 103), rays.uniform(1, 103)]
        d = rays.standard_normal(3)
        d[0] = abs(d[0]) + 0.1
        d = (d / np.sqrt((d ** 2).sum())).astype(np.float32)
        a = oracle_libm.trace(job, pos, d)
        b = Ref("oct104").trace(job, pos, d)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))


def test_random_rays_octree(oracle_libm, oracle_soc):
    o8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    job = Job(o8, np.linspace(1, -1, 2500))
    r = Ref("oct8")
    rays = np.random.default_rng(2)
    for _ in range(200):
        pos = rays.uniform(0.01, 7.99, 3)
        d = rays.standard_normal(3)
        d = (d / np.sqrt((d ** 2).sum())).astype(np.float32)
        d[np.abs(d) < 5e-5] = 5e-5
        a, b, c = oracle_libm.trace(job, pos, d), r.trace(job, pos, d), oracle_soc.trace(job, pos, d)
        for x in (a, c):
            assert np.array_equal(x[0], b[0]) and np.array_equal(x[1], b[1])
            assert np.array_equal(x[2].view(np.uint32), b[2].view(np.uint32))
