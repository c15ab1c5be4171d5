"""Scattered-light oracle (oracle/soc_oracle.c: walk_packet_sca) against the golden images
generated from the x86 builds of the reference's kernel_ASOC_sca.c (tests/golden/sca.npz,
made by tests/golden/make_golden.py --sca): bit-exact in libm mode (sequential work items,
hence the same fp32 summation order), 1e-5 relative on the image total in soc mode (same
algorithm, product math, slightly different trajectories)."""
import os

import numpy as np
import pytest

import cases
from oracle.pyoracle import RefSca, oracle_sim_sca

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sca.npz"))


@pytest.mark.parametrize("name", sorted(cases.SCA_CASES))
def test_sca_oracle_bit_exact_vs_reference_golden(name, oracle_libm):
    ref, kind, mk, vkw = cases.SCA_CASES[name]
    job, view = mk(), cases.sca_view(**vkw)
    OUT, n = oracle_sim_sca(oracle_libm, job, view, kind)
    want = GOLD[name].ravel()
    assert n > 1000
    assert np.array_equal(OUT.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name", sorted(cases.SCA_CASES))
def test_sca_soc_math_statistically_equal(name, oracle_soc):
    """The product math (soc_math.h) changes last bits, so trajectories diverge after a few
    boundary decisions; image totals must still agree to Monte Carlo accuracy."""
    ref, kind, mk, vkw = cases.SCA_CASES[name]
    job, view = mk(), cases.sca_view(**vkw)
    OUT, n = oracle_sim_sca(oracle_soc, job, view, kind)
    want = GOLD[name].ravel()
    assert abs(OUT.sum(dtype=np.float64) / want.sum(dtype=np.float64) - 1.0) < 1e-2


def test_sca_threaded_oracle_matches_sequential(oracle_soc):
    ref, kind, mk, vkw = cases.SCA_CASES["sca_bg_oct8"]
    job, view = mk(), cases.sca_view(**vkw)
    a, na = oracle_sim_sca(oracle_soc, job, view, kind)
    b, nb = oracle_sim_sca(oracle_soc, job, view, kind, nthreads=4)
    assert na == nb
    np.testing.assert_allclose(b, a, rtol=2e-5, atol=1e-6 * a.max())


def test_sca_workitem_ranges_add_up(oracle_soc):
    """Work items are independent: two halves of a launch sum to the whole (what the
    multi-GPU split relies on)."""
    ref, kind, mk, vkw = cases.SCA_CASES["sca_ps_ext1_c8"]
    job, view = mk(), cases.sca_view(**vkw)
    whole, n = oracle_sim_sca(oracle_soc, job, view, kind)
    a, na = oracle_sim_sca(oracle_soc, job, view, kind, 0, 128)
    b, nb = oracle_sim_sca(oracle_soc, job, view, kind, 128, 256)
    assert na + nb == n
    np.testing.assert_allclose(a + b, whole, rtol=2e-5, atol=1e-6 * whole.max())


@pytest.mark.skipif(not RefSca.available("c8"), reason="reference builds (oracle/_ref) not present")
@pytest.mark.parametrize("name", ["sca_bg_c8", "sca_ps_ext2_c8", "sca_cl_oct8_emw"])
def test_sca_live_reference(name, oracle_libm):
    ref, kind, mk, vkw = cases.SCA_CASES[name]
    job, view = mk(), cases.sca_view(**vkw)
    OUT, _ = oracle_sim_sca(oracle_libm, job, view, kind, 0, 64)
    want = RefSca(ref).sim(job, view, kind, 0, 64)
    assert np.array_equal(OUT.view(np.uint32), want.view(np.uint32))
