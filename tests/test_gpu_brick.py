"""Brick-sweep execution (LDS tallies, packets sorted by brick) against the oracle and against
the direct kernel: same logical work items and RNG streams -> identical trajectories."""
import numpy as np
import pytest

import cases
from oracle.pyoracle import Job
from soc_amd import synth
from util import run_engine, assert_tally_close

pytestmark = pytest.mark.gpu

CART = [n for n, (ref, kind, mk) in sorted(cases.CASES.items()) if kind == 0 and "oct" not in n and "mirror" not in n]


@pytest.mark.parametrize("name", CART)
@pytest.mark.parametrize("lb", [2, 3])
def test_brick_sweep_matches_oracle(name, lb, engine, oracle_soc):
    ref, kind, mk = cases.CASES[name]
    job = mk()
    T, I, n = oracle_soc.sim(job, kind)
    Tg, Ig, st = run_engine(engine, job, kind, exec_mode=1, brick_log2=lb)
    assert engine.last_passes() > 0
    assert st["tally_events"] == n, "trajectories diverged from the oracle"
    assert_tally_close(Tg, T, rtol=1e-5)
    if job.WITH_INT:
        assert_tally_close(Ig, I, rtol=1e-5)


@pytest.mark.parametrize("lb", [3, 4])
def test_brick_sweep_c32_and_sharding(lb, engine, oracle_soc):
    c32 = synth.cartesian_cloud(32, seed=21)
    _, csc = synth.hg_scattering_table(0.6)
    job = Job(c32, csc, ABS=2e-5, SCA=6e-5, SOURCE=1, BATCH=6, SEED=0.3711)
    T, _, n = oracle_soc.sim(job, 0)
    Tg, _, st = run_engine(engine, job, 0, exec_mode=1, brick_log2=lb)
    assert st["tally_events"] == n and st["packets"] == job.GLOBAL * 6
    assert_tally_close(Tg, T, rtol=1e-5)
    # work-item ranges (multi-GPU partition) in brick mode
    h = job.GLOBAL // 3 + 7
    Ta, _, sa = run_engine(engine, job, 0, gid_first=0, gid_count=h, exec_mode=1, brick_log2=lb)
    Tb, _, sb = run_engine(engine, job, 0, gid_first=h, gid_count=job.GLOBAL - h, zero=False, exec_mode=1, brick_log2=lb)
    assert sa["tally_events"] + sb["tally_events"] == n
    assert_tally_close(Tb, T, rtol=1e-5)


def test_brick_sweep_rejects_octree(engine):
    from soc_amd.lib import SocError
    ref, kind, mk = cases.CASES["bg_oct8"]
    with pytest.raises(SocError, match="not applicable"):
        run_engine(engine, mk(), kind, exec_mode=1, brick_log2=2)
    engine.set_exec(-1, 4)


def test_brick_sweep_refuses_mirror(engine):
    """reflecting faces are handled by the direct kernel only: forcing the brick sweep is an error,
    automatic mode falls back"""
    from soc_amd.lib import SocError
    ref, kind, mk = cases.CASES["bg_c8_mirror"]
    with pytest.raises(SocError):
        run_engine(engine, mk(), kind, exec_mode=1)
    run_engine(engine, mk(), kind, exec_mode=-1)
    assert engine.last_passes() == 0
