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


def test_deferred_launches_equal_sequential(engine, oracle_soc):
    """soc_batch_begin/end: launches with different seeds, opacities, scattering tables, weights and
    sources run in ONE sweep and give the tallies of running them one after the other (same packets,
    same per-launch RNG streams); the oracle sum pins both."""
    c32 = synth.cartesian_cloud(32, seed=21)
    d6, csc6 = synth.hg_scattering_table(0.6)
    d0, csc0 = synth.hg_scattering_table(0.1)
    ps = np.array([[16.3, 16.2, 16.1], [40.0, 16.0, 16.0]], np.float32)
    jobs = [Job(c32, csc6, ABS=2e-5, SCA=6e-5, SOURCE=1, BATCH=4, SEED=0.3711, BG=1.0, TW=1.0),
            Job(c32, csc0, ABS=5e-5, SCA=2e-5, SOURCE=1, BATCH=3, SEED=0.11, BG=2.5, TW=0.5),
            Job(c32, csc6, ABS=1e-5, SCA=9e-5, SOURCE=0, BATCH=6, SEED=0.77, TW=2.0, GLOBAL=65536, PSPOS=ps, PS=[3.0, 1.0]),
            Job(c32, csc0, ABS=2e-5, SCA=6e-5, SOURCE=1, BATCH=5, SEED=0.5, BG=0.7, TW=1.5)]

    def launch_all(batch):
        engine.set_cloud(c32)
        engine.set_features(0, 0, 0)
        engine.set_mirror(0)
        engine.set_opt(None)
        engine.set_exec(1, 3)
        engine.zero(0)
        engine.stats(reset=True)
        if batch:
            engine.batch_begin(batch)
        for j in jobs:
            engine.set_scatter_table(None, j.CSC)
            engine.set_optical(j.ABS, j.SCA)
            engine.sim_pb(j.SOURCE, 0, j.BATCH, j.SEED, j.BG, j.TW, PSPOS=j.PSPOS[:, :3], PS=j.PS, GLOBAL=j.GLOBAL)
        if batch:
            engine.batch_end()
        return engine.read_tally(0), engine.stats()

    want = np.zeros(c32.CELLS, np.float32)
    n = 0
    for j in jobs:
        _, _, k = oracle_soc.sim(j, 0, TABS=want)
        n += k
    Tseq, sseq = launch_all(0)
    assert sseq["tally_events"] == n
    assert_tally_close(Tseq, want, rtol=2e-5)
    for batch in (2, 4, 8):
        Tb, sb = launch_all(batch)
        assert sb == sseq                                   # identical trajectories, launch by launch
        assert_tally_close(Tb, want, rtol=2e-5)


def test_deferred_launches_flush_on_state_access(engine):
    """reading a tally inside a batch executes what is pending first; launches that cannot be deferred
    (here: INT tally requested) run immediately and in order"""
    c32 = synth.cartesian_cloud(32, seed=21)
    _, csc = synth.hg_scattering_table(0.6)
    engine.set_cloud(c32)
    engine.set_features(0, 0, 0)
    engine.set_mirror(0)
    engine.set_opt(None)
    engine.set_scatter_table(None, csc)
    engine.set_optical(2e-5, 6e-5)
    engine.set_exec(1, 3)
    engine.zero(0)
    engine.batch_begin(8)
    engine.sim_pb(1, 0, 2, 0.3, 1.0, 1.0, GLOBAL=8 * c32.AREA)
    a = engine.read_tally(0)                                  # flushes
    assert a.sum() > 0
    engine.sim_pb(1, 0, 2, 0.4, 1.0, 1.0, GLOBAL=8 * c32.AREA)
    engine.set_features(1, 0, 0)                              # INT tally: not deferrable; flushes the pending one
    engine.zero(1)
    engine.sim_pb(1, 0, 2, 0.5, 1.0, 1.0, GLOBAL=8 * c32.AREA)
    engine.batch_end()
    b = engine.read_tally(0)
    i = engine.read_tally(1)
    assert b.sum() > 2.5 * a.sum() and i.sum() > 0
    engine.set_features(0, 0, 0)
