"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden vectors.

Tolerances: integer work (RNG streams, cell indices, tally-event counts) is bit-exact; fp32
single-ray walks are bit-exact; per-cell tallies are compared at rtol 1e-5 (the north-star
tolerance): with identical trajectories the only difference left is the order of the fp32
atomic adds."""
import os

import numpy as np
import pytest

import cases
from oracle.pyoracle import Job
from soc_amd import synth
from util import run_engine, assert_tally_close

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
RNG = np.load(os.path.join(G, "rng.npz"))
RAYS = np.load(os.path.join(G, "rays.npz"))
SIMS = np.load(os.path.join(G, "sims.npz"))


def test_native_library_is_the_thing_running(engine):
    assert b"gfx950" in engine.lib.soc_version()


def test_rng_streams_match_reference_golden(engine):
    for i, s in enumerate(RNG["seeds"]):
        for j, g in enumerate(RNG["gids"]):
            st, dr = engine.probe_rng(s, int(g), 1, 8)
            assert tuple(st[0]) == tuple(RNG["states"][i, j])
            assert np.array_equal(dr[0], RNG["draws"][i, j])


def test_rng_streams_contiguous_block_vs_oracle(engine, oracle_soc):
    st, dr = engine.probe_rng(0.7853981634, 786000, 600, 4)
    for k in (0, 1, 63, 64, 255, 256, 431, 599):
        s = oracle_soc.seed(0.7853981634, 786000 + k)
        assert tuple(st[k]) == s
        assert np.array_equal(dr[k], oracle_soc.draws(*s, 4)[0])


@pytest.mark.parametrize("fn", ["exp", "log", "sin", "cos", "acos", "sqrt", "fmod1", "expm1", "pow15", "logd"])
def test_device_math_bit_identical_to_host_build(fn, engine, oracle_soc):
    rng = np.random.default_rng(11)
    x = {"exp": np.concatenate([rng.uniform(-90, 5, 200000), -np.logspace(-9, 1, 20000)]),
         "log": np.concatenate([rng.uniform(0, 1, 200000), np.logspace(-38, 30, 20000), [0.0, 1.0]]),
         "sin": rng.uniform(-7, 7, 200000), "cos": rng.uniform(-7, 7, 200000),
         "acos": np.concatenate([rng.uniform(-1, 1, 200000), [1.0, -1.0, 0.5, -0.5]]),
         "sqrt": rng.uniform(0, 1e4, 200000), "fmod1": rng.uniform(-300, 300, 200000),
         "expm1": -np.concatenate([rng.uniform(0, 40, 100000), np.logspace(-30, 0, 100000)]),
         "pow15": rng.uniform(0.1, 3.0, 200000),
         "logd": np.concatenate([rng.uniform(0, 1, 200000), np.logspace(-38, 30, 20000)])}[fn].astype(np.float32)
    assert np.array_equal(engine.probe_math(fn, x).view(np.uint32), oracle_soc.math(fn, x).view(np.uint32))


def test_device_division_correctly_rounded(engine):
    x = np.random.default_rng(12).uniform(-1e3, 1e3, 200000).astype(np.float32)
    x[x == 0] = 1
    assert np.array_equal(engine.probe_math("rcp", x), np.float32(1.0) / x)


@pytest.mark.parametrize("name", sorted(cases.RAYS))
def test_ray_walk_bit_exact_vs_reference_golden(name, engine):
    ref, mk, pos, d = cases.RAYS[name]
    engine.set_cloud(mk())
    lev, ind, ds, end = engine.probe_trace(RAYS[name + "_pos"], RAYS[name + "_dir"])
    assert np.array_equal(lev, RAYS[name + "_lev"])
    assert np.array_equal(ind, RAYS[name + "_ind"])
    assert np.array_equal(ds.view(np.uint32), RAYS[name + "_ds"].view(np.uint32))
    assert np.array_equal(end.view(np.uint32), RAYS[name + "_end"].view(np.uint32))


def test_parents_table(engine):
    engine.set_cloud(synth.kat_octree())
    assert np.array_equal(engine.read_par(), RAYS["par_oct4"])
    engine.set_cloud(synth.octree_cloud(8, levels=3, frac=0.15, seed=7))
    assert np.array_equal(engine.read_par(), RAYS["par_oct8"])


def test_random_rays_octree_and_double_index(engine, oracle_soc):
    for cloud in (synth.octree_cloud(8, levels=3, frac=0.15, seed=7),
                  synth.octree_cloud(104, levels=3, frac=0.002, seed=11)):     # NX>DIMLIM: double Index
        job = Job(cloud, np.linspace(1, -1, 2500))
        engine.set_cloud(cloud)
        rays = np.random.default_rng(2)
        for _ in range(60):
            pos = rays.uniform(0.01, cloud.NX - 0.01, 3)
            d = rays.standard_normal(3)
            d = (d / np.sqrt((d ** 2).sum())).astype(np.float32)
            d[np.abs(d) < 5e-5] = 5e-5
            a = engine.probe_trace(pos, d)
            b = oracle_soc.trace(job, pos, d)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_simulation_vs_oracle(name, engine, oracle_soc):
    ref, kind, mk = cases.CASES[name]
    job = mk()
    T, I, n = oracle_soc.sim(job, kind)
    Tg, Ig, st = run_engine(engine, job, kind)
    assert st["tally_events"] == n, "trajectories diverged from the oracle"
    assert_tally_close(Tg, T, rtol=1e-5)
    if job.WITH_INT:
        assert_tally_close(Ig, I, rtol=1e-5)
    else:
        assert not Ig.any()
    if job.WITH_INT == 2:
        # signed sums: cancellation makes the per-cell relative error meaningless where |sum| << sum of |terms| = INT
        assert np.abs(job.INTV).sum() > 0
        for k in range(3):
            assert np.abs(job.INTV_gpu[k] - job.INTV[k]).max() <= 2e-6 * np.abs(I).max()
            assert np.all(np.abs(job.INTV_gpu[k] - job.INTV[k]) <= 1e-5 * np.maximum(I, 1e-3 * I.max()))
    if job.WITH_ALI:
        assert_tally_close(job.XAB_gpu, job.XAB, rtol=1e-5)      # job.XAB was filled by the oracle run above
    if job.ROI is not None:
        # the record of packets entering ROI: same entries touched, same sums up to the order of the fp32 adds
        assert np.array_equal(job.ROI_SAVE_gpu != 0, job.ROI_SAVE != 0)
        assert_tally_close(job.ROI_SAVE_gpu, job.ROI_SAVE, rtol=1e-5)


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_simulation_vs_reference_golden_statistical(name, engine):
    """Against the reference's own output (libm transcendentals): same physics, packets that
    hit a one-ulp difference in log/exp/sincos/acos take another path."""
    ref, kind, mk = cases.CASES[name]
    job = mk()
    Tg, Ig, st = run_engine(engine, job, kind)
    want = SIMS[name + "_TABS"]
    assert abs(Tg.sum(dtype=np.float64) / want.sum(dtype=np.float64) - 1) < 2e-3
    big = want > 0.05 * want.max()
    rel = np.abs(Tg[big] - want[big]) / want[big]
    assert np.median(rel) < (5e-3 if job.MIRROR else 1e-3) and rel.max() < 0.1   # reflected packets: longer paths, more diverge


def test_double_index_simulation(engine, oracle_soc):
    o = synth.octree_cloud(104, levels=3, frac=0.002, seed=11)
    _, csc = synth.hg_scattering_table(0.6)
    job = Job(o, csc, ABS=1e-6, SCA=5e-6, SOURCE=1, BATCH=1, SEED=0.41)
    T, _, n = oracle_soc.sim(job, 0, gid0=0, gid1=20000)
    Tg, _, st = run_engine(engine, job, 0, gid_first=0, gid_count=20000)
    assert st["tally_events"] == n
    assert_tally_close(Tg, T, rtol=1e-5)


def test_sharded_launch_equals_whole_launch(engine, oracle_soc):
    """Work-item ranges (the multi-GPU partition) reproduce the single launch: same streams."""
    ref, kind, mk = cases.CASES["bg_oct8"]
    job = mk()
    T, _, n = oracle_soc.sim(job, kind)
    cuts = [0, 1000, 1001, job.GLOBAL // 2 + 13, job.GLOBAL]
    Tg, _, st = run_engine(engine, job, kind, gid_first=cuts[0], gid_count=cuts[1] - cuts[0])
    events = st["tally_events"]
    for a, b in zip(cuts[1:-1], cuts[2:]):
        Tg, _, st = run_engine(engine, job, kind, gid_first=a, gid_count=b - a, zero=False)
        events += st["tally_events"]
    assert events == n
    assert_tally_close(Tg, T, rtol=1e-5)


def test_empty_and_ragged_launches(engine):
    ref, kind, mk = cases.CASES["bg_c8"]
    job = mk()
    Tg, _, st = run_engine(engine, job, kind, gid_first=5, gid_count=0)
    assert st["packets"] == 0 and not Tg.any()
    job.BATCH = 0
    Tg, _, st = run_engine(engine, job, kind)
    assert st["packets"] == 0 and not Tg.any()
    # GLOBAL padded beyond 8*AREA (the reference pads to a multiple of 64): extra ids return at once
    job = mk()
    job.GLOBAL = 8 * job.cloud.AREA + 64
    job.BATCH = 2
    Tg, _, st = run_engine(engine, job, kind)
    assert st["packets"] == 8 * job.cloud.AREA * 2


def test_errors_are_reported_not_fatal(engine):
    from soc_amd.lib import SocError
    c = synth.cartesian_cloud(4, uniform=1.0)
    bad = c.DENS.copy()
    bad[3] = -1.0                       # a "link" in a one-level cloud
    with pytest.raises(SocError, match="not a valid child link"):
        engine.set_grid(4, 4, 4, 1, c.LCELLS, bad)
    with pytest.raises(SocError):
        engine.set_features(ps_method=3)
    engine.set_cloud(c)
    with pytest.raises(SocError, match="outside GLOBAL"):
        engine.lib.soc_sim_pb  # keep flake8 quiet
        engine.set_scatter_table(None, np.linspace(1, -1, 100))
        engine.set_optical(1e-3, 1e-3)
        engine.sim_pb(1, 0, 1, 0.5, 1.0, 1.0, GLOBAL=100, gid_first=90, gid_count=20)
    # the switches of round 2: arguments are checked, state that does not fit is reported at the launch
    with pytest.raises(SocError, match="SW_A"):
        engine.set_step_weight(1, 0.0, 0.0)            # `stepweight 0.5 ...`: the reference passes -D SW_A=int(0.5)=0 and divides by it
    with pytest.raises(SocError, match="SW_B"):
        engine.set_step_weight(2, 0.7, 1.0)
    with pytest.raises(SocError, match="mode"):
        engine.set_step_weight(3, 0.7, 0.5)
    engine.set_step_weight(0)
    csc2 = np.stack([np.linspace(1, -1, 100), np.linspace(1, -1, 100)]).astype(np.float32)
    engine.set_scatter_tables(None, csc2)               # two species ...
    with pytest.raises(SocError, match="WITH_MSF"):
        engine.sim_pb(1, 0, 1, 0.5, 1.0, 1.0, GLOBAL=96)   # ... but no abundances / per-species cross sections
    engine.set_abundances(np.ones((c.CELLS, 3), np.float32))
    engine.set_optical_abu([1e-3] * 3, [1e-3] * 3)
    with pytest.raises(SocError, match="WITH_MSF"):
        engine.sim_pb(1, 0, 1, 0.5, 1.0, 1.0, GLOBAL=96)   # three species of abundances, two tables
    engine.set_abundances(None)
    engine.set_opt(None)
    engine.set_scatter_table(None, np.linspace(1, -1, 100))
    with pytest.raises(SocError, match="soc_set_grid first|with_int 2"):
        from soc_amd.lib import Engine
        e2 = Engine(0)
        try:
            e2.set_features(2, 0, 0)                    # intensity vectors are sized by the grid
        finally:
            e2.close()
    with pytest.raises(SocError):
        engine.read_tally(3)                            # INTX without with_int == 2
    with pytest.raises(SocError, match="tuning|unknown|name"):
        engine.set_tuning(no_such_knob=1)


def test_opt_is_half(engine, oracle_soc):
    """-D OPT_IS_HALF: OPT rounded to fp16 exactly as numpy does for the reference's upload (ASOC.py:1158-1159), for the
    device-built and the uploaded OPT, including values fp16 holds as subnormals, zeros, and overflow to inf"""
    cl = cases._c8()
    rr = np.random.default_rng(9)
    ABU = rr.uniform(0.0, 1.0, (cl.CELLS, 2)).astype(np.float32)
    AFABS = np.asarray([1e-4, 3e-7], np.float32)
    AFSCA = np.asarray([3e-4, 2e-8], np.float32)
    OPT = np.zeros((cl.CELLS, 2), np.float32)
    for d in range(2):
        OPT[:, 0] += ABU[:, d] * AFABS[d]
        OPT[:, 1] += ABU[:, d] * AFSCA[d]
    H = np.asarray(np.asarray(OPT, np.float16), np.float32)
    assert not np.array_equal(H, OPT)
    engine.set_cloud(cl)
    engine.set_opt_half(True)
    try:
        engine.set_abundances(ABU)
        engine.set_optical_abu(AFABS, AFSCA)
        assert np.array_equal(engine.read_opt().view(np.uint32), H.view(np.uint32))
        X = OPT.copy()
        X.ravel()[:8] = [0.0, 1e-9, 6e-8, 5.96e-8, 65504.0, 65519.9, 65520.0, 1e6]     # -> 0, 0, subnormals, max, max, inf, inf
        with np.errstate(over="ignore"):
            XH = np.asarray(np.asarray(X, np.float16), np.float32)
        engine.set_opt(X)
        assert np.array_equal(engine.read_opt().view(np.uint32), XH.view(np.uint32))
        # a launch with the rounded opacities is the oracle's launch with them
        job = Job(cl, cases._CSC, SOURCE=1, BATCH=6, SEED=0.3, OPT=H)
        T, _, n = oracle_soc.sim(job, 0)
        engine.set_opt(OPT)
        Tg, _, st = run_engine(engine, job, 0)       # uploads job.OPT = H (already fp16 values: rounding is idempotent)
        assert st["tally_events"] == n
        assert_tally_close(Tg, T, rtol=1e-5)
    finally:
        engine.set_opt_half(False)
        engine.set_abundances(None)
        engine.set_opt(None)


@pytest.mark.parametrize("single", [False, True])
def test_opt_from_abundances_on_device(single, engine, oracle_soc):
    """soc_set_abundances + soc_set_optical_abu: OPT built on the device is the numpy OPT of ASOC.py:1146-1160
    bit for bit, and a launch with it is the launch with the uploaded OPT"""
    cl = cases._oct8()
    rr = np.random.default_rng(8)
    ndust = 2 if single else 3
    ABU = rr.uniform(0.0, 1.0, cl.CELLS if single else (cl.CELLS, ndust)).astype(np.float32)
    AFABS = (1e-4 * rr.uniform(0.3, 3, ndust)).astype(np.float32)
    AFSCA = (3e-4 * rr.uniform(0.3, 3, ndust)).astype(np.float32)
    OPT = np.zeros((cl.CELLS, 2), np.float32)
    if single:
        OPT[:, 0] += ABU * AFABS[0] + (1.0 - ABU) * AFABS[1]
        OPT[:, 1] += ABU * AFSCA[0] + (1.0 - ABU) * AFSCA[1]
    else:
        for d in range(ndust):
            OPT[:, 0] += ABU[:, d] * AFABS[d]
            OPT[:, 1] += ABU[:, d] * AFSCA[d]
    engine.set_cloud(cl)
    engine.set_abundances(ABU, single=single)
    engine.set_optical_abu(AFABS, AFSCA)
    assert np.array_equal(engine.read_opt().view(np.uint32), OPT.view(np.uint32))
    job = Job(cl, cases._CSC, SOURCE=1, BATCH=6, SEED=0.3, OPT=OPT)
    T, _, n = oracle_soc.sim(job, 0)
    engine.set_features(0, 0, 0)
    engine.set_scatter_table(job.DSC, job.CSC)
    engine.set_optical(0.0, 0.0)
    engine.set_mirror(0)
    engine.set_exec(0, 4)
    engine.zero(0)
    engine.stats(reset=True)
    engine.sim_pb(1, 0, job.BATCH, job.SEED, job.BG, job.TW, GLOBAL=job.GLOBAL)
    engine.sync()
    assert engine.stats()["tally_events"] == n
    assert_tally_close(engine.read_tally(0), T, rtol=1e-5)
    engine.set_abundances(None)
    engine.set_opt(None)
    engine.set_exec(-1, 4)
