"""The walk on brick-local hierarchies (soc_lbrick_walk, soc_ltree.h) on the GPU: hierarchies whose Index() the reference
evaluates in double (NX > 100, >= 3 levels).  Same bar as the other sweeps: identical trajectories (tally-event counts
equal to the oracle's), tallies equal to fp32 summation order; plus the older form of the sweep (`global_tree`) as a
second witness and the slow-step queue forced on every few steps."""
import numpy as np
import pytest

import cases
from oracle.pyoracle import Job
from soc_amd import synth
from util import assert_tally_close, run_engine

pytestmark = pytest.mark.gpu

_CLOUD = {}


def cloud104():
    if "c" not in _CLOUD:
        _CLOUD["c"] = synth.octree_cloud(104, levels=4, frac=0.08, seed=3)
    return _CLOUD["c"]


kw_form = {}


@pytest.fixture(autouse=True, params=[0, 64, 1], ids=["park4096", "park64", "nopark"])
def walk_form(request, engine):
    """every test of this file runs with short brick queues parked until they hold 4096 packets (the default), 64, or never"""
    engine.set_tuning(park_below=request.param)
    kw_form["base"] = 3
    yield
    engine.set_tuning(park_below=0)
    kw_form.pop("base", None)


def _sweep(engine, job, kind, **kw):
    T, I, st = run_engine(engine, job, kind, exec_mode=1, **kw)
    assert engine.last_passes() > 0
    assert engine.last_form() == kw_form.get("form", kw_form.get("base", 3))
    return T, I, st


@pytest.mark.parametrize("tune", [dict(), dict(brick_cells=700), dict(slow_every=3), dict(steps_per_visit=3, hash_slots=64),
                                  dict(brick_cells=12288, threads=1024), dict(chunk=64, threads=64)])
def test_background_packets(tune, engine, oracle_soc, tuned):
    cl = cloud104()
    job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=1, BATCH=4, SEED=0.377)
    g0, g1 = 100000, 106000
    T, _, n = oracle_soc.sim(job, 0, gid0=g0, gid1=g1, nthreads=8)
    tuned(**tune)
    Tg, _, st = _sweep(engine, job, 0, gid_first=g0, gid_count=g1 - g0)
    assert st["tally_events"] == n, "trajectories diverged from the oracle"
    assert_tally_close(Tg, T, rtol=1e-5)
    engine.set_exec(-1, 4)


@pytest.mark.parametrize("sw", [(1, 0.5, 0.0), (2, 0.7, 0.4)])
def test_weighted_free_paths(sw, engine, oracle_soc):
    """-D STEP_WEIGHT in the event workgroups of the brick-local form: creation and scattering draw the weighted path"""
    cl = cloud104()
    job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=1, BATCH=3, SEED=0.52, STEP_WEIGHT=sw)
    g0, g1 = 200000, 205000
    T, _, n = oracle_soc.sim(job, 0, gid0=g0, gid1=g1, nthreads=8)
    Tg, _, st = _sweep(engine, job, 0, gid_first=g0, gid_count=g1 - g0)
    assert st["tally_events"] == n and st["scatterings"] > 100
    assert_tally_close(Tg, T, rtol=1e-5)
    engine.set_step_weight(0)
    engine.set_exec(-1, 4)


def test_older_sweep_is_a_second_witness(engine, tuned):
    cl = cloud104()
    job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=1, BATCH=2, SEED=0.91)
    g0, g1 = 0, 40000
    Ta, _, sa = _sweep(engine, job, 0, gid_first=g0, gid_count=g1 - g0)
    tuned(global_tree=1)
    kw_form["form"] = 2
    try:
        Tb, _, sb = _sweep(engine, job, 0, gid_first=g0, gid_count=g1 - g0)
    finally:
        kw_form.pop("form", None)
    assert sa["tally_events"] == sb["tally_events"] and sa["packets"] == sb["packets"] and sa["scatterings"] == sb["scatterings"]
    assert_tally_close(Ta, Tb, rtol=1e-5)
    engine.set_exec(-1, 4)


@pytest.mark.parametrize("slow", [0, 5])
def test_point_sources_inside_and_outside(slow, engine, oracle_soc, tuned):
    cl = cloud104()
    ps = np.array([[52.3, 51.7, 50.2], [52.0, 52.0, 300.0]], np.float32)
    job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=0, BATCH=30, SEED=0.2, GLOBAL=512, PSPOS=ps, PS=[1.0, 2.5], PS_METHOD=0,
              WITH_INT=1, TW=1.5)
    T, I, n = oracle_soc.sim(job, 0, nthreads=8)
    tuned(slow_every=slow)
    Tg, Ig, st = _sweep(engine, job, 0)
    assert st["tally_events"] == n
    assert_tally_close(Tg, T, rtol=1e-5)
    assert_tally_close(Ig, I, rtol=1e-5)
    engine.set_exec(-1, 4)


@pytest.mark.parametrize("emw", [0, 1])
def test_cell_emission(emw, engine, oracle_soc, tuned):
    """SimRAM_CL: without emission weights packets also start "in" refined cells (kernel_ASOC.c:1318-1355)"""
    cl = cloud104()
    emit = np.where(cl.DENS > 0, cl.DENS * 1e-3, 1e-4).astype(np.float32)
    emwei = np.random.default_rng(5).uniform(0, 2.5, cl.CELLS).astype(np.float32) if emw else None
    job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=2, BATCH=1, SEED=0.9, GLOBAL=8192, EMIT=emit, EMWEI=emwei, USE_EMWEIGHT=emw)
    g0, g1 = 4000, 4096
    T, _, n = oracle_soc.sim(job, 1, gid0=g0, gid1=g1, nthreads=8)
    tuned(slow_every=7)
    Tg, _, st = _sweep(engine, job, 1, gid_first=g0, gid_count=g1 - g0)
    assert st["tally_events"] == n
    assert_tally_close(Tg, T, rtol=1e-5)
    engine.set_exec(-1, 4)


def test_healpix_background(engine, oracle_soc):
    cl = cloud104()
    sky, P = cases.hp_sky(weighted=True)
    job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=1, BATCH=8, SEED=0.11, GLOBAL=4096, HPBG=sky, HPBGP=P, TW=1.2)
    T, _, n = oracle_soc.sim(job, 2, nthreads=8)
    Tg, _, st = _sweep(engine, job, 2)
    assert st["tally_events"] == n
    assert_tally_close(Tg, T, rtol=1e-5)
    engine.set_exec(-1, 4)


def test_deferred_launches_with_and_without_int(engine, oracle_soc):
    """several launches in one sweep (soc_batch_begin) and with their own INT tallies (soc_batch_begin_int)"""
    cl = cloud104()
    d6, csc6 = synth.hg_scattering_table(0.6)
    d1, csc1 = synth.hg_scattering_table(0.1)
    jobs = [Job(cl, csc6, ABS=3e-6, SCA=3e-5, SOURCE=1, BATCH=2, SEED=0.377, TW=1.0),
            Job(cl, csc1, ABS=1e-6, SCA=6e-5, SOURCE=1, BATCH=3, SEED=0.52, TW=0.5, BG=2.0)]
    g0, g1 = 200000, 203000
    want = np.zeros(cl.CELLS, np.float32)
    ints, n = [], 0
    for j in jobs:
        j.WITH_INT = 1
        T, I, k = oracle_soc.sim(j, 0, gid0=g0, gid1=g1, nthreads=8)
        want += T
        ints.append(I)
        n += k
    for keep_int in (0, 1):
        e = engine
        e.set_cloud(cl)
        e.set_features(with_int=keep_int, ps_method=0, use_emweight=0)
        e.set_opt(None)
        e.set_exec(1, 4)
        e.zero(0)
        e.stats(reset=True)
        (e.batch_begin_int if keep_int else e.batch_begin)(4)
        for j in jobs:
            e.set_scatter_table(j.DSC, j.CSC)
            e.set_optical(j.ABS, j.SCA)
            e.sim_pb(1, 0, j.BATCH, j.SEED, j.BG, j.TW, GLOBAL=j.GLOBAL, gid_first=g0, gid_count=g1 - g0)
        e.batch_end()
        st = e.stats()
        assert e.last_passes() > 0 and e.last_form() == kw_form.get("base", 3) and st["tally_events"] == n
        assert_tally_close(e.read_tally(0), want, rtol=1e-5)
        if keep_int:
            for k, I in enumerate(ints):
                assert_tally_close(e.batch_read_int(k), I, rtol=1e-5)
    engine.set_exec(-1, 4)


@pytest.mark.parametrize("mask,kind", [(25, 0), (6, 0), (21, 1)], ids=["xYZ_bg", "Xy_bg", "xyz_cl"])
def test_reflecting_faces_in_the_sweep(mask, kind, engine, oracle_soc):
    """`mirror`: a packet whose step took it out of the model arrives in its launch's creation queue; the event workgroups
    reflect it there (Mirror, kernel_ASOC_aux.c:1054-1083, on the root-grid position Index() leaves behind) and send it on to the
    brick of the cell it re-enters.  Same trajectories as the oracle: event counts equal.  (Masks with both faces of an axis make
    the reference's unconditional direction flips cancel: such a packet leaves again at once and comes back until its free path
    ends -- correct here too, but a pass of the sweep per bounce; not exercised.)"""
    cl = cloud104()
    if kind == 0:
        job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=1, BATCH=3, SEED=0.4177, MIRROR=mask)
        g0, g1 = 200000, 204000
    else:
        emit = np.where(cl.DENS > 0, cl.DENS * 1e-3, 0).astype(np.float32)
        job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=2, BATCH=1, SEED=0.913, GLOBAL=1 << 20, EMIT=emit, MIRROR=mask)
        g0, g1 = 300000, 303000
    T, _, n = oracle_soc.sim(job, kind, gid0=g0, gid1=g1, nthreads=8)
    Tg, _, st = _sweep(engine, job, kind, gid_first=g0, gid_count=g1 - g0)
    assert st["tally_events"] == n, "trajectories diverged from the oracle"
    assert_tally_close(Tg, T, rtol=1e-5)
    engine.set_exec(-1, 4)


@pytest.mark.parametrize("kind", [0, 1])
def test_intensity_vectors_in_the_sweep(kind, engine, oracle_soc):
    """-D SAVE_INTENSITY=2 (kernel_ASOC.c:604-612, :724-732): INT and the vector sums INTX, INTY, INTZ beside TABS in the LDS of the
    walk (24 B per cell: smaller bricks), the scattering block's deposits from the event workgroups; background and cell emission"""
    cl = cloud104()
    if kind == 0:
        job = Job(cl, cases._CSC, ABS=3e-5, SCA=6e-5, SOURCE=1, BATCH=3, SEED=0.377, WITH_INT=2, TW=1.7)
        g0, g1 = 100000, 106000
    else:
        emit = np.where(cl.DENS > 0, cl.DENS * 1e-3, 1e-4).astype(np.float32)
        job = Job(cl, cases._CSC, ABS=3e-5, SCA=6e-5, SOURCE=2, BATCH=1, SEED=0.9, GLOBAL=8192, EMIT=emit, WITH_INT=2)
        g0, g1 = 4000, 4064
    T, I, n = oracle_soc.sim(job, kind, gid0=g0, gid1=g1, nthreads=8)
    want = np.array(job.INTV, np.float32).copy()
    Tg, Ig, st = _sweep(engine, job, kind, gid_first=g0, gid_count=g1 - g0)
    assert st["tally_events"] == n and st["scatterings"] > 100
    assert_tally_close(Tg, T, rtol=1e-5)
    assert_tally_close(Ig, I, rtol=1e-5)
    assert np.abs(want).sum() > 0
    for k in range(3):      # signed sums: the error is measured against the sum of the absolute terms, INT (as tests/test_gpu_parity.py)
        got = engine.read_tally(3 + k)
        assert np.abs(got - want[k]).max() <= 2e-6 * np.abs(I).max()
        assert np.all(np.abs(got - want[k]) <= 1e-5 * np.maximum(I, 1e-3 * I.max()))
    engine.set_features(0, 0, 0)
    engine.set_exec(-1, 4)


def test_groups_of_launches_with_an_int_tally_each(engine, oracle_soc):
    """soc_batch_begin_int_groups / soc_batch_next_int: several 'frequencies' in one sweep, the launches of a frequency -- a point-source
    and a cell-emission launch here -- tallying into that frequency's INT array (brick queues per group); TABS integrates over all of them.
    Every group's INT equals the one its launches give alone, and the oracle's for the first group."""
    cl = cloud104()
    ps = np.array([[52.3, 51.7, 50.2]], np.float32)
    emit = np.where(cl.DENS > 0, cl.DENS * 1e-3, 1e-4).astype(np.float32)
    groups = []
    for f in range(3):
        kw = dict(ABS=(3 + f) * 1e-6, SCA=(3 - 0.5 * f) * 1e-5, WITH_INT=1, TW=1.0 + 0.5 * f)
        groups.append([(0, Job(cl, cases._CSC, SOURCE=0, BATCH=20, SEED=0.2 + 0.1 * f, GLOBAL=512, PSPOS=ps, PS=[1.0 + f], PS_METHOD=0, **kw), 0, 512),
                       (1, Job(cl, cases._CSC, SOURCE=2, BATCH=1, SEED=0.6 + 0.1 * f, GLOBAL=8192, EMIT=emit * (1 + f), **kw), 4000, 4048)])
    alone, stats, tabs = [], dict(tally_events=0, packets=0, scatterings=0), None
    for g in groups:
        I = None
        for kind, job, g0, g1 in g:
            T, Ik, st = _sweep(engine, job, kind, gid_first=g0, gid_count=g1 - g0)
            I = Ik.astype(np.float64) if I is None else I + Ik
            tabs = T.astype(np.float64) if tabs is None else tabs + T
            for key in stats:
                stats[key] += st[key]
        alone.append(I)
    T0, I0, n0 = oracle_soc.sim(groups[0][0][1], 0, nthreads=8)
    T1, I1, n1 = oracle_soc.sim(groups[0][1][1], 1, gid0=4000, gid1=4048, nthreads=8)
    assert_tally_close(alone[0], I0.astype(np.float64) + I1, rtol=1e-5)
    # the same launches in one sweep
    e = engine
    e.set_exec(1, 4)
    e.set_features(1, 0, 0)
    e.zero(0)
    e.stats(reset=True)
    e.batch_begin_int_groups(0)
    for g in groups:
        e.batch_next_int()
        for kind, job, g0, g1 in g:
            e.set_optical(job.ABS, job.SCA)
            e.set_scatter_table(job.DSC, job.CSC)
            if kind == 0:
                e.sim_pb(0, job.PACKETS, job.BATCH, job.SEED, job.BG, job.TW, PSPOS=job.PSPOS[:, :3], PS=job.PS, GLOBAL=job.GLOBAL, gid_first=g0, gid_count=g1 - g0)
            else:
                e.set_emission(job.EMIT, None)
                e.sim_cl(2, job.PACKETS, job.BATCH, job.SEED, job.TW, job.GLOBAL, gid_first=g0, gid_count=g1 - g0)
    assert e.last_passes() == 0
    e.batch_end()
    st = e.stats()
    assert e.last_form() == 3 and e.last_passes() > 0
    for key in stats:
        assert st[key] == stats[key]
    assert_tally_close(e.read_tally(0), tabs, rtol=1e-5)
    for k in range(3):
        assert_tally_close(e.batch_read_int(k), alone[k], rtol=1e-5)
    e.set_features(0, 0, 0)
    e.set_exec(-1, 4)


@pytest.mark.parametrize("kind", [0, 1])
def test_record_of_packets_entering_roi_in_the_sweep(kind, engine, oracle_soc, tuned):
    """-D WITH_ROI_SAVE (kernel_ASOC.c:615-642, :1510-1535) in the brick-local sweep: the walk sees a packet step from outside the region of
    interest into it (integer root-cell coordinates) and hands it to a fourth event queue of its launch, whose lanes add it to the record
    (surface element, Healpix direction, photons) and send it back; slow steps make the test themselves.  Same entries, same sums."""
    cl = cloud104()
    roi = [40, 63, 45, 70, 38, 60]
    if kind == 0:
        job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=1, BATCH=3, SEED=0.377, ROI=roi, ROI_STEP=2, ROI_NSIDE=2)
        g0, g1 = 100000, 106000
    else:
        emit = np.where(cl.DENS > 0, cl.DENS * 1e-3, 1e-4).astype(np.float32)
        job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=2, BATCH=1, SEED=0.9, GLOBAL=8192, EMIT=emit, ROI=roi, ROI_STEP=1, ROI_NSIDE=4)
        g0, g1 = 4000, 4064
    T, _, n = oracle_soc.sim(job, kind, gid0=g0, gid1=g1, nthreads=8)
    want = np.array(job.ROI_SAVE, np.float32).copy()
    for tune in (dict(), dict(slow_every=3)):
        tuned(**tune)
        Tg, _, st = _sweep(engine, job, kind, gid_first=g0, gid_count=g1 - g0)
        assert st["tally_events"] == n
        assert_tally_close(Tg, T, rtol=1e-5)
        assert want.sum() > 0 and np.array_equal(job.ROI_SAVE_gpu != 0, want != 0)
        assert_tally_close(job.ROI_SAVE_gpu, want, rtol=1e-5)
    engine.set_roi_save(None)
    engine.set_exec(-1, 4)


def test_loaded_roi_record_with_a_record_saved_in_the_sweep(engine, oracle_soc):
    """nested runs on a brick-local hierarchy: packets of a loaded region-of-interest record (SOURCE == 3, created by the event workgroups)
    while the packets entering another region are recorded (WITH_ROI_LOAD + WITH_ROI_SAVE, with the INT tally)"""
    cl = cloud104()
    dim, nside = (6, 6, 6), 2
    a = cases.roi_load(dim, nside)
    job = Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=3, PACKETS=a.shape[0], GLOBAL=100 * a.shape[0], BATCH=12 * nside * nside,
              ROI_LOAD=a, ROI_DIM=dim, ROI_NSIDE=nside, SEED=0.71, WITH_INT=1, TW=1.2, ROI=[30, 70, 35, 66, 40, 75], ROI_STEP=1)
    T, I, n = oracle_soc.sim(job, 0, nthreads=8)
    want = np.array(job.ROI_SAVE, np.float32).copy()
    Tg, Ig, st = _sweep(engine, job, 0)
    assert st["tally_events"] == n and st["packets"] == job.GLOBAL * job.BATCH
    assert_tally_close(Tg, T, rtol=1e-5)
    assert_tally_close(Ig, I, rtol=1e-5)
    assert want.sum() > 0 and np.array_equal(job.ROI_SAVE_gpu != 0, want != 0)
    assert_tally_close(job.ROI_SAVE_gpu, want, rtol=1e-5)
    engine.set_features(0, 0, 0)
    engine.set_exec(-1, 4)


@pytest.mark.parametrize("wint", [0, 1])
def test_ali_tally_in_the_sweep(wint, engine, oracle_soc, tuned):
    """-D WITH_ALI (kernel_ASOC.c:1394-1396, :1486-1494): what a SimRAM_CL packet deposits in the cell that emitted it goes to XAB, not TABS --
    in the walk by the cell numbers of the brick's slots in LDS, in the scattering block of the event workgroups by the cell index"""
    cl = cloud104()
    emit = np.where(cl.DENS > 0, cl.DENS * 1e-3, 1e-4).astype(np.float32)
    job = Job(cl, cases._CSC, ABS=3e-4, SCA=6e-4, SOURCE=2, BATCH=2, SEED=0.9, GLOBAL=8192, EMIT=emit, WITH_ALI=1, WITH_INT=wint, TW=1.3)
    g0, g1 = 4000, 4064
    T, I, n = oracle_soc.sim(job, 1, gid0=g0, gid1=g1, nthreads=8)
    want = np.array(job.XAB, np.float32).copy()
    assert want.sum() > 0
    for tune in (dict(), dict(slow_every=3)):
        tuned(**tune)
        Tg, Ig, st = _sweep(engine, job, 1, gid_first=g0, gid_count=g1 - g0)
        assert st["tally_events"] == n and st["scatterings"] > 100
        assert_tally_close(Tg, T, rtol=1e-5)
        assert_tally_close(job.XAB_gpu, want, rtol=1e-5)
        if wint:
            assert_tally_close(Ig, I, rtol=1e-5)
    engine.set_features(0, 0, 0)
    engine.set_exec(-1, 4)
