"""Brick-sweep execution (LDS tallies, packets sorted by brick) against the oracle and against
the direct kernel: same logical work items and RNG streams -> identical trajectories."""
import numpy as np
import pytest

import cases
from oracle.pyoracle import Job
from soc_amd import synth
from util import run_engine, assert_tally_close

pytestmark = pytest.mark.gpu

CART = [n for n, (ref, kind, mk) in sorted(cases.CASES.items()) if kind == 0 and "oct" not in n and "mirror" not in n and "roi" not in n and "int2" not in n]


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


@pytest.mark.parametrize("name", ["bg_oct8", "bg_oct4"])
@pytest.mark.parametrize("cap,hs", [(8, 0), (100, 0), (8192, 0), (100, 64), (8, 8)])
def test_brick_sweep_octree_matches_oracle(name, cap, hs, engine, oracle_soc, tuned):
    """hierarchies: bricks of <= cap leaves (cap 8 splits the subtree of a refined root cell of the
    known-answer tree), tally slots from the cell -> slot map; hs: hash-table form of the arrival counts
    (8 entries: overflows into the direct global count)"""
    tuned(brick_cells=cap, hash_slots=hs)
    ref, kind, mk = cases.CASES[name]
    job = mk()
    T, I, n = oracle_soc.sim(job, kind)
    Tg, Ig, st = run_engine(engine, job, kind, exec_mode=1)
    assert engine.last_passes() > 0
    assert st["tally_events"] == n, "trajectories diverged from the oracle"
    assert_tally_close(Tg, T, rtol=1e-5)
    engine.set_exec(-1, 4)


def test_brick_sweep_octree_abu_int_point_sources(engine, oracle_soc, tuned):
    """per-cell opacities, INT tally and point sources (inside and outside) on a hierarchy"""
    tuned(brick_cells=200)
    oct8 = cases._oct8()
    job = Job(oct8, cases._CSC, SOURCE=0, BATCH=25, SEED=0.2, GLOBAL=512, PSPOS=cases._PS_EXT, PS=[1.0, 2.0], PS_METHOD=0,
              XPS=cases._XPS2, OPT=cases._opt(oct8.CELLS), WITH_INT=1, TW=1.5)
    T, I, n = oracle_soc.sim(job, 0)
    Tg, Ig, st = run_engine(engine, job, 0, exec_mode=1)
    assert engine.last_passes() > 0 and st["tally_events"] == n
    assert_tally_close(Tg, T, rtol=1e-5)
    assert_tally_close(Ig, I, rtol=1e-5)
    engine.set_exec(-1, 4)


def test_brick_sweep_octree_double_index(engine, oracle_soc):
    """NX > 100 with >= 3 levels: Index() in double (DIMLIM), a range of the work items"""
    cl = synth.octree_cloud(104, levels=3, frac=0.03, seed=5)
    job = Job(cl, cases._CSC, ABS=2e-6, SCA=2e-5, SOURCE=1, BATCH=3, SEED=0.37)
    g0, g1 = 40000, 44000
    T, _, n = oracle_soc.sim(job, 0, gid0=g0, gid1=g1, nthreads=8)
    Tg, _, st = run_engine(engine, job, 0, gid_first=g0, gid_count=g1 - g0, exec_mode=1)
    assert engine.last_passes() > 0 and st["tally_events"] == n
    assert_tally_close(Tg, T, rtol=1e-5)
    Td, _, sd = run_engine(engine, job, 0, gid_first=g0, gid_count=g1 - g0, exec_mode=0)
    assert sd["tally_events"] == n
    engine.set_exec(-1, 4)


def test_brick_sweep_refuses_mirror_off_brick_local_hierarchies(engine):
    """reflecting faces in the sweep: the event workgroups of brick-local hierarchies handle them (tests/test_gpu_ltree.py); on
    Cartesian grids and hierarchies walked in global memory forcing the brick sweep is an error, automatic mode falls back"""
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


def test_deferred_launches_on_hierarchy(engine, oracle_soc):
    """automatic mode on a hierarchy: launches deferred inside soc_batch_begin/end share one brick sweep
    (a single one goes to the direct kernel); tallies = the oracle's sum of the launches"""
    cl = synth.octree_cloud(40, levels=3, frac=0.1, seed=3)       # 8*6*40^2 work items: above the automatic threshold
    d6, csc6 = synth.hg_scattering_table(0.6)
    d0, csc0 = synth.hg_scattering_table(0.1)
    jobs = [Job(cl, csc6, ABS=2e-5, SCA=6e-5, SOURCE=1, BATCH=2, SEED=0.3711, BG=1.0, TW=1.0),
            Job(cl, csc0, ABS=5e-5, SCA=2e-5, SOURCE=1, BATCH=1, SEED=0.11, BG=2.5, TW=0.5),
            Job(cl, csc6, ABS=1e-5, SCA=9e-5, SOURCE=1, BATCH=2, SEED=0.77, BG=0.7, TW=2.0)]
    T = np.zeros(cl.CELLS, np.float32)
    n = 0
    for j in jobs:
        _, _, k = oracle_soc.sim(j, 0, TABS=T, nthreads=8)
        n += k

    def run(njobs, batch):
        engine.set_cloud(cl)
        engine.set_features(0, 0, 0)
        engine.set_opt(None)
        engine.set_mirror(0)
        engine.set_exec(-1, 4)
        engine.zero(0)
        engine.stats(reset=True)
        if batch:
            engine.batch_begin(0)
        for j in jobs[:njobs]:
            engine.set_scatter_table(j.DSC, j.CSC)
            engine.set_optical(j.ABS, j.SCA)
            engine.sim_pb(1, j.PACKETS, j.BATCH, j.SEED, j.BG, j.TW, GLOBAL=j.GLOBAL)
        if batch:
            engine.batch_end()
        engine.sync()
        return engine.read_tally(0), engine.stats(), engine.last_passes()

    Tb, sb, pb = run(3, True)
    assert pb > 0, "deferred launches on a hierarchy did not use the brick sweep"
    assert sb["tally_events"] == n
    assert_tally_close(Tb, T, rtol=1e-5)
    _, s1, p1 = run(1, True)
    assert p1 == 0, "a single deferred launch on a hierarchy goes to the direct kernel"
    Td, sd, pd = run(3, False)
    assert pd == 0 and sd["tally_events"] == n
    assert_tally_close(Td, T, rtol=1e-5)


@pytest.mark.parametrize("octree", [False, True])
def test_deferred_launches_with_abundances(octree, engine, oracle_soc):
    """launches with per-cell opacities (a different OPT each, built on the device from abundances) share one
    sweep: every deferred launch keeps its own copy of OPT"""
    cl = synth.octree_cloud(40, levels=3, frac=0.1, seed=3) if octree else synth.cartesian_cloud(40, seed=21)
    rr = np.random.default_rng(5)
    ABU = rr.uniform(0.1, 1.0, (cl.CELLS, 2)).astype(np.float32)
    AF = [((2e-5, 1e-5), (6e-5, 2e-5)), ((1e-5, 4e-5), (3e-5, 5e-5)), ((3e-5, 2e-5), (2e-5, 8e-5))]
    _, csc = synth.hg_scattering_table(0.6)
    T = np.zeros(cl.CELLS, np.float32)
    n = 0
    for k, (fa, fs) in enumerate(AF):
        OPT = np.zeros((cl.CELLS, 2), np.float32)
        for d in range(2):
            OPT[:, 0] += ABU[:, d] * np.float32(fa[d])
            OPT[:, 1] += ABU[:, d] * np.float32(fs[d])
        _, _, m = oracle_soc.sim(Job(cl, csc, SOURCE=1, BATCH=2, SEED=0.2 + 0.3 * k, BG=1.0 + k, OPT=OPT), 0, TABS=T, nthreads=8)
        n += m
    engine.set_cloud(cl)
    engine.set_features(0, 0, 0)
    engine.set_mirror(0)
    engine.set_scatter_table(None, csc)
    engine.set_optical(0.0, 0.0)
    engine.set_abundances(ABU)
    engine.set_exec(-1, 4)
    engine.zero(0)
    engine.stats(reset=True)
    engine.batch_begin(0)
    for k, (fa, fs) in enumerate(AF):
        engine.set_optical_abu(fa, fs)
        engine.sim_pb(1, 0, 2, 0.2 + 0.3 * k, 1.0 + k, 1.0, GLOBAL=8 * cl.AREA)
    engine.batch_end()
    engine.sync()
    assert engine.last_passes() > 0, "the launches were not deferred into a brick sweep"
    assert engine.stats()["tally_events"] == n
    assert_tally_close(engine.read_tally(0), T, rtol=1e-5)
    engine.set_abundances(None)
    engine.set_opt(None)


def _clustered_hierarchy(NX, NY, NZ, levels, seed=2):
    """rectangular root grid; the corner cells (root coordinates < 3) and a few scattered ones are refined, and of
    every further level the first 216 cells again: a deep, clustered hierarchy whose first root cells carry
    subtrees of hundreds of leaves"""
    rr = np.random.default_rng(seed)
    H = [rr.uniform(500.0, 2000.0, NX * NY * NZ)]
    for l in range(levels - 1):
        cur = H[l]
        if l == 0:
            x, y, z = np.arange(NX * NY * NZ) % NX, (np.arange(NX * NY * NZ) // NX) % NY, np.arange(NX * NY * NZ) // (NX * NY)
            sel = (x < 3) & (y < 3) & (z < 3)
            sel |= rr.uniform(size=sel.size) < 0.05
        else:
            sel = np.ones(cur.size, bool)
            sel[8 * 27:] = False                                 # children of the scattered cells stay leaves
        parents = np.flatnonzero(sel)
        kids = (cur[parents][:, None] * rr.uniform(0.7, 1.3, (parents.size, 8))).reshape(-1)
        H.append(kids)
        cur = cur.astype(np.float32)
        cur[parents] = -synth.I2F((8 * np.arange(parents.size)).astype(np.int32)).astype(np.float32)
        H[l] = cur
    return synth.Cloud(NX, NY, NZ, [np.asarray(h, np.float32) for h in H])


@pytest.mark.parametrize("cap", [8, 40, 700])
def test_brick_sweep_on_a_clustered_rectangular_hierarchy(cap, engine, oracle_soc, tuned):
    """brick builder: root grid not a multiple of the 16-cell cube, subtrees far above the cap (split level by
    level), bricks spanning several cubes"""
    tuned(brick_cells=cap)
    cl = _clustered_hierarchy(19, 7, 5, 4)
    assert cl.LEVELS == 4 and cl.LCELLS[2] == 1728 and cl.LCELLS[3] == 1728
    job = Job(cl, cases._CSC, ABS=2e-5, SCA=6e-5, SOURCE=1, BATCH=3, SEED=0.3711)
    T, _, n = oracle_soc.sim(job, 0, nthreads=8)
    Tg, _, st = run_engine(engine, job, 0, exec_mode=1)
    assert engine.last_passes() > 0 and st["tally_events"] == n
    assert_tally_close(Tg, T, rtol=1e-5)
    assert (Tg[cl.DENS <= 0] == 0).all()
    engine.set_exec(-1, 4)


HP = [n for n, (ref, kind, mk) in sorted(cases.CASES.items()) if kind == 2 and "mirror" not in n and "int2" not in n]


@pytest.mark.parametrize("name", HP)
def test_brick_sweep_healpix_background(name, engine, oracle_soc, tuned):
    """SimRAM_HP through the brick sweep: the walk is SimRAM_PB's, the event workgroups create the packets from the
    Healpix sky (uniform and weighted pixel selection, Cartesian and hierarchy, with the INT tally)"""
    tuned(brick_cells=300)
    ref, kind, mk = cases.CASES[name]
    job = mk()
    T, I, n = oracle_soc.sim(job, kind)
    Tg, Ig, st = run_engine(engine, job, kind, exec_mode=1, brick_log2=2)
    assert engine.last_passes() > 0
    assert st["tally_events"] == n, "trajectories diverged from the oracle"
    assert_tally_close(Tg, T, rtol=1e-5)
    if job.WITH_INT:
        assert_tally_close(Ig, I, rtol=1e-5)
    engine.set_exec(-1, 4)


def test_deferred_healpix_launches(engine, oracle_soc):
    """two SimRAM_HP launches with different skies (one weighted) and a SimRAM_PB one in the same sweep"""
    cl = synth.cartesian_cloud(40, seed=21)
    _, csc = synth.hg_scattering_table(0.6)
    skyA, _ = cases.hp_sky(weighted=False)
    skyB, PB = cases.hp_sky(weighted=True)
    G = 8 * cl.AREA
    jobs = [(2, Job(cl, csc, ABS=2e-5, SCA=6e-5, BATCH=2, SEED=0.21, TW=1.0, GLOBAL=G, HPBG=skyA)),
            (0, Job(cl, csc, ABS=3e-5, SCA=5e-5, SOURCE=1, BATCH=1, SEED=0.43, BG=2.0, TW=0.5, GLOBAL=G)),
            (2, Job(cl, csc, ABS=1e-5, SCA=8e-5, BATCH=1, SEED=0.65, TW=2.0, GLOBAL=G, HPBG=0.5 * skyB, HPBGP=PB))]
    T = np.zeros(cl.CELLS, np.float32)
    n = 0
    for kind, j in jobs:
        _, _, m = oracle_soc.sim(j, kind, TABS=T, nthreads=8)
        n += m
    engine.set_cloud(cl)
    engine.set_features(0, 0, 0)
    engine.set_opt(None)
    engine.set_mirror(0)
    engine.set_exec(-1, 4)
    engine.set_scatter_table(None, csc)
    engine.zero(0)
    engine.stats(reset=True)
    engine.batch_begin(0)
    for kind, j in jobs:
        engine.set_optical(j.ABS, j.SCA)
        if kind == 2:
            engine.set_hpbg(j.HPBG, j.HPBGP)
            engine.sim_hp(0, j.BATCH, j.SEED, j.TW, G)
        else:
            engine.sim_pb(1, 0, j.BATCH, j.SEED, j.BG, j.TW, GLOBAL=G)
    engine.batch_end()
    engine.sync()
    assert engine.last_passes() > 0 and engine.stats()["tally_events"] == n
    assert_tally_close(engine.read_tally(0), T, rtol=1e-5)


@pytest.mark.parametrize("name", ["bg_oct8_sw2", "bg_oct8_msf", "ps_in_oct8"])
def test_brick_sweep_octree_weighting_and_species(name, engine, oracle_soc, tuned):
    """-D STEP_WEIGHT and -D WITH_MSF in the sweep's event workgroups (hierarchy in global memory; per-cell opacities)"""
    tuned(brick_cells=100)
    ref, kind, mk = cases.CASES[name]
    job = mk()
    T, I, n = oracle_soc.sim(job, kind)
    Tg, Ig, st = run_engine(engine, job, kind, exec_mode=1)
    assert engine.last_passes() > 0
    assert st["tally_events"] == n, "trajectories diverged from the oracle"
    assert_tally_close(Tg, T, rtol=1e-5)
    engine.set_exec(-1, 4)


CLB = [n for n, (ref, kind, mk) in sorted(cases.CASES.items())
       if kind == 1 and not any(t in n for t in ("mirror", "emw2", "ali", "roi", "int2"))]


@pytest.mark.parametrize("name", CLB)
def test_brick_sweep_cell_emission(name, engine, oracle_soc, tuned):
    """SimRAM_CL through the brick sweep: the event workgroups step through the work item's cells (EMWEIGHT 0 and 1),
    no nudge after a failed step, packets dropped before the deposit of the 21st scattering"""
    tuned(brick_cells=300)
    ref, kind, mk = cases.CASES[name]
    job = mk()
    T, I, n = oracle_soc.sim(job, kind)
    Tg, Ig, st = run_engine(engine, job, kind, exec_mode=1, brick_log2=2)
    assert engine.last_passes() > 0
    assert st["tally_events"] == n, "trajectories diverged from the oracle"
    assert_tally_close(Tg, T, rtol=1e-5)
    engine.set_exec(-1, 4)


@pytest.mark.parametrize("octree", [False, True])
def test_cell_emission_large_global_and_deferred(octree, engine, oracle_soc):
    """`global` raised to the cell count: automatic mode takes the brick sweep; two frequencies (own EMIT, optical
    depths, scattering tables) deferred into one sweep equal the oracle's sum"""
    cl = synth.octree_cloud(56, levels=3, frac=0.1, seed=3) if octree else synth.cartesian_cloud(64, seed=21)   # >= 262144 cells
    rr = np.random.default_rng(6)
    d6, csc6 = synth.hg_scattering_table(0.6)
    d0, csc0 = synth.hg_scattering_table(0.1)
    leaf = cl.DENS > 0
    G = launch_fix(cl.CELLS)
    jobs = []
    for k, (csc, a, s) in enumerate(((csc6, 2e-5, 6e-5), (csc0, 4e-5, 3e-5))):
        EMIT = np.where(leaf, rr.uniform(0.5, 2.0, cl.CELLS), 0.0).astype(np.float32)
        jobs.append(Job(cl, csc, ABS=a, SCA=s, SOURCE=2, BATCH=1 + k, SEED=0.2 + 0.4 * k, TW=1.0 + k, GLOBAL=G, EMIT=EMIT))
    T = np.zeros(cl.CELLS, np.float32)
    n = 0
    for j in jobs:
        _, _, m = oracle_soc.sim(j, 1, TABS=T, nthreads=8)
        n += m
    engine.set_cloud(cl)
    engine.set_features(0, 0, 0)
    engine.set_opt(None)
    engine.set_mirror(0)
    engine.set_ali(0)
    engine.set_exec(-1, 4)
    engine.zero(0)
    engine.stats(reset=True)
    engine.batch_begin(0)
    for j in jobs:
        engine.set_scatter_table(j.DSC, j.CSC)
        engine.set_optical(j.ABS, j.SCA)
        engine.set_emission(j.EMIT, None)
        engine.sim_cl(2, 0, j.BATCH, j.SEED, j.TW, G)
    engine.batch_end()
    engine.sync()
    assert engine.last_passes() > 0, "cell emission with GLOBAL ~ CELLS did not take the brick sweep"
    assert engine.stats()["tally_events"] == n
    assert_tally_close(engine.read_tally(0), T, rtol=1e-5)


def launch_fix(n):
    from soc_amd import launch
    return launch.Fix(n, 64)


@pytest.mark.parametrize("octree", [False, True])
def test_deferred_launches_with_int_tally(octree, engine, oracle_soc):
    """soc_batch_begin_int: launches that keep the per-frequency INT tally share a sweep -- brick queues per launch, an
    INT tally per launch, TABS shared; every INT equals the oracle's for that launch"""
    cl = synth.octree_cloud(40, levels=3, frac=0.1, seed=3) if octree else synth.cartesian_cloud(40, seed=21)
    d6, csc6 = synth.hg_scattering_table(0.6)
    d0, csc0 = synth.hg_scattering_table(0.1)
    G = 8 * cl.AREA
    jobs = [Job(cl, csc6, ABS=2e-5, SCA=6e-5, SOURCE=1, BATCH=2, SEED=0.3711, BG=1.0, TW=1.0, WITH_INT=1),
            Job(cl, csc0, ABS=5e-5, SCA=2e-5, SOURCE=1, BATCH=1, SEED=0.11, BG=2.5, TW=0.5, WITH_INT=1),
            Job(cl, csc6, ABS=1e-5, SCA=9e-5, SOURCE=1, BATCH=2, SEED=0.77, BG=0.7, TW=2.0, WITH_INT=1)]
    T = np.zeros(cl.CELLS, np.float32)
    INT, n = [], 0
    for j in jobs:
        I = np.zeros(cl.CELLS, np.float32)
        _, _, m = oracle_soc.sim(j, 0, TABS=T, INT=I, nthreads=8)
        INT.append(I)
        n += m
    engine.set_cloud(cl)
    engine.set_features(1, 0, 0)
    engine.set_opt(None)
    engine.set_mirror(0)
    engine.set_exec(-1, 4)
    engine.zero(0)
    engine.zero(1)
    engine.stats(reset=True)
    engine.batch_begin_int(3)
    for j in jobs:
        engine.set_scatter_table(j.DSC, j.CSC)
        engine.set_optical(j.ABS, j.SCA)
        engine.sim_pb(1, j.PACKETS, j.BATCH, j.SEED, j.BG, j.TW, GLOBAL=G)
    from soc_amd.lib import SocError
    with pytest.raises(SocError, match="read"):            # a fourth launch would have no INT slot to keep
        engine.sim_pb(1, 0, 1, 0.5, 1.0, 1.0, GLOBAL=G)
    engine.batch_end()
    engine.sync()
    assert engine.last_passes() > 0 and engine.stats()["tally_events"] == n
    assert_tally_close(engine.read_tally(0), T, rtol=1e-5)
    for k in range(3):
        assert_tally_close(engine.batch_read_int(k), INT[k], rtol=1e-5)
    assert not engine.read_tally(1).any()                   # the shared INT tally was not touched
    engine.set_features(0, 0, 0)


def test_absorbed_file_run_with_int_batches(engine, tmp_path):
    """asoc.py with an absorbed file (per-frequency INT tallies) on a Cartesian model: the frequencies of a source
    block go through soc_batch_begin_int; file and tallies equal the oracle engine's run"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_host import _write_model
    from oracle_engine import OracleEngine
    from soc_amd import files
    from soc_amd.asoc import AbsorptionRun
    from soc_amd.ini import User
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(40, seed=5)
    ini = _write_model(d, cloud, extra="gridlength 2e-6\nbgpackets 300000\n")
    os.chdir(d)

    class Threaded(OracleEngine):
        threads = 8
    Cw, Fw = AbsorptionRun(User(ini), Threaded("soc"), verbose=0).run()
    want = files.read_absorbed(os.path.join(d, "abs.data")).copy()
    passes = []
    real = engine.batch_end

    def spy():
        real()
        passes.append(engine.last_passes())
    engine.batch_end = spy
    try:
        Cg, Fg = AbsorptionRun(User(ini), engine, verbose=0).run()
    finally:
        engine.batch_end = real
    assert passes and max(passes) > 0, "the INT batches did not go through the brick sweep"
    got = files.read_absorbed(os.path.join(d, "abs.data"))
    assert_tally_close(Cg, Cw, rtol=1e-5)
    assert np.allclose(got, want, rtol=2e-5, atol=1e-6 * np.abs(want).max())
    engine.set_features(0, 0, 0)


@pytest.mark.parametrize("lb", [2, 3])
def test_packets_of_a_loaded_roi_record_in_the_sweep(lb, engine, oracle_soc):
    """SOURCE == 3 (-D WITH_ROI_LOAD, kernel_ASOC.c:141-179, :469-501) created by the event workgroups of the Cartesian sweep"""
    ref, kind, mk = cases.CASES["roi_c8_load"]
    job = mk()
    T, I, n = oracle_soc.sim(job, kind)
    Tg, Ig, st = run_engine(engine, job, kind, exec_mode=1, brick_log2=lb)
    assert engine.last_passes() > 0 and engine.last_form() == 1
    assert st["tally_events"] == n and st["packets"] == job.GLOBAL * job.BATCH
    assert_tally_close(Tg, T, rtol=1e-5)
