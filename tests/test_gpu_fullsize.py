"""Size-independent properties at the bench configuration (128^3, ~1e8 packets): the oracle
cannot run this size, so the checks are invariants of the algorithm."""
import numpy as np
import pytest

from soc_amd import synth

pytestmark = pytest.mark.gpu

N = 128
ABS, SCA = 8.9084e-7, 5.4552e-6          # tmp.dust row 33 at GL = 0.01 pc


@pytest.fixture(scope="module")
def big(engine):
    cloud = synth.cartesian_cloud(N, seed=1234)
    _, csc = synth.hg_scattering_table(0.6)
    engine.set_cloud(cloud)
    engine.set_features(0, 0, 0)
    engine.set_mirror(0)
    engine.set_exec(-1, 4)
    engine.set_scatter_table(None, csc)
    engine.set_optical(ABS, SCA)
    engine.set_opt(None)
    return cloud


def run(engine, cloud, BATCH, SEED, BG=1.0, TW=1.0, first=0, count=None, zero=True):
    G = 8 * cloud.AREA
    if zero:
        engine.zero(0)
    engine.stats(reset=True)
    engine.sim_pb(1, 0, BATCH, SEED, BG, TW, GLOBAL=G, gid_first=first, gid_count=count)
    return engine.read_tally(0), engine.stats()


def test_full_size_invariants(engine, big):
    T, st = run(engine, big, 127, 0.6004384)
    assert st["packets"] == 8 * big.AREA * 127 == 99876864           # SURVEY.md 8(c) launch arithmetic
    assert np.isfinite(T).all() and (T >= 0).all() and (T > 0).all()
    # identical trajectories on a re-run: integer event counts equal, tallies equal to summation order
    T2, st2 = run(engine, big, 127, 0.6004384)
    assert st2 == st
    assert np.allclose(T, T2, rtol=2e-5, atol=0)
    # linearity in the packet weight and the frequency weight (exact powers of two: bit-for-bit per add)
    T3, st3 = run(engine, big, 127, 0.6004384, BG=4.0, TW=0.5)
    assert st3 == st
    assert np.allclose(T3, 2.0 * T, rtol=2e-5, atol=0)
    # mean number of cell steps per packet ~ N (mean chord 2N/3 x mean |dx|+|dy|+|dz| = 3/2), SURVEY.md 7.4
    assert 0.9 * N < st["tally_events"] / st["packets"] < 1.15 * N
    # absorbed energy: optically thin estimate  sum(TABS) ~ packets * BG * <tau_abs along the path>
    mean_rho = big.DENS.mean(dtype=np.float64)
    est = st["packets"] * ABS * mean_rho * (2.0 * N / 3.0)
    assert abs(T.sum(dtype=np.float64) / est - 1) < 0.15


def test_sharding_and_seed_change(engine, big):
    G = 8 * big.AREA
    Tw, stw = run(engine, big, 8, 0.25)
    h = G // 2 + 64
    Ta, sta = run(engine, big, 8, 0.25, first=0, count=h)
    Tb, stb = run(engine, big, 8, 0.25, first=h, count=G - h, zero=False)
    assert sta["tally_events"] + stb["tally_events"] == stw["tally_events"]
    assert np.allclose(Tb, Tw, rtol=2e-5, atol=0)
    # another seed: statistically the same field, different packets
    Ts, sts = run(engine, big, 8, 0.75)
    assert sts["tally_events"] != stw["tally_events"]
    assert abs(Ts.sum(dtype=np.float64) / Tw.sum(dtype=np.float64) - 1) < 5e-3


def test_full_size_direct_kernel_equals_brick_sweep(engine, big):
    """the two execution modes run the same packets: integer event counts equal, tallies to summation order;
    and four deferred launches in one sweep equal the four run one after the other"""
    engine.set_exec(0, 4)
    Td, sd = run(engine, big, 16, 0.41)
    assert engine.last_passes() == 0
    engine.set_exec(1, 4)
    Tb, sb = run(engine, big, 16, 0.41)
    assert engine.last_passes() > 0
    assert sb == sd
    assert np.allclose(Tb, Td, rtol=2e-5, atol=0)
    G = 8 * big.AREA
    engine.set_exec(-1, 4)
    engine.zero(0)
    engine.stats(reset=True)
    engine.batch_begin(0)
    for k in range(4):
        engine.sim_pb(1, 0, 8, 0.1 + 0.2 * k, 1.0 + k, 1.0, GLOBAL=G)
    engine.batch_end()
    Tq, sq = engine.read_tally(0), engine.stats()
    engine.zero(0)
    engine.stats(reset=True)
    for k in range(4):
        engine.sim_pb(1, 0, 8, 0.1 + 0.2 * k, 1.0 + k, 1.0, GLOBAL=G)
    Ts, ss = engine.read_tally(0), engine.stats()
    assert sq == ss
    assert np.allclose(Tq, Ts, rtol=2e-5, atol=0)


def test_full_size_hierarchy_direct_equals_brick_sweep(engine):
    """BASELINE configs[2] geometry (256^3 roots, 4 levels, Index in double): 3.1e6 work items through the
    direct kernel, the brick sweep, and two launches deferred into one sweep"""
    cloud = synth.octree_cloud(256, levels=4, frac=0.10, seed=1234)
    _, csc = synth.hg_scattering_table(0.6)
    engine.set_cloud(cloud)
    engine.set_features(0, 0, 0)
    engine.set_mirror(0)
    engine.set_scatter_table(None, csc)
    engine.set_optical(0.5 * ABS, 0.5 * SCA)
    engine.set_opt(None)
    G = 8 * cloud.AREA
    engine.set_exec(0, 4)
    Td, sd = run(engine, cloud, 2, 0.41)
    assert sd["packets"] == 2 * G
    steps = sd["tally_events"] / sd["packets"]
    assert 256 < steps < 330                                     # ~N for the root grid + the refined cells on the way
    engine.set_exec(1, 4)
    Tb, sb = run(engine, cloud, 2, 0.41)
    assert engine.last_passes() > 0
    assert sb == sd
    leaf = cloud.DENS > 0
    assert (Tb[~leaf] == 0).all() and (Td[~leaf] == 0).all()    # refined cells never receive a tally
    assert np.allclose(Tb[leaf], Td[leaf], rtol=2e-5, atol=1e-6 * Td.max())
    engine.set_exec(-1, 4)
    engine.zero(0)
    engine.stats(reset=True)
    engine.batch_begin(0)
    engine.sim_pb(1, 0, 2, 0.41, 1.0, 1.0, GLOBAL=G)
    engine.sim_pb(1, 0, 1, 0.77, 3.0, 1.0, GLOBAL=G)
    engine.batch_end()
    Tq, sq = engine.read_tally(0), engine.stats()
    assert engine.last_passes() > 0
    engine.set_exec(0, 4)
    T2, s2 = run(engine, cloud, 1, 0.77, BG=3.0)
    assert sq["tally_events"] == sd["tally_events"] + s2["tally_events"] and sq["packets"] == 3 * G
    assert np.allclose(Tq[leaf], (Td + T2)[leaf], rtol=2e-5, atol=1e-6 * Td.max())
    engine.set_exec(-1, 4)


@pytest.fixture(scope="module")
def c3(engine):
    """BASELINE configs[2] as bench.py states it: 256^3-root octree, 4 levels, 50-frequency dust table, point source + diffuse"""
    import bench
    return bench.c3_workload(4194304)


def _c3_launch(engine, work, i, first, count, batch=None):
    from soc_amd import launch
    s = work["step"](i)
    L = s["L"]
    engine.set_optical(s["ABS"], s["SCA"])
    engine.set_scatter_table(s["DSC"], s["CSC"])
    seed = launch.launch_seed(work["SEED"], s["IFREQ"])
    if s["kind"] == "ps":
        engine.sim_pb(0, 0, batch or L["BATCH"], seed, 0.0, s["TW"], PSPOS=s["PSPOS"], PS=s["PS"], GLOBAL=L["GLOBAL"], gid_first=first, gid_count=count)
    else:
        engine.set_emission(s["EMIT"], None)
        engine.sim_cl(2, 0, batch or L["BATCH"], seed, s["TW"], L["GLOBAL"], gid_first=first, gid_count=count)


@pytest.mark.parametrize("step,first,count,batch", [(60, 0, 4194304, 4), (61, 1000000, 1200000, 2)])
def test_c3_point_source_and_diffuse_launches_direct_equals_sweep(step, first, count, batch, engine, c3):
    """the stated workload of config 3 on its geometry: a point-source launch (all packets born in one cell: the hot
    brick) and a diffuse-emission launch (packets from every cell, refined ones included), frequency 30 of the table,
    with the per-frequency INT tally: brick sweep (brick-local hierarchies) == direct kernel -- integer event counts
    equal, both tallies equal to summation order"""
    cloud = c3["cloud"]
    engine.set_cloud(cloud)
    engine.set_features(1, 0, 0)
    engine.set_mirror(0)
    engine.set_opt(None)
    out = {}
    for mode in (0, 1):
        engine.set_exec(mode, 4)
        engine.zero(0)
        engine.zero(1)
        engine.stats(reset=True)
        _c3_launch(engine, c3, step, first, count, batch)
        out[mode] = (engine.read_tally(0), engine.read_tally(1), engine.stats(), engine.last_form())
    assert out[0][3] == 0 and out[1][3] == 3
    assert out[0][2] == out[1][2] and out[0][2]["packets"] > 2e6
    for k in (0, 1):
        # same packets, same events; the order of the fp32 additions differs.  Cells next to the point source take
        # millions of additions each (4 packets x 4194304 work items start in ONE cell), where one atomic after the
        # other (1.7e7 fp32 additions into one number lose its last digits) and per-brick partial sums differ by more
        # than anywhere else: 2e-5 for 99.9 % of the cells, a handful of cells around the source beyond 1e-4, 5 % at worst
        a, b = np.asarray(out[0][k], np.float64), np.asarray(out[1][k], np.float64)
        rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-6 * np.abs(a).max())
        assert np.quantile(rel, 0.999) < 2e-5 and (rel > 1e-4).sum() < 200 and rel.max() < 5e-2, (np.quantile(rel, 0.999), (rel > 1e-4).sum(), rel.max())
        assert abs(a.sum() / b.sum() - 1.0) < 5e-3               # (the source cell alone holds a large share of the total)
    T, I = out[1][0], out[1][1]
    assert np.isfinite(T).all() and (T >= 0).all() and T.sum() > 0
    assert (I[cloud.DENS <= 0] == 0).all()      # (the workload's refined cells emit nothing: their packets carry no photons)
    engine.set_features(0, 0, 0)
    engine.set_exec(-1, 4)


def test_c3_source_cell_against_fp64_partial_sums(engine, c3):
    """Which execution mode is right where they differ by percents: the cell of the point source, into which every packet of a
    launch makes its first deposit (1.7e7 fp32 additions into ONE number in the direct kernel).  Reference: the same launch in 128
    work-item ranges through the direct kernel, the tally read after each range (<= 1.3e5 additions into a zeroed cell: accurate to
    ~1e-6) and added up on the host in fp64.  The brick sweep adds workgroup partial sums (<= 16384 packets each, in LDS) and must
    agree with that reference to 1e-5 in the hot cells; the one-launch direct kernel is the one that is off (a float cannot take
    1.7e7 increments of its own 2^-24: they fall below half an ulp)."""
    cloud = c3["cloud"]
    engine.set_cloud(cloud)
    engine.set_features(0, 0, 0)
    engine.set_mirror(0)
    engine.set_opt(None)
    step, G, batch = 60, 4194304, 4
    out = {}
    for mode in (0, 1):
        engine.set_exec(mode, 4)
        engine.zero(0)
        _c3_launch(engine, c3, step, 0, G, batch)
        out[mode] = np.asarray(engine.read_tally(0), np.float64)
    hot = np.argsort(out[1])[-8:]                                  # the source cell and the cells around it
    ref = np.zeros(hot.size, np.float64)
    engine.set_exec(0, 4)
    nr = 128
    for r in range(nr):
        engine.zero(0)
        _c3_launch(engine, c3, step, r * (G // nr), G // nr, batch)
        ref += np.asarray(engine.read_tally(0), np.float64)[hot]
    err_sweep = np.abs(out[1][hot] - ref) / ref
    err_direct = np.abs(out[0][hot] - ref) / ref
    assert err_sweep.max() < 1e-5, (err_sweep, err_direct)
    assert err_direct.max() > 10 * err_sweep.max()                 # the direct kernel's single accumulator is where the percents come from
    engine.set_exec(-1, 4)


def test_c3_lone_launches_with_int_use_the_sweep_in_automatic_mode(engine, c3):
    """automatic mode: on this hierarchy a lone launch of >= 1e6 work items goes through the brick sweep, with the INT tally"""
    cloud = c3["cloud"]
    engine.set_cloud(cloud)
    engine.set_features(1, 0, 0)
    engine.set_opt(None)
    engine.set_exec(-1, 4)
    engine.zero(0)
    engine.zero(1)
    _c3_launch(engine, c3, 60, 0, 4194304, 1)
    assert engine.last_passes() > 0 and engine.last_form() == 3
    _c3_launch(engine, c3, 60, 0, 65536, 1)
    engine.sync()
    assert engine.last_passes() == 0
    engine.set_features(0, 0, 0)


def test_c1_known_answer(engine):
    """BASELINE.md section 2 / SURVEY.md 8(c): 32^3, uniform density 1e3, tmp.dust row 33, BG = TW = 1, SEED = 0.6004384,
    GLOBAL 49152 x BATCH 20 = 983040 packets: the reference kernel (x86 build) gave sum(TABS) = 1.844163e+04 and 32.1 tally
    events per packet.  Our transcendentals differ from libm in the last bit, so the comparison is statistical."""
    cloud = synth.cartesian_cloud(32, uniform=1.0e3)
    import zlib                                                    # noqa: F401  (keep the import block honest)
    _, csc = synth.hg_scattering_table(0.6)
    engine.set_cloud(cloud)
    engine.set_features(0, 0, 0)
    engine.set_mirror(0)
    engine.set_opt(None)
    engine.set_scatter_table(None, csc)
    engine.set_optical(ABS, SCA)
    for mode in (0, 1):
        engine.set_exec(mode, 4)
        engine.zero(0)
        engine.stats(reset=True)
        engine.sim_pb(1, 983040, 20, 0.6004384, 1.0, 1.0, GLOBAL=49152)
        T, st = engine.read_tally(0), engine.stats()
        assert st["packets"] == 983040
        assert abs(st["tally_events"] / st["packets"] - 32.1) < 0.15
        assert abs(T.sum(dtype=np.float64) / 1.844163e4 - 1.0) < 5e-3
        assert 0.44 < T.min() and T.max() < 0.70                   # reference: min 0.4785, max 0.6499
    engine.set_exec(-1, 4)


def test_c4_scattered_light_on_the_c3_geometry(engine, c3):
    """BASELINE configs[3] at its stated size: the scattered-light kernels (forced first scattering, peel-off towards three
    observers, 256^2 pixels) on the 256^3-root octree -- background, point-source and cell-emission launches.  No oracle at
    this size; the properties that hold for any size: work-item ranges add up to the whole launch (same streams, integer
    counts equal), images are linear in the source strength (a factor two is exact in fp32), nothing but finite,
    non-negative pixels."""
    import math
    from soc_amd import launch
    cloud = c3["cloud"]
    N = cloud.NX
    s = c3["step"](60)
    engine.set_cloud(cloud)
    engine.set_features(0, 0, 0)
    engine.set_mirror(0)
    engine.set_opt(None)
    engine.set_optical(s["ABS"], s["SCA"])
    engine.set_scatter_table(s["DSC"], s["CSC"])
    th = [math.radians(30 + 25 * i) for i in range(3)]
    ph = [math.radians(40 * i) for i in range(3)]
    _, OD, RA, DE = launch.set_observer_directions(th, ph)
    engine.sca_set_view(OD, RA, DE, (256, 256), 1.5, (N / 2, N / 2, N / 2), 1)
    AREA = 6 * N * N
    GLOBAL = launch.Fix(8 * AREA, 64)
    emit = np.where(cloud.DENS > 0, 1.0e-3 * cloud.DENS, 0).astype(np.float32)

    def shoot(kind, scale, first, count):
        engine.stats(reset=True)
        if kind == "bg":
            engine.sca_sim_pb(1, 8 * AREA, 1, 0.31, scale, GLOBAL=GLOBAL, gid_first=first, gid_count=count)
        elif kind == "ps":
            engine.sca_sim_ps(1048576 * 2, 2, 0.32, 0.0, s["PSPOS"], [scale * float(s["PS"][0])], GLOBAL=1048576, gid_first=first, gid_count=count)
        else:
            engine.set_emission(emit * np.float32(scale))
            engine.sca_sim_cl(2, cloud.CELLS, 1, 0.33, 1048576, gid_first=first, gid_count=count)
        engine.sync()
        return engine.stats()
    for kind, G in (("bg", GLOBAL), ("ps", 1048576), ("cl", 1048576)):
        engine.sca_zero()
        st = shoot(kind, 1.0, 0, G)
        whole = engine.sca_read_out()
        assert np.isfinite(whole).all() and (whole >= 0).all() and whole.sum() > 0 and st["tally_events"] > 1e5, kind
        if kind == "bg":
            assert st["packets"] == 8 * AREA
        elif kind == "cl":
            assert st["packets"] == cloud.CELLS
        # two work-item ranges into one image
        engine.sca_zero()
        cut = G // 3 + 11
        sa = shoot(kind, 1.0, 0, cut)
        sb = shoot(kind, 1.0, cut, G - cut)
        both = engine.sca_read_out()
        assert all(sa[k] + sb[k] == st[k] for k in ("packets", "tally_events", "scatterings")), kind
        tol = 2e-5 * np.abs(whole) + 1e-9 * whole.max()
        assert (np.abs(both - whole) <= tol).mean() > 0.9999 and abs(both.sum(dtype=np.float64) / whole.sum(dtype=np.float64) - 1) < 1e-6, kind
        # linearity
        engine.sca_zero()
        st2 = shoot(kind, 2.0, 0, G)
        twice = engine.sca_read_out()
        assert st2 == st, kind
        assert (np.abs(twice - 2.0 * whole) <= 2.0 * tol).mean() > 0.9999, kind
    engine.set_emission(emit)


def test_c4_rays_in_one_batch_equal_the_direct_kernel(engine, c3):
    """Config 4 as soc_amd.asocs runs it on this geometry: the launches of a source block deferred into ONE sweep of rays on the
    brick-local hierarchies (soc_batch_begin ... soc_batch_end, an image per frequency).  No oracle at this size; the direct kernel
    (held to the oracle at oracle sizes, tests/test_gpu_sca.py) is the witness: integer counts equal, every image to summation order."""
    import math
    from soc_amd import launch
    cloud = c3["cloud"]
    N = cloud.NX
    engine.set_cloud(cloud)
    engine.set_features(0, 0, 0)
    engine.set_mirror(0)
    engine.set_opt(None)
    th = [math.radians(30 + 25 * i) for i in range(3)]
    ph = [math.radians(40 * i) for i in range(3)]
    _, OD, RA, DE = launch.set_observer_directions(th, ph)
    engine.sca_set_view(OD, RA, DE, (256, 256), 1.5, (N / 2, N / 2, N / 2), 1)
    AREA = 6 * N * N
    GLOBAL = launch.Fix(8 * AREA, 64)
    emit = np.where(cloud.DENS > 0, 1.0e-3 * cloud.DENS, 0).astype(np.float32)
    steps = [c3["step"](k) for k in (40, 60, 90)]                 # three frequencies: own opacities and scattering functions

    def launches(f, s):
        engine.set_optical(s["ABS"], s["SCA"])
        engine.set_scatter_table(s["DSC"], s["CSC"])
        engine.sca_sim_pb(1, 8 * AREA, 1, 0.31 + 0.1 * f, 1.0 + f, GLOBAL=GLOBAL)
        engine.sca_sim_ps(1048576 * 2, 2, 0.32 + 0.1 * f, 0.0, s["PSPOS"], [float(s["PS"][0])], GLOBAL=1048576)
        engine.set_emission(emit * np.float32(1 + f))
        engine.sca_sim_cl(2, cloud.CELLS, 1, 0.33 + 0.1 * f, 4194304)

    res = {}
    for mode in (0, 1):
        engine.set_exec(mode, 4)
        engine.stats(reset=True)
        engine.batch_begin(0)
        engine.sca_batch_images(len(steps))
        for f, s in enumerate(steps):
            engine.sca_batch_select(f)
            launches(f, s)
        engine.batch_end()
        st = engine.stats()
        assert (engine.last_form() == 3) == (mode == 1)
        res[mode] = (st, [engine.sca_batch_read(f) for f in range(len(steps))])
        engine.sca_batch_images(0)
    engine.set_exec(-1, 4)
    assert res[0][0] == res[1][0] and res[0][0]["packets"] == 3 * (8 * AREA + 2 * 1048576 + cloud.CELLS)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.isfinite(b).all() and b.sum() > 0
        tol = 2e-5 * np.abs(a) + 1e-9 * a.max()
        assert (np.abs(b - a) <= tol).mean() > 0.9999 and abs(b.sum(dtype=np.float64) / a.sum(dtype=np.float64) - 1) < 1e-6
    engine.set_emission(emit)


def test_c5_stochastic_heating_at_its_stated_size(engine, c3, oracle_soc):
    """BASELINE configs[4] at its stated size: 128 enthalpy bins x 50 frequencies, absorptions of 65536 cells taken from a
    point-source launch on the config-3 geometry (so that the dynamic range is the real one: cells next to the source
    down to cells that saw a few packets).  Cells are independent; a sample of them -- spread over the range, including
    the brightest and the faintest -- must equal the CPU oracle bit for bit, and the cell batch must not matter."""
    from oracle.pyoracle import a2e_oracle_dosolve
    from soc_amd import synth
    cloud = c3["cloud"]
    engine.set_cloud(cloud)
    engine.set_features(1, 0, 0)
    engine.set_opt(None)
    engine.set_exec(-1, 4)
    engine.zero(0)
    engine.zero(1)
    _c3_launch(engine, c3, 60, 0, 4194304, 2)
    INT = engine.read_tally(1)
    engine.set_features(0, 0, 0)
    NE, NFREQ, NCELL = 128, 50, 65536
    sol = synth.synth_solver(NFREQ=NFREQ, NE=NE, NSIZE=2, seed=5)
    # the 65536 cells around the source cell along the storage order of the root grid, refined cells left out
    src = (128 * 256 + 128) * 256 + 128
    pick = np.arange(src - NCELL, src + NCELL)
    pick = pick[cloud.DENS[pick] > 0][:NCELL]
    a = INT[pick].astype(np.float64)
    assert (a > 0).mean() > 0.5 and a.max() / a[a > 0].min() > 1e4
    shape = (sol["FREQ"] / 1e13) ** -1.0 * np.random.default_rng(1).lognormal(0, 0.3, NFREQ)
    ABS = np.asarray(1e-3 * (a / a[a > 0].mean())[:, None] * shape[None, :], np.float32)
    AF = synth.a2e_absorption_fraction(sol, 1)
    engine.a2e_set_size(NE, NFREQ, sol["sizes"][1], AF)
    got = engine.a2e_solve(ABS)
    assert got.shape == (NCELL, NFREQ)
    order = np.argsort(a)
    sample = np.unique(np.concatenate([order[:8], order[-8:], order[np.linspace(0, NCELL - 1, 48).astype(int)]]))
    want = a2e_oracle_dosolve(oracle_soc, NE, NFREQ, sol["sizes"][1], AF, ABS[sample])
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(got[sample]), ok)
    assert np.array_equal(got[sample][ok].view(np.uint32), want[ok].view(np.uint32))
    lit = a > 0
    assert np.isfinite(got[lit]).all() and (got[lit] >= 0).all() and got[lit].sum() > 0
    again = engine.a2e_solve(ABS[1000:1777])                 # another batch size and offset: same cells, same bits
    assert np.array_equal(again.view(np.uint32), got[1000:1777].view(np.uint32))
