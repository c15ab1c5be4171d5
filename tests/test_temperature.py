"""Equilibrium temperature and emission: oracle against the reference's
EqTemperature / Emission2 (x86 build, bit-exact in libm mode), host E<->T table against an independent
evaluation, the iteration loop of asoc.py on the oracle-backed engine, and (GPU) the HIP kernels against
the oracle."""
import math
import os

import numpy as np
import pytest

from oracle.pyoracle import Job, Oracle, Ref
from soc_amd import files, launch, synth
from soc_amd.asoc import AbsorptionRun
from soc_amd.ini import User

FF = np.logspace(np.log10(3e11), np.log10(3e15), 40)
FABS = 1e-5 * (FF / 1e13) ** 1.6
GL = 0.01


def _absorbed(cloud, Emin, kE, TTT, FACTOR, LENGTH, seed=4):
    """absorbed energies that put the leaves at temperatures between 5 and 300 K"""
    NE = len(TTT)
    T_target = np.random.default_rng(seed).uniform(5, 300, cloud.CELLS)
    Ein = np.interp(T_target, TTT.astype(float), Emin * kE ** np.arange(NE))
    lev = np.zeros(cloud.CELLS, int)
    for l in range(cloud.LEVELS):
        lev[cloud.OFF[l]:cloud.OFF[l] + cloud.LCELLS[l]] = l
    scale = 6.62607e-27 * float(FACTOR) / float(LENGTH)
    return (Ein * np.abs(cloud.DENS) / (scale * 8.0 ** lev)).astype(np.float32), T_target


def test_kernel_literals_and_table():
    FACTOR, LENGTH = launch.kernel_literals(GL)
    assert FACTOR == np.float32(1.0e20) and LENGTH == np.float32(3.08568e16)     # "%.5e" % (0.01*3.08567758e18)
    Emin, kE, TTT = launch.temperature_table(FF, FABS, GL, NE=6000)
    # independent: energy emitted at T, trapezoid over the frequency grid
    def Eout(T):
        B = 2.0e-20 * 7.372496678e-28 * FF ** 3 / np.expm1(np.clip(4.79924335e-11 * FF / T, -100, 100))
        return 4.0 * np.pi * 1.0e20 / (GL * 3.08567758e18) * np.trapezoid(FABS * B, FF)
    assert abs(Emin / Eout(1.0) - 1) < 1e-9
    for i in (3000, 4500, 5999):                                                 # T > 20 K: E(T) is smooth on the 0.27 K grid
        assert abs(Eout(float(TTT[i])) / (Emin * kE ** i) - 1) < 2e-3
    assert TTT[0] == 1.0 and 1590 < TTT[-1] <= 1601


@pytest.mark.skipif(not Ref.available("oct8"), reason="reference builds (oracle/_ref) not present")
def test_oracle_bit_exact_vs_reference_kernels():
    o8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    job = Job(o8, np.linspace(1, -1, 2500))
    Emin, kE, TTT = launch.temperature_table(FF, FABS, GL, NE=3000)
    FACTOR, LENGTH = launch.kernel_literals(GL)
    EABS, T_target = _absorbed(o8, Emin, kE, TTT, FACTOR, LENGTH)
    ol = Oracle("libm")
    T = ol.eqtemp(job, 1.0, kE, Emin, TTT, FACTOR, LENGTH, EABS)
    assert np.array_equal(T.view(np.uint32), Ref("oct8").eqtemp(job, 1.0, kE, Emin, TTT, EABS).view(np.uint32))
    leaf = o8.DENS > 0
    assert np.abs(T[leaf] - T_target[leaf]).max() < 0.01 and (T[~leaf] == 10.0).all()
    E = ol.emission(FF, FABS, FACTOR, LENGTH, T)
    assert np.array_equal(E.view(np.uint32), Ref("oct8").emission(job, FF, FABS, T).view(np.uint32))
    # -D CR_HEATING=1 -D CR_HEATING_RATE=2.5 (ini key CR_HEATING): a constant added to the absorbed energy (kernel_ASOC_aux.c:769-773)
    job.CR_HEATING_RATE = 2.5
    Tc = ol.eqtemp(job, 1.0, kE, Emin, TTT, FACTOR, LENGTH, 1e-6 * EABS)
    assert np.array_equal(Tc.view(np.uint32), Ref("oct8cr").eqtemp(job, 1.0, kE, Emin, TTT, 1e-6 * EABS).view(np.uint32))
    job.CR_HEATING_RATE = 0.0
    T0 = ol.eqtemp(job, 1.0, kE, Emin, TTT, FACTOR, LENGTH, 1e-6 * EABS)
    assert (Tc[leaf] >= T0[leaf] - 1e-3).all() and (Tc[leaf] > T0[leaf]).mean() > 0.5
    # product math: same algorithm, last-bit differences in log10/pown/exp only
    Ts = Oracle("soc").eqtemp(job, 1.0, kE, Emin, TTT, FACTOR, LENGTH, EABS)
    assert np.abs(Ts[leaf] / T[leaf] - 1).max() < 2e-5


def test_iteration_loop_on_oracle_engine(tmp_path):
    """constant sources -> T -> emission -> cell-emission packets -> T: files and arithmetic of one cycle"""
    from oracle_engine import OracleEngine
    from test_host import _write_model
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    extra = ("CLT\nCLE\nnoabsorbed\niterations 2\ncellpackets %d\ntemperature %s/T.bin\nemitted %s/em.bin\nglobal 128\n" % (2 * cloud.CELLS, d, d))
    ini = _write_model(d, cloud, extra=extra).replace("nosolve\n", "")
    txt = open(ini).read().replace("nosolve\n", "").replace("absorbed %s/abs.data\n" % d, "")
    open(ini, "w").write(txt)
    os.chdir(d)
    eng = OracleEngine("soc")
    run = AbsorptionRun(User(ini), eng)
    CTABS, FABS_ = run.run()
    assert FABS_ is None and run.TNEW is not None
    # files
    head = np.fromfile(os.path.join(d, "T.bin"), np.int32, 5)
    assert list(head) == [6, 6, 6, 2, cloud.CELLS]
    em = files.mmap_emitted(os.path.join(d, "em.bin"), cloud.CELLS, 3)
    assert np.array_equal(np.array(em), run.EMITTED)
    leaf = cloud.DENS > 0
    assert (run.TNEW[leaf] >= 3.0).all() and (run.TNEW[leaf] <= 1600.0).all() and (run.TNEW[~leaf] == 10.0).all()
    # the emission is the modified black body of those temperatures (Emission kernel formula)
    FFREQ, _, AFABS, _ = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    FACTOR, LENGTH = launch.kernel_literals(0.5)
    want = Oracle("soc").emission(FFREQ, AFABS[0], FACTOR, LENGTH, run.TNEW)
    assert np.array_equal(run.EMITTED, want)
    # second iteration used cell emission: temperatures are at least those from the constant sources alone
    Emin, kE, TTT = launch.temperature_table(FFREQ, AFABS[0], 0.5)
    T0 = Oracle("soc").eqtemp(Job(cloud, np.linspace(1, -1, 8)), 1.0, kE, Emin, TTT, FACTOR, LENGTH, CTABS)
    assert (run.TNEW[leaf] >= T0[leaf] - 1e-3).all() and (run.TNEW[leaf] > T0[leaf]).any()


@pytest.mark.gpu
def test_hip_temperature_and_emission_match_oracle(engine):
    o8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    job = Job(o8, np.linspace(1, -1, 2500))
    Emin, kE, TTT = launch.temperature_table(FF, FABS, GL, NE=3000)
    FACTOR, LENGTH = launch.kernel_literals(GL)
    EABS, _ = _absorbed(o8, Emin, kE, TTT, FACTOR, LENGTH)
    engine.set_cloud(o8)
    T = engine.solve_temperature(1.0, kE, Emin, TTT, FACTOR, LENGTH, EABS)
    osoc = Oracle("soc")
    want = osoc.eqtemp(job, 1.0, kE, Emin, TTT, FACTOR, LENGTH, EABS)
    assert np.array_equal(T.view(np.uint32), want.view(np.uint32))          # same header, same operation order
    E = engine.emission(FF, FABS, FACTOR, LENGTH)
    assert np.array_equal(E.view(np.uint32), osoc.emission(FF, FABS, FACTOR, LENGTH, want).view(np.uint32))
    # CR_HEATING: a constant term in the absorbed energy (kernel_ASOC_aux.c:769-773)
    engine.set_cr_heating(2.5)
    try:
        Tc = engine.solve_temperature(1.0, kE, Emin, TTT, FACTOR, LENGTH, 1e-6 * EABS)
    finally:
        engine.set_cr_heating(0.0)
    job.CR_HEATING_RATE = 2.5
    wc = osoc.eqtemp(job, 1.0, kE, Emin, TTT, FACTOR, LENGTH, 1e-6 * EABS)
    assert np.array_equal(Tc.view(np.uint32), wc.view(np.uint32))
    job.CR_HEATING_RATE = 0.0
    assert not np.array_equal(wc, osoc.eqtemp(job, 1.0, kE, Emin, TTT, FACTOR, LENGTH, 1e-6 * EABS))


@pytest.mark.gpu
def test_iteration_loop_end_to_end(engine, tmp_path):
    from oracle_engine import OracleEngine
    from test_host import _write_model
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    extra = ("CLT\nCLE\nnoabsorbed\niterations 2\ncellpackets %d\ntemperature %s/T.bin\nemitted %s/em.bin\nglobal 128\n" % (2 * cloud.CELLS, d, d))
    ini = _write_model(d, cloud, extra=extra)
    txt = open(ini).read().replace("nosolve\n", "").replace("absorbed %s/abs.data\n" % d, "")
    open(ini, "w").write(txt)
    os.chdir(d)
    want = AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0)
    want.run()
    os.remove(os.path.join(d, "em.bin"))          # an existing emitted file is the starting point of the next run
    got = AbsorptionRun(User(ini), engine, verbose=0)
    got.run()
    leaf = cloud.DENS > 0
    assert np.abs(got.TNEW[leaf] / want.TNEW[leaf] - 1).max() < 2e-5
    assert np.allclose(got.EMITTED[leaf], want.EMITTED[leaf], rtol=2e-3)


def _iter_ini(d, cloud, extra):
    from test_host import _write_model
    ini = _write_model(d, cloud, extra="CLT\nCLE\nnoabsorbed\niterations 2\ntemperature %s/T.bin\nemitted %s/em.bin\nglobal 128\n" % (d, d) + extra)
    txt = open(ini).read().replace("nosolve\n", "").replace("absorbed %s/abs.data\n" % d, "")
    open(ini, "w").write(txt)
    return ini


def test_emweight2_and_ali_iterations_on_oracle_engine(tmp_path):
    """host loops of USE_EMWEIGHT==2 (EMINDEX lists, 100 packets per cell and launch, ASOC.py:1811-1840) and of
    ALI (XEM, escape probability, host temperature solve with beta, ASOC.py:1606,1741,1939-1942,2042-2073)"""
    from oracle_engine import OracleEngine
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(5, seed=2)
    os.chdir(d)
    # EMWEIGHT 2: every cell emits a multiple of 100 packets; energy is conserved against the plain run
    runs = {}
    for key, extra in (("plain", "cellpackets %d\n" % (200 * cloud.CELLS)),
                       ("emw2", "cellpackets %d\nemweight 2\n" % (200 * cloud.CELLS))):
        if os.path.exists(os.path.join(d, "em.bin")):
            os.remove(os.path.join(d, "em.bin"))
        eng = OracleEngine("soc")
        run = AbsorptionRun(User(_iter_ini(d, cloud, extra)), eng, verbose=0)
        run.run()
        runs[key] = run
    a, b = runs["plain"].TNEW, runs["emw2"].TNEW
    assert np.abs(b / a - 1).max() < 0.05 and not np.array_equal(a, b)
    # ALI: same energy budget, temperatures from the host solve with the escape probability
    os.remove(os.path.join(d, "em.bin"))
    eng = OracleEngine("soc")
    run = AbsorptionRun(User(_iter_ini(d, cloud, "cellpackets %d\nali 1\n" % (20 * cloud.CELLS))), eng, verbose=0)
    run.run()
    assert eng.ali == 1 and eng.T[2].sum() > 0                      # XAB was tallied
    # self-absorbed emission is taken out of the budget and the solve divides by the escape probability:
    # in this opaque toy model the temperatures come out higher than without ALI
    assert (run.TNEW > a).all() and np.abs(run.TNEW / a - 1).max() < 0.4


def test_reference_field_iterations_on_oracle_engine(tmp_path):
    """`reference 1` (ASOC.py:796-812, :1607-1632, :1728-1735, :1965-1975): the packets carry EMITTED - k*previous, the
    host adds k*(absorptions of the previous emission) back.  One iteration: k = 0, the plain run bit for bit; three
    iterations: the same temperatures to Monte Carlo accuracy; `reference 300` writes the state a run with
    `reference 302` continues from."""
    from oracle_engine import OracleEngine
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(5, seed=2)
    os.chdir(d)

    def go(extra, iterations):
        if os.path.exists(os.path.join(d, "em.bin")):
            os.remove(os.path.join(d, "em.bin"))
        ini = _iter_ini(d, cloud, "cellpackets %d\n" % (40 * cloud.CELLS) + extra)
        txt = open(ini).read().replace("iterations 2\n", "iterations %d\n" % iterations)
        open(ini, "w").write(txt)
        run = AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0)
        run.run()
        return run
    plain1, ref1 = go("", 1), go("reference 1\n", 1)
    assert np.array_equal(plain1.TNEW, ref1.TNEW)
    plain3, ref3 = go("", 3), go("reference 1\n", 3)
    assert not np.array_equal(plain3.TNEW, ref3.TNEW)
    assert np.abs(ref3.TNEW / plain3.TNEW - 1).max() < 0.03
    # AABB form: 3 iterations in total, this run starts at iteration 0 -> same damping as `reference 1`, state saved
    cont = go("reference 300\n", 2)
    assert os.path.getsize(os.path.join(d, "OEMITTED.save")) == 4 * cloud.CELLS * 3
    OT = np.fromfile(os.path.join(d, "OTABS.save"), np.float32)
    assert OT.shape == (cloud.CELLS,) and (OT >= 0).all() and (OT > 0).mean() > 0.5
    em = np.array(cont.EMITTED)
    ini = _iter_ini(d, cloud, "cellpackets %d\nreference 302\n" % (40 * cloud.CELLS))      # keeps em.bin of the first part
    txt = open(ini).read().replace("iterations 2\n", "iterations 1\n")
    open(ini, "w").write(txt)
    last = AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0)
    last.run()
    assert np.abs(last.TNEW / plain3.TNEW - 1).max() < 0.03
    assert not np.array_equal(np.fromfile(os.path.join(d, "OTABS.save"), np.float32), OT) and em.shape == last.EMITTED.shape


def test_default_solver_is_the_host_loop_of_the_reference(tmp_path):
    """without the keys `CLT` / `CLE` the reference solves temperatures and emission on the host (ASOC.py:2042-2073,
    :2214-2227): parents get T = 0, the weight is the host loop's, the emission the double-precision Planck formula;
    `MPE` replaces T < 3 by 10 first (:2211); `remit` that cuts low frequencies fails there and is refused"""
    from oracle_engine import OracleEngine
    from test_host import _write_model
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    os.chdir(d)

    def go(keys):
        extra = keys + "noabsorbed\niterations 1\ntemperature %s/T.bin\nemitted %s/em.bin\nglobal 128\n" % (d, d)
        ini = _write_model(d, cloud, extra=extra)
        txt = open(ini).read().replace("nosolve\n", "").replace("absorbed %s/abs.data\n" % d, "")
        open(ini, "w").write(txt)
        if os.path.exists(os.path.join(d, "em.bin")):
            os.remove(os.path.join(d, "em.bin"))
        run = AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0)
        return run, run.run()[0]
    host, CTABS = go("")
    dev, _ = go("CLT\nCLE\n")
    leaf = cloud.DENS > 0
    assert (host.TNEW[~leaf] == 0.0).all() and (dev.TNEW[~leaf] == 10.0).all()
    FFREQ, _, AFABS, _ = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    Emin, kE, TTT = launch.temperature_table(FFREQ, AFABS[0], 0.5)
    want = launch.solve_temperature_host(CTABS, cloud, Emin, kE, TTT, 0.5)
    assert np.array_equal(host.TNEW[leaf], np.clip(want[leaf], 3.0, 1600.0))
    assert np.array_equal(host.EMITTED, launch.emission_host(FFREQ, AFABS[0], host.TNEW, 0.5))
    # host and device emission formulas agree where the temperatures do, to the constants they are written with
    # (h/k = 4.7995074e-11 in the kernel, 4.79924335e-11 on the host)
    mixed, _ = go("CLT\n")
    assert np.array_equal(mixed.TNEW, dev.TNEW)
    assert np.allclose(mixed.EMITTED[leaf], dev.EMITTED[leaf], rtol=1e-2)
    mpe, _ = go("MPE\n")
    assert (mpe.TNEW[~leaf] == 10.0).all() and np.array_equal(mpe.TNEW[leaf], host.TNEW[leaf])
    with pytest.raises(ValueError, match="CLE"):
        go("remit 0.5 0.7\n")


def test_host_temperature_solve_matches_device_formula_away_from_its_quirk():
    """launch.solve_temperature_host (the reference's host loop) lands within one table step of the device kernel"""
    o8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    Emin, kE, TTT = launch.temperature_table(FF, FABS, GL, NE=3000)
    FACTOR, LENGTH = launch.kernel_literals(GL)
    EABS, _ = _absorbed(o8, Emin, kE, TTT, FACTOR, LENGTH)
    Th = launch.solve_temperature_host(EABS, o8, Emin, kE, TTT, GL)
    Td = Oracle("soc").eqtemp(Job(o8, np.linspace(1, -1, 8)), 1.0, kE, Emin, TTT, FACTOR, LENGTH, EABS)
    leaf = o8.DENS > 0
    assert (Th[~leaf] == 0).all()
    assert np.abs(Th[leaf] - Td[leaf]).max() < 1.2 * np.diff(TTT).max()


@pytest.mark.gpu
def test_iteration_loop_end_to_end_with_cell_emission_in_the_brick_sweep(engine, tmp_path):
    """`global` = number of cells on a 64^3 model: the dust-emission launches of asoc.py are deferred per frequency
    and run through the brick sweep (SimRAM_CL kind); temperatures equal the oracle engine's run"""
    from oracle_engine import OracleEngine
    from test_host import _write_model
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(64, seed=9)
    extra = ("CLT\nCLE\ngridlength 2e-6\nnoabsorbed\niterations 2\ncellpackets %d\ntemperature %s/T.bin\nemitted %s/em.bin\nglobal %d\n"
             "bgpackets 400000\n" % (cloud.CELLS, d, d, cloud.CELLS))
    ini = _write_model(d, cloud, extra=extra)
    txt = open(ini).read().replace("nosolve\n", "").replace("absorbed %s/abs.data\n" % d, "")
    open(ini, "w").write(txt)
    os.chdir(d)

    class Counting(OracleEngine):
        threads = 8

    want = AbsorptionRun(User(ini), Counting("soc"), verbose=0)
    want.run()
    os.remove(os.path.join(d, "em.bin"))
    passes = []
    real_batch_end = engine.batch_end

    def spy():
        real_batch_end()
        passes.append(engine.last_passes())
    engine.batch_end = spy
    try:
        got = AbsorptionRun(User(ini), engine, verbose=0)
        got.run()
    finally:
        engine.batch_end = real_batch_end
    assert max(passes) > 0, "no launch of the run went through the brick sweep"
    assert np.abs(got.TNEW / want.TNEW - 1).max() < 5e-5
    assert np.allclose(got.EMITTED, want.EMITTED, rtol=5e-3)
