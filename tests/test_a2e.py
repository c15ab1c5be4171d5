"""A2E (stochastically heated grains): oracle pinned to the reference's DoSolve/EqTemperature
(bit-exact in libm mode), file formats, and -- on the GPU -- the HIP kernels against the oracle.

On the GPU both kernels are held to the oracle bit for bit: EqTemperature is one lane per cell with the shared math
header; DoSolve keeps the reference's fp32 operations and their order everywhere -- the forward substitution sums
each row in one lane (i ascending, mul then add), finished values handed down by v_readlane -- and uses +,*,/,max
only, so it equals the reference's own golden output as well."""


def _same_bits(got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    ok = np.isfinite(want)
    return np.array_equal(np.isfinite(got), ok) and np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
import os
import sys

import numpy as np
import pytest

from oracle.pyoracle import RefA2E, a2e_oracle_dosolve, a2e_oracle_eqtemp
from soc_amd import files, synth

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import a2e_case  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "a2e.npz"))


@pytest.mark.parametrize("tag", ["ne16", "ne64", "ne128"])
def test_dosolve_oracle_bit_exact_vs_reference_golden(tag, oracle_libm, oracle_soc):
    m, sol, ABS = a2e_case(tag)
    assert np.array_equal(ABS, GOLD[tag + "_abs"])
    for isize in range(2):
        AF = synth.a2e_absorption_fraction(sol, isize)
        for orc in (oracle_libm, oracle_soc):          # DoSolve has no transcendentals: both modes identical
            e = a2e_oracle_dosolve(orc, m["NE"], m["NFREQ"], sol["sizes"][isize], AF, ABS)
            assert np.array_equal(e.view(np.uint32), GOLD["%s_s%d_emit" % (tag, isize)].view(np.uint32))
            assert np.isfinite(e).all() and (e >= 0).all() and e.max() > 0


def test_eqtemp_oracle_vs_reference_golden(oracle_libm, oracle_soc):
    m, sol, ABS = a2e_case("ne64")
    Emin, kE, oplgkE = GOLD["eq_scal"]
    T, E = a2e_oracle_eqtemp(oracle_libm, 0, m["CELLS"], m["NIP"], 1e20, kE, oplgkE, Emin, sol["FREQ"], GOLD["eq_kabs"],
                             GOLD["eq_TTT"], GOLD["eq_abs"])
    assert np.array_equal(T.view(np.uint32), GOLD["eq_T"].view(np.uint32))
    assert np.array_equal(E.view(np.uint32), GOLD["eq_emit"].view(np.uint32))
    T2, E2 = a2e_oracle_eqtemp(oracle_soc, 0, m["CELLS"], m["NIP"], 1e20, kE, oplgkE, Emin, sol["FREQ"], GOLD["eq_kabs"],
                               GOLD["eq_TTT"], GOLD["eq_abs"])
    assert np.allclose(T2, GOLD["eq_T"], rtol=2e-5)     # soc_pownf/soc_log10f vs libm powf/log10f
    ok = GOLD["eq_emit"] > 1e-30 * GOLD["eq_emit"].max()
    assert np.allclose(E2[ok], GOLD["eq_emit"][ok], rtol=2e-3)   # Wien tail amplifies dT/T by h nu / kT


@pytest.mark.skipif(not RefA2E.available("ne16"), reason="reference builds (oracle/_ref) not present")
def test_dosolve_live_vs_reference_random_inputs(oracle_libm):
    m, sol, _ = a2e_case("ne16")
    rng = np.random.default_rng(9)
    ABS = (rng.lognormal(0, 2, (40, m["NFREQ"])) * 1e-2).astype(np.float32)
    ABS[3] = 0.0                                       # a cell without absorptions
    AF = synth.a2e_absorption_fraction(sol, 0)
    a = a2e_oracle_dosolve(oracle_libm, m["NE"], m["NFREQ"], sol["sizes"][0], AF, ABS)
    b = RefA2E("ne16").dosolve(sol["sizes"][0], AF, ABS)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_solver_file_round_trip(tmp_path):
    sol = synth.synth_solver(NFREQ=12, NE=16, NSIZE=3, seed=2)
    fn = str(tmp_path / "x.solver")
    synth.write_solver(fn, sol)
    back = files.read_solver(fn)
    assert back["NFREQ"] == 12 and back["NE"] == 16 and back["NSIZE"] == 3 and len(back["sizes"]) == 3
    assert np.array_equal(back["FREQ"], sol["FREQ"]) and np.array_equal(back["SK_ABS"], sol["SK_ABS"])
    for a, b in zip(back["sizes"], sol["sizes"]):
        for k in ("Iw", "L1", "L2", "Tdown", "EA", "Ibeg"):
            assert np.array_equal(a[k], b[k])
    # windows and weight count are consistent (what soc_a2e_set_size validates on the host)
    for s in sol["sizes"]:
        L1, L2 = s["L1"].reshape(16, 16), s["L2"].reshape(16, 16)
        n = sum(max(0, L2[l, u] - L1[l, u] + 1) for l in range(15) for u in range(l + 1, 16))
        assert n == len(s["Iw"])


# ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["ne16", "ne64", "ne128"])
def test_gpu_dosolve_vs_oracle_and_golden(tag, engine, oracle_soc):
    m, sol, ABS = a2e_case(tag)
    for isize in range(2):
        AF = synth.a2e_absorption_fraction(sol, isize)
        want = a2e_oracle_dosolve(oracle_soc, m["NE"], m["NFREQ"], sol["sizes"][isize], AF, ABS)
        engine.a2e_set_size(m["NE"], m["NFREQ"], sol["sizes"][isize], AF)
        got = engine.a2e_solve(ABS)
        assert np.isfinite(got).all()
        assert _same_bits(got, want)
        assert _same_bits(got, GOLD["%s_s%d_emit" % (tag, isize)])


@pytest.mark.gpu
def test_gpu_dosolve_ragged_and_edge_inputs(engine, oracle_soc):
    m, sol, _ = a2e_case("ne64")
    rng = np.random.default_rng(4)
    ABS = (rng.lognormal(0, 2, (131, m["NFREQ"])) * 1e-2).astype(np.float32)
    ABS[0] = 0.0
    ABS[1] *= 1e12                                     # triggers the 1e-20 rescaling branch
    AF = synth.a2e_absorption_fraction(sol, 1)
    want = a2e_oracle_dosolve(oracle_soc, m["NE"], m["NFREQ"], sol["sizes"][1], AF, ABS)
    engine.a2e_set_size(m["NE"], m["NFREQ"], sol["sizes"][1], AF)
    got = engine.a2e_solve(ABS)
    ok = np.isfinite(want)
    assert _same_bits(got, want)
    one = engine.a2e_solve(ABS[5:6])                   # batch of one cell
    assert _same_bits(one[0], want[5])
    from soc_amd.lib import SocError
    bad = dict(sol["sizes"][1])
    bad["L2"] = bad["L2"].copy()
    bad["L2"][1] = 10 ** 6
    with pytest.raises(SocError, match="outside"):
        engine.a2e_set_size(m["NE"], m["NFREQ"], bad, AF)


@pytest.mark.gpu
def test_gpu_eqtemp_bit_identical_to_oracle(engine, oracle_soc):
    m, sol, ABS = a2e_case("ne64")
    Emin, kE, oplgkE = GOLD["eq_scal"]
    T, E = a2e_oracle_eqtemp(oracle_soc, 0, m["CELLS"], m["NIP"], 1e20, kE, oplgkE, Emin, sol["FREQ"], GOLD["eq_kabs"],
                             GOLD["eq_TTT"], GOLD["eq_abs"])
    Tg, Eg = engine.a2e_eqtemp(0, m["CELLS"], m["NIP"], 1e20, kE, oplgkE, Emin, sol["FREQ"], GOLD["eq_kabs"],
                               GOLD["eq_TTT"], GOLD["eq_abs"])
    assert np.array_equal(Tg.view(np.uint32), T.view(np.uint32))
    assert np.array_equal(Eg.view(np.uint32), E.view(np.uint32))
    assert np.allclose(Tg, GOLD["eq_T"], rtol=2e-5)


@pytest.mark.gpu
def test_gpu_a2e_host_program(engine, oracle_soc, tmp_path):
    """files in -> files out, stochastic + equilibrium sizes, against an oracle-driven evaluation."""
    from soc_amd import a2e
    sol = synth.synth_solver(NFREQ=12, NE=16, NSIZE=3, seed=2)
    rng = np.random.default_rng(3)
    ABS = (rng.lognormal(0, 1, (300, 12)) * 1e-3).astype(np.float32)
    E, _ = a2e.run(engine, sol, ABS, NSTOCH=2, batch=128, verbose=False)
    A = ABS.copy()
    A[:, 11] = np.clip(A[:, 11], 0.0, 0.2 * A[:, 10])
    want = np.zeros_like(E)
    for isize in range(2):
        want += a2e_oracle_dosolve(oracle_soc, 16, 12, sol["sizes"][isize], synth.a2e_absorption_fraction(sol, isize), A)
    Emin, kE, oplgkE, TTT, KABS = a2e.eq_table(sol, 2)
    AF = synth.a2e_absorption_fraction(sol, 2)
    T, e2 = a2e_oracle_eqtemp(oracle_soc, 0, 300, a2e.NIP, 1e20, kE, oplgkE, Emin, sol["FREQ"], KABS, TTT,
                              np.asarray(A * AF, np.float32))
    want += e2 * (sol["GD"] * sol["S_FRAC"][2])
    assert np.allclose(E, want, rtol=1e-6, atol=1e-30 * want.max())


@pytest.mark.gpu
def test_gpu_a2e_resident_cells_equal_the_batches_to_the_bit(engine):
    """soc_a2e_resident_*: the absorptions of all cells in device memory, every size one launch that adds to the emission sum there
    (what a2e.run does where the memory allows) against the reference's shape -- a batch up, a size, a batch down, the host adds."""
    from soc_amd import a2e
    sol = synth.synth_solver(NFREQ=12, NE=16, NSIZE=4, seed=7)
    rng = np.random.default_rng(9)
    ABS = (rng.lognormal(0, 1, (1000, 12)) * 1e-3).astype(np.float32)

    class Batches:                                        # the engine without the resident calls
        def __getattr__(self, name):
            if name.startswith("a2e_resident"):
                raise AttributeError(name)
            return getattr(engine, name)
    for ifreq in (-1, 5):
        got, _ = a2e.run(engine, sol, ABS, NSTOCH=3, IFREQ=ifreq, batch=256, verbose=False)
        want, _ = a2e.run(Batches(), sol, ABS, NSTOCH=3, IFREQ=ifreq, batch=256, verbose=False)
        assert np.isfinite(got).all() and got.any()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # chunked upload / download at odd offsets, and the refusals
    engine.a2e_resident_begin(1000, 12)
    engine.a2e_resident_upload(0, ABS[:333])
    engine.a2e_resident_upload(333, ABS[333:])
    with pytest.raises(Exception):
        engine.a2e_resident_upload(900, ABS[:200])
    engine.a2e_set_size(16, 12, sol["sizes"][0], synth.a2e_absorption_fraction(sol, 0))
    engine.a2e_resident_solve()
    one = engine.a2e_solve(ABS)
    assert np.array_equal(engine.a2e_resident_download(17, 500).view(np.uint32), one[17:517].view(np.uint32))
    engine.a2e_resident_end()
    with pytest.raises(Exception):
        engine.a2e_resident_solve()


@pytest.mark.gpu
def test_gpu_a2e_sharded_program_equals_in_memory_run(engine, tmp_path):
    """python -m soc_amd.a2e: memory-mapped absorbed file, this rank's cell range, its part of the emitted file"""
    from soc_amd import a2e, files
    from soc_amd.lib import Engine
    d = str(tmp_path)
    sol = synth.synth_solver(NFREQ=12, NE=16, NSIZE=3, seed=2)
    synth.write_solver(os.path.join(d, "x.solver"), sol)
    ABS = (np.random.default_rng(3).lognormal(0, 1, (300, 12)) * 1e-3).astype(np.float32)
    files.write_absorbed(os.path.join(d, "abs.bin"), ABS)
    n, _ = a2e.run_sharded(lambda: Engine(0), os.path.join(d, "x.solver"), os.path.join(d, "abs.bin"),
                           os.path.join(d, "em.bin"), NSTOCH=2, verbose=False)
    want, _ = a2e.run(engine, files.read_solver(os.path.join(d, "x.solver")), ABS, NSTOCH=2, verbose=False)
    assert n == 300 and np.array_equal(files.read_absorbed(os.path.join(d, "em.bin")), want)


@pytest.mark.gpu
def test_eqsolver_hip_equals_oracle(engine, oracle_soc, tmp_path):
    """kernel_eqsolver.c (equilibrium dust components of a multi-dust run): HIP == oracle, operation for operation"""
    import os
    from oracle.pyoracle import oracle_eqsolver
    from soc_amd import driver
    from soc_amd.launch import FACTOR
    nf = 24
    FREQ = np.logspace(np.log10(1.5e11), np.log10(2.0e15), nf)
    dust = os.path.join(str(tmp_path), "eq.dust")
    with open(dust, "w") as fp:
        fp.write("eqdust\n 1.0e-7\n 1.0e-4\n%d\n" % nf)
        for f in FREQ:
            fp.write(" %.5e  0.5  %.5e  %.5e\n" % (f, 2.0e-2 * (f / 1e14) ** 1.2, 5.0e-2 * (f / 1e14) ** 1.5))
    Fq, KABS, Emin, kE, oplgkE, TTT = driver.eq_dust_table(dust)
    rng = np.random.default_rng(2)
    ABS = (10.0 ** rng.uniform(-2, 5, (3000, nf)) * (FREQ[None, :] / 1e14) ** -1.0).astype(np.float32)
    ABS[7, :] = 0.0                                       # no absorbed energy: T = 2.7
    ABS[9, :] = -1.0e20                                   # a refined cell of the absorbed file
    Tw, Ew = oracle_eqsolver(oracle_soc, 100, 3050, driver.NE_EQ, FACTOR, kE, oplgkE, Emin, Fq, KABS, TTT, ABS)
    Tg, Eg = engine.eqsolver(100, 3050, driver.NE_EQ, FACTOR, kE, oplgkE, Emin, Fq, KABS, TTT, ABS)
    assert Tw[7] == np.float32(2.7) and 3 < np.median(Tw) < 300
    assert np.array_equal(Tg[:2950].view(np.uint32), Tw[:2950].view(np.uint32))
    assert np.array_equal(Eg[:2950].view(np.uint32), Ew[:2950].view(np.uint32))
    assert (Eg[2950:] == 0).all()                         # cells beyond CELLS are not touched
