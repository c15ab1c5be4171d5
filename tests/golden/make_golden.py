"""Generate the golden vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container: it executes the x86 builds of the reference's kernels
(oracle/_ref, compiled from /root/reference by oracle/build.py) on the seeded cases of
tests/cases.py and stores inputs + outputs as small .npz files.  The vectors are data --
no reference source travels.  Work items are executed sequentially, so the fp32 summation
order (and with it every bit of TABS) is reproducible.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from oracle.pyoracle import Ref, Job  # noqa: E402
from soc_amd import synth            # noqa: E402
import cases                         # noqa: E402


def main():
    # ---- RNG: stream states and draws (integer, exact) ----
    r = Ref("c8")
    seeds = [0.6004384, 0.7853981634, 0.25, 1.0]
    gids = [0, 1, 2, 63, 64, 255, 256, 65535, 65536, 49151, 786431, 3145727, 16777215, 16777216]
    states, draws, bases = [], [], []
    for s in seeds:
        for g in gids:
            x, c = r.seed(s, g)
            u, _ = r.draws(x, c, 8)
            states.append((x, c))
            draws.append(u)
    np.savez(os.path.join(HERE, "rng.npz"), seeds=np.asarray(seeds, np.float32), gids=np.asarray(gids, np.int64),
             states=np.asarray(states, np.uint32).reshape(len(seeds), len(gids), 2),
             draws=np.asarray(draws, np.uint32).reshape(len(seeds), len(gids), 8))

    # ---- fixed rays: (level, ind, ds) per step ----
    out = {}
    for name, (ref, mk, pos, d) in cases.RAYS.items():
        cloud = mk()
        job = Job(cloud, np.linspace(1, -1, 2500))
        lev, ind, ds, end = Ref(ref).trace(job, pos, np.asarray(d, np.float32))
        out[name + "_pos"] = np.asarray(pos, np.float32)
        out[name + "_dir"] = np.asarray(d, np.float32)
        out[name + "_lev"], out[name + "_ind"], out[name + "_ds"], out[name + "_end"] = lev, ind, ds, end
    k = synth.kat_octree()
    out["par_oct4"] = Ref("oct4").parents(Job(k, np.linspace(1, -1, 2500)))
    o8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    out["par_oct8"] = Ref("oct8").parents(Job(o8, np.linspace(1, -1, 2500)))
    np.savez(os.path.join(HERE, "rays.npz"), **out)

    # ---- Scatter / Deflect on fixed inputs ----
    dsc, csc = synth.hg_scattering_table(0.6)
    rng = np.random.default_rng(99)
    dirs = rng.standard_normal((64, 3))
    dirs /= np.sqrt((dirs ** 2).sum(1, keepdims=True))
    dirs = dirs.astype(np.float32)
    dirs[0] = [0, 0, 1]
    dirs[1] = [0, 0, -1]
    dirs[2] = [1, 0, 0]
    dirs[3] = [5e-5, 5e-5, 1.0]
    sc_out, sc_state = [], []
    for i, d in enumerate(dirs):
        x, c = r.seed(0.3, i)
        nd, st = r.scatter(d, csc, x, c)
        sc_out.append(nd)
        sc_state.append(st)
    ct = rng.uniform(-1, 1, 64).astype(np.float32)
    ph = rng.uniform(0, 2 * np.pi, 64).astype(np.float32)
    df_out = [r.deflect(d, a, b) for d, a, b in zip(dirs, ct, ph)]
    np.savez(os.path.join(HERE, "scatter.npz"), dirs=dirs, csc=csc, scatter_out=np.asarray(sc_out, np.float32),
             scatter_state=np.asarray(sc_state, np.uint32), cos_theta=ct, phi=ph,
             deflect_out=np.asarray(df_out, np.float32))

    # ---- full simulations ----
    out = {}
    for name, (ref, kind, mk) in cases.CASES.items():
        job = mk()
        T, I = Ref(ref).sim(job, kind)
        out[name + "_TABS"] = T
        if job.WITH_INT:
            out[name + "_INT"] = I
        if job.INTV is not None:
            out[name + "_INTV"] = job.INTV.copy()
        if job.WITH_ALI:
            out[name + "_XAB"] = job.XAB.copy()
        if job.ROI is not None:
            out[name + "_ROISAVE"] = job.ROI_SAVE.copy()
        out[name + "_DENS"] = job.DENS          # pins the synthetic-cloud generator as well
        print("%-14s sum(TABS) = %.9e   nonzero cells %d / %d" % (name, T.sum(dtype=np.float64), (T != 0).sum(), T.size))
    np.savez_compressed(os.path.join(HERE, "sims.npz"), **out)


def a2e_case(tag):
    """Seeded inputs of one A2E golden case (shared with tests/test_a2e.py)."""
    from oracle import build as ob
    m = ob.a2e_ref_models()[tag]
    sol = synth.synth_solver(NFREQ=m["NFREQ"], NE=m["NE"], NSIZE=2, seed=5)
    rng = np.random.default_rng(1)
    batch = min(m["CELLS"], 37)
    ABS = (rng.lognormal(0, 1, (batch, m["NFREQ"])) * 1e-3 * (sol["FREQ"][None, :] / 1e13) ** -1.0).astype(np.float32)
    return m, sol, ABS


def a2e_main():
    from oracle.pyoracle import RefA2E
    sys.path.insert(0, REPO)
    from soc_amd import a2e as a2e_host
    out = {}
    for tag in ("ne16", "ne64", "ne128"):
        m, sol, ABS = a2e_case(tag)
        r = RefA2E(tag)
        for isize in range(2):
            AF = synth.a2e_absorption_fraction(sol, isize)
            out["%s_s%d_emit" % (tag, isize)] = r.dosolve(sol["sizes"][isize], AF, ABS)
        out[tag + "_abs"] = ABS
    # EqTemperature on the ne64 build (NIP 5000)
    m, sol, ABS = a2e_case("ne64")
    Emin, kE, oplgkE, TTT, KABS = a2e_host.eq_table(sol, 1)
    AF = synth.a2e_absorption_fraction(sol, 1)
    tmp = np.asarray(ABS * AF, np.float32)
    T, E = RefA2E("ne64").eqtemp(0, kE, oplgkE, Emin, sol["FREQ"], KABS, TTT, tmp)
    out.update(eq_T=T, eq_emit=E, eq_TTT=TTT, eq_scal=np.asarray([Emin, kE, oplgkE], np.float64), eq_abs=tmp, eq_kabs=KABS)
    np.savez_compressed(os.path.join(HERE, "a2e.npz"), **out)
    print("a2e golden: T range", T.min(), T.max())


def a2e_pre_cases():
    """Inputs of the solver-preprocessing golden cases (shared with tests/test_a2e_pre.py): (FREQ, Ef, SKABS per grain, E, T)
    of one grain size of soc_amd.a2e_pre.AnalyticDust on the temperature grid of A2E_pre.py:206-207."""
    from soc_amd import a2e_pre
    out = {}
    for name, (NFREQ, NE, NSIZE, isize) in dict(small=(24, 16, 3, 0), big=(40, 48, 3, 2), wide=(64, 32, 4, 1)).items():
        dust = a2e_pre.AnalyticDust(NSIZE=NSIZE)
        FREQ = np.logspace(np.log10(1.5e11), np.log10(2.0e15), NFREQ).astype(np.float32)
        Ef = np.asarray(a2e_pre.PLANCK * FREQ, np.float32)
        T = dust.TMIN[isize] + (dust.TMAX[isize] - dust.TMIN[isize]) * (np.arange(NE + 1) / float(NE)) ** 2.0
        E = dust.T2E(isize, T)
        SK1 = np.asarray(dust.SKabs(isize, FREQ), np.float32)
        out[name] = (FREQ, Ef, SK1, np.asarray(E, np.float32), np.asarray(T, np.float32))
    return out


def a2e_pre_main():
    """Golden arrays of the solver preprocessing from the x86 build of kernel_A2E_pre.c."""
    from oracle.pyoracle import RefA2EPre
    sys.path.insert(0, REPO)
    R = RefA2EPre()
    out = {}
    for name, (FREQ, Ef, SK1, E, T) in a2e_pre_cases().items():
        k = R.pre(FREQ, Ef, SK1, E, T)
        for key in ("Iw", "L1", "L2", "Tdown", "noIw"):
            out["%s_%s" % (name, key)] = k[key]
        print("%-6s weights %d  pairs %d  Tdown %.3e .. %.3e" % (name, k["Iw"].size, (k["L1"] >= 0).sum(), k["Tdown"][1], k["Tdown"][-1]))
    np.savez_compressed(os.path.join(HERE, "a2e_pre.npz"), **out)


def sca_main():
    """Scattered-light images of tests/cases.py:SCA_CASES from the x86 builds of kernel_ASOC_sca.c."""
    from oracle.pyoracle import RefSca
    out = {}
    for name, (ref, kind, mk, vkw) in cases.SCA_CASES.items():
        job, view = mk(), cases.sca_view(**vkw)
        OUT = RefSca(ref).sim(job, view, kind)
        out[name] = OUT if view.nside else OUT.reshape(view.NDIR, view.NPIX[1], view.NPIX[0])
        print("%-18s sum(OUT) = %.9e   nonzero pixels %d / %d" % (name, OUT.sum(dtype=np.float64), (OUT != 0).sum(), OUT.size))
    np.savez_compressed(os.path.join(HERE, "sca.npz"), **out)


def maps_main():
    """Golden maps of tests/test_maps.py:MAP_CASES from the x86 build of kernel_ASOC_map.c."""
    from oracle.pyoracle import RefMap, Oracle
    import test_maps
    out = {}
    ol = Oracle("libm")
    for name, (ref, mk, kw) in test_maps.MAP_CASES.items():
        R = RefMap(ref, NSIDE=kw.get("healpix", 8) or 8)
        job, m, t = test_maps.run_case(name, lambda job, emit, d, r, e, npix, dx, c, io, cd, hp:
                                       R.mapping(job, ol.parents(job), emit, d, r, e, npix, dx, c, io, cd, hp))
        out[name + "_map"], out[name + "_tau"] = m, t
        print("%-20s sum(MAP) = %.6e  nonzero %d / %d" % (name, m.sum(dtype=np.float64), (m > 0).sum(), m.size))
    np.savez_compressed(os.path.join(HERE, "maps.npz"), **out)


if __name__ == "__main__":
    if "--maps" in sys.argv:
        maps_main()
        sys.exit(0)
    if "--sca" in sys.argv:
        sca_main()
        sys.exit(0)
    if "--a2e-pre" in sys.argv:
        a2e_pre_main()
        sys.exit(0)
    if "--a2e" in sys.argv:
        a2e_main()
        sys.exit(0)
    main()
