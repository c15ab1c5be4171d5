"""Map making: the oracle's restatement of kernel_ASOC_map.c against the x86 build of
that file (bit-exact, libm mode) and the golden maps made from it; the HIP kernel against the oracle
(soc mode; same header, same operation order -> bit-identical); the map files of asoc.py."""
import math
import os

import numpy as np
import pytest

from oracle.pyoracle import Job, Oracle, RefMap, oracle_mapping, NO_INTOBS
from soc_amd import launch, synth

_, CSC = synth.hg_scattering_table(0.6)
N, OD, RA, DE = launch.set_observer_directions([math.radians(30), math.radians(90), 0.0], [math.radians(40), 0.0, 0.0])
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "maps.npz"))


def _opt(cells):
    rr = np.random.default_rng(5)
    o = np.zeros((cells, 2), np.float32)
    o[:, 0] = 1e-3 * rr.uniform(0.5, 2, cells)
    o[:, 1] = 3e-3 * rr.uniform(0.5, 2, cells)
    return o


MAP_CASES = {
    # name: (ref build, cloud factory, kwargs)
    "map_c8": ("c8", lambda: synth.cartesian_cloud(8, seed=3), {}),
    "map_oct8": ("oct8", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), {}),
    "map_c8_abu": ("c8abu", lambda: synth.cartesian_cloud(8, seed=3), dict(abu=True)),
    "map_oct8_inside": ("oct8", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), dict(intobs=(4.2, 4.1, 3.9), npix=(16, 9))),
    "map_oct8_colden": ("oct8", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), dict(colden=1)),
    "map_oct8_healpix": ("oct8", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), dict(intobs=(4.2, 4.1, 3.9), healpix=8)),
    "map_oct8_roimap": ("oct8roi", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), dict(roi=[2, 5, 1, 6, 3, 4])),    # -D ROI_MAP=1
    "map_oct8_roimap_healpix": ("oct8roi", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7),
                                dict(roi=[2, 5, 1, 6, 3, 4], intobs=(4.2, 4.1, 3.9), healpix=8)),
    "map_oct8_threshold": ("oct8thr", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), dict(threshold=1)),   # -D LEVEL_THRESHOLD=1
    # -D MAP_INTERPOLATION=1|2 (ini key mapint): density and emission blended with two neighbours across the ray
    "map_c8_mapint1": ("c8mi1", lambda: synth.cartesian_cloud(8, seed=3), dict(mapint=1)),
    "map_oct8_mapint1": ("oct8mi1", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), dict(mapint=1)),
    "map_oct8_mapint2": ("oct8mi2", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), dict(mapint=2)),
    "map_oct8_mapint1_inside": ("oct8mi1", lambda: synth.octree_cloud(8, levels=3, frac=0.15, seed=7), dict(mapint=1, intobs=(4.2, 4.1, 3.9), npix=(16, 9))),
    "map_oct104_mapint2_double": ("oct104mi2", lambda: synth.octree_cloud(104, levels=3, frac=0.002, seed=11), dict(mapint=2, npix=(12, 12), dx=8.0)),
    "map_oct104_double": ("oct104", lambda: synth.octree_cloud(104, levels=3, frac=0.002, seed=11), dict(npix=(24, 24), dx=4.0)),
    "map_c208_entry": ("c208", lambda: synth.cartesian_cloud(208, uniform=1.0), dict(npix=(10, 10), dx=20.0)),
}
LENGTH = np.float32(3.08568e16)


def run_case(name, mapper):
    ref, mk, kw = MAP_CASES[name]
    cloud = mk()
    job = Job(cloud, CSC, ABS=1e-3, SCA=3e-3, OPT=_opt(cloud.CELLS) if kw.get("abu") else None)
    job.ROI_MAP = kw.get("roi")                             # `roimap`: only the emission of cells inside ROI
    job.MAP_INTERPOLATION = kw.get("mapint", 0)             # `mapint` key
    job.LEVEL_THRESHOLD = kw.get("threshold", 0)            # `threshold` key: emission of coarser levels left out of the maps
    emit = np.where(cloud.DENS > 0, np.abs(cloud.DENS) * 1e-3 * np.random.default_rng(1).uniform(0.5, 2, cloud.CELLS), 0).astype(np.float32)
    c = (cloud.NX / 2, cloud.NY / 2, cloud.NZ / 2)
    out = []
    for k in range(1 if kw.get("healpix") else N):
        out.append(mapper(job, emit, OD[k], RA[k], DE[k], kw.get("npix", (12, 10)), kw.get("dx", 0.9), c,
                          kw.get("intobs", NO_INTOBS), kw.get("colden", 0), kw.get("healpix", 0)))
    return job, np.concatenate([o[0].ravel() for o in out]), np.concatenate([o[1].ravel() for o in out])


@pytest.mark.parametrize("name", sorted(MAP_CASES))
def test_map_oracle_bit_exact_vs_reference_golden(name, oracle_libm):
    job, m, t = run_case(name, lambda job, emit, d, r, e, npix, dx, c, io, cd, hp:
                         oracle_mapping(oracle_libm, job, emit, d, r, e, npix, dx, c, io, cd, LENGTH, hp))
    assert (m > 0).sum() > 30
    assert np.array_equal(m.view(np.uint32), GOLD[name + "_map"].view(np.uint32))
    assert np.array_equal(t.view(np.uint32), GOLD[name + "_tau"].view(np.uint32))


@pytest.mark.parametrize("name", ["map_c8", "map_oct8_inside", "map_oct8_healpix"])
def test_map_soc_math_close(name, oracle_soc):
    job, m, t = run_case(name, lambda job, emit, d, r, e, npix, dx, c, io, cd, hp:
                         oracle_mapping(oracle_soc, job, emit, d, r, e, npix, dx, c, io, cd, LENGTH, hp))
    assert np.abs(m - GOLD[name + "_map"]).max() < 1e-5 * GOLD[name + "_map"].max()


def test_map_files_of_the_driver(tmp_path):
    """asoc.py writes map_dir_XX.bin (int32 NPIX.x, NPIX.y + one float32 image per emitted frequency, ASOC.py:2992-2996,
    3134) with the scaling of ASOC.py:2997-2998, 3099"""
    from oracle_engine import OracleEngine
    from soc_amd import files
    from soc_amd.asoc import AbsorptionRun
    from soc_amd.ini import User
    from test_host import _write_model
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    extra = ("noabsorbed\niterations 1\ntemperature %s/T.bin\nemitted %s/em.bin\nmapping 12 10 0.8\ndirection 30 40\ndirection 90 0\n" % (d, d))
    ini = _write_model(d, cloud, extra=extra)
    txt = open(ini).read().replace("nosolve\n", "").replace("nomap\n", "").replace("absorbed %s/abs.data\n" % d, "")
    open(ini, "w").write(txt)
    os.chdir(d)
    run = AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0)
    run.run()
    for idir in (0, 1):
        raw = np.fromfile("map_dir_%02d.bin" % idir, np.int32, 2)
        assert list(raw) == [12, 10]
        maps = np.fromfile("map_dir_%02d.bin" % idir, np.float32, offset=8).reshape(3, 10, 12)
        assert (maps >= 0).all() and maps.sum() > 0
    # one image re-derived: EMIT = KK*FREQ*EMITTED, KK = 1e23/FACTOR*PLANCK/(4 pi)*GL*PARSEC
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    KK = (1.0e23 / launch.FACTOR) * launch.PLANCK / (4.0 * np.pi) * 0.5 * launch.PARSEC
    i = 1
    emit = np.asarray(KK * float(FFREQ[i]) * run.EMITTED[:, i], np.float32)
    _, OD2, RA2, DE2 = launch.set_observer_directions([math.radians(30), math.radians(90)], [math.radians(40), 0.0])
    job = Job(cloud, CSC, ABS=AFABS[0][i], SCA=AFSCA[0][i])
    want, _ = oracle_mapping(Oracle("soc"), job, emit, OD2[0], RA2[0], DE2[0], (12, 10), 0.8, (3.0, 3.0, 3.0))
    maps = np.fromfile("map_dir_00.bin", np.float32, offset=8).reshape(3, 10, 12)
    assert np.array_equal(maps[i].ravel(), want)
    # `mapint 1`: the same run with -D MAP_INTERPOLATION=1 (ASOC_aux.py:330, ASOC.py:352,362)
    open(ini, "w").write(txt + "mapint 1\n")
    os.remove(os.path.join(d, "em.bin"))
    run2 = AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0)
    run2.run()
    assert np.array_equal(run2.EMITTED, run.EMITTED)
    job.MAP_INTERPOLATION = 1
    want1, _ = oracle_mapping(Oracle("soc"), job, emit, OD2[0], RA2[0], DE2[0], (12, 10), 0.8, (3.0, 3.0, 3.0))
    maps1 = np.fromfile("map_dir_00.bin", np.float32, offset=8).reshape(3, 10, 12)
    assert np.array_equal(maps1[i].ravel(), want1) and not np.array_equal(want1, want)
    assert abs(want1.sum() / want.sum() - 1) < 0.2           # a blend of neighbouring cells, not another quantity


def test_fits_maps_and_column_density(tmp_path):
    """`fits ra de prefix` + `mapum`: one FITS image per selected frequency instead of map_dir_00.bin (ASOC.py:2977-2996,
    3143-3148); `savetau file -1 <um>`: column density with the first mapped frequency, always a FITS image (:3152-3159),
    optical depth at <um> (:3160-3171).  Headers as MakeFits writes them (ASOC_aux.py:1723-1768)."""
    from oracle_engine import OracleEngine
    from soc_amd import files
    from soc_amd.asoc import AbsorptionRun
    from soc_amd.ini import User
    from test_host import _write_model
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    extra = ("noabsorbed\niterations 1\ntemperature %s/T.bin\nemitted %s/em.bin\nmapping 12 10 0.8\ndirection 30 40\n"
             "fits 83.8 -5.4 img\nmapum 0.641\nsavetau %s/sv -1 0.75\ndistance 400\n" % (d, d, d))
    ini = _write_model(d, cloud, extra=extra)
    txt = open(ini).read().replace("nosolve\n", "").replace("nomap\n", "").replace("absorbed %s/abs.data\n" % d, "")
    open(ini, "w").write(txt)
    os.chdir(d)
    U = User(ini)
    assert U.FITS == 1 and len(U.SINGLE_MAP_FREQ) == 1
    run = AbsorptionRun(U, OracleEngine("soc"), verbose=0)
    run.run()
    assert not os.path.exists("map_dir_00.bin")
    hdr, img = files.read_fits("img_0.64.fits")
    assert img.shape == (10, 12) and hdr["CTYPE1"] == "RA---TAN" and hdr["CRVAL1"] == 83.8 and hdr["CRVAL2"] == -5.4
    pix = 0.5 * 0.8 / 400.0
    assert abs(hdr["CDELT2"] - pix * 180.0 / math.pi) < 1e-15 and hdr["CDELT1"] == -hdr["CDELT2"] and hdr["CRPIX1"] == 7.0 and hdr["CRPIX2"] == 6.0
    assert not os.path.exists("img_0.75.fits") and not os.path.exists("img_0.56.fits")      # mapum picks one frequency
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    KK = (1.0e23 / launch.FACTOR) * launch.PLANCK / (4.0 * np.pi) * 0.5 * launch.PARSEC
    _, OD, RA, DE = launch.set_observer_directions([math.radians(30)], [math.radians(40)])
    i = 1
    emit = np.asarray(KK * float(FFREQ[i]) * run.EMITTED[:, i], np.float32)
    want, _ = oracle_mapping(Oracle("soc"), Job(cloud, CSC, ABS=AFABS[0][i], SCA=AFSCA[0][i]), emit, OD[0], RA[0], DE[0], (12, 10), 0.8, (3.0, 3.0, 3.0))
    assert np.array_equal(img.ravel(), want)
    # column density (first mapped frequency, 0.75 um, which also asks for tau: tau wins there, ASOC.py:3056) ...
    _, tau = files.read_fits(os.path.join(d, "sv_tau_0.75.fits"))
    _, wtau = oracle_mapping(Oracle("soc"), Job(cloud, CSC, ABS=AFABS[0][0], SCA=AFSCA[0][0]), np.zeros(cloud.CELLS, np.float32), OD[0], RA[0], DE[0],
                             (12, 10), 0.8, (3.0, 3.0, 3.0))
    assert np.array_equal(tau.ravel(), wtau) and tau.max() > 0
    assert not os.path.exists(os.path.join(d, "sv_colden.fits"))
    # ... and without a tau request at the first frequency the column density is written
    open(ini, "w").write(txt.replace("savetau %s/sv -1 0.75" % d, "savetau %s/sv -1" % d))
    AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0).run()
    _, col = files.read_fits(os.path.join(d, "sv_colden.fits"))
    _, LENGTH = launch.kernel_literals(0.5)
    _, wcol = oracle_mapping(Oracle("soc"), Job(cloud, CSC, ABS=AFABS[0][0], SCA=AFSCA[0][0]), np.zeros(cloud.CELLS, np.float32), OD[0], RA[0], DE[0],
                             (12, 10), 0.8, (3.0, 3.0, 3.0), save_colden=1, LENGTH=LENGTH)
    assert np.array_equal(col.ravel(), wcol) and col.max() > 0


_PS4 = np.array([[4.3, 4.2, 4.1], [2.5, 6.5, 3.3], [0.2, 0.3, 7.9], [7.7, 0.4, 0.6]], np.float32)


@pytest.mark.skipif(not RefMap.available("oct8"), reason="reference builds (oracle/_ref) not present")
def test_pstau_oracle_bit_exact_vs_reference(oracle_libm):
    """PSTau (kernel_ASOC_map.c:1545-1584) compiled from the reference, scalar and per-cell opacities"""
    from oracle.pyoracle import oracle_pstau
    o8 = synth.octree_cloud(8, levels=3, frac=0.15, seed=7)
    LEN = float("%.5e" % (0.01 * 3.08567758e18))              # the -D LENGTH of the build (oracle/build.py)
    for ref, opt in (("oct8", None), ("c8abu", True)):
        cloud = o8 if ref == "oct8" else synth.cartesian_cloud(8, seed=3)
        job = Job(cloud, CSC, ABS=1e-3, SCA=3e-3, OPT=_opt(cloud.CELLS) if opt else None)
        for k in range(N):
            c, t = oracle_pstau(oracle_libm, job, _PS4, OD[k], LEN)
            c2, t2 = RefMap(ref).pstau(job, oracle_libm.parents(job), _PS4, OD[k])
            assert np.array_equal(c.view(np.uint32), c2.view(np.uint32)) and np.array_equal(t.view(np.uint32), t2.view(np.uint32))
            assert (t > 0).all() and (c > 0).all()


@pytest.mark.gpu
def test_pstau_hip_bit_identical_to_oracle(engine, oracle_soc):
    from oracle.pyoracle import oracle_pstau
    for cloud, opt in ((synth.octree_cloud(8, levels=3, frac=0.15, seed=7), False), (synth.cartesian_cloud(8, seed=3), True),
                       (synth.octree_cloud(104, levels=3, frac=0.002, seed=11), False)):
        job = Job(cloud, CSC, ABS=1e-3, SCA=3e-3, OPT=_opt(cloud.CELLS) if opt else None)
        ps = _PS4 * np.float32(cloud.NX / 8.0)
        engine.set_cloud(cloud)
        engine.set_opt(job.OPT)
        for k in range(N):
            c, t = engine.ps_tau(ps, OD[k], job.ABS, job.SCA, LENGTH)
            c2, t2 = oracle_pstau(oracle_soc, job, ps, OD[k], LENGTH)
            assert np.array_equal(c.view(np.uint32), c2.view(np.uint32)) and np.array_equal(t.view(np.uint32), t2.view(np.uint32))
    engine.set_opt(None)


def test_pssavetau_file_of_the_driver(tmp_path):
    """`pssavetau file um`: <file>_<idir>.dat, one line per point source (ASOC.py:3576-3645)"""
    from oracle.pyoracle import oracle_pstau
    from oracle_engine import OracleEngine
    from soc_amd import files
    from soc_amd.asoc import AbsorptionRun
    from soc_amd.ini import User
    from test_host import _write_model
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    ini = _write_model(d, cloud, with_ps=True, extra="mapping 12 10 0.8\ndirection 30 40\ndirection 90 0\npssavetau %s/pst 0.64\n" % d)
    os.chdir(d)
    U = User(ini)
    run = AbsorptionRun(U, OracleEngine("soc"), verbose=0)
    run.run()
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    i = int(np.argmin(np.abs(FFREQ - U.pssavetau_freq)))
    assert i == 1
    _, OD2, _, _ = launch.set_observer_directions([math.radians(30), math.radians(90)], [math.radians(40), 0.0])
    _, LEN = launch.kernel_literals(0.5)
    for idir in (0, 1):
        rows = [l.split() for l in open(os.path.join(d, "pst_%d.dat" % idir))]
        assert len(rows) == 1 and rows[0][0] == "0"
        c, t = oracle_pstau(Oracle("soc"), Job(cloud, CSC, ABS=AFABS[0][i], SCA=AFSCA[0][i]), U.PSPOS[:1, :3], OD2[idir], LEN)
        assert rows[0][1] == "%.4e" % c[0] and rows[0][2] == "%.4e" % t[0] and t[0] > 0


def test_healpix_map_file_of_the_driver(tmp_path):
    """`mapping 4 -1 1.0` + `perspective x y z`: map_dir_00_H.bin = int32 [NSIDE, -1], int32 [frequencies, LEVELS], one
    float32 [12*NSIDE^2] all-sky map per emitted frequency inside `wavelength` (the loop of ASOC.py:3240-3309)"""
    from oracle_engine import OracleEngine
    from soc_amd import files
    from soc_amd.asoc import AbsorptionRun
    from soc_amd.ini import User
    from test_host import _write_model
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    extra = ("noabsorbed\niterations 1\ntemperature %s/T.bin\nemitted %s/em.bin\nmapping 4 -1 1.0\nperspective 0.5 3.1 2.9\n"
             "wavelength 0.6 0.8\n" % (d, d))
    ini = _write_model(d, cloud, extra=extra)
    txt = open(ini).read().replace("nosolve\n", "").replace("nomap\n", "").replace("absorbed %s/abs.data\n" % d, "")
    open(ini, "w").write(txt)
    os.chdir(d)
    U = User(ini)
    assert U.NPIX[1] < 0
    run = AbsorptionRun(U, OracleEngine("soc"), verbose=0)
    run.run()
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    sel = [i for i in range(3) if U.MAP_FREQ[0] <= FFREQ[i] <= U.MAP_FREQ[1]]
    assert 0 < len(sel) < 3                                   # `wavelength` restricts the frequencies in the file
    head = np.fromfile("map_dir_00_H.bin", np.int32, 4)
    assert list(head) == [4, -1, len(sel), cloud.LEVELS]
    maps = np.fromfile("map_dir_00_H.bin", np.float32, offset=16).reshape(len(sel), 12 * 16)
    assert (maps >= 0).all() and (maps[0] > 0).mean() > 0.5   # the observer sits in a heated surface cell of this opaque toy cloud
    KK = (1.0e23 / launch.FACTOR) * launch.PLANCK / (4.0 * np.pi) * 0.5 * launch.PARSEC
    i = sel[0]
    emit = np.asarray(run.EMITTED[:, i] * np.float32(KK) * np.float32(FFREQ[i]), np.float32)
    _, OD, RA, DE = launch.set_observer_directions(U.OBS_THETA, U.OBS_PHI)
    job = Job(cloud, CSC, ABS=AFABS[0][i], SCA=AFSCA[0][i])
    want, _ = oracle_mapping(Oracle("soc"), job, emit, OD[0], RA[0], DE[0], (4, -1), 1.0, (3.0, 3.0, 3.0), U.INTOBS, 0, 1.0, 4)
    assert np.array_equal(maps[0], want.ravel())


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MAP_CASES))
def test_map_hip_bit_identical_to_oracle(name, engine, oracle_soc):
    def gpu(job, emit, d, r, e, npix, dx, c, io, cd, hp):
        engine.set_cloud(job.cloud)
        engine.set_opt(job.OPT)
        engine.set_map_threshold(job.LEVEL_THRESHOLD)
        engine.set_map_interpolation(job.MAP_INTERPOLATION)
        engine.set_map_roi(job.ROI_MAP)
        return engine.map(emit, d, r, e, npix, dx, c, job.ABS, job.SCA, INTOBS=io, save_colden=cd, LENGTH=LENGTH, healpix=hp)
    job, m, t = run_case(name, gpu)
    engine.set_map_threshold(0)
    engine.set_map_interpolation(0)
    engine.set_map_roi(None)
    _, mo, to = run_case(name, lambda job, emit, d, r, e, npix, dx, c, io, cd, hp:
                         oracle_mapping(oracle_soc, job, emit, d, r, e, npix, dx, c, io, cd, LENGTH, hp))
    assert np.array_equal(m.view(np.uint32), mo.view(np.uint32))
    assert np.array_equal(t.view(np.uint32), to.view(np.uint32))
    engine.set_opt(None)
