"""Host side of the scattering run (soc_amd/asocs.py) on the oracle-backed engine: launch
formulas, seeds, unit conversion and the outcoming.socs layout of ASOCS.py, each launch
re-derived independently here from the reference's formulas (file:line in the comments)."""
import math
import os

import numpy as np
import pytest

from soc_amd import files, launch, synth
from soc_amd.asoc import UnsupportedOption
from soc_amd.asocs import ScatteringRun
from soc_amd.ini import User
from test_host import _write_model

SCA_INI = ("direction 30 40\ndirection 90 0\npoints 12 10\nmapum 0.5 0.7\nremit 0.1 1000\n")


def _ini(d, cloud, **kw):
    extra = kw.pop("extra", "")
    ini = _write_model(d, cloud, extra="mapping 12 10 0.8\ndirection 30 40\ndirection 90 0\n" + extra, **kw)
    return ini


def test_observer_directions_pins():
    """ASOC_aux.py:1155-1176: theta from +Z, phi from +X; RA to the right, DE up."""
    n, OD, RA, DE = launch.set_observer_directions([0.5 * math.pi], [0.0])
    assert n == 1
    np.testing.assert_allclose(OD[0, :3], [1.0, 1e-5, 1e-5], atol=1e-7)     # observer on +X, zeros nudged
    np.testing.assert_allclose(RA[0, :3], [0.0, 1.0, 0.0], atol=1e-7)       # +Y to the right
    np.testing.assert_allclose(DE[0, :3], [0.0, 0.0, 1.0], atol=1e-7)       # +Z up
    n, OD, RA, DE = launch.set_observer_directions([0.0], [0.0])            # from +Z
    np.testing.assert_allclose(OD[0, :3], [1e-5, 1e-5, 1.0], atol=1e-6)
    np.testing.assert_allclose(DE[0, :3], [-1.0, 0.0, 0.0], atol=1e-6)
    n, OD, RA, DE = launch.set_observer_directions([], [])                  # default: one observer on +X
    assert n == 1 and abs(OD[0, 0] - 1) < 1e-6
    for M in (launch.set_observer_directions([0.3, 1.1], [0.2, 4.0]),):
        _, O, R, D = M
        for i in range(2):                                                  # orthonormal triad
            o, r, dd = O[i, :3].astype(float), R[i, :3].astype(float), D[i, :3].astype(float)
            assert abs(o @ r) < 1e-6 and abs(o @ dd) < 1e-6 and abs(r @ dd) < 1e-6
            np.testing.assert_allclose(np.cross(r, dd), o, atol=1e-6)


def test_scattering_run_on_oracle_engine(tmp_path):
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle, ScaView, oracle_sim_sca
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    ini = _ini(d, cloud, with_ps=True)
    os.chdir(d)
    U = User(ini)
    run = ScatteringRun(U, OracleEngine("soc"))
    OUTC = run.run()
    # packet counts (ASOCS.py:84-93) and packet.info
    assert run.GLOBAL_0 == 128 * 8 and run.PSPAC == 4000 and run.BGPAC == launch.Fix(launch.Fix(20000, 216), 32)
    assert list(np.fromfile("packet.info", np.int32)) == [run.BGPAC, run.PSPAC, run.DFPAC, run.CLPAC]
    # file layout (ASOCS.py:409-416)
    head = np.fromfile("outcoming.socs", np.int32, 3)
    assert list(head) == [10, 12, 3]
    FF, data = files.read_outcoming("outcoming.socs", 2)
    assert data.shape == (3, 2, 10, 12) and np.array_equal(data, OUTC)
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    assert np.array_equal(FF, FFREQ.astype(np.float32))
    # independent evaluation: PS block + BG block of frequency 1
    orc = Oracle("soc")
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 3, 500)
    _, OD, RA, DE = launch.set_observer_directions([math.radians(30), math.radians(90)], [math.radians(40), 0.0])
    view = ScaView(OD, RA, DE, NPIX=(12, 10), MAP_DX=0.8, CENTRE=(3.0, 3.0, 3.0), FFS=1)
    i = 1
    FREQ = float(FFREQ[i])
    seed = math.fmod(math.pi / 4 + launch.SEED0 + i * launch.SEED1, 1.0)               # ASOCS.py:634
    # point sources: ASOCS.py:457-465 -- GLOBAL work items x BATCH packets per source
    GLOBAL = 1024
    BATCH = int(max(1, 4000 / GLOBAL))
    WPS = 1.0 / (launch.PLANCK * GLOBAL * BATCH * (0.5 * launch.PARSEC) ** 2)
    LPS = np.fromfile(os.path.join(d, "ps.bin"), np.float32)
    PS = np.asarray([LPS[i]], np.float32) * np.float32(WPS) / np.float32(FREQ)
    job = Job(cloud, FCSC[0, i], ABS=AFABS[0][i], SCA=AFSCA[0][i], SOURCE=0, BATCH=BATCH, SEED=seed, GLOBAL=GLOBAL,
              PSPOS=np.array([[3.3, 3.2, 3.1]], np.float32), PS=PS, DSC=FDSC[0, i])
    img, _ = oracle_sim_sca(orc, job, view, 2)
    # background: ASOCS.py:470-478
    IBG = np.fromfile(os.path.join(d, "bg.bin"), np.float32)
    L = launch.bg_launch(run.BGPAC, 216)
    job = Job(cloud, FCSC[0, i], ABS=AFABS[0][i], SCA=AFSCA[0][i], SOURCE=1, BATCH=L["BATCH"], SEED=seed,
              GLOBAL=L["GLOBAL"], BG=np.float32(float(IBG[i]) * L["WBG"] / FREQ), DSC=FDSC[0, i])
    img2, _ = oracle_sim_sca(orc, job, view, 0)
    want = (img + img2).reshape(2, 10, 12) * np.float32(FREQ * 1.0e23 * launch.PLANCK / (0.8 * 0.8))   # ASOCS.py:890
    np.testing.assert_allclose(OUTC[i], want, rtol=1e-6)
    assert OUTC[i].sum() > 0


def test_scattering_run_with_dust_emission(tmp_path):
    """CLPAC block (ASOCS.py:733-881): EMIT from the emitted file, seed without SEED0, BATCH = CLPAC/CELLS."""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle, ScaView, oracle_sim_sca
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(5, seed=2)
    emitted = (1e-3 * cloud.DENS[:, None] * np.array([1.0, 2.0, 0.5])[None, :]).astype(np.float32)
    files.write_emitted(os.path.join(d, "em.bin"), emitted)
    ini = _ini(d, cloud, extra="emitted %s/em.bin\ncellpackets %d\nbgpackets 0\nglobal 128\n" % (d, 2 * cloud.CELLS))
    os.chdir(d)
    run = ScatteringRun(User(ini), OracleEngine("soc"))
    OUTC = run.run()
    assert run.CLPAC == launch.Fix(launch.Fix(2 * cloud.CELLS, cloud.CELLS), 32) and run.BGPAC == 0
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 3, 500)
    _, OD, RA, DE = launch.set_observer_directions([math.radians(30), math.radians(90)], [math.radians(40), 0.0])
    view = ScaView(OD, RA, DE, NPIX=(12, 10), MAP_DX=0.8, CENTRE=(2.5, 2.5, 2.5), FFS=1)
    i = 2
    EMIT = (emitted[:, i] * np.float32(1.0e-20 * 0.5 * launch.PARSEC) * cloud.DENS).astype(np.float32)
    job = Job(cloud, FCSC[0, i], ABS=AFABS[0][i], SCA=AFSCA[0][i], SOURCE=2, BATCH=max(1, int(run.CLPAC / cloud.CELLS)),
              SEED=math.fmod(math.pi / 4 + i * launch.SEED1, 1.0), GLOBAL=1024, EMIT=EMIT, DSC=FDSC[0, i])
    img, n = oracle_sim_sca(Oracle("soc"), job, view, 1)
    want = img.reshape(2, 10, 12) * np.float32(float(FFREQ[i]) * 1.0e23 * launch.PLANCK / (0.8 * 0.8))
    np.testing.assert_allclose(OUTC[i], want, rtol=2e-6)
    assert n > 0 and OUTC[i].sum() > 0


def test_scattering_run_refuses_what_it_cannot_do(tmp_path):
    from oracle_engine import OracleEngine
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(4, seed=1)
    for extra in ("roipac 100\n", "split 1\n"):
        with pytest.raises(UnsupportedOption):
            ScatteringRun(User(_ini(d, cloud, extra=extra)), OracleEngine("soc"))


def test_perspective_and_healpix_background(tmp_path):
    """`perspective x y z` -> one Healpix map (NSIDE outnside) with the header of ASOCS.py:418-426 and the
    solid-angle scaling of :894-896; `hpbg` -> sca SimRAM_HP with the launch of ASOCS.py:480-497."""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle, ScaView, oracle_sim_sca
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(6, seed=4)
    sky = np.random.default_rng(3).lognormal(0, 1, (3, 49152)).astype(np.float32) * 1e-13
    sky.tofile(os.path.join(d, "sky.bin"))
    ini = _write_model(d, cloud, extra="perspective 2.5 3.5 9.0\noutnside 4\nhpbg %s/sky.bin\nbgpackets 30000\n" % d)
    os.chdir(d)
    run = ScatteringRun(User(ini), OracleEngine("soc"))
    OUTC = run.run()
    assert OUTC.shape == (3, 192)
    head = np.fromfile("outcoming.socs", np.int32, 2)
    assert list(head) == [4, 3]
    data = np.fromfile("outcoming.socs", np.float32, offset=8 + 12).reshape(3, 192)
    assert np.array_equal(data, OUTC)
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 3, 500)
    L = launch.hpbg_sca_launch(run.BGPAC, 6, 6, 6)
    assert L["BATCH"] == 1 and L["GLOBAL"] == launch.Fix(run.BGPAC, 64)
    assert np.isclose(L["WBG"], np.pi * 4 * np.pi * (0.5 * math.sqrt(108.0)) ** 2 / (launch.PLANCK * L["GLOBAL"]))
    z = np.zeros((1, 4), np.float32)
    o = np.zeros((1, 4), np.float32)
    o[0, :3] = [2.5, 3.5, 9.0]
    view = ScaView(o, z, z, NPIX=(1, 1), FFS=1, nside=4)
    i = 0
    bg = (np.float32(L["WBG"] / float(FFREQ[i])) * sky[i]).astype(np.float32)
    job = Job(cloud, FCSC[0, i], ABS=AFABS[0][i], SCA=AFSCA[0][i], BATCH=1, SEED=math.fmod(math.pi / 4 + launch.SEED0, 1.0),
              GLOBAL=L["GLOBAL"], HPBG=bg, DSC=FDSC[0, i])
    img, n = oracle_sim_sca(Oracle("soc"), job, view, 3)
    want = img * np.float32(float(FFREQ[i]) * 1.0e23 * launch.PLANCK / (4.0 * np.pi / (12.0 * 16.0)))
    np.testing.assert_allclose(OUTC[i], want, rtol=2e-6)
    assert n > 0 and OUTC[i].sum() > 0


def test_scattering_run_with_several_scattering_functions(tmp_path):
    """-D WITH_MSF in the scattering run (ASOCS.py:26-30, :622-629): the tables of all species per frequency"""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle, ScaView, oracle_sim_sca
    d = str(tmp_path)
    os.chdir(d)
    cloud = synth.cartesian_cloud(5, seed=2)
    rr = np.random.default_rng(4)
    abu = rr.uniform(0.2, 1.0, cloud.CELLS).astype(np.float32)
    abu.tofile(os.path.join(d, "a.abu"))
    GL = 5.0e-7
    ini = _ini(d, cloud, nfreq=2, extra="gridlength %g\n" % GL)
    with open(os.path.join(d, "m2.dust"), "w") as fp:
        fp.write("eqdust\n 1.0e-7\n 0.7e-4\n2\n 4.00000e+14  0.6  2.0e-2  1.2e-1\n 4.67700e+14  0.6  2.5e-2  1.0e-1\n")
    dsc, csc = synth.hg_scattering_table(0.1, 500)
    files.write_scattering_functions(os.path.join(d, "m2.dsc"), np.tile(dsc, (2, 1)), np.tile(csc, (2, 1)))
    txt = open(ini).read().replace("optical %s/m.dust\n" % d, "optical %s/m.dust %s/a.abu\noptical %s/m2.dust\n" % (d, d, d))
    open(ini, "w").write(txt.replace("dsc %s/m.dsc 500\n" % d, "dsc %s/m.dsc 500\ndsc %s/m2.dsc 500\n" % (d, d)))
    run = ScatteringRun(User(ini), OracleEngine("soc"))
    assert run.WITH_MSF
    OUTC = run.run()
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust"), os.path.join(d, "m2.dust")], GL)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc"), os.path.join(d, "m2.dsc")], 2, 500)
    _, OD, RA, DE = launch.set_observer_directions([math.radians(30), math.radians(90)], [math.radians(40), 0.0])
    view = ScaView(OD, RA, DE, NPIX=(12, 10), MAP_DX=0.8, CENTRE=(2.5, 2.5, 2.5), FFS=1)
    ABU = np.stack([abu, np.ones(cloud.CELLS, np.float32)], 1)
    IBG = np.fromfile(os.path.join(d, "bg.bin"), np.float32)
    L = launch.bg_launch(run.BGPAC, cloud.AREA)
    for i in range(2):
        FREQ = float(FFREQ[i])
        OPT = np.zeros((cloud.CELLS, 2), np.float32)
        for k in range(2):
            OPT[:, 0] += ABU[:, k] * AFABS[k][i]
            OPT[:, 1] += ABU[:, k] * AFSCA[k][i]
        msf = ([AFABS[0][i], AFABS[1][i]], [AFSCA[0][i], AFSCA[1][i]], FCSC[:, i, :], ABU)
        seed = math.fmod(math.pi / 4 + launch.SEED0 + i * launch.SEED1, 1.0)
        job = Job(cloud, None, SOURCE=1, BATCH=L["BATCH"], SEED=seed, GLOBAL=L["GLOBAL"],
                  BG=np.float32(float(IBG[i]) * L["WBG"] / FREQ), OPT=OPT, MSF=msf, DSC=FDSC[:, i, :])
        img, n = oracle_sim_sca(Oracle("soc"), job, view, 0)
        want = img.reshape(2, 10, 12) * np.float32(FREQ * 1.0e23 * launch.PLANCK / (0.8 * 0.8))
        assert n > 1000
        np.testing.assert_allclose(OUTC[i], want, rtol=1e-6)


def test_scattering_run_writes_the_fits_cube(tmp_path):
    """`fits` with ONE direction: <scattering>.fits, a cube [frequency, y, x] of the images with the frequencies as header
    comments (ASOCS.py:882-892, MakeFits), beside outcoming.socs"""
    from oracle_engine import OracleEngine
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(5, seed=2)
    ini = _write_model(d, cloud, nfreq=2, extra="mapping 12 10 0.8\ndirection 30 40\nfits 10.0 20.0\nscattering %s/sca\n" % d)
    os.chdir(d)
    U = User(ini)
    assert U.FITS == 1 and U.file_scattering == os.path.join(d, "sca")
    run = ScatteringRun(U, OracleEngine("soc"))
    OUTC = run.run()
    hdr, cube = files.read_fits(os.path.join(d, "sca.fits"))
    assert cube.shape == (2, 10, 12) and np.array_equal(cube, OUTC[:, 0]) and cube.sum() > 0
    assert hdr["NAXIS3"] == 2 and hdr["CTYPE3"] == "channel" and hdr["CRVAL1"] == 10.0 and hdr["CRVAL2"] == 20.0
    assert abs(hdr["CDELT2"] - (0.5 * 0.8 / 1000.0) * 180.0 / math.pi) < 1e-15              # nominal 1 kpc without `distance`
    FFREQ, _, _, _ = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    assert hdr["COMMENT"] == ["F[ %3d ] = %.4e" % (i, f) for i, f in enumerate(FFREQ)]
    assert os.path.exists("outcoming.socs")
