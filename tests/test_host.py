"""Host side of the drop-in: ini parsing, file formats, launch arithmetic, the ASOC driver
loop (on an oracle-backed engine) -- no GPU needed."""
import math
import os

import numpy as np
import pytest

from soc_amd import files, launch, synth
from soc_amd.asoc import AbsorptionRun, UnsupportedOption
from soc_amd.ini import User

MY_INI = """
gridlength      0.01              # root grid cells have a size of 0.01 pc each
cloud           tmp.cloud          # density field (reference to a file)
mapping         64 64 1.0          # output 64x64 pixels, pixel size == root-grid cell size
density         1.0e3              # scale values read from tmp.cloud
seed           -1.0                # random seed for random numbers
directions      0.0  0.0           # observer in direction (theta,phi)
optical         tmp.dust           # dust optical parameters
dsc             tmp.dsc  2500      # dust scattering function
bgpackets       999999             # photon packages simulated from the background
background      bg_intensity.bin   # background intensity at the
iterations      1                  # one iteration is enough
prefix          tmp                # prefix for output files
absorbed        absorbed.data      # save absorptions to a file
emitted         emitted.data       # save dust emission to a file
noabsorbed                         # actually, we integrate absorptions on the fly and skip the *file*
temperature     tmp.T              # save dust temperatures
device          g                  # run calculations on a GPU
CLT                                # temperature caclulations done on the device
CLE                                # emission calculations done on the device
"""


def test_example_ini_matches_hand_derived_values():
    """Expected attributes derived by hand from ASOC_aux.py:239-537 for soc_example.zip/my.ini."""
    U = User(text=MY_INI)
    assert U.GL == 0.01 and U.file_cloud == 'tmp.cloud' and U.KDENSITY == 1.0e3
    assert U.NPIX == (64, 64) and U.MAP_DX == 1.0
    assert U.SEED == -1.0
    assert U.OBS_THETA == [0.0] and U.OBS_PHI == [0.0]
    assert U.file_optical == ['tmp.dust'] and U.file_abundance == ['#']
    assert U.file_scafunc == ['tmp.dsc'] and U.DSC_BINS == 2500
    assert U.BGPAC == 999999 and U.file_background == 'bg_intensity.bin'
    assert U.ITERATIONS == 1 and U.file_absorbed == 'absorbed.data' and U.file_emitted == 'emitted.data'
    assert U.NOABSORBED == 1 and U.file_temperature == 'tmp.T' and U.DEVICES == 'g'
    assert 'CLT' in U.KEYS and 'CLE' in U.KEYS and U.KEYS['prefix'] == ['tmp']
    assert U.NOSOLVE == 0 and U.NOMAP == 0
    assert U.Validate()


def test_prefix_matching_gotchas():
    """SURVEY.md Appendix A, "gotchas worth a unit test each"."""
    U = User(text="density 5\ndens 7\n")                       # both hit `dens`
    assert U.KDENSITY == 7.0
    assert User(text="emitted x.e\n").file_emitted == 'x.e'     # emit matches emitted
    assert User(text="scatter out.s\n").file_scattering == 'out.s'
    U = User(text="background b.bin 2.0\nhpbg h.bin 3.0 1\n")   # share scale_background
    assert U.scale_background == 3.0 and U.HPBG_WEIGHTED == 1 and U.file_hpbg == 'h.bin'
    U = User(text="diffpack 500\ncellpackets 1e3\n")            # cellpac>0 overrides diffpac
    assert U.CLPAC == 1000 and U.DFPAC == 1000
    with pytest.raises(ValueError):
        User(text="diffpack 1e3\n")                             # int('1e3') fails, as in the reference
    U = User(text="cloud c\npointsource 1 2 3 ps.bin 2.5\npspackets 0\n")
    assert U.NO_PS == 1 and U.PS_SCALING[0] == 2.5 and list(U.PSPOS[0, :3]) == [1.0, 2.0, 3.0]
    U.Validate()
    assert U.NO_PS == 0                                         # pspac==0 discards point sources
    assert User(text="seed 17\n").SEED == 1.0                   # clipped to [-1,1]
    assert User(text="bgpackets 1e8\nbgpac 2e6\n").BGPAC == 2000000
    U = User(text="simum 0.60 0.68\nremit 10 1000\n")
    assert abs(U.SIM_F[0] - launch.um2f(0.68)) < 1 and abs(U.SIM_F[1] - launch.um2f(0.60)) < 1
    U = User(text="optical a.dust a.abu\noptical b.dust #\noptical c.dust\n")
    assert U.file_abundance == ['a.abu', '#', '#']
    U = User(text="directions 30 60\ndirewei 1 0.5\nnoabsorbed\nNOMAP\ndustem\n")
    assert len(U.OBS_THETA) == 1 and U.DIR_WEIGHT == [1, 0.5] and U.NOMAP == 1 and U.SAVE_INTENSITY == 1
    U = User(text="emweight 1 0.5 10 0.1 4\nlevels 3\npsmethod 2\nglobal 65536\n")
    assert U.USE_EMWEIGHT == 1 and U.EMWEIGHT_LIM == [0.5, 10.0, 0.1] and U.EMWEIGHT_SKIP == 4
    assert U.LEVELS == 3 and U.PS_METHOD == 2 and U.GLOBAL == 65536


def test_launch_arithmetic_pins():
    """C1 AREA 6144, GLOBAL 49152, BATCH 20, BGPAC 983040; C2 786432/127/99876864;
    256^3: 3145728/318/1000341504; seed(pi/4, IFREQ 0) = 0.6004964104."""
    for N, bg, G, B, P in ((32, 1000000, 49152, 20, 983040), (128, 100000000, 786432, 127, 99876864),
                           (256, 1000000000, 3145728, 318, 1000341504)):
        L = launch.bg_launch(launch.packet_counts(bg, 0, 0, 0, 6 * N * N, N ** 3)["BGPAC"], 6 * N * N)
        assert (L["GLOBAL"], L["BATCH"], L["PACKETS"]) == (G, B, P)
        assert L["WBG"] == math.pi / (launch.PLANCK * 8 * B)
    assert abs(launch.launch_seed(math.pi / 4, 0) - 0.6004964104) < 1e-9
    L = launch.ps_launch(launch.Fix(1000000000, 32), 1, 0.02)
    assert (L["GLOBAL"], L["BATCH"], L["PACKETS"]) == (32768, 30517, 999981056)
    assert launch.Fix(10, 8) == 16 and launch.Fix(16, 8) == 16
    F = np.asarray([1.0, 2.0, 4.0, 8.0], np.float32)
    assert launch.trapezoid_weight(F, 0) == 0.5 and launch.trapezoid_weight(F, 1) == 2 * 1.5 and launch.trapezoid_weight(F, 3) == 8 * 2.0
    for G in (100, 49152, 786432, 3145729):
        for W in (1, 2, 3, 8):
            parts = [launch.shard_range(G, r, W) for r in range(W)]
            assert sum(c for _, c in parts) == G
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(W - 1))


def test_levels_keyword_cuts_the_hierarchy(tmp_path):
    """`levels N` (ASOC_aux.py:748-761, OT_cut_levels :651-713, AverageParent kernel_OT_tools.c:5-24): a cloud with more
    levels is cut from the bottom up, parents become leaves with the float32 mean of their eight children; the cut
    cloud is written as <cloud>.MAX<N> and used"""
    cloud = synth.octree_cloud(6, levels=4, frac=0.3, seed=5)
    fn = str(tmp_path / "c.cloud")
    cloud.write(fn)
    cut = files.read_cloud(fn, max_levels=2)
    assert os.path.exists(fn + ".MAX2") and cut.LEVELS == 2
    assert np.array_equal(cut.LCELLS, cloud.LCELLS[:2]) and cut.CELLS == int(cloud.LCELLS[:2].sum())
    # scalar restatement, bottom up
    H = [np.array(cloud.DENS[cloud.OFF[l]:cloud.OFF[l] + cloud.LCELLS[l]]) for l in range(cloud.LEVELS)]
    for i in (2, 1):
        for j in range(len(H[i])):
            if not (H[i][j] > np.float32(1e-9)):
                k = int(np.float32(-H[i][j]).view(np.int32))
                f = np.float32(0.0)
                for c in H[i + 1][k:k + 8]:
                    f = np.float32(f + c)
                H[i][j] = np.float32(f / np.float32(8.0))
    assert np.array_equal(cut.DENS.view(np.uint32), np.concatenate(H[:2]).view(np.uint32))
    assert (cut.DENS[cut.OFF[1]:] > 0).all()                      # the last level holds leaves only
    links = cut.DENS[:cut.OFF[1]] <= 0
    assert np.array_equal(links, cloud.DENS[:cloud.OFF[1]] <= 0)  # root links stay
    # mass is conserved to rounding: volume-weighted density
    def mass(c):
        return sum(float(np.sum(np.where(c.DENS[c.OFF[l]:c.OFF[l] + c.LCELLS[l]] > 0, c.DENS[c.OFF[l]:c.OFF[l] + c.LCELLS[l]], 0).astype(np.float64))) / 8.0 ** l
                   for l in range(c.LEVELS))
    assert abs(mass(cut) / mass(cloud) - 1) < 1e-6
    same = files.read_cloud(fn, max_levels=4)
    assert same.LEVELS == 4 and not os.path.exists(fn + ".MAX4")


def test_cloud_file_round_trip(tmp_path):
    for cloud in (synth.cartesian_cloud(5, seed=1, NY=4, NZ=3), synth.octree_cloud(6, levels=3, frac=0.2, seed=2)):
        fn = str(tmp_path / "c.cloud")
        cloud.write(fn)
        raw = np.fromfile(fn, np.int32, 5)
        assert list(raw) == [cloud.NX, cloud.NY, cloud.NZ, cloud.LEVELS, cloud.CELLS]
        back = files.read_cloud(fn)
        assert np.array_equal(back.DENS.view(np.uint32), cloud.DENS.view(np.uint32))
        assert np.array_equal(back.OFF, cloud.OFF) and np.array_equal(back.LCELLS, cloud.LCELLS)
        scaled = files.read_cloud(fn, kdensity=1e-9)
        leaf = cloud.DENS > 0
        assert np.array_equal(scaled.DENS[~leaf].view(np.uint32), cloud.DENS[~leaf].view(np.uint32))   # links untouched
        assert (scaled.DENS[leaf] >= np.float32(1e-6)).all()                                            # clip floor


def _write_model(d, cloud, nfreq=3, with_ps=False, with_diffuse=False, extra=""):
    cloud.write(os.path.join(d, "m.cloud"))
    freq = np.asarray([4.0e14, 4.677e14, 5.4e14][:nfreq])
    with open(os.path.join(d, "m.dust"), "w") as fp:
        fp.write("eqdust\n 1.0e-7\n 1.0e-4\n%d\n" % nfreq)
        for f in freq:
            fp.write(" %.5e  0.6  %.5e  %.5e\n" % (f, 3.0e-2, 9.0e-2))
    dsc, csc = synth.hg_scattering_table(0.6, 500)
    files.write_scattering_functions(os.path.join(d, "m.dsc"), np.tile(dsc, (nfreq, 1)), np.tile(csc, (nfreq, 1)))
    np.asarray([1e-13, 2e-13, 1.5e-13][:nfreq], np.float32).tofile(os.path.join(d, "bg.bin"))
    ini = ("gridlength 0.5\ncloud %s/m.cloud\noptical %s/m.dust\ndsc %s/m.dsc 500\nbackground %s/bg.bin\n"
           "bgpackets 20000\nseed 0.7853981634\niterations 1\nnosolve\nnomap\nabsorbed %s/abs.data\n"
           "csave %s/ctabs.bin\ndevice g\nverbose 0\n" % (d, d, d, d, d, d))
    if with_ps:
        np.asarray([1e20, 2e20, 1.5e20][:nfreq], np.float32).tofile(os.path.join(d, "ps.bin"))
        ini += "pointsource 3.3 3.2 3.1 %s/ps.bin\npspackets 4000\nglobal 128\n" % d
    if with_diffuse:
        files.write_diffuserad(os.path.join(d, "diff.bin"),
                               np.where(cloud.DENS > 0, 1e-30 * cloud.DENS, 0)[:, None] * np.ones((1, nfreq)))
        ini += "diffuse %s/diff.bin\ndiffpack %d\nglobal 128\n" % (d, 2 * cloud.CELLS)
    ini += extra
    with open(os.path.join(d, "m.ini"), "w") as fp:
        fp.write(ini)
    return os.path.join(d, "m.ini")


def test_driver_loop_on_oracle_engine(tmp_path):
    """The II x IFREQ loop, weights, seeds and output files, with every launch independently
    re-derived from the reference formulas here."""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    ini = _write_model(d, cloud, with_ps=True, with_diffuse=True)
    U = User(ini)
    run = AbsorptionRun(U, OracleEngine("soc"))
    os.chdir(d)
    CTABS, FABS = run.run()
    assert list(np.fromfile(os.path.join(d, "packet.info"), np.int32)) == [run.BGPAC, run.PSPAC, run.DFPAC, run.CLPAC]
    assert run.BGPAC == launch.Fix(launch.Fix(20000, 216), 32) and run.PSPAC == 4000
    # independent evaluation of one block: background, all three frequencies
    orc = Oracle("soc")
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 3, 500)
    IBG = np.fromfile(os.path.join(d, "bg.bin"), np.float32)
    L = launch.bg_launch(run.BGPAC, 216)
    T = np.zeros(cloud.CELLS, np.float32)
    F = np.zeros((cloud.CELLS, 3), np.float32)
    for i in range(3):
        job = Job(cloud, FCSC[0, i], ABS=AFABS[0][i], SCA=AFSCA[0][i], SOURCE=1, BATCH=L["BATCH"],
                  SEED=launch.launch_seed(math.pi / 4, i), BG=np.float32(float(IBG[i]) * L["WBG"] / float(FFREQ[i])),
                  TW=launch.trapezoid_weight(FFREQ, i), GLOBAL=L["GLOBAL"], WITH_INT=1)
        I = np.zeros(cloud.CELLS, np.float32)
        orc.sim(job, 0, TABS=T, INT=I)
        F[:, i] = I
    # the file holds PS + BG + diffuse; check the structure, then the BG block through a BG-only run
    absd = files.read_absorbed(os.path.join(d, "abs.data"))
    assert absd.shape == (cloud.CELLS, 3)
    assert (absd[cloud.DENS <= 0] == np.float32(-1e20)).all() and (absd[cloud.DENS > 0] >= 0).all()
    assert np.array_equal(np.fromfile(os.path.join(d, "ctabs.bin"), np.float32), CTABS)
    ini2 = _write_model(d, cloud)            # background only
    run2 = AbsorptionRun(User(ini2), OracleEngine("soc"))
    C2, F2 = run2.run()
    assert np.array_equal(C2, T)
    want = files.scale_absorbed(F.copy(), cloud, 0.5)
    assert np.allclose(files.read_absorbed(os.path.join(d, "abs.data")), want, rtol=1e-6)


def test_unsupported_options_fail_loudly(tmp_path):
    from oracle_engine import OracleEngine
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(4, seed=1)
    for extra in ("split 1\n", "stepweight 1 0.5 3\n", "direweight 1 0.5\n", "psmethod 3\n", "absthin 4\nnnmake 1\nmmapabs 1\n", "polmap bx by bz\n",
                  "mapping 12 10 0.8 2\n", "mapint 3\n", "DEFS -D X=1\n",
                  "reference 1\nsaveint 1\n"):
        with pytest.raises(UnsupportedOption):
            AbsorptionRun(User(_write_model(d, cloud, extra=extra)), OracleEngine("soc"))
    # keys without effect in the reference are accepted: absthin without nnmake (ASOC.py:100-101), and keys that only reach
    # branches refused above or nothing at all
    AbsorptionRun(User(_write_model(d, cloud, extra="absthin 4\ninterpolate 1\nexternalmask m.bin\nsourcemap s.bin\nbgmethod 1\nyshear 0.1\n")),
                  OracleEngine("soc"))


def test_nnmake_writes_the_absorptions_of_every_nth_cell(tmp_path):
    """`nnmake` + `absthin N` (ASOC.py:100-105, :632-638, :1496, :2815-2836, :2871-2873): the absorbed file holds the cells
    0, N, 2N, ... with the scaling and the -1e20 marks of the full file; `absthin` alone changes nothing"""
    from oracle_engine import OracleEngine
    d = str(tmp_path)
    cloud = synth.octree_cloud(5, levels=2, frac=0.15, seed=4)
    os.chdir(d)
    out = {}
    for key, extra in (("full", ""), ("alone", "absthin 3\n"), ("thin", "absthin 3\nnnmake 1\n")):
        ini = _write_model(d, cloud, extra=extra)
        AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0).run()
        out[key] = files.read_absorbed(os.path.join(d, "abs.data"))
    assert out["full"].shape[0] == cloud.CELLS and np.array_equal(out["alone"], out["full"])
    assert out["thin"].shape == ((cloud.CELLS + 2) // 3, out["full"].shape[1])
    assert np.array_equal(out["thin"], out["full"][0::3])
    assert (out["thin"] == -1.0e20).any() and (out["thin"] > 0).any()


def test_healpix_background_block(tmp_path):
    """hpbg runs: launch size and weight of ASOC.py:1050-1059, per-frequency sky arrays of
    ASOC.py:1196-1214, SimRAM_HP launch -- re-derived here for one frequency."""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(6, seed=4)
    sky = np.random.default_rng(3).lognormal(0, 1, (3, 49152)).astype(np.float32) * 1e-13
    sky.tofile(os.path.join(d, "sky.bin"))
    for weighted in (0, 1):
        ini = _write_model(d, cloud, extra="hpbg %s/sky.bin 2.0 %d\nbgpackets 60000\n" % (d, weighted))
        run = AbsorptionRun(User(ini), OracleEngine("soc"))
        os.chdir(d)
        CTABS, FABS = run.run()
        L = launch.hpbg_launch(run.BGPAC, 6, 6, 6)
        assert L["BATCH"] == 100 and L["GLOBAL"] == launch.Fix(run.BGPAC / 100, 64)
        assert np.isclose(L["WBG"], np.pi / launch.PLANCK / (L["GLOBAL"] * 100 / 216.0))
        FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
        FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 3, 500)
        T = np.zeros(cloud.CELLS, np.float32)
        for i in range(3):
            s = np.float32(2.0) * sky[i]
            if weighted:
                p = np.clip(s.astype(np.float64) / s.astype(np.float64).mean(), 1e-3, 1e4)
                p /= p.sum()
                bg = (L["WBG"] / float(FFREQ[i]) * s * ((1.0 / 49152.0) / p)).astype(np.float32)
                P = np.cumsum(p)
                P[-1] = 1.00001
                P = P.astype(np.float32)
            else:
                bg, P = (np.float32(L["WBG"] / float(FFREQ[i])) * s).astype(np.float32), None
            job = Job(cloud, FCSC[0, i], ABS=AFABS[0][i], SCA=AFSCA[0][i], BATCH=100, SEED=launch.launch_seed(math.pi / 4, i),
                      TW=launch.trapezoid_weight(FFREQ, i), GLOBAL=L["GLOBAL"], HPBG=bg, HPBGP=P, WITH_INT=1)
            Oracle("soc").sim(job, 2, TABS=T, INT=np.zeros(cloud.CELLS, np.float32))
        assert np.array_equal(CTABS, T) and T.sum() > 0


def test_mirror_key_reaches_the_engine(tmp_path):
    """`mirror` -> bit mask (ASOC.py:319-321) -> soc_set_mirror; reflected packets deposit (a little, the test model is opaque) more energy."""
    from oracle_engine import OracleEngine
    assert launch.mirror_mask("xY") == 1 + 8 and launch.mirror_mask("xXyYzZ") == 63 and launch.mirror_mask("") == 0
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(6, seed=4)
    os.chdir(d)
    plain = AbsorptionRun(User(_write_model(d, cloud)), OracleEngine("soc")).run()[0]
    eng = OracleEngine("soc")
    mir = AbsorptionRun(User(_write_model(d, cloud, extra="mirror xZ\n")), eng).run()[0]
    assert eng.mirror == 1 + 32
    assert not np.array_equal(mir, plain) and mir.sum(dtype=np.float64) > plain.sum(dtype=np.float64)


def test_nested_run_roi_save_then_load(tmp_path):
    """roi/roisave in an outer run, roiload/roipac in the nested one (ASOC.py:909-944, 1094-1105, 1301, 1415-1475):
    file format, per-frequency records (x GL^2), the SOURCE 3 launch and its scaling -- re-derived here."""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle
    d = str(tmp_path)
    os.chdir(d)
    outer = synth.cartesian_cloud(8, seed=3)
    ROI = [2, 5, 2, 5, 3, 4]
    GL = 5.0e-7                                                    # the model of _write_model is opaque at 0.5 pc per cell
    ini = _write_model(d, outer, extra="gridlength %g\nroi %d %d %d %d %d %d\nroisave %s/roi.save 2\nroinside 2\n" % (GL, *ROI, d))
    U = User(ini)
    assert U.WITH_ROI_SAVE == 1 and U.ROI_STEP == 2 and U.ROI_NSIDE == 2 and list(U.ROI) == ROI
    run = AbsorptionRun(U, OracleEngine("soc"))
    run.run()
    hdr = np.fromfile(os.path.join(d, "roi.save"), np.int32, 5)
    assert list(hdr) == [8, 8, 4, 2, 3]                                    # (nx, ny, nz) elements, nside, nfreq
    nelem = 8 * 8 + 8 * 4 + 4 * 8
    rec = np.fromfile(os.path.join(d, "roi.save"), np.float32, offset=20).reshape(3, nelem * 48)
    assert (rec > 0).sum() > 100
    # independent evaluation of frequency 1
    orc = Oracle("soc")
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], GL)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 3, 500)
    IBG = np.fromfile(os.path.join(d, "bg.bin"), np.float32)
    L = launch.bg_launch(run.BGPAC, outer.AREA)
    job = Job(outer, FCSC[0, 1], ABS=AFABS[0][1], SCA=AFSCA[0][1], SOURCE=1, BATCH=L["BATCH"],
              SEED=launch.launch_seed(math.pi / 4, 1), BG=np.float32(float(IBG[1]) * L["WBG"] / float(FFREQ[1])),
              TW=launch.trapezoid_weight(FFREQ, 1), GLOBAL=L["GLOBAL"], WITH_INT=1, ROI=ROI, ROI_STEP=2, ROI_NSIDE=2)
    orc.sim(job, 0)
    assert np.array_equal(rec[1], job.ROI_SAVE * np.float32(GL * GL))

    # nested run: the ROI region at twice the resolution, fed by the record
    inner = synth.cartesian_cloud(8, seed=4, NZ=4)
    d2 = os.path.join(d, "inner")
    os.makedirs(d2)
    ini2 = _write_model(d2, inner, extra="gridlength %g\nroiload %s/roi.save 1.5\nroipac 500000\nroinside 2\nbgpackets 0\n" % (GL, d))
    U2 = User(ini2)
    assert U2.WITH_ROI_LOAD == 1 and U2.ROI_LOAD_SCALE == 1.5 and U2.ROIPAC == 500000
    run2 = AbsorptionRun(U2, OracleEngine("soc"))
    C2, _ = run2.run()
    L3 = launch.roi_launch(500000, nelem, 2)
    assert L3 == dict(GLOBAL=launch.Fix(100 * nelem, 32), BATCH=max(1, int(500000 / (100.0 * 48 * nelem))) * 48, PACKETS=nelem)
    T = np.zeros(inner.CELLS, np.float32)
    for i in range(3):
        job = Job(inner, FCSC[0, i], ABS=AFABS[0][i], SCA=AFSCA[0][i], SOURCE=3, BATCH=L3["BATCH"], PACKETS=nelem,
                  SEED=launch.launch_seed(math.pi / 4, i), BG=0.0, TW=launch.trapezoid_weight(FFREQ, i), GLOBAL=L3["GLOBAL"],
                  WITH_INT=1, ROI_LOAD=np.asarray(rec[i] * 1.5 / (GL * GL), np.float32), ROI_DIM=[8, 8, 4], ROI_NSIDE=2)
        orc.sim(job, 0, TABS=T)
    assert T.sum() > 0 and np.array_equal(C2, T)


@pytest.mark.parametrize("single", [False, True])
def test_abundance_run(single, tmp_path):
    """optical <dust> <abundance file> (+ singleabu): per-cell opacities = sum of abundance x cross section
    (ASOC.py:1146-1160), built by the engine from abundances handed over once"""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle
    d = str(tmp_path)
    os.chdir(d)
    cloud = synth.cartesian_cloud(6, seed=2)
    rr = np.random.default_rng(4)
    abu = rr.uniform(0.2, 1.0, cloud.CELLS).astype(np.float32)
    abu.tofile(os.path.join(d, "a.abu"))
    GL = 5.0e-7
    ini = _write_model(d, cloud, nfreq=2, extra="gridlength %g\n" % GL)
    with open(os.path.join(d, "m2.dust"), "w") as fp:
        fp.write("eqdust\n 1.0e-7\n 0.7e-4\n2\n 4.00000e+14  0.6  2.0e-2  1.2e-1\n 4.67700e+14  0.6  2.5e-2  1.0e-1\n")
    txt = open(ini).read().replace("optical %s/m.dust\n" % d, "optical %s/m.dust %s/a.abu\noptical %s/m2.dust\n" % (d, d, d))
    txt = txt.replace("dsc %s/m.dsc 500\n" % d, "dsc %s/m.dsc 500\n" % d) + ("singleabu\n" if single else "")
    open(ini, "w").write(txt)
    U = User(ini)
    assert U.file_abundance == [os.path.join(d, "a.abu"), "#"] and U.SINGLE_ABU == int(single)
    run = AbsorptionRun(U, OracleEngine("soc"))
    CT, _ = run.run()
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust"), os.path.join(d, "m2.dust")], GL)
    second = (1.0 - abu) if single else np.ones(cloud.CELLS, np.float32)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 2, 500)
    IBG = np.fromfile(os.path.join(d, "bg.bin"), np.float32)
    L = launch.bg_launch(run.BGPAC, cloud.AREA)
    T = np.zeros(cloud.CELLS, np.float32)
    for i in range(2):
        OPT = np.zeros((cloud.CELLS, 2), np.float32)
        OPT[:, 0] = abu * AFABS[0][i] + second * AFABS[1][i]
        OPT[:, 1] = abu * AFSCA[0][i] + second * AFSCA[1][i]
        job = Job(cloud, FCSC[0, i], SOURCE=1, BATCH=L["BATCH"], SEED=launch.launch_seed(math.pi / 4, i),
                  BG=np.float32(float(IBG[i]) * L["WBG"] / float(FFREQ[i])), TW=launch.trapezoid_weight(FFREQ, i),
                  GLOBAL=L["GLOBAL"], WITH_INT=1, OPT=OPT)
        Oracle("soc").sim(job, 0, TABS=T)
    assert T.sum() > 0 and np.allclose(CT, T, rtol=2e-6)
    if not single:
        # `optishalf`: the same run with OPT rounded to fp16 (-D OPT_IS_HALF, ASOC.py:1158-1159)
        open(ini, "a").write("optishalf\n")
        U = User(ini)
        assert U.OPT_IS_HALF == 1
        eng = OracleEngine("soc")
        CH, _ = AbsorptionRun(U, eng).run()
        assert np.array_equal(eng.OPT, np.asarray(np.asarray(OPT, np.float16), np.float32))
        assert not np.array_equal(CH, CT) and abs(CH.sum(dtype=np.float64) / CT.sum(dtype=np.float64) - 1) < 1e-2


def test_stepweight_key(tmp_path):
    """`stepweight a b c` reaches the kernels as -D SW_A=int(a) -D SW_B=b -D STEP_WEIGHT=int(c) (ASOC.py:348,357 -- not in
    the order the key's comment suggests, ASOC_aux.py:476-481): the drop-in hands the engine those values"""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle
    d = str(tmp_path)
    os.chdir(d)
    cloud = synth.cartesian_cloud(5, seed=3)
    U = User(_write_model(d, cloud, nfreq=2, extra="stepweight 2 0.4 2\n"))
    assert U.STEP_WEIGHT == [2, 0.4, 2.0]
    eng = OracleEngine("soc")
    run = AbsorptionRun(U, eng)
    CT, _ = run.run()
    assert eng.step_weight == (2, 2.0, 0.4)
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 2, 500)
    IBG = np.fromfile(os.path.join(d, "bg.bin"), np.float32)
    L = launch.bg_launch(run.BGPAC, cloud.AREA)
    T = np.zeros(cloud.CELLS, np.float32)
    T0 = np.zeros(cloud.CELLS, np.float32)
    for i in range(2):
        kw = dict(ABS=AFABS[0][i], SCA=AFSCA[0][i], SOURCE=1, BATCH=L["BATCH"], SEED=launch.launch_seed(math.pi / 4, i),
                  BG=np.float32(float(IBG[i]) * L["WBG"] / float(FFREQ[i])), TW=launch.trapezoid_weight(FFREQ, i),
                  GLOBAL=L["GLOBAL"], WITH_INT=1)
        Oracle("soc").sim(Job(cloud, FCSC[0, i], STEP_WEIGHT=(2, 2.0, 0.4), **kw), 0, TABS=T)
        Oracle("soc").sim(Job(cloud, FCSC[0, i], **kw), 0, TABS=T0)
    assert T.sum() > 0 and np.array_equal(CT, T)
    assert not np.array_equal(T, T0) and abs(T.sum(dtype=np.float64) / T0.sum(dtype=np.float64) - 1) < 0.05    # same physics, other weights
    # a key absent from the ini file switches the weighting off
    eng2 = OracleEngine("soc")
    AbsorptionRun(User(_write_model(d, cloud, nfreq=2)), eng2).setup_engine()
    assert eng2.step_weight is None


def test_several_scattering_functions(tmp_path):
    """one dsc file per dust species = -D WITH_MSF (ASOC.py:132-138): abundances are required, the engine receives the
    tables of all species for every frequency and draws the scatterer per event"""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle
    d = str(tmp_path)
    os.chdir(d)
    cloud = synth.octree_cloud(5, levels=2, frac=0.1, seed=4)
    rr = np.random.default_rng(4)
    abu = rr.uniform(0.2, 1.0, cloud.CELLS).astype(np.float32)
    abu.tofile(os.path.join(d, "a.abu"))
    GL = 5.0e-7
    ini = _write_model(d, cloud, nfreq=2, extra="gridlength %g\n" % GL)
    with open(os.path.join(d, "m2.dust"), "w") as fp:
        fp.write("eqdust\n 1.0e-7\n 0.7e-4\n2\n 4.00000e+14  0.6  2.0e-2  1.2e-1\n 4.67700e+14  0.6  2.5e-2  1.0e-1\n")
    dsc, csc = synth.hg_scattering_table(0.1, 500)
    files.write_scattering_functions(os.path.join(d, "m2.dsc"), np.tile(dsc, (2, 1)), np.tile(csc, (2, 1)))
    txt = open(ini).read().replace("optical %s/m.dust\n" % d, "optical %s/m.dust %s/a.abu\noptical %s/m2.dust\n" % (d, d, d))
    open(ini, "w").write(txt.replace("dsc %s/m.dsc 500\n" % d, "dsc %s/m.dsc 500\ndsc %s/m2.dsc 500\n" % (d, d)))
    run = AbsorptionRun(User(ini), OracleEngine("soc"))
    assert run.WITH_MSF and run.NDUST == 2
    CT, _ = run.run()
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust"), os.path.join(d, "m2.dust")], GL)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc"), os.path.join(d, "m2.dsc")], 2, 500)
    IBG = np.fromfile(os.path.join(d, "bg.bin"), np.float32)
    L = launch.bg_launch(run.BGPAC, cloud.AREA)
    ABU = np.stack([abu, np.ones(cloud.CELLS, np.float32)], 1)
    T = np.zeros(cloud.CELLS, np.float32)
    for i in range(2):
        OPT = np.zeros((cloud.CELLS, 2), np.float32)
        for k in range(2):
            OPT[:, 0] += ABU[:, k] * AFABS[k][i]
            OPT[:, 1] += ABU[:, k] * AFSCA[k][i]
        msf = ([AFABS[0][i], AFABS[1][i]], [AFSCA[0][i], AFSCA[1][i]], FCSC[:, i, :], ABU)
        job = Job(cloud, None, SOURCE=1, BATCH=L["BATCH"], SEED=launch.launch_seed(math.pi / 4, i),
                  BG=np.float32(float(IBG[i]) * L["WBG"] / float(FFREQ[i])), TW=launch.trapezoid_weight(FFREQ, i),
                  GLOBAL=L["GLOBAL"], WITH_INT=1, OPT=OPT, MSF=msf)
        Oracle("soc").sim(job, 0, TABS=T)
    assert T.sum() > 0 and np.array_equal(CT, T)
    # without abundances the reference stops (ASOC.py:168-170)
    txt = open(ini).read().replace(" %s/a.abu" % d, "")
    open(ini, "w").write(txt)
    with pytest.raises(ValueError, match="variable abundances"):
        AbsorptionRun(User(ini), OracleEngine("soc"))


@pytest.mark.parametrize("mode", [1, 2])
def test_saveint_writes_the_intensity_file(mode, tmp_path):
    """`saveint 1|2 file`: mean intensity per cell and frequency from the INT tally, with saveint 2 also the net flux
    direction (INTX, INTY, INTZ over INT) -- ASOC.py:990-1000, :1499-1515, :2733-2757"""
    from oracle_engine import OracleEngine
    from oracle.pyoracle import Job, Oracle
    d = str(tmp_path)
    os.chdir(d)
    cloud = synth.octree_cloud(5, levels=2, frac=0.1, seed=4)
    U = User(_write_model(d, cloud, nfreq=2, with_ps=True, extra="saveint %d %s/isrf.dat\n" % (mode, d)))
    assert U.SAVE_INTENSITY == mode
    run = AbsorptionRun(U, OracleEngine("soc"))
    run.run()
    got = files.read_intensity(os.path.join(d, "isrf.dat"))
    assert got.shape == ((cloud.CELLS, 2) if mode == 1 else (cloud.CELLS, 2, 4))
    FFREQ, _, AFABS, AFSCA = files.read_dust([os.path.join(d, "m.dust")], 0.5)
    FDSC, FCSC = files.read_scattering_functions([os.path.join(d, "m.dsc")], 2, 500)
    IBG = np.fromfile(os.path.join(d, "bg.bin"), np.float32)
    LPS = np.fromfile(os.path.join(d, "ps.bin"), np.float32)
    Lb = launch.bg_launch(run.BGPAC, cloud.AREA)
    Lp = launch.ps_launch(run.PSPAC, 1, 0.5, run.GLOBAL_0)
    want = np.zeros((cloud.CELLS, 2, 4), np.float32)
    orc = Oracle("soc")
    for i in range(2):
        FREQ = float(FFREQ[i])
        kw = dict(ABS=AFABS[0][i], SCA=AFSCA[0][i], SEED=launch.launch_seed(math.pi / 4, i), TW=launch.trapezoid_weight(FFREQ, i),
                  WITH_INT=2)
        jobs = [Job(cloud, FCSC[0, i], SOURCE=0, BATCH=Lp["BATCH"], GLOBAL=Lp["GLOBAL"], PSPOS=np.array([[3.3, 3.2, 3.1]], np.float32),
                    PS=np.asarray([LPS[i]], np.float32) * np.float32(Lp["WPS"]) / np.float32(FREQ), **kw),
                Job(cloud, FCSC[0, i], SOURCE=1, BATCH=Lb["BATCH"], GLOBAL=Lb["GLOBAL"],
                    BG=np.float32(float(IBG[i]) * Lb["WBG"] / FREQ), **kw)]
        for job in jobs:
            _, I, _ = orc.sim(job, 0)
            for k, v in enumerate([I, job.INTV[0], job.INTV[1], job.INTV[2]]):
                for level in range(cloud.LEVELS):
                    a, b = int(cloud.OFF[level]), int(cloud.OFF[level] + cloud.LCELLS[level])
                    coeff = np.float32(launch.PLANCK * FREQ) / np.float32(AFABS[0][i]) * np.float32(8.0 ** level)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        want[a:b, i, k] += coeff * v[a:b] / cloud.DENS[a:b]
    leaf = cloud.DENS > 0
    if mode == 1:
        assert np.array_equal(got[leaf], want[leaf][:, :, 0]) and (got[leaf] > 0).mean() > 0.5
    else:
        for k in (1, 2, 3):
            want[:, :, k] /= (want[:, :, 0] + 1.0e-33)
        assert np.array_equal(got[leaf], want[leaf])
        assert np.abs(got[leaf][:, :, 1:]).max() <= 1.0 + 1e-5 and np.abs(got[leaf][:, :, 1:]).mean() > 0.01
