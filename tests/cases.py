"""Named simulation cases shared by the golden-vector generator and the parity tests.

Every case is a small, seeded model that the CPU oracle finishes in well under a second.
``ref`` names the x86 build of the reference kernels (oracle/build.py: ref_models) whose
baked-in geometry matches the case; ``kind`` 0 = SimRAM_PB, 1 = SimRAM_CL.
"""
import numpy as np

from oracle.pyoracle import Job
from soc_amd import synth

_DSC, _CSC = synth.hg_scattering_table(0.6)
_DSC0, _CSC0 = synth.hg_scattering_table(0.0)


def _c8():
    return synth.cartesian_cloud(8, seed=3)


def _oct8():
    return synth.octree_cloud(8, levels=3, frac=0.15, seed=7)


def _opt(cells, seed=5):
    rr = np.random.default_rng(seed)
    opt = np.zeros((cells, 2), np.float32)
    opt[:, 0] = 1e-4 * rr.uniform(0.5, 2, cells)
    opt[:, 1] = 3e-4 * rr.uniform(0.5, 2, cells)
    return opt


_PS_IN = np.array([[4.3, 4.2, 4.1], [2.5, 6.5, 3.3]], np.float32)
_PS_EXT = np.array([[4.0, 4.0, 20.0], [4.3, 4.2, 4.1]], np.float32)
_XPS2 = (np.array([1, 0], np.int32), np.array([4, 0, 0, 0, 0, 0], np.int32),
         np.array([1.0, 1, 1, 1, 1, 1], np.float32))
_XPS5 = (np.array([1, 0], np.int32), np.array([4, 0, 0, 0, 0, 0], np.int32),
         np.array([0.9, 1, 1, 1, 1, 1], np.float32))


def _ps_in_oct8():
    """one source in an unrefined root cell of the oct8 cloud, one inside a refined one"""
    c = _oct8()
    root = np.asarray(c.DENS[:512])
    leafs, refined = np.flatnonzero(root > 0), np.flatnonzero(root <= 0)
    out = []
    for i, frac in ((int(leafs[len(leafs) // 2]), (0.3, 0.2, 0.1)), (int(refined[len(refined) // 2]), (0.31, 0.77, 0.52))):
        out.append([i % 8 + frac[0], (i // 8) % 8 + frac[1], i // 64 + frac[2]])
    return np.asarray(out, np.float32)


def _emit(cloud):
    return np.where(cloud.DENS > 0, cloud.DENS * 1e-3, 0).astype(np.float32)


def _emwei(cloud, seed=5):
    rr = np.random.default_rng(seed)
    return rr.uniform(0, 3, cloud.CELLS).astype(np.float32)


def hp_sky(seed=8, weighted=False):
    """Synthetic Healpix sky (NSIDE 64) in photons per package; with weighted=True the pair
    (BG*weight, cumulative probability) the host prepares at ASOC.py:1198-1211."""
    sky = np.random.default_rng(seed).lognormal(0, 1, 49152).astype(np.float32)
    if not weighted:
        return sky, None
    tmp = np.asarray(sky, np.float64)
    tmp = tmp / tmp.mean()
    tmp = np.clip(tmp, 1.0e-3, 1.0e4)
    tmp /= tmp.sum()
    W = (1.0 / 49152.0) / tmp
    P = np.cumsum(tmp)
    P[-1] = 1.00001
    return np.asarray(sky * W, np.float32), np.asarray(P, np.float32)


def _hpjob(cloud, ref_weighted, **kw):
    bg, P = hp_sky(weighted=ref_weighted)
    return Job(cloud, _CSC, ABS=1e-4, SCA=3e-4, HPBG=bg, HPBGP=P, **kw)


def _hpjob_msf():
    return Job(_oct8(), None, HPBG=hp_sky()[0], BATCH=6, SEED=0.11, GLOBAL=4096, **msf_inputs(_oct8(), dsc=True))


def _emindex(cloud, n=300, seed=3):
    """USE_EMWEIGHT==2 inputs: a list of emitting leaf cells (terminated by -1) and packet weights"""
    rr = np.random.default_rng(seed)
    pick = rr.choice(np.flatnonzero(cloud.DENS > 0), n, replace=False)
    EMINDEX = -np.ones(cloud.CELLS, np.int32)
    EMINDEX[:n] = pick
    EMWEI = (1.0 / (100 * rr.integers(1, 4, cloud.CELLS) + 1e-10)).astype(np.float32)
    return EMINDEX, EMWEI


def roi_load(dim, nside, seed=9):
    """SOURCE == 3 input: photons per surface element of the (nx, ny, nz) discretisation and Healpix pixel; a third
    of the entries are empty (those packets are skipped without a draw)"""
    rr = np.random.default_rng(seed)
    nelem = dim[0] * dim[1] + dim[1] * dim[2] + dim[2] * dim[0]
    a = rr.uniform(0.5, 2.0, (nelem, 12 * nside * nside)).astype(np.float32)
    a[rr.uniform(size=a.shape) < 0.33] = 0.0
    return a


def _roil(cloud, dim, nside, nbatch, **kw):
    a = roi_load(dim, nside)
    return Job(cloud, _CSC, ABS=1e-4, SCA=3e-4, SOURCE=3, PACKETS=a.shape[0], GLOBAL=100 * a.shape[0],
               BATCH=nbatch * 12 * nside * nside, ROI_LOAD=a, ROI_DIM=dim, ROI_NSIDE=nside, **kw)


def msf_inputs(cloud, ndust=3, seed=6, dsc=False):
    """-D WITH_MSF inputs: per-dust cross sections, scattering functions (different asymmetries), abundances per cell,
    and the OPT array the host sums from them (ASOC.py:1146-1165)"""
    rr = np.random.default_rng(seed)
    ABS = (1e-4 * rr.uniform(0.5, 2, ndust)).astype(np.float32)
    SCA = (3e-4 * rr.uniform(0.5, 2, ndust)).astype(np.float32)
    tabs = [synth.hg_scattering_table(g) for g in np.linspace(0.1, 0.7, ndust)]
    CSC = np.stack([t[1] for t in tabs]).astype(np.float32)
    ABU = rr.uniform(0.2, 1.5, (cloud.CELLS, ndust)).astype(np.float32)
    OPT = np.zeros((cloud.CELLS, 2), np.float32)
    for i in range(ndust):
        OPT[:, 0] += ABU[:, i] * ABS[i]
        OPT[:, 1] += ABU[:, i] * SCA[i]
    if dsc:
        return dict(OPT=OPT, MSF=(ABS, SCA, CSC, ABU), DSC=np.stack([t[0] for t in tabs]).astype(np.float32))
    return dict(OPT=OPT, MSF=(ABS, SCA, CSC, ABU))


CASES = {
    # name: (ref build, kind, job factory); kind 0 = SimRAM_PB, 1 = SimRAM_CL, 2 = SimRAM_HP
    "bg_c8": ("c8", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=50, SEED=0.6004384)),
    "bg_c8_thin": ("c8", 0, lambda: Job(_c8(), _CSC0, ABS=1e-7, SCA=2e-7, SOURCE=1, BATCH=20, SEED=0.25, BG=3.0)),
    "bg_c8_int": ("c8int", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=20, SEED=0.123,
                                            WITH_INT=1, TW=2.5)),
    "bg_c8_abu": ("c8abu", 0, lambda: Job(_c8(), _CSC, SOURCE=1, BATCH=20, SEED=0.3, OPT=_opt(512))),
    "bg_r654": ("r654", 0, lambda: Job(synth.cartesian_cloud(6, seed=4, NY=5, NZ=4), _CSC, ABS=1e-4, SCA=3e-4,
                                         SOURCE=1, BATCH=30, SEED=0.77)),
    # region of interest (nested runs): save what enters ROI, load such a record as SOURCE 3
    "bg_c8_roisave": ("c8roi", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=30, SEED=0.6004384,
                                                ROI=[2, 5, 2, 4, 3, 6], ROI_STEP=2, ROI_NSIDE=2)),
    "cl_oct8_roisave": ("oct8roi", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                                    EMIT=_emit(_oct8()), ROI=[1, 3, 2, 5, 0, 4], ROI_STEP=1, ROI_NSIDE=4)),
    "roi_c8_load": ("c8roil", 0, lambda: _roil(_c8(), (4, 4, 4), 2, 1, SEED=0.33, TW=1.2)),
    "roi_oct8_load_save": ("oct8roils", 0, lambda: _roil(_oct8(), (2, 2, 2), 2, 2, SEED=0.71, WITH_INT=1,
                                                          ROI=[3, 4, 3, 5, 2, 5], ROI_STEP=2)),
    "bg_oct8": ("oct8", 0, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=30, SEED=0.41)),
    "bg_oct4": ("oct4", 0, lambda: Job(synth.kat_octree(), _CSC, ABS=2e-3, SCA=4e-3, SOURCE=1, BATCH=40, SEED=0.51)),
    "ps_in_c8": ("c8ps0", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=40, SEED=0.2, GLOBAL=256,
                                           PSPOS=_PS_IN, PS=[1.0, 2.0])),
    # point sources inside a hierarchy (config 3's source sits in one): the second source lies in a refined root cell
    "ps_in_oct8": ("oct8ps0", 0, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=40, SEED=0.21, GLOBAL=256,
                                              PSPOS=_ps_in_oct8(), PS=[1.0, 2.0], WITH_INT=1, TW=1.5)),
    "ps_ext0_c8": ("c8ps0", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=40, SEED=0.2, GLOBAL=256,
                                             PSPOS=_PS_EXT, PS=[1.0, 2.0], PS_METHOD=0, XPS=_XPS2)),
    "ps_ext1_c8": ("c8ps1", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=40, SEED=0.2, GLOBAL=256,
                                             PSPOS=_PS_EXT, PS=[1.0, 2.0], PS_METHOD=1, XPS=_XPS2)),
    "ps_ext2_c8": ("c8ps2", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=40, SEED=0.2, GLOBAL=256,
                                             PSPOS=_PS_EXT, PS=[1.0, 2.0], PS_METHOD=2, XPS=_XPS2)),
    "ps_ext4_c8": ("c8ps4", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=40, SEED=0.2, GLOBAL=256,
                                             PSPOS=_PS_EXT, PS=[1.0, 2.0], PS_METHOD=4, XPS=_XPS2)),
    "ps_ext5_c8": ("c8ps5", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=40, SEED=0.2, GLOBAL=256,
                                             PSPOS=_PS_EXT, PS=[1.0, 2.0], PS_METHOD=5, XPS=_XPS5)),
    "cl_oct8": ("oct8", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                         EMIT=_emit(_oct8()))),
    "cl_oct8_emw": ("oct8emw", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9,
                                                GLOBAL=128, EMIT=_emit(_oct8()), EMWEI=_emwei(_oct8()),
                                                USE_EMWEIGHT=1)),
    "hp_c8": ("c8", 2, lambda: _hpjob(_c8(), False, BATCH=10, SEED=0.37, TW=1.5, GLOBAL=3072 + 64)),
    "hp_c8_weighted": ("c8hpw", 2, lambda: _hpjob(_c8(), True, BATCH=10, SEED=0.37, TW=1.5, GLOBAL=3072)),
    "hp_oct8_weighted_int": ("oct8hpw", 2, lambda: _hpjob(_oct8(), True, BATCH=6, SEED=0.11, TW=1.5, GLOBAL=3072, WITH_INT=1)),
    "hp_oct8": ("oct8", 2, lambda: _hpjob(_oct8(), False, BATCH=6, SEED=0.11, GLOBAL=3072)),
    # reflecting faces (`mirror` key): mask bits x,X,y,Y,z,Z = 1,2,4,8,16,32
    "bg_c8_mirror": ("c8mir", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=10, SEED=0.6, MIRROR=25)),
    "bg_oct8_mirror": ("oct8mir", 0, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=6, SEED=0.61, MIRROR=6)),
    "cl_oct8_mirror": ("oct8mir", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                                  EMIT=_emit(_oct8()), MIRROR=6)),
    "hp_c8_mirror": ("c8mir", 2, lambda: _hpjob(_c8(), False, BATCH=6, SEED=0.3, GLOBAL=3072, MIRROR=25)),
    "cl_oct8_emw2": ("oct8emw2", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=64,
                                                  EMIT=_emit(_oct8()), EMWEI=_emindex(_oct8())[1], USE_EMWEIGHT=2,
                                                  EMINDEX=_emindex(_oct8())[0])),
    "cl_oct8_ali": ("oct8ali", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                                EMIT=_emit(_oct8()), WITH_ALI=1)),
    "cl_c8": ("c8", 1, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=4, SEED=0.35, GLOBAL=64,
                                     EMIT=_emit(_c8()))),
    # -D SAVE_INTENSITY=2: INT and the vector sums INTX, INTY, INTZ (kernel_ASOC.c:604-612, :724-732)
    "bg_c8_int2": ("c8int2", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=20, SEED=0.123, WITH_INT=2, TW=2.5)),
    "cl_oct8_int2": ("oct8int2", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                                 EMIT=_emit(_oct8()), WITH_INT=2)),
    "hp_oct8_int2": ("oct8int2", 2, lambda: _hpjob(_oct8(), False, BATCH=4, SEED=0.11, GLOBAL=3072, WITH_INT=2)),
    # weighted free paths (-D STEP_WEIGHT=1|2 with SW_A, SW_B: kernel_ASOC.c:516-535) and per-dust scattering functions
    "bg_c8_sw1": ("c8sw1", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=30, SEED=0.44, STEP_WEIGHT=(1, 0.5, 0.0))),
    "bg_oct8_sw2": ("oct8sw2", 0, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=20, SEED=0.45,
                                               STEP_WEIGHT=(2, 0.7, 0.4))),
    "cl_oct8_sw2": ("oct8sw2", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                               EMIT=_emit(_oct8()), STEP_WEIGHT=(2, 0.7, 0.4))),
    "bg_oct8_msf": ("oct8msf", 0, lambda: Job(_oct8(), None, SOURCE=1, BATCH=20, SEED=0.46, **msf_inputs(_oct8()))),
    "cl_oct8_msf": ("oct8msf", 1, lambda: Job(_oct8(), None, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128, EMIT=_emit(_oct8()),
                                               **msf_inputs(_oct8()))),
    "hp_oct8_msf": ("oct8msf", 2, lambda: Job(_oct8(), None, HPBG=hp_sky()[0], BATCH=4, SEED=0.12, GLOBAL=3072, **msf_inputs(_oct8()))),
}

# ---- scattered-light images (kernel_ASOC_sca.c) --------------------------------------------
# name: (ref build of oracle/build.py: sca_ref_models, kernel kind, job factory, view kwargs)
# kind 0 = SimRAM_PB, 1 = SimRAM_CL, 2 = SimRAM_PS, 3 = SimRAM_HP
_PS_EXT2 = np.array([[14.3, 4.2, 4.1], [4.0, -3.0, 20.0]], np.float32)


def sca_view(NPIX=(12, 10), MAP_DX=1.1, FFS=1, angles=((30.0, 40.0), (90.0, 0.0), (0.0, 0.0)), healpix=None):
    """healpix = (nside, (x, y, z)): one Healpix map seen from that position instead of orthographic maps"""
    import math
    from oracle.pyoracle import ScaView
    from soc_amd import launch
    if healpix is not None:
        z = np.zeros((1, 4), np.float32)
        o = np.zeros((1, 4), np.float32)
        o[0, :3] = healpix[1]
        return ScaView(o, z, z, NPIX=(1, 1), MAP_DX=1.0, CENTRE=(0.0, 0.0, 0.0), FFS=FFS, nside=healpix[0])
    th = [math.radians(a[0]) for a in angles]
    ph = [math.radians(a[1]) for a in angles]
    _, OD, RA, DE = launch.set_observer_directions(th, ph)
    return ScaView(OD, RA, DE, NPIX=NPIX, MAP_DX=MAP_DX, CENTRE=(4.0, 4.0, 4.0), FFS=FFS)


def _xps(ps, meth):
    from soc_amd import files
    return files.analyse_external_point_sources(8, 8, 8, ps, len(ps), meth)


def _psjob(meth, ps=_PS_EXT2, lum=(1.0, 2.0), **kw):
    return Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=20, SEED=0.2, GLOBAL=256, PSPOS=ps, PS=list(lum), DSC=_DSC,
               PS_METHOD=meth, XPS=_xps(ps, meth), **kw)


SCA_CASES = {
    "sca_bg_c8": ("c8", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=10, SEED=0.6004384, DSC=_DSC), {}),
    "sca_bg_c8_noffs": ("c8noffs", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=10, SEED=0.25, DSC=_DSC),
                        dict(FFS=0)),
    "sca_bg_c8_abu": ("c8abu", 0, lambda: Job(_c8(), _CSC, SOURCE=1, BATCH=8, SEED=0.3, OPT=_opt(512), DSC=_DSC), {}),
    "sca_bg_c8_thick": ("c8", 0, lambda: Job(_c8(), _CSC0, ABS=2e-2, SCA=2e-1, SOURCE=1, BATCH=2, SEED=0.77, DSC=_DSC0),
                        dict(angles=((60.0, 200.0),))),
    "sca_bg_oct8": ("oct8", 0, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=8, SEED=0.41, DSC=_DSC), {}),
    "sca_ps_in_c8": ("c8ps", 2, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=20, SEED=0.2, GLOBAL=256,
                                            PSPOS=_PS_IN, PS=[1.0, 2.0], DSC=_DSC), {}),
    "sca_ps_ext0_c8": ("c8ps", 2, lambda: _psjob(0), {}),
    "sca_ps_ext1_c8": ("c8ps1", 2, lambda: _psjob(1), {}),
    "sca_ps_ext2_c8": ("c8ps2", 2, lambda: _psjob(2), {}),
    "sca_ps_ext4_c8": ("c8ps4", 2, lambda: _psjob(4, ps=np.array([[4.0, 4.0, 15.0]], np.float32), lum=(1.0,)), {}),
    "sca_ps_ext5_c8": ("c8ps5", 2, lambda: _psjob(5), {}),
    "sca_pbps_ext2_c8": ("c8ps2", 0, lambda: _psjob(2), {}),
    "sca_bg_c8_mirror": ("c8mir", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=6, SEED=0.6, DSC=_DSC, MIRROR=5), {}),
    "sca_cl_c8_mirror": ("c8mir", 1, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                                  EMIT=_emit(_c8()), DSC=_DSC, MIRROR=5), {}),
    "sca_ps_c8_mirror": ("c8mir", 2, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=20, SEED=0.2, GLOBAL=256,
                                                  PSPOS=[[4.3, 4.2, 4.1]], PS=[1.0], DSC=_DSC, MIRROR=5), {}),
    # Healpix maps seen from a position inside / outside the cloud (`perspective`)
    "sca_hpx_bg_c8_in": ("c8", 0, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=6, SEED=0.6, DSC=_DSC),
                         dict(healpix=(8, (4.2, 4.1, 3.9)))),
    "sca_hpx_bg_oct8_out": ("oct8", 0, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=1, BATCH=6, SEED=0.41, DSC=_DSC),
                            dict(healpix=(8, (20.0, 5.0, 5.5)))),
    "sca_hpx_cl_oct8_in": ("oct8", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                                   EMIT=_emit(_oct8()), DSC=_DSC), dict(healpix=(8, (4.2, 4.1, 3.9)))),
    "sca_hpx_ps_c8_in": ("c8ps", 2, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=0, BATCH=20, SEED=0.2, GLOBAL=256,
                                                 PSPOS=_PS_EXT, PS=[1.0, 2.0], DSC=_DSC), dict(healpix=(4, (4.2, 4.1, 3.9)))),
    # Healpix background (sca SimRAM_HP), kind 3
    "sca_hp_c8": ("c8", 3, lambda: _hpjob(_c8(), False, BATCH=8, SEED=0.37, GLOBAL=4096, DSC=_DSC), {}),
    "sca_hp_c8_weighted": ("c8hpw", 3, lambda: _hpjob(_c8(), True, BATCH=8, SEED=0.37, GLOBAL=4096, DSC=_DSC), {}),
    "sca_hp_oct8_hpx": ("oct8", 3, lambda: _hpjob(_oct8(), False, BATCH=8, SEED=0.11, GLOBAL=4096, DSC=_DSC),
                        dict(healpix=(8, (4.2, 4.1, 3.9)))),
    "sca_cl_c8": ("c8", 1, lambda: Job(_c8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                        EMIT=_emit(_c8()), DSC=_DSC), {}),
    "sca_cl_oct8": ("oct8", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                            EMIT=_emit(_oct8()), DSC=_DSC), {}),
    # -D WITH_MSF: the species is drawn per peel-off (its DSC row) and per scattering (its CSC row)
    "sca_bg_oct8_msf": ("oct8msf", 0, lambda: Job(_oct8(), None, SOURCE=1, BATCH=8, SEED=0.41, **msf_inputs(_oct8(), dsc=True)), {}),
    "sca_hpx_bg_oct8_msf": ("oct8msf", 0, lambda: Job(_oct8(), None, SOURCE=1, BATCH=6, SEED=0.43, **msf_inputs(_oct8(), dsc=True)),
                            dict(healpix=(8, (20.0, 5.0, 5.5)))),
    "sca_ps_c8_msf": ("c8msf", 2, lambda: Job(_c8(), None, SOURCE=0, BATCH=20, SEED=0.2, GLOBAL=256, PSPOS=_PS_IN, PS=[1.0, 2.0],
                                               **msf_inputs(_c8(), dsc=True)), {}),
    "sca_cl_oct8_msf": ("oct8msf", 1, lambda: Job(_oct8(), None, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128, EMIT=_emit(_oct8()),
                                                   **msf_inputs(_oct8(), dsc=True)), {}),
    "sca_hp_oct8_msf": ("oct8msf", 3, lambda: _hpjob_msf(), {}),
    "sca_cl_oct8_emw": ("oct8emw", 1, lambda: Job(_oct8(), _CSC, ABS=1e-4, SCA=3e-4, SOURCE=2, BATCH=3, SEED=0.9, GLOBAL=128,
                                                   EMIT=_emit(_oct8()), EMWEI=_emwei(_oct8()), USE_EMWEIGHT=1, DSC=_DSC), {}),
}

# fixed rays for step-by-step traces: (ref build, cloud factory, pos, dir)
RAYS = {
    "ray_c32": ("c32", lambda: synth.cartesian_cloud(32, uniform=1.0), [1e-4, 10.3, 20.7], [0.8, 0.36, 0.48]),
    "ray_oct4": ("oct4", synth.kat_octree, [1e-4, 1.3, 1.2], list(np.array([0.9, 0.2, 0.25]) / np.sqrt(0.9 ** 2 + 0.2 ** 2 + 0.25 ** 2))),
    "ray_oct8_a": ("oct8", _oct8, [1e-4, 3.37, 5.21], [0.70, 0.55, -0.4555217]),
    "ray_oct8_b": ("oct8", _oct8, [7.9999, 6.1, 2.2], [-0.6, -0.3, 0.7416198]),
    "ray_oct8_c": ("oct8", _oct8, [4.4, 1e-4, 4.6], [0.02, 0.9995999, -0.02]),
}
