"""HIP scattered-light kernels (soc_amd/csrc/soc_sca.hip through the C ABI) against the CPU
oracle in soc math mode: identical trajectories (same contribution / packet / scattering
counts) and images equal to fp32 summation-order accuracy (1e-5 per pixel), on every case
that is pinned bit-exactly against the reference (tests/test_sca_oracle.py)."""
import os

import numpy as np
import pytest

import cases
from oracle.pyoracle import oracle_sim_sca

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sca.npz"))


def run_sca(eng, job, view, kind, gid_first=0, gid_count=None, zero=True, rebind=None):
    eng.set_cloud(job.cloud)
    eng.set_features(with_int=0, ps_method=job.PS_METHOD, use_emweight=job.USE_EMWEIGHT)
    eng.set_optical(job.ABS, job.SCA)
    if job.MSF is not None:                      # -D WITH_MSF: one DSC/CSC row per species, OPT built from the abundances
        eng.set_abundances(job.MSF[3])
        eng.set_optical_abu(job.MSF[0], job.MSF[1])
        eng.set_scatter_tables(job.DSC, job.MSF[2])
    else:
        eng.set_scatter_table(job.DSC, job.CSC)
        eng.set_opt(job.OPT)
    eng.set_mirror(job.MIRROR)
    if view.nside:
        eng.sca_set_healpix(view.nside, view.ODIR[0, :3], view.FFS)
    else:
        eng.sca_set_view(view.ODIR, view.RA, view.DE, view.NPIX, view.MAP_DX, view.CENTRE, view.FFS)
    if rebind:
        eng.sca_bind_out(rebind)                 # caller-owned image (soc_sca_set_view allocates the library's)
    if zero:
        eng.sca_zero()
    eng.stats(reset=True)
    gid_count = job.GLOBAL - gid_first if gid_count is None else gid_count
    xps = (job.XPS_NSIDE, job.XPS_SIDE, job.XPS_AREA)
    if kind == 3:
        eng.set_hpbg(job.HPBG, job.HPBGP)
        eng.sca_sim_hp(job.PACKETS, job.BATCH, job.SEED, job.GLOBAL, gid_first=gid_first, gid_count=gid_count)
    elif kind == 2:
        eng.sca_sim_ps(job.PACKETS, job.BATCH, job.SEED, job.BG, job.PSPOS[:, :3], job.PS, XPS=xps, GLOBAL=job.GLOBAL,
                       gid_first=gid_first, gid_count=gid_count)
    elif kind == 0:
        eng.sca_sim_pb(job.SOURCE, job.PACKETS, job.BATCH, job.SEED, job.BG, job.PSPOS[:, :3], job.PS, XPS=xps,
                       GLOBAL=job.GLOBAL, gid_first=gid_first, gid_count=gid_count)
    else:
        eng.set_emission(job.EMIT, job.EMWEI)
        eng.sca_sim_cl(job.SOURCE, job.PACKETS, job.BATCH, job.SEED, job.GLOBAL, gid_first=gid_first, gid_count=gid_count)
    eng.sync()
    if job.MSF is not None:
        eng.set_scatter_table(None, job.MSF[2][0])
        eng.set_opt(None)
        eng.set_abundances(None)
    return eng.sca_read_out(), eng.stats()


def assert_image_close(got, want, rtol=1e-5):
    got = np.asarray(got, np.float64).ravel()
    want = np.asarray(want, np.float64).ravel()
    tol = rtol * np.abs(want) + 1e-6 * rtol * np.abs(want).max()
    bad = np.abs(got - want) > tol
    assert not bad.any(), "%d of %d pixels differ, worst rel %.3e" % (
        bad.sum(), bad.size, (np.abs(got - want) / np.maximum(np.abs(want), 1e-300))[bad].max())


@pytest.mark.parametrize("name", sorted(cases.SCA_CASES))
def test_sca_hip_matches_oracle(name, engine, oracle_soc):
    ref, kind, mk, vkw = cases.SCA_CASES[name]
    job, view = mk(), cases.sca_view(**vkw)
    want, n = oracle_sim_sca(oracle_soc, job, view, kind)
    got, st = run_sca(engine, job, view, kind)
    assert st["tally_events"] == n                       # identical trajectories
    assert_image_close(got, want)
    # and, at Monte Carlo accuracy, the reference's own image
    assert abs(got.sum(dtype=np.float64) / GOLD[name].sum(dtype=np.float64) - 1.0) < 1e-2


def test_sca_split_launch_adds_up(engine, oracle_soc):
    ref, kind, mk, vkw = cases.SCA_CASES["sca_bg_oct8"]
    job, view = mk(), cases.sca_view(**vkw)
    whole, st = run_sca(engine, job, view, kind)
    a, sa = run_sca(engine, job, view, kind, 0, 1024)
    b, sb = run_sca(engine, job, view, kind, 1024, job.GLOBAL - 1024, zero=False)
    assert sa["tally_events"] + sb["tally_events"] == st["tally_events"]
    assert_image_close(b, whole, rtol=2e-5)               # b accumulated on top of a


def test_sca_full_size_properties(engine):
    """A uniform 64^3 Cartesian cloud, 8 work items per surface element: the image of an optically thin uniform cloud
    seen along the three axes has equal totals by symmetry, and doubling BG doubles every pixel (power of two).
    (Config 4 at its stated size -- 256^3-root octree, three launch kinds -- is tests/test_gpu_fullsize.py.)"""
    from oracle.pyoracle import Job
    from soc_amd import synth
    cloud = synth.cartesian_cloud(64, uniform=1.0)
    dsc, csc = synth.hg_scattering_table(0.0)
    view = cases.sca_view(NPIX=(64, 64), MAP_DX=1.0, angles=((90.0, 0.0), (90.0, 90.0), (0.0, 0.0)))
    view.CENTRE = (np.float32(32), np.float32(32), np.float32(32))
    job = Job(cloud, csc, ABS=1e-4, SCA=2e-3, SOURCE=1, BATCH=4, SEED=0.31, DSC=dsc, BG=1.0)
    a, st = run_sca(engine, job, view, 0)
    job2 = Job(cloud, csc, ABS=1e-4, SCA=2e-3, SOURCE=1, BATCH=4, SEED=0.31, DSC=dsc, BG=2.0)
    b, st2 = run_sca(engine, job2, view, 0)
    assert st == st2 and st["packets"] == 8 * 6 * 64 * 64 * 4
    tot = a.reshape(3, -1).sum(axis=1, dtype=np.float64)
    assert np.abs(tot / tot.mean() - 1).max() < 0.01
    np.testing.assert_allclose(b, 2.0 * a, rtol=1e-5, atol=1e-6 * a.max())


def test_scattering_run_end_to_end(engine, tmp_path):
    """python -m soc_amd.asocs on the GPU equals the same host loop on the oracle engine."""
    import os
    from oracle_engine import OracleEngine
    from soc_amd import synth, files
    from soc_amd.asocs import ScatteringRun
    from soc_amd.ini import User
    from test_host_sca import _ini
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    ini = _ini(d, cloud, with_ps=True)
    os.chdir(d)
    want = ScatteringRun(User(ini), OracleEngine("soc"), verbose=0).run()
    got = ScatteringRun(User(ini), engine, verbose=0).run()
    for i in range(want.shape[0]):
        assert_image_close(got[i], want[i], rtol=2e-5)
    _, data = files.read_outcoming("outcoming.socs", 2)
    assert np.array_equal(data, got)
