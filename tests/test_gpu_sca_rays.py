"""Scattered-light images through the sweep of rays on brick-local hierarchies (soc_brick.hip: soc_lbrick_walk<., RAY> +
soc_sca_events) against the CPU oracle: identical trajectories (image contributions, packets, scatterings equal) and images
equal to fp32 summation order (1e-5 per pixel) -- the bar of tests/test_gpu_sca.py for the direct kernel, which is the second
witness here.  The hierarchy is one whose Index() the reference evaluates in double (104^3 roots, 4 levels)."""
import math

import numpy as np
import pytest

import cases
from oracle.pyoracle import Job, ScaView, oracle_sim_sca
from soc_amd import launch
from test_gpu_ltree import cloud104
from test_gpu_sca import assert_image_close, run_sca

pytestmark = pytest.mark.gpu


def view104(FFS=1, angles=((30.0, 40.0), (90.0, 0.0), (0.0, 0.0)), NPIX=(40, 36)):
    th = [math.radians(a[0]) for a in angles]
    ph = [math.radians(a[1]) for a in angles]
    _, OD, RA, DE = launch.set_observer_directions(th, ph)
    return ScaView(OD, RA, DE, NPIX=NPIX, MAP_DX=3.1, CENTRE=(52.0, 52.0, 52.0), FFS=FFS)


@pytest.fixture(params=[0, 64, 1], ids=["park4096", "park64", "nopark"])
def parking(request, engine):
    engine.set_tuning(park_below=request.param)
    yield
    engine.set_tuning(park_below=0)


def _rays(engine, job, view, kind, g0, g1):
    engine.set_exec(1, 4)
    try:
        img, st = run_sca(engine, job, view, kind, g0, g1 - g0)
        assert engine.last_passes() > 0 and engine.last_form() == 3
        assert engine.sca_ray_steps() > st["packets"]
    finally:
        engine.set_exec(-1, 4)
    return img, st


def _direct(engine, job, view, kind, g0, g1):
    engine.set_exec(0, 4)
    try:
        img, st = run_sca(engine, job, view, kind, g0, g1 - g0)
        assert engine.last_passes() == 0
    finally:
        engine.set_exec(-1, 4)
    return img, st


@pytest.mark.parametrize("tune", [dict(), dict(slow_every=3), dict(brick_cells=900, steps_per_visit=5), dict(chunk=64, threads=64)])
@pytest.mark.parametrize("ffs", [1, 0])
def test_background_rays(ffs, tune, engine, oracle_soc, tuned, parking):
    cl = cloud104()
    k = 2.0 / (104 * float(cl.DENS[:104 ** 3][cl.DENS[:104 ** 3] > 0].mean()))
    job = Job(cl, cases._CSC, ABS=0.3 * k, SCA=k, SOURCE=1, BATCH=3, SEED=0.377, DSC=cases._DSC, BG=1.0)
    view = view104(FFS=ffs)
    g0, g1 = 100000, 103000
    want, n = oracle_sim_sca(oracle_soc, job, view, 0, gid0=g0, gid1=g1, nthreads=8)
    tuned(**tune)
    got, st = _rays(engine, job, view, 0, g0, g1)
    assert st["tally_events"] == n and st["packets"] == 3 * (g1 - g0) and st["scatterings"] > 1000
    assert_image_close(got, want)


def test_point_source_rays(engine, oracle_soc, parking):
    cl = cloud104()
    k = 2.0 / (104 * float(cl.DENS[:104 ** 3][cl.DENS[:104 ** 3] > 0].mean()))
    ps = np.array([[52.3, 51.7, 50.2], [52.0, 52.0, 300.0]], np.float32)
    job = Job(cl, cases._CSC, ABS=0.3 * k, SCA=k, SOURCE=0, BATCH=12, SEED=0.2, GLOBAL=512, PSPOS=ps, PS=[1.0, 2.5], PS_METHOD=0, DSC=cases._DSC)
    view = view104()
    for kind in (2, 0):                                   # SimRAM_PS, and the point sources of SimRAM_PB
        want, n = oracle_sim_sca(oracle_soc, job, view, kind, nthreads=8)
        got, st = _rays(engine, job, view, kind, 0, 512)
        assert st["tally_events"] == n and st["packets"] == 12 * 512
        assert_image_close(got, want)


@pytest.mark.parametrize("emw", [0, 1])
def test_cell_emission_rays(emw, engine, oracle_soc, parking):
    cl = cloud104()
    k = 2.0 / (104 * float(cl.DENS[:104 ** 3][cl.DENS[:104 ** 3] > 0].mean()))
    emit = np.where(cl.DENS > 0, cl.DENS * 1e-3, 1e-4).astype(np.float32)
    emwei = np.random.default_rng(5).uniform(0, 2.5, cl.CELLS).astype(np.float32) if emw else None
    job = Job(cl, cases._CSC, ABS=0.3 * k, SCA=k, SOURCE=2, BATCH=1, SEED=0.9, GLOBAL=8192, EMIT=emit, EMWEI=emwei, USE_EMWEIGHT=emw, DSC=cases._DSC)
    view = view104()
    g0, g1 = 4000, 4024
    want, n = oracle_sim_sca(oracle_soc, job, view, 1, gid0=g0, gid1=g1, nthreads=8)
    got, st = _rays(engine, job, view, 1, g0, g1)
    assert st["tally_events"] == n and st["packets"] > 1000
    assert_image_close(got, want)


def test_reflecting_faces_and_the_direct_kernel_as_second_witness(engine, parking):
    """a larger launch than the oracle takes in seconds: both execution modes of the library, event counts equal, images to summation order;
    with reflecting faces (mask 21: one face per axis, see test_gpu_ltree.test_reflecting_faces_in_the_sweep) and without"""
    cl = cloud104()
    k = 2.0 / (104 * float(cl.DENS[:104 ** 3][cl.DENS[:104 ** 3] > 0].mean()))
    view = view104(angles=((60.0, 200.0), (10.0, 80.0)))
    for mirror in (0, 21):
        job = Job(cl, cases._CSC, ABS=0.3 * k, SCA=k, SOURCE=1, BATCH=2, SEED=0.61, DSC=cases._DSC, BG=1.0, MIRROR=mirror)
        a, sa = _direct(engine, job, view, 0, 0, 60000)
        b, sb = _rays(engine, job, view, 0, 0, 60000)
        assert sa == sb
        assert_image_close(b, a)
    engine.set_mirror(0)


def test_a_batch_of_launches_with_an_image_per_frequency(engine, parking):
    """soc_batch_begin ... soc_batch_end around scattered-light launches: point-source, background and cell-emission launches of several
    'frequencies' (own opacities, scattering functions, emission, seeds) deferred into ONE sweep of rays, each frequency adding to its own
    image (soc_sca_batch_images / _select / _read).  Every image equals the one the same launches give one at a time."""
    from soc_amd import synth
    cl = cloud104()
    k = 2.0 / (104 * float(cl.DENS[:104 ** 3][cl.DENS[:104 ** 3] > 0].mean()))
    view = view104(angles=((60.0, 200.0), (10.0, 80.0)))
    ps = np.array([[52.3, 51.7, 50.2]], np.float32)
    emit = np.where(cl.DENS > 0, cl.DENS * 1e-3, 1e-4).astype(np.float32)
    tabs = [synth.hg_scattering_table(g) for g in (0.6, 0.2, 0.4)]                # (DSC, CSC)
    freqs = []
    for f in range(3):
        kw = dict(ABS=(0.2 + 0.1 * f) * k, SCA=(1.0 - 0.2 * f) * k, DSC=tabs[f][0])
        freqs.append([(2, Job(cl, tabs[f][1], SOURCE=0, BATCH=6, SEED=0.2 + 0.1 * f, GLOBAL=4096, PSPOS=ps, PS=[1.0 + f], **kw), 0, 4096),
                      (0, Job(cl, tabs[f][1], SOURCE=1, BATCH=2, SEED=0.3 + 0.1 * f, BG=1.0 + f, **kw), 1000, 21000),
                      (1, Job(cl, tabs[f][1], SOURCE=2, BATCH=1, SEED=0.4 + 0.1 * f, GLOBAL=16384, EMIT=emit * (1 + f), **kw), 0, 16384)])
    # one at a time (rays too), one image
    single, stats = [], []
    for launches in freqs:
        img = None
        tot = dict(tally_events=0, packets=0, scatterings=0)
        for kind, job, g0, g1 in launches:
            engine.set_exec(1, 4)
            a, st = run_sca(engine, job, view, kind, g0, g1 - g0)
            img = a.astype(np.float64) if img is None else img + a
            for key in tot:
                tot[key] += st[key]
        single.append(img)
        stats.append(tot)
    # the same launches in one batch (soc_set_exec(1): in automatic mode a batch this small would go through the direct kernel)
    engine.set_exec(1, 4)
    engine.stats(reset=True)
    engine.batch_begin(0)
    engine.sca_batch_images(3)
    for f, launches in enumerate(freqs):
        engine.sca_batch_select(f)
        for kind, job, g0, g1 in launches:
            _defer(engine, job, view, kind, g0, g1 - g0)
    assert engine.last_passes() == 0                              # nothing has run yet (a launch run at once as rays would have left its passes) ...
    engine.batch_end()
    st = engine.stats()
    assert engine.last_form() == 3 and engine.last_passes() > 0   # ... and all of it ran as one sweep
    for key in ("tally_events", "packets", "scatterings"):
        assert st[key] == sum(s[key] for s in stats)
    for f in range(3):
        assert_image_close(engine.sca_batch_read(f), single[f], rtol=2e-5)
    engine.sca_batch_images(0)
    engine.set_exec(-1, 4)


def _defer(eng, job, view, kind, gid_first, gid_count):
    """the calls of run_sca without the ones that would run what is pending (zero, stats, sync, read)"""
    eng.set_optical(job.ABS, job.SCA)
    eng.set_scatter_table(job.DSC, job.CSC)
    xps = (job.XPS_NSIDE, job.XPS_SIDE, job.XPS_AREA)
    if kind == 2:
        eng.sca_sim_ps(job.PACKETS, job.BATCH, job.SEED, job.BG, job.PSPOS[:, :3], job.PS, XPS=xps, GLOBAL=job.GLOBAL, gid_first=gid_first, gid_count=gid_count)
    elif kind == 0:
        eng.sca_sim_pb(job.SOURCE, job.PACKETS, job.BATCH, job.SEED, job.BG, job.PSPOS[:, :3], job.PS, XPS=xps, GLOBAL=job.GLOBAL, gid_first=gid_first, gid_count=gid_count)
    else:
        eng.set_emission(job.EMIT, job.EMWEI)
        eng.sca_sim_cl(job.SOURCE, job.PACKETS, job.BATCH, job.SEED, job.GLOBAL, gid_first=gid_first, gid_count=gid_count)
