"""Helpers shared by the parity tests: run one oracle ``Job`` on the HIP engine."""
import numpy as np


def run_engine(eng, job, kind=0, gid_first=0, gid_count=None, zero=True, exec_mode=0, brick_log2=4):
    """Execute ``job`` (oracle.pyoracle.Job) through the C ABI.  Returns (TABS, INT, stats)."""
    cl = job.cloud
    eng.set_cloud(cl)
    eng.set_features(with_int=job.WITH_INT, ps_method=job.PS_METHOD, use_emweight=job.USE_EMWEIGHT)
    eng.set_optical(job.ABS, job.SCA)
    eng.set_step_weight(*getattr(job, "STEP_WEIGHT", (0, 0.0, 0.0)))
    if getattr(job, "MSF", None) is not None:            # -D WITH_MSF: abundances once, cross sections + tables per frequency
        ABS, SCA, CSC, ABU = job.MSF
        eng.set_abundances(ABU)
        eng.set_optical_abu(ABS, SCA)
        eng.set_scatter_tables(None, CSC)
        assert np.array_equal(eng.read_opt().view(np.uint32), np.asarray(job.OPT, np.float32).view(np.uint32))
    else:
        eng.set_scatter_table(job.DSC, job.CSC)
        eng.set_opt(job.OPT)
    eng.set_mirror(getattr(job, "MIRROR", 0))
    eng.set_exec(exec_mode, brick_log2)
    if job.ROI is not None:
        eng.set_roi_save(job.ROI, job.ROI_STEP, job.ROI_NSIDE)
    if job.ROI_LOAD is not None:
        eng.set_roi_load(job.ROI_DIM, job.ROI_NSIDE, job.ROI_LOAD)
    if zero:
        eng.zero(0)
        eng.zero(1)
    eng.stats(reset=True)
    gid_count = job.GLOBAL - gid_first if gid_count is None else gid_count
    if kind == 0:
        xps = (job.XPS_NSIDE, job.XPS_SIDE, job.XPS_AREA)
        eng.sim_pb(job.SOURCE, job.PACKETS, job.BATCH, job.SEED, job.BG, job.TW,
                   PSPOS=job.PSPOS[:, :3], PS=job.PS, XPS=xps, GLOBAL=job.GLOBAL,
                   gid_first=gid_first, gid_count=gid_count)
    elif kind == 2:
        eng.set_hpbg(job.HPBG, job.HPBGP)
        eng.sim_hp(job.PACKETS, job.BATCH, job.SEED, job.TW, job.GLOBAL, gid_first=gid_first, gid_count=gid_count)
    else:
        eng.set_emission(job.EMIT, job.EMWEI)
        eng.set_ali(job.WITH_ALI)
        if job.WITH_ALI:
            eng.zero(0)                                  # clears XAB together with TABS
        if job.EMINDEX is not None:
            eng.set_emindex(job.EMINDEX)
        eng.sim_cl(job.SOURCE, job.PACKETS, job.BATCH, job.SEED, job.TW, job.GLOBAL,
                   gid_first=gid_first, gid_count=gid_count)
    eng.sync()
    if job.ROI is not None:
        job.ROI_SAVE_gpu = eng.roi_read()
        eng.set_roi_save(None)
    if job.ROI_LOAD is not None:
        eng.set_roi_load(None, 0, None)
    if kind == 1 and job.WITH_ALI:
        job.XAB_gpu = eng.read_tally(2)
        eng.set_ali(0)
    if getattr(job, "STEP_WEIGHT", (0,))[0] > 0:
        eng.set_step_weight(0)                           # the session's engine goes back to the plain state
    if getattr(job, "MSF", None) is not None:
        eng.set_scatter_table(None, job.MSF[2][0])
        eng.set_opt(None)
        eng.set_abundances(None)
    if job.WITH_INT == 2:
        job.INTV_gpu = np.stack([eng.read_tally(3 + k) for k in range(3)])
    return eng.read_tally(0), eng.read_tally(1), eng.stats()


def assert_tally_close(got, want, rtol=1e-5, floor_frac=1e-6):
    """Per-cell comparison: relative tolerance on cells that carry signal, absolute floor
    (a fraction of the largest tally) elsewhere.  Differences between identical-trajectory
    runs come only from the order of fp32 atomic adds."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = np.abs(want).max() if want.size else 0.0
    tol = rtol * np.abs(want) + floor_frac * rtol * scale
    bad = np.abs(got - want) > tol
    assert not bad.any(), "%d of %d cells differ; worst rel %.3e" % (
        bad.sum(), bad.size, (np.abs(got - want) / np.maximum(np.abs(want), 1e-300))[bad].max())
