"""The multi-GPU plumbing on ONE GPU: tallies and the scattered-light image bound to torch tensors
(soc_bind_tally / soc_sca_bind_out), the engine running on a torch stream (soc_set_stream), as soc_amd/dist.py sets
it up for RCCL.  Results must equal the ones with library-owned memory on the engine's own stream: direct kernel,
brick sweep, deferred launches (TABS only and with per-launch INT tallies)."""
import numpy as np
import pytest

import cases
from oracle.pyoracle import Job
from soc_amd import synth
from util import assert_tally_close, run_engine

pytestmark = pytest.mark.gpu


def _bound(engine, cells):
    import torch
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    tabs = torch.zeros(cells, dtype=torch.float32, device="cuda")
    ints = torch.zeros(cells, dtype=torch.float32, device="cuda")
    engine.set_stream(stream.cuda_stream)
    engine.bind_tally(0, tabs.data_ptr(), cells)
    engine.bind_tally(1, ints.data_ptr(), cells)
    return torch, stream, tabs, ints


def _unbind(engine, torch):
    engine.sync()
    engine.bind_tally(0, None)
    engine.bind_tally(1, None)
    engine.set_stream(0)
    torch.cuda.set_stream(torch.cuda.default_stream())


@pytest.mark.parametrize("mode", [0, 1])
def test_bound_tallies_on_a_torch_stream(mode, engine):
    cl = synth.cartesian_cloud(32, seed=21)
    job = Job(cl, cases._CSC, ABS=2e-5, SCA=6e-5, SOURCE=1, BATCH=6, SEED=0.31, WITH_INT=1, TW=1.5)
    T0, I0, s0 = run_engine(engine, job, 0, exec_mode=mode)
    torch, stream, tabs, ints = _bound(engine, cl.CELLS)
    try:
        T1, I1, s1 = run_engine(engine, job, 0, exec_mode=mode)
        assert s1 == s0
        assert_tally_close(T1, T0, rtol=1e-5)
        assert_tally_close(I1, I0, rtol=1e-5)
        # the tensor IS the tally: a torch op on the same stream sees the kernel's result without a host sync
        total = float((tabs.sum() + ints.sum()).item())
        assert abs(total - float(T0.astype(np.float64).sum() + I0.astype(np.float64).sum())) <= 1e-4 * abs(total)
        # ... and a collective-like in-place update by torch is what the engine reads back
        tabs.mul_(2.0)
        assert_tally_close(engine.read_tally(0), 2.0 * T0, rtol=1e-5)
        with pytest.raises(Exception):
            engine.bind_tally(0, tabs.data_ptr(), cl.CELLS - 1)          # wrong element count
        with pytest.raises(Exception):
            engine.set_cloud(synth.cartesian_cloud(16, seed=2))          # other cell count while a caller-owned tally is bound
        # the refused call left the handle as it was: the old grid (densities, links, brick cache) still runs
        engine.zero(0)
        engine.zero(1)
        engine.stats(reset=True)
        engine.sim_pb(job.SOURCE, job.PACKETS, job.BATCH, job.SEED, job.BG, job.TW, GLOBAL=job.GLOBAL)
        engine.sync()
        assert engine.stats() == s0
        assert_tally_close(engine.read_tally(0), T0, rtol=1e-5)
    finally:
        _unbind(engine, torch)
    engine.set_exec(-1, 4)
    T2, _, s2 = run_engine(engine, job, 0, exec_mode=mode)
    assert s2 == s0
    assert_tally_close(T2, T0, rtol=1e-5)
    engine.set_exec(-1, 4)


def test_bound_tally_with_deferred_launches(engine):
    """soc_batch_begin/end and soc_batch_begin_int with a bound TABS on a torch stream (the bench's N > 1 path)"""
    cl = synth.octree_cloud(104, levels=4, frac=0.08, seed=3)
    jobs = [Job(cl, cases._CSC, ABS=3e-6, SCA=3e-5, SOURCE=1, BATCH=2, SEED=0.377 + 0.1 * k, TW=1.0 + k) for k in range(3)]
    g0, n = 150000, 4000
    res = {}
    for bound in (False, True):
        if bound:
            torch, stream, tabs, ints = _bound(engine, cl.CELLS)
        try:
            for keep_int in (0, 1):
                e = engine
                e.set_cloud(cl) if not bound else None
                e.set_features(with_int=keep_int, ps_method=0, use_emweight=0)
                e.set_opt(None)
                e.set_exec(1, 4)
                e.zero(0)
                e.stats(reset=True)
                (e.batch_begin_int if keep_int else e.batch_begin)(4)
                for j in jobs:
                    e.set_scatter_table(j.DSC, j.CSC)
                    e.set_optical(j.ABS, j.SCA)
                    e.sim_pb(1, 0, j.BATCH, j.SEED, j.BG, j.TW, GLOBAL=j.GLOBAL, gid_first=g0, gid_count=n)
                e.batch_end()
                got = (e.read_tally(0), [e.batch_read_int(k) for k in range(len(jobs))] if keep_int else [], e.stats())
                if bound:
                    assert got[2] == res[keep_int][2]
                    assert_tally_close(got[0], res[keep_int][0], rtol=1e-5)
                    for a, b in zip(got[1], res[keep_int][1]):
                        assert_tally_close(a, b, rtol=1e-5)
                    assert_tally_close(tabs.cpu().numpy(), res[keep_int][0], rtol=1e-5)
                else:
                    res[keep_int] = got
        finally:
            if bound:
                _unbind(engine, torch)
    engine.set_exec(-1, 4)


def test_bound_image_on_a_torch_stream(engine):
    import torch
    from test_gpu_sca import assert_image_close, run_sca
    ref, kind, mk, vkw = cases.SCA_CASES["sca_bg_c8"]
    job, view = mk(), cases.sca_view(**vkw)
    want, st0 = run_sca(engine, job, view, kind)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    try:
        engine.set_stream(stream.cuda_stream)
        img = torch.zeros(want.size, dtype=torch.float32, device="cuda")
        engine.sca_set_view(view.ODIR, view.RA, view.DE, view.NPIX, view.MAP_DX, view.CENTRE, view.FFS)
        engine.sca_bind_out(img.data_ptr())
        got, st1 = run_sca(engine, job, view, kind, rebind=img.data_ptr())
        assert st1 == st0
        assert_image_close(got, want)
        assert_image_close(img.cpu().numpy(), want)
    finally:
        engine.sync()
        engine.sca_bind_out(None)                                        # library-owned image again
        engine.set_stream(0)
        torch.cuda.set_stream(torch.cuda.default_stream())


def test_library_and_torch_load_in_either_order():
    """torch ships its own HIP runtime; the one of two runtimes that is initialised second finds no device.  soc_amd.lib loads torch's
    first wherever torch is installed, so a caller may create the engine before or after `import torch` (a child process each: the
    session's fixture has loaded both already)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = ("import sys; sys.path.insert(0, %r)\n"
            "%s\n"
            "import numpy as np\n"
            "from soc_amd import synth\n"
            "eng.set_cloud(synth.cartesian_cloud(8, seed=1))\n"
            "t = torch.ones(4, device='cuda') * 2\n"
            "assert torch.cuda.is_available() and float(t.sum().item()) == 8.0\n"
            "eng.close()\nprint('order ok')\n")
    first_engine = "from soc_amd.lib import Engine\neng = Engine(0)\nimport torch"
    first_torch = "import torch\nfrom soc_amd.lib import Engine\neng = Engine(0)"
    for order in (first_engine, first_torch):
        out = subprocess.run([sys.executable, "-c", body % (repo, order)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "order ok" in out.stdout, out.stderr[-2000:]
