"""N>1 path on CPU: two gloo ranks share one logical launch by work-item ranges and
all-reduce the tallies; the result must equal the single-process run (same RNG streams)."""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import numpy as np
from soc_amd.ini import User
from soc_amd.asoc import AbsorptionRun
from soc_amd.dist import Comm
from oracle_engine import OracleEngine
comm = Comm(backend="gloo")
os.chdir(sys.argv[2] + "/r%d" % comm.rank)
run = AbsorptionRun(User(sys.argv[1]), OracleEngine("soc"), comm, verbose=0, shard=(sys.argv[3] if len(sys.argv) > 3 else "items"))
C, F = run.run()
np.save("ctabs_rank%d.npy" % comm.rank, C)
comm.close()
"""


import pytest


@pytest.mark.parametrize("grid", ["octree", "cartesian"])
def test_two_rank_sharded_run_equals_single(grid, tmp_path):
    """octree: one launch per frequency, INT all-reduced per frequency; cartesian: frequencies batched with their own INT
    tallies (soc_batch_begin_int), each summed over the ranks when it is read"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_host import _write_model
    from oracle_engine import OracleEngine
    from soc_amd import synth
    from soc_amd.ini import User
    from soc_amd.asoc import AbsorptionRun
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9) if grid == "octree" else synth.cartesian_cloud(6, seed=9)
    ini = _write_model(d, cloud, with_ps=True, with_diffuse=True)
    for r in (0, 1):
        os.makedirs(os.path.join(d, "r%d" % r))
    os.makedirs(os.path.join(d, "single"))
    os.chdir(os.path.join(d, "single"))
    C1, F1 = AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0).run()
    script = os.path.join(d, "worker.py")
    with open(script, "w") as fp:
        fp.write(WORKER.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", script, ini, d],
                          env=env, timeout=600)
    for r in (0, 1):
        C = np.load(os.path.join(d, "r%d" % r, "ctabs_rank%d.npy" % r))
        assert np.allclose(C, C1, rtol=1e-5, atol=1e-7 * np.abs(C1).max())
    # rank 0 wrote the absorbed file; per-frequency tallies were all-reduced before they were pulled
    from soc_amd import files
    A = files.read_absorbed(os.path.join(d, "abs.data"))
    want = F1                                   # (run() has scaled the array it returns, as it wrote it)
    assert np.allclose(A, want, rtol=1e-5, atol=1e-7 * np.abs(want).max())


@pytest.mark.parametrize("noabsorbed", [False, True], ids=["absorbed_file", "tabs_only"])
def test_two_ranks_share_the_launch_sequence(noabsorbed, tmp_path):
    """shard="launches": an absorbed-file run gives every frequency to one rank, which simulates it whole and writes its column
    of the file itself (no collective for the per-frequency absorptions); a TABS-only run gives each rank a contiguous share of the
    launch sequence.  Same packets and streams as one process: results equal to summation order."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_host import _write_model
    from oracle_engine import OracleEngine
    from soc_amd import synth, files
    from soc_amd.ini import User
    from soc_amd.asoc import AbsorptionRun
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    ini = _write_model(d, cloud, with_ps=True, with_diffuse=True, extra="noabsorbed\n" if noabsorbed else "")
    for r in (0, 1):
        os.makedirs(os.path.join(d, "r%d" % r))
    os.makedirs(os.path.join(d, "single"))
    os.chdir(os.path.join(d, "single"))
    C1, F1 = AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0).run()
    if not noabsorbed:
        os.remove(os.path.join(d, "abs.data"))
    script = os.path.join(d, "worker.py")
    with open(script, "w") as fp:
        fp.write(WORKER.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", script, ini, d, "launches"],
                          env=env, timeout=600)
    for r in (0, 1):
        C = np.load(os.path.join(d, "r%d" % r, "ctabs_rank%d.npy" % r))
        assert np.allclose(C, C1, rtol=1e-5, atol=1e-7 * np.abs(C1).max())
    if not noabsorbed:
        A = files.read_absorbed(os.path.join(d, "abs.data"))
        assert A.shape == F1.shape and np.allclose(A, F1, rtol=1e-5, atol=1e-7 * np.abs(F1).max())


def test_launch_shares_partition_the_sequence():
    from soc_amd import launch
    G = [16777216, 4096, 100000, 16777216, 64]
    W = [1e9, 3e5, 7e6, 9.9e8, 4e3]
    for world in (1, 2, 3, 8):
        tot = [0] * len(G)
        per_rank = []
        for r in range(world):
            sh = launch.shard_launches(G, W, r, world)
            assert all(f % 64 == 0 for f, c in sh if c)
            for i, (f, c) in enumerate(sh):
                tot[i] += c
            per_rank.append(sum(c * W[i] / G[i] for i, (f, c) in enumerate(sh)))
        assert tot == G                                          # every work item exactly once
        assert max(per_rank) - min(per_rank) <= 0.02 * sum(W) / world + 64 * 1e9 / 16777216      # equal packets per rank


SCA_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import numpy as np
from soc_amd.ini import User
from soc_amd.asocs import ScatteringRun
from soc_amd.dist import Comm
from oracle_engine import OracleEngine
comm = Comm(backend="gloo")
os.chdir(sys.argv[2] + "/s%d" % comm.rank)
OUT = ScatteringRun(User(sys.argv[1]), OracleEngine("soc"), comm, verbose=0).run()
np.save("out_rank%d.npy" % comm.rank, OUT)
comm.close()
"""


def test_two_rank_scattering_run_equals_single(tmp_path):
    """Scattered-light images: two ranks split every launch by work-item range and all-reduce
    the image; same streams, so the sum equals the single-process image."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_host_sca import _ini
    from oracle_engine import OracleEngine
    from soc_amd import synth, files
    from soc_amd.ini import User
    from soc_amd.asocs import ScatteringRun
    d = str(tmp_path)
    cloud = synth.octree_cloud(6, levels=2, frac=0.1, seed=9)
    ini = _ini(d, cloud, with_ps=True)
    for r in (0, 1):
        os.makedirs(os.path.join(d, "s%d" % r))
    os.makedirs(os.path.join(d, "single"))
    os.chdir(os.path.join(d, "single"))
    O1 = ScatteringRun(User(ini), OracleEngine("soc"), verbose=0).run()
    script = os.path.join(d, "sca_worker.py")
    with open(script, "w") as fp:
        fp.write(SCA_WORKER.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29537", script, ini, d],
                          env=env, timeout=600)
    for r in (0, 1):
        O = np.load(os.path.join(d, "s%d" % r, "out_rank%d.npy" % r))
        assert np.allclose(O, O1, rtol=1e-5, atol=1e-7 * np.abs(O1).max())
    _, data = files.read_outcoming(os.path.join(d, "s0", "outcoming.socs"), 2)
    assert np.allclose(data, O1, rtol=1e-5, atol=1e-7 * np.abs(O1).max())
    assert not os.path.exists(os.path.join(d, "s1", "outcoming.socs"))       # only rank 0 writes


def test_two_rank_roi_record_equals_single(tmp_path):
    """roisave with two ranks: every rank records the packets of its work items, the per-frequency records are
    summed over the ranks (Comm.all_reduce_host) and rank 0 writes the file"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_host import _write_model
    from oracle_engine import OracleEngine
    from soc_amd import synth
    from soc_amd.ini import User
    from soc_amd.asoc import AbsorptionRun
    d = str(tmp_path)
    cloud = synth.cartesian_cloud(8, seed=3)
    ini = _write_model(d, cloud, extra="gridlength 5e-7\nroi 2 5 2 5 3 4\nroisave %s/roi.save 1\nroinside 2\n" % d)
    for r in (0, 1):
        os.makedirs(os.path.join(d, "r%d" % r))
    os.makedirs(os.path.join(d, "single"))
    os.chdir(os.path.join(d, "single"))
    AbsorptionRun(User(ini), OracleEngine("soc"), verbose=0).run()
    single = np.fromfile(os.path.join(d, "roi.save"), np.float32, offset=20)
    assert (single > 0).sum() > 100
    os.remove(os.path.join(d, "roi.save"))
    script = os.path.join(d, "worker.py")
    with open(script, "w") as fp:
        fp.write(WORKER.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29535", script, ini, d],
                          env=env, timeout=600)
    assert list(np.fromfile(os.path.join(d, "roi.save"), np.int32, 5)) == [4, 4, 2, 2, 3]
    both = np.fromfile(os.path.join(d, "roi.save"), np.float32, offset=20)
    assert np.array_equal(both > 0, single > 0)
    assert np.allclose(both, single, rtol=1e-5, atol=1e-7 * single.max())


A2E_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
from soc_amd import a2e
from soc_amd.dist import Comm
from oracle_engine import OracleA2E
comm = Comm(backend="gloo")
a2e.run_sharded(OracleA2E, sys.argv[1], sys.argv[2], sys.argv[3], NSTOCH=2, comm=comm, verbose=False)
comm.close()
"""


def test_two_rank_a2e_equals_single(tmp_path):
    """A2E on N GPUs: the cells are split over the ranks, every rank writes its part of the emitted file"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle_engine import OracleA2E
    from soc_amd import a2e, files, synth
    d = str(tmp_path)
    sol = synth.synth_solver(NFREQ=12, NE=16, NSIZE=3, seed=2)
    synth.write_solver(os.path.join(d, "x.solver"), sol)
    ABS = (np.random.default_rng(3).lognormal(0, 1, (301, 12)) * 1e-3).astype(np.float32)
    files.write_absorbed(os.path.join(d, "abs.bin"), ABS)
    assert a2e.cell_range(301, 0, 2) == (0, 151) and a2e.cell_range(301, 1, 2) == (151, 301) and a2e.cell_range(3, 3, 8) == (3, 3)
    n, _ = a2e.run_sharded(OracleA2E, os.path.join(d, "x.solver"), os.path.join(d, "abs.bin"), os.path.join(d, "em1.bin"),
                           NSTOCH=2, verbose=False)
    assert n == 301
    E1 = files.read_absorbed(os.path.join(d, "em1.bin"))
    want, _ = a2e.run(OracleA2E(), files.read_solver(os.path.join(d, "x.solver")), ABS, NSTOCH=2, verbose=False)
    assert E1.shape == (301, 12) and np.array_equal(E1, want) and (E1 > 0).any()
    script = os.path.join(d, "worker.py")
    with open(script, "w") as fp:
        fp.write(A2E_WORKER.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29537", script,
                           os.path.join(d, "x.solver"), os.path.join(d, "abs.bin"), os.path.join(d, "em2.bin")],
                          env=env, timeout=600)
    E2 = files.read_absorbed(os.path.join(d, "em2.bin"))
    assert np.array_equal(E2, E1)                      # cells are independent: bit-identical
