#!/usr/bin/env python3
"""bench.py -- photon packets / second of the HIP packet path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input = the
per-frequency body of the reference's simulation loop (ASOC.py:1120-1461) for the workload:

  C2 (BASELINE.json configs[1]): 128^3 Cartesian cloud (lognormal density, seed 1234),
  one frequency (tmp.dust row 33, 4.677e14 Hz, GL = 0.01 pc), `bgpackets 1e8` ->
  GLOBAL = 8*AREA = 786432 work items x BATCH 127 = 99 876 864 packets of isotropic
  background (ASOC.py:1061-1064), HG(g=0.6) scattering table with 2500 bins, noabsorbed.

Multi-GPU (weak scaling): every rank simulates the full launch with its own stream seed
-- the reference's vestigial DEVICES/ID seed term, ASOC.py:1247 -- and the per-cell
absorption buffer is summed with ONE RCCL all-reduce per step (= per frequency).
`--scaling strong` instead splits the work items of one logical launch across ranks
(identical result to one GPU).

Inputs are resident in HBM before the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from soc_amd import launch, synth           # noqa: E402
from soc_amd.lib import Engine               # noqa: E402

# tmp.dust row 33 (4.677e14 Hz) at gridlength 0.01 pc: optical depth per unit density per
# root cell (ASOC_aux.py:582-587)
C2_ABS, C2_SCA = 8.9084e-7, 5.4552e-6
C2_FREQ = 4.677e14
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_TALLY_EVENT = 12                   # 4 B density read + 8 B tally read-modify-write


def c2_workload():
    cloud = synth.cartesian_cloud(128, seed=1234)
    dsc, csc = synth.hg_scattering_table(0.6, 2500)
    L = launch.bg_launch(100000000, cloud.AREA)
    return dict(name="C2: 128^3 Cartesian cloud, 1 frequency, bgpackets 1e8 (786432 work items x BATCH 127 = "
                     "99876864 packets), isotropic background, HG g=0.6 2500-bin scattering table, noabsorbed",
                cloud=cloud, DSC=dsc, CSC=csc, ABS=C2_ABS, SCA=C2_SCA, launch=L, SEED=0.7853981634)


def c3_workload():
    """BASELINE.json configs[2], the background part: 256^3-root octree with 3 refinement levels (the densest
    10 % of the cells of every level refined: 4.95e7 cells, 4.54e7 leaves), `bgpackets 1e9` -> 3 145 728 work
    items x BATCH 318, same dust row at GL = 0.005 pc (same optical depth across the model as C2)."""
    cloud = synth.octree_cloud(256, levels=4, frac=0.10, seed=1234)
    dsc, csc = synth.hg_scattering_table(0.6, 2500)
    L = launch.bg_launch(1000000000, cloud.AREA)
    return dict(name="C3 (background part): 256^3-root octree, 4 levels (49.5e6 cells), 1 frequency per step, bgpackets 1e9 "
                     "(%d work items x BATCH %d = %d packets), isotropic background, HG g=0.6 2500-bin scattering table, "
                     "noabsorbed" % (L["GLOBAL"], L["BATCH"], L["GLOBAL"] * L["BATCH"]),
                cloud=cloud, DSC=dsc, CSC=csc, ABS=0.5 * C2_ABS, SCA=0.5 * C2_SCA, launch=L, SEED=0.7853981634)


def host_cores():
    """CPU threads this process may really use: scheduler affinity capped by the cgroup quota
    (the GPU box exposes 256 hardware threads but grants a 16-CPU share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(work, budget_s=15.0):
    """Time the reference's own kernel (x86 build of kernel_ASOC.c, oracle/_ref) -- or, where
    that build is absent, the C restatement -- on an evenly strided sample of the work items
    of the same launch, on all host cores available to this process."""
    from oracle.pyoracle import Job, Oracle, Ref
    ncores = host_cores()
    L = work["launch"]
    job = Job(work["cloud"], work["CSC"], ABS=work["ABS"], SCA=work["SCA"], SOURCE=1, BATCH=L["BATCH"],
              SEED=launch.launch_seed(work["SEED"], 0), BG=1.0, TW=1.0, GLOBAL=L["GLOBAL"], DSC=work["DSC"])
    kind = "reference"
    try:
        if work["cloud"].LEVELS > 1:
            raise RuntimeError("reference builds bake the geometry in; only the C2 one is kept")
        runner = Ref("c128")
        run = lambda stride: runner.sim(job, 0, 0, L["GLOBAL"], nthreads=ncores, stride=stride)      # noqa: E731
    except Exception:
        kind = "port"
        runner = Oracle("libm")
        run = lambda stride: runner.sim(job, 0, 0, L["GLOBAL"], nthreads=ncores, stride=stride)      # noqa: E731
    # calibrate on 1/512 of the work items, then size the sample for ~budget_s
    t0 = time.time()
    run(512)
    t_cal = max(time.time() - t0, 1e-3)
    n_cal = (L["GLOBAL"] + 511) // 512
    rate_items = n_cal / t_cal
    stride = max(1, int(L["GLOBAL"] / max(rate_items * budget_s, 1)))
    t0 = time.time()
    run(stride)
    dt = time.time() - t0
    n_items = (L["GLOBAL"] + stride - 1) // stride
    packets = n_items * L["BATCH"]
    return dict(value=packets / dt, unit="packets/s", cores=ncores, kind=kind,
                sample="every %d-th of the %d work items of the same launch (%d packets) in %.1f s; "
                       "x86 build of the reference kernel_ASOC.c SimRAM_PB, %d threads"
                       % (stride, L["GLOBAL"], packets, dt, ncores) if kind == "reference" else
                       "every %d-th of the %d work items (%d packets) in %.1f s; C restatement, %d threads"
                       % (stride, L["GLOBAL"], packets, dt, ncores))


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed
    under profiles/ (collected separately, as the profiling guide prescribes)."""
    p = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--in-flight", type=int, default=0,
                    help="steps executed together in one brick sweep (soc_batch_begin/end); 1 = one launch at a time; "
                         "0 = the K steps in equal sweeps of at most 16 (on Cartesian grids 2.7e6 packets are in "
                         "flight, the next launches' work items are admitted as the first finish)")
    ap.add_argument("--workload", choices=["C2", "C3"], default="C2",
                    help="C2 = BASELINE.json configs[1] (the headline); C3 = the background part of configs[2]")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SOC_BENCH_REHEARSE_ON_ONE_GPU"):
            # rehearsal of the N > 1 code path on a one-GPU box: every rank on device 0, gloo instead of RCCL
            # (RCCL refuses two ranks on one device).  Not a measurement.
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    work = c3_workload() if args.workload == "C3" else c2_workload()
    if args.in_flight == 0:
        # sweeps of equal size, at most 16 launches each (K = 20 -> 10 + 10 rather than 16 + 4)
        args.in_flight = max(1, -(-args.steps // -(-args.steps // 16))) if args.steps > 0 else 16
    cloud, L = work["cloud"], work["launch"]
    eng = Engine(local_rank)
    eng.set_cloud(cloud)
    eng.set_features(with_int=0, ps_method=0, use_emweight=0)      # noabsorbed: TABS only
    eng.set_scatter_table(work["DSC"], work["CSC"])
    eng.set_optical(work["ABS"], work["SCA"])

    tabs = None
    if world > 1:
        # tally lives in a torch tensor so RCCL reduces it in place; kernels run on torch's stream
        tabs = torch.zeros(cloud.CELLS, dtype=torch.float32, device="cuda")
        eng.bind_tally(0, tabs.data_ptr())
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.zero(0)

    FFREQ = [C2_FREQ * 0.9, C2_FREQ, C2_FREQ * 1.1]
    TW = np.float32(launch.trapezoid_weight(FFREQ, 1))
    KDEV = 1.0 / world if args.scaling == "weak" else 1.0            # ASOC.py:180,1501
    BG = np.float32(1.0e-12 * L["WBG"] / C2_FREQ * KDEV)             # I_bg = 1e-12 cgs (ASOC.py:1194)
    if args.scaling == "weak":
        first, count = 0, L["GLOBAL"]
    else:
        first, count = launch.shard_range(L["GLOBAL"], rank, world)

    def steps(i0, n):
        """n steps = n launches of the workload with the seeds of (frequency i, device): the per-frequency
        body of the reference's loop.  They are handed to the engine together (deferred launches), which
        runs up to --in-flight of them per brick sweep; then one all-reduce of TABS."""
        dev_id = rank if args.scaling == "weak" else 0
        ndev = world if args.scaling == "weak" else 1
        eng.timer_start()
        eng.batch_begin(args.in_flight)
        for i in range(i0, i0 + n):
            seed = launch.launch_seed(work["SEED"], i, DEVICES=ndev, ID=dev_id)
            eng.sim_pb(1, L["PACKETS"], L["BATCH"], seed, BG, TW, GLOBAL=L["GLOBAL"], gid_first=first, gid_count=count)
        eng.batch_end()
        ms = eng.timer_stop()
        if world > 1:
            dist.all_reduce(tabs)
        return ms

    def fence():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        else:
            eng.sync()

    if args.warmup:
        steps(0, args.warmup)
    eng.stats(reset=True)
    fence()
    t0 = time.perf_counter()
    kernel_ms = [steps(args.warmup, args.steps) / max(args.steps, 1)]      # HIP-event span of all K steps / K
    fence()
    elapsed = time.perf_counter() - t0
    st = eng.stats()

    packets_rank = st["packets"]
    events_rank = st["tally_events"]
    if world > 1:
        t = torch.tensor([elapsed, float(packets_rank), float(events_rank)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        packets_total = int(t[1])
    else:
        packets_total = packets_rank

    passes = eng.last_passes()
    kernel_name = ("soc_brick_pass<scalar-opacity,TABS-only> (+ soc_brick_scan, soc_brick_scatter), %d passes in the last "
                   "sweep of up to %d steps: time is the HIP-event span of all kernels of the K steps / K" % (passes, args.in_flight)) if passes else \
        "soc_sim_pb_kernel<Cartesian,float,scalar-opacity,TABS-only>"
    if rank == 0:
        kavg_s = float(np.mean(kernel_ms)) * 1e-3 if kernel_ms else float("nan")
        alg_bytes = events_rank / max(args.steps, 1) * BYTES_PER_TALLY_EVENT
        achieved = alg_bytes / kavg_s / 1e9
        out = {
            "metric": "photon packets/sec",
            "value": packets_total / elapsed,
            "unit": "packets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": work["name"], "packets_per_step_per_gpu": packets_rank // max(args.steps, 1),
                       "cells": cloud.CELLS, "tally_events_per_packet": events_rank / max(packets_rank, 1),
                       "steps_in_flight": args.in_flight,
                       "parallelism": "1 process per GPU; %s" % (
                           "replicas with per-rank seeds + 1 RCCL all-reduce of TABS after the K steps (TABS integrates over frequency on the device)" if args.scaling == "weak"
                           else "work-item ranges of one launch + 1 RCCL all-reduce of TABS after the K steps")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic() if args.workload == "C2" else None,
                         "kernel": kernel_name,
                         "kernel_ms": kavg_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(work, args.cpu_budget)
            except Exception as e:                      # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "packets/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %s" % e}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
