#!/usr/bin/env python3
"""bench.py -- photon packets / second of the HIP packet path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input = one kernel launch of the
reference's simulation loop (source block II x frequency IFREQ, ASOC.py:1028-1461) for the workload:

  C3 (default; BASELINE.json configs[2]): 256^3-root octree with 3 refinement levels
  (LEVELS = 4; the densest 10 % of the cells of every level refined: 4.95e7 cells), GL = 0.02 pc, the
  50-frequency optical table soc_amd/data/c3_dust50.txt (1.5e11 .. 2e15 Hz, log-log interpolation of the
  reference's example dust; per-frequency ABS, SCA and a Henyey-Greenstein scattering table of the row's g),
  `pspackets 1e9` from one point source at (128.3, 128.3, 128.3) (SimRAM_PB SOURCE 0, ASOC.py:1036-1044) and
  `diffpack 1e9` of diffuse emission 1e-30 * density photons/Hz/cm3 (SimRAM_CL, ASOC.py:1086-1090), noabsorbed.
  Step i simulates frequency (21 * (i // 2) + 45) % 50 -- a stride through the whole table, so that the driver's 5 + 20
  steps hit optically thin and thick frequencies alike (the scattering block, kernel_ASOC.c:695-815, runs in the timed
  region); even steps are the point-source launch, odd steps the diffuse one.
  Launch size: the reference's absorption script hard-codes GLOBAL_0 = 32768 work items for both (ASOC.py:86) -- 512
  wavefronts, half a wavefront per SIMD of an MI355X.  ASOC.py parses the ini key `global` (ASOC_aux.py:400) but never
  reads it; only ASOCS.py:82 does.  The bench runs --global work items (default 2^24: same packets, same sources, another
  partition into RNG streams): a launch shape the reference's ASOC.py can only be given by editing ASOC.py:86.  The rate
  at the reference's own launch shape is measured beside it on a shortened launch (config.reference_launch_shape).

  C4 (--workload C4; BASELINE.json configs[3], the ASOCS scattering path): see run_c4.

  C2 (--workload C2; BASELINE.json configs[1]): 128^3 Cartesian cloud, one frequency, `bgpackets 1e8` ->
  786432 work items x BATCH 127 of isotropic background, noabsorbed.

The K steps are handed to the engine together (soc_batch_begin/end): on the config-3 hierarchy point-source and diffuse
launches share one brick sweep (up to 128 launches: the two source blocks of a 50-frequency run, ASOC.py:1028-1545).

Multi-GPU: photon packets are independent units, so the default (--scaling auto = weak) keeps the per-GPU work fixed: every rank
runs the K steps with the reference's per-device seed term and the weight 1/N of its multi-device design (ASOC.py:180,1247,1501) -- N
times the packets, the same per-rank population as one GPU -- and ONE RCCL all-reduce of the per-cell absorption buffer after the
timed sweeps.  --scaling strong splits the work items of every launch across the ranks (identical result to one GPU),
--scaling launches gives every rank a share of the launch (frequency) sequence; both fix the total work.

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

from soc_amd import files, launch, synth    # noqa: E402
from soc_amd.lib import Engine               # noqa: E402

# tmp.dust row 33 (4.677e14 Hz) at gridlength 0.01 pc: optical depth per unit density per
# root cell (ASOC_aux.py:582-587)
C2_ABS, C2_SCA = 8.9084e-7, 5.4552e-6
C2_FREQ = 4.677e14
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_TALLY_EVENT = 12                   # 4 B density read + 8 B tally read-modify-write; + 8 B with the INT tally
C3_GL = 0.02                                 # pc
C3_CELLS = 49526352                          # synth.octree_cloud(256, levels=4, frac=0.10, seed=1234); oracle/build.py ref "oct256"
R_SUN, T_SUN = 6.957e10, 5800.0


def c2_workload():
    cloud = synth.cartesian_cloud(128, seed=1234)
    dsc, csc = synth.hg_scattering_table(0.6, 2500)
    L = launch.bg_launch(100000000, cloud.AREA)
    FF = [C2_FREQ * 0.9, C2_FREQ, C2_FREQ * 1.1]
    step = dict(kind="bg", L=L, ABS=np.float32(C2_ABS), SCA=np.float32(C2_SCA), CSC=csc, DSC=dsc,
                TW=np.float32(launch.trapezoid_weight(FF, 1)), BG=np.float32(1.0e-12 * L["WBG"] / C2_FREQ), IFREQ=0)
    return dict(name="C2: 128^3 Cartesian cloud, 1 frequency, bgpackets 1e8 (786432 work items x BATCH 127 = "
                     "99876864 packets), isotropic background, HG g=0.6 2500-bin scattering table, noabsorbed",
                cloud=cloud, step=lambda i: step, SEED=0.7853981634, ref_tag="c128", kinds=("bg",))


def c3_workload(GLOBAL_0):
    cloud = synth.octree_cloud(256, levels=4, frac=0.10, seed=1234)
    FFREQ, AFG, AFABS, AFSCA = files.read_dust([os.path.join(REPO, "soc_amd", "data", "c3_dust50.txt")], C3_GL)
    NFREQ = len(FFREQ)
    tables = [synth.hg_scattering_table(float(g), 2500) for g in AFG[0]]
    LOCAL = launch.LOCAL_GPU
    pc = launch.packet_counts(0, 1000000000, 0, 1000000000, cloud.AREA, cloud.CELLS, LOCAL, 0)     # ASOC.py:234-250
    LPS = launch.ps_launch(pc["PSPAC"], 1, C3_GL, GLOBAL_0)
    LCL = launch.cl_launch(pc["DFPAC"], cloud.CELLS, GLOBAL_0)
    # point source: L_nu = 4 pi R^2 pi B_nu(5800 K) -> photons per package (ASOC.py:1236)
    f64 = np.asarray(FFREQ, np.float64)
    Bnu = 2.0 * launch.PLANCK * f64 ** 3 / launch.C_LIGHT ** 2 / np.expm1(np.clip(launch.H_K * f64 / T_SUN, 1e-10, 600.0))
    Lnu = 4.0 * np.pi * R_SUN ** 2 * np.pi * Bnu
    PS = np.asarray(Lnu * LPS["WPS"] / f64, np.float32)
    PSPOS = np.asarray([[128.3, 128.3, 128.3]], np.float32)
    # diffuse emission 1e-30 * density photons / Hz / cm3, flat spectrum -> photons per cell (ASOC.py:1262-1276)
    EMIT = np.zeros(cloud.CELLS, np.float32)
    for level in range(cloud.LEVELS):
        a, b = int(cloud.OFF[level]), int(cloud.OFF[level] + cloud.LCELLS[level])
        d = cloud.DENS[a:b]
        EMIT[a:b] = np.where(d > 0.0, 1.0e-30 * d, 0.0) * (C3_GL * launch.PARSEC / (8.0 ** level))

    def step_for(f, kind):
        dsc, csc = tables[f]
        return dict(kind=kind, L=LPS if kind == "ps" else LCL, ABS=AFABS[0][f], SCA=AFSCA[0][f], CSC=csc, DSC=dsc,
                    TW=np.float32(launch.trapezoid_weight(FFREQ, f)), BG=np.float32(0.0), IFREQ=f,
                    PS=PS[f:f + 1], PSPOS=PSPOS, EMIT=EMIT)

    def step(i):
        return step_for((21 * (i // 2) + 45) % NFREQ, "ps" if (i % 2) == 0 else "cl")
    name = ("C3: 256^3-root octree, LEVELS 4 (%d cells), GL 0.02 pc, 50 frequencies 1.5e11-2e15 Hz (own ABS, SCA, HG(g) "
            "table each), noabsorbed; step i = frequency (21*(i//2)+45)%%50, even: point source at (128.3,128.3,128.3) pspackets 1e9 "
            "= %d work items x BATCH %d = %d packets; odd: diffuse emission diffpack 1e9 = BATCH %d per cell = %d packets, "
            "%d work items; GLOBAL_0 = %d work items per launch -- NOT a shape the reference's ASOC.py can be given without "
            "editing it (ASOC.py:86 hard-codes 32768; the `global` key is read by ASOCS.py:82 only)"
            % (cloud.CELLS, LPS["GLOBAL"], LPS["BATCH"], LPS["PACKETS"], LCL["BATCH"], LCL["BATCH"] * cloud.CELLS,
               min(LCL["GLOBAL"], cloud.CELLS), GLOBAL_0))
    return dict(name=name, cloud=cloud, step=step, step_for=step_for, NFREQ=NFREQ, SEED=0.7853981634, ref_tag="oct256", kinds=("ps", "cl"))


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
    """Time the reference's own kernel (x86 build of kernel_ASOC.c, oracle/_ref) -- or, where that build is
    absent, the C restatement -- on an evenly strided sample of the work items of the workload's launches
    (C3: half the budget on the point-source launch, half on the diffuse one), on all host cores available."""
    from oracle.pyoracle import Job, Oracle, Ref
    ncores = host_cores()
    cloud = work["cloud"]
    kind = "reference"
    try:
        runner = Ref(work["ref_tag"])
        if runner.model["CELLS"] != cloud.CELLS:
            raise RuntimeError("reference build %s is for another hierarchy" % work["ref_tag"])
    except Exception:
        kind = "port"
        runner = Oracle("libm")
    packets, seconds, parts = 0, 0.0, []
    for k, which in enumerate(work["kinds"]):
        s = work["step"](k if len(work["kinds"]) > 1 else 0)
        L = s["L"]
        seed = launch.launch_seed(work["SEED"], s["IFREQ"])
        if which == "cl":
            job = Job(cloud, s["CSC"], ABS=s["ABS"], SCA=s["SCA"], SOURCE=2, BATCH=L["BATCH"], SEED=seed, TW=s["TW"],
                      GLOBAL=L["GLOBAL"], DSC=s["DSC"], EMIT=s["EMIT"])
            okind, nitems, per_item = 1, min(L["GLOBAL"], cloud.CELLS), L["BATCH"] * max(1.0, cloud.CELLS / L["GLOBAL"])
        elif which == "ps":
            job = Job(cloud, s["CSC"], ABS=s["ABS"], SCA=s["SCA"], SOURCE=0, BATCH=L["BATCH"], SEED=seed, BG=0.0, TW=s["TW"],
                      GLOBAL=L["GLOBAL"], DSC=s["DSC"], PSPOS=s["PSPOS"], PS=s["PS"])
            okind, nitems, per_item = 0, L["GLOBAL"], L["BATCH"]
        else:
            job = Job(cloud, s["CSC"], ABS=s["ABS"], SCA=s["SCA"], SOURCE=1, BATCH=L["BATCH"], SEED=seed, BG=1.0, TW=1.0,
                      GLOBAL=L["GLOBAL"], DSC=s["DSC"])
            okind, nitems, per_item = 0, L["GLOBAL"], L["BATCH"]
        run = lambda stride: runner.sim(job, okind, 0, nitems, nthreads=ncores, stride=stride)      # noqa: E731
        # calibrate on a thin sample of the work items, then size the sample for its share of the budget
        cal = max(512, nitems // (64 * ncores))
        t0 = time.time()
        run(cal)
        t_cal = max(time.time() - t0, 1e-3)
        rate_items = ((nitems + cal - 1) // cal) / t_cal
        stride = max(1, int(nitems / max(rate_items * budget_s / len(work["kinds"]), 1)))
        t0 = time.time()
        run(stride)
        dt = time.time() - t0
        n_items = (nitems + stride - 1) // stride
        packets += n_items * per_item
        seconds += dt
        parts.append("%s: every %d-th of %d work items (%d packets) in %.1f s" % (which, stride, nitems, n_items * per_item, dt))
    what = "x86 build of the reference kernel_ASOC.c (SimRAM_PB / SimRAM_CL)" if kind == "reference" else "C restatement"
    return dict(value=packets / seconds, unit="packets/s", cores=ncores, kind=kind,
                sample="; ".join(parts) + "; %s, %d threads" % (what, ncores))


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed
    under profiles/ (collected separately, as the profiling guide prescribes)."""
    p = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            t = json.load(open(p))
            t = t.get(workload, t) if isinstance(t.get(workload), dict) else (t if workload == "C2" else {})
            return t.get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None



def run_c4(args, world, rank, local_rank, dist, torch):
    """--workload C4 (BASELINE.json configs[3]): the scattered-light kernels on the config-3 hierarchy -- 3 observers, 256^2 pixels,
    forced first scattering, the 50-frequency dust with its scattering functions.  Step i is one launch: frequency (21*(i//3)+45)%50,
    kind background / point source / cell emission in turn, as soc_amd.asocs runs a source block -- the K steps deferred into one batch
    with an image per frequency (soc_batch_begin ... soc_sca_batch_images ... soc_batch_end): one sweep of rays on the brick-local
    hierarchies.  value = packets/s; roofline: 4 B (a density) per cell step of a ray."""
    import math
    work = c3_workload(args.global0)
    cloud = work["cloud"]
    N = cloud.NX
    eng = Engine(local_rank)
    eng.set_cloud(cloud)
    eng.set_features(0, 0, 0)
    eng.set_opt(None)
    th = [math.radians(30 + 25 * i) for i in range(3)]
    ph = [math.radians(40 * i) for i in range(3)]
    _, OD, RA, DE = launch.set_observer_directions(th, ph)
    eng.sca_set_view(OD, RA, DE, (256, 256), N / 256.0 * 1.5, (N / 2, N / 2, N / 2), 1)
    AREA = 6 * N * N
    GBG = launch.Fix(8 * AREA, 64)
    shapes = {"bg": dict(GLOBAL=GBG, BATCH=4, packets=8 * AREA * 4), "ps": dict(GLOBAL=2097152, BATCH=8, packets=2097152 * 8),
              "cl": dict(GLOBAL=8388608, BATCH=1, packets=cloud.CELLS)}
    kinds = ("bg", "ps", "cl")
    ps = np.array([[N / 2 + 0.3, N / 2 + 0.2, N / 2 + 0.1]], np.float32)
    emit = np.where(cloud.DENS > 0, cloud.DENS * 1e-3, 0).astype(np.float32)
    eng.set_emission(emit)
    dev_id, ndev = (rank, world)

    def step(i):
        f = (21 * (i // 3) + 45) % work["NFREQ"]
        return f, kinds[i % 3], work["step_for"](f, "ps")

    def run_steps(i0, n):
        freqs = sorted({step(i)[0] for i in range(i0, i0 + n)})
        eng.timer_start()
        eng.batch_begin(0)
        eng.sca_batch_images(len(freqs))
        for i in range(i0, i0 + n):
            f, kind, s = step(i)
            sh = shapes[kind]
            seed = launch.launch_seed(work["SEED"], f, DEVICES=ndev, ID=dev_id)
            eng.sca_batch_select(freqs.index(f))
            eng.set_optical(s["ABS"], s["SCA"])
            eng.set_scatter_table(s["DSC"], s["CSC"])
            if kind == "bg":
                eng.sca_sim_pb(1, sh["packets"], sh["BATCH"], seed, np.float32(1.0 / world), GLOBAL=sh["GLOBAL"])
            elif kind == "ps":
                eng.sca_sim_ps(sh["packets"], sh["BATCH"], seed, 0.0, ps, [np.float32(1.0 / world)], GLOBAL=sh["GLOBAL"])
            else:
                eng.sca_sim_cl(2, cloud.CELLS, sh["BATCH"], seed, sh["GLOBAL"])
        eng.batch_end()
        ms = eng.timer_stop()
        imgs = np.stack([eng.sca_batch_read(k) for k in range(len(freqs))])
        eng.sca_batch_images(0)
        return ms, imgs, freqs

    def fence():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        else:
            eng.sync()

    if args.warmup:
        run_steps(0, args.warmup)
    eng.stats(reset=True)
    fence()
    t0 = time.perf_counter()
    first = args.first_step if args.first_step >= 0 else args.warmup
    kernel_ms, imgs, freqs = run_steps(first, args.steps)
    if world > 1:                                           # the images of the ranks add up (weight 1/N each): one all-reduce
        t = torch.from_numpy(imgs)
        t = t if os.environ.get("SOC_BENCH_REHEARSE_ON_ONE_GPU") else t.cuda()
        dist.all_reduce(t)
    fence()
    elapsed = time.perf_counter() - t0
    st = eng.stats()
    steps_rays = eng.sca_ray_steps()
    form, passes = eng.last_form(), eng.last_passes()
    packets_total = st["packets"]
    if world > 1:
        dev = "cpu" if os.environ.get("SOC_BENCH_REHEARSE_ON_ONE_GPU") else "cuda"
        t = torch.tensor([elapsed, float(st["packets"])], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, packets_total = float(tmax[0]), int(t[1])
    if rank == 0:
        alg = 4.0 * steps_rays / max(args.steps, 1)
        kavg_s = kernel_ms * 1e-3 / max(args.steps, 1)
        out = {"metric": "photon packets/sec", "value": packets_total / elapsed, "unit": "packets/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "C4: scattered-light images (kernel_ASOC_sca.c) on the config-3 hierarchy (256^3 roots, LEVELS 4, %d cells), 3 observers, "
                                      "256^2 pixels, forced first scattering, 50-frequency dust with its HG(g) scattering functions; step i = one launch at frequency "
                                      "(21*(i//3)+45)%%50: background (8*AREA = %d work items x BATCH 4), point source (2097152 x 8), cell emission (8388608 work items, "
                                      "one packet per cell) in turn; all steps deferred into one batch, an image per frequency" % (cloud.CELLS, 8 * AREA),
                          "packets_per_step_per_gpu": st["packets"] // max(args.steps, 1), "cells": cloud.CELLS,
                          "image_contributions_per_packet": st["tally_events"] / max(st["packets"], 1), "scatterings_per_packet": st["scatterings"] / max(st["packets"], 1),
                          "ray_steps_per_packet": steps_rays / max(st["packets"], 1), "frequencies_in_the_timed_steps": freqs,
                          "parallelism": "1 process per GPU; replicas with per-rank seeds, weight 1/N + 1 all-reduce of the images"},
               "roofline": {"bound": "hbm", "achieved": alg / kavg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / kavg_s / 1e9 / HBM_PEAK_GBS,
                            "traffic": None, "kernel": ("soc_lray_pass (rays on brick-local hierarchies: soc_lbrick_walk<RAY> + soc_sca_events) (+ soc_brick_scan, "
                                                        "soc_brick_scatter), %d passes" % passes) if form == 3 else "soc_sca_kernel (direct)",
                            "kernel_ms": kavg_s * 1e3, "algorithmic_bytes_per_launch": alg}}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline_c4(work, eng, shapes, step, ps, emit, (OD, RA, DE, N), args.cpu_budget)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "packets/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def cpu_baseline_c4(work, eng, shapes, step, ps, emit, view, budget_s):
    """the C restatement of kernel_ASOC_sca.c (oracle/, pinned bit-exactly on x86 builds of the reference at test sizes) on a strided sample of
    the background launch's work items, all host cores"""
    from oracle.pyoracle import Job, Oracle, ScaView, oracle_sim_sca
    OD, RA, DE, N = view
    cloud = work["cloud"]
    f, kind, s = step(0)
    sh = shapes["bg"]
    ncores = host_cores()
    orc = Oracle("libm")
    v = ScaView(OD, RA, DE, NPIX=(256, 256), MAP_DX=N / 256.0 * 1.5, CENTRE=(N / 2, N / 2, N / 2), FFS=1)
    job = Job(cloud, s["CSC"], ABS=s["ABS"], SCA=s["SCA"], SOURCE=1, BATCH=sh["BATCH"], SEED=launch.launch_seed(work["SEED"], f), BG=1.0,
              GLOBAL=sh["GLOBAL"], DSC=s["DSC"])
    nitems = 8 * 6 * N * N
    cal = max(512, nitems // (16 * ncores))
    t0 = time.time()
    oracle_sim_sca(orc, job, v, 0, 0, nitems, nthreads=ncores, stride=cal)
    t_cal = max(time.time() - t0, 1e-3)
    stride = max(1, int(nitems / max(((nitems + cal - 1) // cal) / t_cal * budget_s, 1)))
    t0 = time.time()
    oracle_sim_sca(orc, job, v, 0, 0, nitems, nthreads=ncores, stride=stride)
    dt = time.time() - t0
    n = ((nitems + stride - 1) // stride) * sh["BATCH"]
    return dict(value=n / dt, unit="packets/s", cores=ncores, kind="port",
                sample="background launch: every %d-th of %d work items (%d packets) in %.1f s; C restatement of kernel_ASOC_sca.c, %d threads" % (stride, nitems, n, dt, ncores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong", "launches", "auto"], default="auto",
                    help="auto = weak: per-GPU work fixed, every rank the K steps with per-rank seeds and weight 1/N (the reference's multi-device "
                         "design); strong = work-item ranges of every launch (total work fixed, identical result to one GPU); launches = "
                         "every rank a contiguous share of the launch sequence (whole launches + a work-item range at either end)")
    ap.add_argument("--int-groups", type=int, default=1, help="C3INT: frequencies (each with its own INT tally) per sweep; soc_amd.asoc runs absorbed-file runs on hierarchies with 1 (more was measured slower: the brick queues are per frequency)")
    ap.add_argument("--separate-kinds", action="store_true", help="point-source and diffuse launches in sweeps of their own (as in round 2)")
    ap.add_argument("--first-step", type=int, default=-1,
                    help="index of the first timed step (default: the number of warm-up steps, i.e. the steps follow the warm-up); "
                         "tools/profile_bench.sh profiles the driver's timed steps without running its warm-up: --warmup 0 --first-step 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-shape", action="store_true")
    ap.add_argument("--no-per-kind", action="store_true", help="skip the per-kind sweeps after the timed region")
    ap.add_argument("--cpu-budget", type=float, default=16.0)
    ap.add_argument("--in-flight", type=int, default=0,
                    help="launches executed together in one brick sweep (soc_batch_begin/end); 1 = one launch at a time; "
                         "0 = all steps in one sweep (at most 128 launches)")
    ap.add_argument("--workload", choices=["C2", "C3", "C3INT", "C4"], default="C3",
                    help="C3 = BASELINE.json configs[2] (the largest single-GPU configuration; default); C2 = configs[1]; C3INT = C3 "
                         "without `noabsorbed`: the per-frequency absorptions INT are kept (the input of config 5) -- the two launches "
                         "of a frequency are one sweep with one INT tally, read back after it (ASOC.py:1482-1498); C4 = configs[3] on one GPU: the "
                         "scattered-light kernels on the config-3 hierarchy, the steps (launches) of a run deferred into one sweep of rays")
    ap.add_argument("--global", dest="global0", type=int, default=16777216,
                    help="C3: GLOBAL_0, work items of the point-source and diffuse launches (ini key `global`; reference: 32768)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if args.scaling == "auto":
        args.scaling = "weak"

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

    if args.workload == "C4":
        return run_c4(args, world, rank, local_rank, dist, torch)
    work = c3_workload(args.global0) if args.workload in ("C3", "C3INT") else c2_workload()
    keep_int = (args.workload == "C3INT")
    if keep_int:
        work["name"] = work["name"].replace("noabsorbed;", "absorbed file (per-frequency INT tally kept, read back after the two launches of a frequency);")
    cloud = work["cloud"]
    eng = Engine(local_rank)
    eng.set_cloud(cloud)
    eng.set_features(with_int=1 if keep_int else 0, ps_method=0, use_emweight=0)      # noabsorbed: TABS only

    tabs = None
    stream = None
    if world > 1:
        # the tally lives in a torch tensor so RCCL reduces it in place; the kernels run on a torch stream that is
        # made current, so the collective is ordered behind them (and the next sweep behind the collective)
        tabs = torch.zeros(cloud.CELLS, dtype=torch.float32, device="cuda")
        eng.bind_tally(0, tabs.data_ptr())
        if not os.environ.get("SOC_BENCH_REHEARSE_ON_ONE_GPU"):
            stream = torch.cuda.Stream()
            torch.cuda.set_stream(stream)
            eng.set_stream(stream.cuda_stream)
    eng.zero(0)

    weak = args.scaling == "weak"
    mixed_sweeps = (args.workload in ("C3", "C3INT")) and not args.separate_kinds     # the config-3 hierarchy: brick-local walk, kinds share sweeps
    inflight = [0]                                                   # work items of this rank in its first sweep
    KDEV = 1.0 / world if weak else 1.0                              # ASOC.py:180,1501

    def run_steps(i0, n, in_flight, kinds=None):
        """Steps i0 .. i0+n-1 (optionally those of some kinds only) in sweeps of at most `in_flight` launches (0: as many as a
        sweep takes, 128).  On the config-3 hierarchy the kinds share sweeps; elsewhere a sweep holds one kind, so the launches
        are ordered by kind.  Several ranks: every launch is split by work items (identical streams to one GPU), or -- --scaling
        launches -- every rank takes a contiguous share of the launch sequence; ONE all-reduce of TABS after the last sweep.
        Returns the HIP-event time of the kernels [ms]."""
        dev_id, ndev = (rank, world) if weak else (0, 1)
        todo = [i for i in range(i0, i0 + n) if kinds is None or work["step"](i)["kind"] in kinds]
        if not mixed_sweeps:
            todo.sort(key=lambda i: work["kinds"].index(work["step"](i)["kind"]))
        if weak or world == 1:
            parts = [(0, work["step"](i)["L"]["GLOBAL"]) for i in todo]
        elif args.scaling == "launches":
            parts = launch.shard_launches([work["step"](i)["L"]["GLOBAL"] for i in todo], [work["step"](i)["L"]["PACKETS"] for i in todo], rank, world)
        else:
            parts = [launch.shard_range(work["step"](i)["L"]["GLOBAL"], rank, world) for i in todo]
        mine = [(i, f, c) for i, (f, c) in zip(todo, parts) if c > 0]
        cap = in_flight if in_flight > 0 else 128
        chunks = []
        for m in mine:
            new = (not chunks or len(chunks[-1]) >= cap
                   or (args.separate_kinds and work["step"](chunks[-1][-1][0])["kind"] != work["step"](m[0])["kind"])
                   or (keep_int and work["step"](chunks[-1][-1][0])["IFREQ"] != work["step"](m[0])["IFREQ"]))      # one INT tally per frequency
            if new:
                chunks.append([])
            chunks[-1].append(m)
        eng.timer_start()
        open_groups = 0                                             # C3INT: frequencies (INT tallies) deferred into the current sweep
        for ic, chunk in enumerate(chunks):
            if keep_int:
                # the launches of one frequency share an INT tally; up to --int-groups frequencies share a sweep
                if open_groups == 0:
                    eng.batch_begin_int_groups(args.int_groups)
                eng.batch_next_int()
                open_groups += 1
            else:
                eng.batch_begin(len(chunk))
            for i, first, count in chunk:
                s = work["step"](i)
                L = s["L"]
                seed = launch.launch_seed(work["SEED"], s["IFREQ"], DEVICES=ndev, ID=dev_id)
                eng.set_optical(s["ABS"], s["SCA"])
                eng.set_scatter_table(s["DSC"], s["CSC"])
                if s["kind"] == "cl":
                    eng.sim_cl(2, L["PACKETS"], L["BATCH"], seed, np.float32(s["TW"] * KDEV), L["GLOBAL"], gid_first=first, gid_count=count)
                elif s["kind"] == "ps":
                    eng.sim_pb(0, L["PACKETS"], L["BATCH"], seed, 0.0, s["TW"], PSPOS=s["PSPOS"], PS=s["PS"] * np.float32(KDEV),
                               GLOBAL=L["GLOBAL"], gid_first=first, gid_count=count)
                else:
                    eng.sim_pb(1, L["PACKETS"], L["BATCH"], seed, np.float32(s["BG"] * KDEV), s["TW"],
                               GLOBAL=L["GLOBAL"], gid_first=first, gid_count=count)
            if keep_int:
                if open_groups >= args.int_groups or ic == len(chunks) - 1:
                    eng.batch_end()
                    for k in range(open_groups):
                        eng.batch_read_int(k)                       # a frequency's column of the absorbed file (device -> host, as ASOC.py:1482)
                    open_groups = 0
            else:
                eng.batch_end()
        if world > 1:
            # TABS integrates over frequency on the device (ASOC.py:1533): one all-reduce per source block, here per call.
            # The local tally is the rank's share only, so the sum over ranks is the one-GPU tally (to summation order).
            if stream is None:
                eng.sync()
            dist.all_reduce(tabs)
        inflight[0] = max(inflight[0], max([sum(c for _, _, c in ch) for ch in chunks] or [0]))
        return eng.timer_stop()

    def fence():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        else:
            eng.sync()

    if "cl" in work["kinds"]:
        eng.set_emission(work["step"](1)["EMIT"], None)           # resident before the timed region (same for every frequency)
    if args.warmup:
        run_steps(0, args.warmup, args.in_flight)
    eng.stats(reset=True)
    fence()
    t0 = time.perf_counter()
    first_step = args.first_step if args.first_step >= 0 else args.warmup
    kernel_ms = run_steps(first_step, args.steps, args.in_flight)
    fence()
    elapsed = time.perf_counter() - t0
    st = eng.stats()
    passes = eng.last_passes()
    form = eng.last_form()                       # of the timed sweeps (the launch-shape measurement below is a direct kernel)

    packets_rank = st["packets"]
    events_rank = st["tally_events"]
    scat_rank = st["scatterings"]
    freqs = sorted({int(work["step"](i)["IFREQ"]) for i in range(first_step, first_step + args.steps)})
    # per-kind rates: the kinds share the timed sweep, so each kind's launches of the first 8 timed steps are run again in a
    # sweep of their own (outside the timed region; fewer launches per sweep than the timed one)
    per_kind = None
    if rank == 0 and world == 1 and len(work["kinds"]) > 1 and not args.no_per_kind and not keep_int:
        per_kind = {}
        for kd in work["kinds"]:
            eng.stats(reset=True)
            ms = run_steps(first_step, min(args.steps, 8), 0, kinds=(kd,))
            eng.sync()
            sk = eng.stats()
            if sk["packets"]:
                per_kind[kd] = {"packets_per_s": sk["packets"] / ms * 1e3, "tally_events_per_packet": sk["tally_events"] / sk["packets"],
                                "scatterings_per_packet": sk["scatterings"] / sk["packets"], "launches_in_the_sweep": min(args.steps, 8) // 2}
    if world > 1:
        dev = "cpu" if os.environ.get("SOC_BENCH_REHEARSE_ON_ONE_GPU") else "cuda"
        t = torch.tensor([elapsed, float(packets_rank), float(events_rank)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        packets_total = int(t[1])
    else:
        packets_total = packets_rank

    # the reference's own launch shape (GLOBAL_0 = 32768), shortened: BATCH cut so that it takes about a second
    ref_shape = None
    if rank == 0 and world == 1 and args.workload == "C3" and not args.no_reference_shape:
        try:
            s = work["step"](0)
            L = launch.ps_launch(32768 * 2000, 1, C3_GL, launch.GLOBAL_0)
            eng.stats(reset=True)
            eng.set_optical(s["ABS"], s["SCA"])
            eng.set_scatter_table(s["DSC"], s["CSC"])
            eng.timer_start()
            eng.sim_pb(0, L["PACKETS"], L["BATCH"], launch.launch_seed(work["SEED"], 0), 0.0, s["TW"], PSPOS=s["PSPOS"], PS=s["PS"], GLOBAL=L["GLOBAL"])
            ms = eng.timer_stop()
            ref_shape = {"packets_per_s": eng.stats()["packets"] / ms * 1e3,
                         "what": "point-source launch with the reference's GLOBAL_0 = 32768 work items, BATCH cut from 30517 to %d "
                                 "(one launch, not deferred)" % L["BATCH"]}
        except Exception as e:
            ref_shape = {"packets_per_s": None, "what": "failed: %s" % e}

    if rank == 0:
        kavg_s = kernel_ms * 1e-3 / max(args.steps, 1)
        alg_bytes = events_rank / max(args.steps, 1) * (BYTES_PER_TALLY_EVENT + (8 if keep_int else 0))
        achieved = alg_bytes / kavg_s / 1e9
        kname = {3: "soc_lbrick_pass<%s> (brick-local hierarchies: soc_lbrick_walk + soc_brick_events)" % ("TABS + INT" if keep_int else "TABS-only"),
                 2: "soc_brick_pass<octree,scalar-opacity,TABS-only>", 1: "soc_brick_pass<Cartesian,scalar-opacity,TABS-only>"}.get(form, "soc_brick_pass")
        kernel_name = ("%s (+ soc_brick_scan, soc_brick_scatter), %d passes in the last sweep: "
                       "time is the HIP-event span of all kernels of the K steps / K" % (kname, passes)) if passes else "soc_sim_pb_kernel / soc_sim_cl_kernel (direct)"
        out = {
            "metric": "photon packets/sec",
            "value": packets_total / elapsed,
            "unit": "packets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": work["name"], "packets_per_step_per_gpu": packets_rank // max(args.steps, 1),
                       "cells": cloud.CELLS, "tally_events_per_packet": events_rank / max(packets_rank, 1),
                       "scatterings_per_packet": scat_rank / max(packets_rank, 1),
                       "frequencies_in_the_timed_steps": freqs,
                       "launches_per_sweep": ("the launches of %d frequenc%s, an INT tally per frequency (soc_batch_begin_int_groups)" % (args.int_groups, "y" if args.int_groups == 1 else "ies")) if keep_int else (
                           args.in_flight if args.in_flight else "all steps in one sweep (point-source and diffuse launches together; at most 128)"),
                       "work_items_in_flight_rank0": inflight[0],
                       "parallelism": "1 process per GPU; %s" % (
                           "replicas with per-rank seeds, weight 1/N + 1 RCCL all-reduce of TABS" if weak
                           else ("a contiguous share of the launch sequence per rank" if args.scaling == "launches" else "work-item ranges of every launch")
                           + " + 1 RCCL all-reduce of TABS after the last sweep (TABS integrates over frequency on the device)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload),
                         "kernel": kernel_name,
                         "kernel_ms": kavg_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if ref_shape is not None:
            out["config"]["reference_launch_shape"] = ref_shape
        if per_kind:
            out["config"]["per_kind"] = per_kind
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
