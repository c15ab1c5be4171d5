#!/usr/bin/env python3
"""Tuning experiments on the C3 geometry (256^3-root octree, 4 levels): steps/s of the brick sweep for a list of
soc_set_tuning settings.

    python tools/exp_c3.py [--kind ps|cl|bg|mix] [--n N] [--launches K] [--packets P] '{"brick_cells": 6144}' '{"global_tree": 1}' ...

Every setting runs K launches (different seeds/frequencies) of about P packets each in one sweep and prints
packets/s, steps/s and the passes.  Not a benchmark of record (bench.py is); used to choose the built-in values.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from soc_amd import launch, synth              # noqa: E402
from soc_amd.lib import Engine                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="ps")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--launches", type=int, default=4)
    ap.add_argument("--packets", type=float, default=2.5e8)
    ap.add_argument("--global0", type=int, default=4194304)
    ap.add_argument("--freq", type=int, default=30)
    ap.add_argument("--fstride", type=int, default=1, help="frequency of launch k: freq + k * fstride (mod 50)")
    ap.add_argument("--int", dest="with_int", type=int, default=0, help="1: launches keep the per-frequency INT tally and run one at a time (read after each); 2: the same in batches of 16 (soc_batch_begin_int: brick queues per launch, every launch its own INT tally, read after the batch)")
    ap.add_argument("tunes", nargs="*")
    a = ap.parse_args()
    t0 = time.time()
    if a.n == 256:
        work = bench.c3_workload(a.global0)
    else:
        raise SystemExit("only --n 256")
    cloud = work["cloud"]
    print("cloud %.1f s, %d cells" % (time.time() - t0, cloud.CELLS), flush=True)
    import ctypes as C
    from soc_amd import lib as soclib
    prof = None
    if os.environ.get("SOC_HIP_LIB"):                     # e.g. a -DSOC_BRICK_PROF build (tools/build_prof.sh)
        L = soclib.load_library(os.environ["SOC_HIP_LIB"])
        if hasattr(L, "soc_prof_read"):
            prof = L.soc_prof_read
    buf = (C.c_ulonglong * 24)()
    eng = Engine(0)
    eng.set_cloud(cloud)
    eng.set_features(1 if a.with_int else 0, 0, 0)
    eng.set_emission(work["step"](1)["EMIT"], None)
    for ts in (a.tunes or ["{}"]):
        tune = json.loads(ts)
        reset = {k: 0 for k in tune}
        eng.set_tuning(**dict(dict(verbose=1), **tune))
        for rep in range(2):
            eng.zero(0)
            eng.stats(reset=True)
            if prof:
                prof(buf, 1)
            eng.timer_start()
            if not a.with_int:
                eng.batch_begin(min(128, a.launches))
            nint = 0
            for k in range(a.launches):
                if a.with_int == 1:
                    eng.zero(1)
                if a.with_int == 2 and nint == 0:
                    eng.batch_begin_int(16)
                kind = a.kind if a.kind != "mix" else ("ps", "cl")[k % 2]      # mix: point-source and diffuse launches share the sweep
                f = (a.freq + (k // 2 if a.kind == "mix" else k) * a.fstride) % 50
                s = work["step_for"](f, "ps" if kind == "ps" else "cl")
                L = dict(s["L"])
                eng.set_optical(s["ABS"], s["SCA"])
                eng.set_scatter_table(s["DSC"], s["CSC"])
                seed = launch.launch_seed(work["SEED"], f)
                if kind == "cl":
                    G = int(min(L["GLOBAL"], cloud.CELLS) * min(1.0, a.packets / (L["BATCH"] * cloud.CELLS)))
                    eng.sim_cl(2, L["PACKETS"], L["BATCH"], seed, s["TW"], L["GLOBAL"], gid_first=0, gid_count=max(G, 65536))
                elif kind == "ps":
                    B = max(1, int(a.packets / L["GLOBAL"]))
                    eng.sim_pb(0, L["GLOBAL"] * B, B, seed, 0.0, s["TW"], PSPOS=s["PSPOS"], PS=s["PS"], GLOBAL=L["GLOBAL"])
                else:
                    Lb = launch.bg_launch(int(a.packets), cloud.AREA)
                    eng.sim_pb(1, Lb["PACKETS"], Lb["BATCH"], seed, 1e-3, s["TW"], GLOBAL=Lb["GLOBAL"])
                if a.with_int == 1:
                    eng.read_tally(1)
                if a.with_int == 2:
                    nint += 1
                    if nint == 16 or k == a.launches - 1:
                        eng.batch_end()
                        for q in range(nint):
                            eng.batch_read_int(q)
                        nint = 0
            if not a.with_int:
                eng.batch_end()
            ms = eng.timer_stop()
            st = eng.stats()
            print("%-60s %s rep %d: %8.1f ms  %.3e packets/s  %.3e steps/s  %.1f steps/packet  passes %d form %d" % (
                ts, a.kind, rep, ms, st["packets"] / ms * 1e3, st["tally_events"] / ms * 1e3,
                st["tally_events"] / max(st["packets"], 1), eng.last_passes(), eng.last_form()), flush=True)
            if prof:
                prof(buf, 0)
                p = list(buf)
                tt = max(sum(p[8:16]), 1)
                print("   wave-iterations %.3e: stepping lanes %.1f, idle lanes %.1f | swap arm every %.1f iterations with %.1f lanes | "
                      "wave cycles: swap %.1f %%, GetStep + tally %.1f %%, Index %.1f %%" % (
                          p[0], p[1] / max(p[0], 1), p[7] / max(p[0], 1), p[0] / max(p[4], 1), p[5] / max(p[4], 1),
                          100 * p[8] / tt, 100 * p[9] / tt, 100 * p[10] / tt), flush=True)
                if p[11] or p[12]:
                    life = max(sum(p[8:16]), 1)
                    print("   a wave's life in its workgroup: prologue %.1f %%, loop %.1f %%, wait for the workgroup's other waves %.1f %%, tallies to global memory %.1f %%, "
                          "ranks of the chunk's packets %.1f %%" % (100 * p[11] / life, 100 * (p[8] + p[9] + p[10] + p[13]) / life, 100 * p[14] / life,
                                                               100 * p[15] / life, 100 * p[12] / life), flush=True)
                if p[16] or p[18] or p[20]:
                    print("   idle lanes by where: chunks < 4 packets per lane %.1f %% of the iterations with %.1f idle lanes; < 16 per lane %.1f %% with %.1f; "
                          "end of a chunk (nothing prefetched in the wave) %.1f %% with %.1f" % (
                              100 * p[16] / max(p[0], 1), p[17] / max(p[16], 1), 100 * p[18] / max(p[0], 1), p[19] / max(p[18], 1),
                              100 * p[20] / max(p[0], 1), p[21] / max(p[20], 1)), flush=True)
        eng.set_tuning(**reset)
    eng.close()


if __name__ == "__main__":
    main()
