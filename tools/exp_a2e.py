#!/usr/bin/env python3
"""Throughput of the A2E DoSolve kernel (config 5): batch resident on the device, kernel alone and with PCIe copies.

    python tools/exp_a2e.py [--ne 128] [--nfreq 50] [--batch 8192] [--reps 5]

Prints one JSON line: cell-sizes/s, and the roofline numbers of SURVEY.md 8(d): algorithmic HBM bytes = 8*NFREQ per
cell and size (absorptions in, emission out; weights and tables are shared by all cells and stay in cache)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soc_amd import synth                 # noqa: E402
from soc_amd.lib import Engine            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ne", type=int, default=128)
    ap.add_argument("--nfreq", type=int, default=50)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--program", type=int, nargs=2, metavar=("CELLS", "NSIZE"), default=None,
                    help="also time soc_amd.a2e.run end to end (host arrays in, host array out) for CELLS cells and NSIZE sizes: with the cells resident on the device, and in batches as the reference does")
    a = ap.parse_args()
    prof = None
    if os.environ.get("SOC_HIP_LIB"):                     # a -DSOC_A2E_PROF build (tools/build_prof.sh): cycles per phase
        import ctypes as C
        from soc_amd import lib as soclib
        L = soclib.load_library(os.environ["SOC_HIP_LIB"])
        if hasattr(L, "soc_a2e_prof_read"):
            prof = L.soc_a2e_prof_read
    eng = Engine(0)
    sol = synth.synth_solver(NFREQ=a.nfreq, NE=a.ne, NSIZE=1, seed=5)
    rng = np.random.default_rng(1)
    ABS = (rng.lognormal(0, 1, (a.batch, a.nfreq)) * 1e-3 * (sol['FREQ'][None, :] / 1e13) ** -1.0).astype(np.float32)
    AF = synth.a2e_absorption_fraction(sol, 0)
    eng.a2e_set_size(a.ne, a.nfreq, sol['sizes'][0], AF)
    eng.a2e_upload(ABS)
    ms = []
    for it in range(a.reps):
        eng.timer_start()
        eng.a2e_run(a.batch)
        ms.append(eng.timer_stop())
    if prof:
        buf = (C.c_ulonglong * 8)()
        prof(buf, 1)
        eng.a2e_run(a.batch)
        eng.sync()
        prof(buf, 0)
        p = list(buf)
        tot = max(sum(p[:5]), 1)
        print("cycles of wave 0 per workgroup phase: heating %.1f %%, suffix sums %.1f %%, substitution %.1f %%, normalisation %.1f %%, emission %.1f %%"
              % tuple(100.0 * x / tot for x in p[:5]), file=sys.stderr)
    t0 = time.time()
    out = eng.a2e_solve(ABS)
    t1 = time.time()
    k = float(np.median(ms))
    alg = 8.0 * a.nfreq * a.batch
    print(json.dumps({"kernel": "soc_a2e_dosolve_kernel", "NE": a.ne, "NFREQ": a.nfreq, "batch": a.batch, "kernel_ms": k,
                      "cell_sizes_per_s": a.batch / k * 1e3, "with_pcie_cell_sizes_per_s": a.batch / (t1 - t0),
                      "roofline": {"bound": "hbm", "algorithmic_bytes_per_launch": alg, "achieved": alg / k * 1e-6, "peak": 8000.0,
                                   "unit": "GB/s", "frac": alg / k * 1e-6 / 8000.0},
                      "note": "LDS/latency-bound: the transition matrix (NE^2/2 floats per cell) never leaves LDS",
                      "finite": bool(np.isfinite(out).all())}))
    if a.program:
        from soc_amd import a2e
        cells, nsize = a.program
        sol = synth.synth_solver(NFREQ=a.nfreq, NE=a.ne, NSIZE=nsize, seed=5)
        ABS = (rng.lognormal(0, 1, (cells, a.nfreq)) * 1e-3 * (sol['FREQ'][None, :] / 1e13) ** -1.0).astype(np.float32)

        class Batches:
            def __getattr__(self, name):
                if name.startswith("a2e_resident"):
                    raise AttributeError(name)
                return getattr(eng, name)
        res = {}
        for tag, e in (("resident", eng), ("batches", Batches())):
            t0 = time.time()
            E, _ = a2e.run(e, sol, ABS, verbose=False)
            res[tag] = (time.time() - t0, E)
        same = bool(np.array_equal(res["resident"][1].view(np.uint32), res["batches"][1].view(np.uint32)))
        print(json.dumps({"program": "soc_amd.a2e.run, host arrays in and out", "NE": a.ne, "NFREQ": a.nfreq, "cells": cells, "sizes": nsize,
                          "resident_s": res["resident"][0], "resident_cell_sizes_per_s": cells * nsize / res["resident"][0],
                          "batches_s": res["batches"][0], "batches_cell_sizes_per_s": cells * nsize / res["batches"][0],
                          "bit_identical": same}))
    eng.close()


if __name__ == "__main__":
    main()
