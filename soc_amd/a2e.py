#!/usr/bin/env python3
"""a2e -- emission of stochastically heated grains on MI355X:

    python -m soc_amd.a2e  dust.solver  absorbed.data  emitted.data  [GPU [NSTOCH [IFREQ]]]

Drop-in for ``A2E.py solver absorbed emitted`` (reference A2E.py:64-66): same solver file
(written by A2E_pre.py), same absorbed/emitted files.  For every grain size the absorptions of
all cells go through the HIP ``DoSolve`` (sizes < NSTOCH) or ``EqTemperature`` (sizes >= NSTOCH)
and the emission is accumulated (A2E.py:453-600).  Differences that are the point of the
rewrite: no run-time kernel build, no 266 MB global scratch matrix (the transition matrix
lives in LDS), batches are sized by memory, not by the scratch buffer.
"""
import sys
import time

import numpy as np
from scipy.interpolate import interp1d

from . import files
from .launch import FACTOR, H_K
from .synth import a2e_absorption_fraction

NIP = 5000                     # interpolation points of the E <-> T table (A2E.py:282)


def planck_safe(f, T):
    """A2E.py:206-209"""
    H_CC20 = 7.372496678e-28
    return 2.0e-20 * ((H_CC20 * f) * f) * f / (np.exp(np.clip(H_K * f / T, -100, +100)) - 1.0)


def eq_table(sol, isize):
    """E -> T lookup table of one size on the host (A2E.py:466-488): Emin, kE, oplgkE, TTT[NIP], KABS."""
    FREQ = np.asarray(sol["FREQ"], np.float64)
    KABS = sol["SK_ABS"][isize, :] / (sol["GD"] * sol["S_FRAC"][isize])
    TSTEP = 1600.0 / NIP
    TT = 4.0 + TSTEP * np.arange(NIP)
    DF = FREQ[2:] - FREQ[:-2]
    Eout = np.zeros(NIP, np.float64)
    for i in range(NIP):
        TMP = FACTOR * KABS * planck_safe(FREQ, TT[i])
        res = TMP[0] * (FREQ[1] - FREQ[0]) + TMP[-1] * (FREQ[-1] - FREQ[-2]) + np.sum(TMP[1:-1] * DF)
        Eout[i] = 4.0 * np.pi * 0.5 * res
    Emin, Emax = Eout[0], Eout[NIP - 1] * 0.9999
    kE = (Emax / Emin) ** (1.0 / (NIP - 1.0))
    oplgkE = 1.0 / np.log10(kE)
    TTT = np.asarray(interp1d(Eout, TT)(Emin * kE ** np.arange(NIP)), np.float32)
    return Emin, kE, oplgkE, TTT, np.asarray(KABS, np.float32)


def run(engine, sol, ABSORBED, NSTOCH=999, IFREQ=-1, batch=65536, verbose=True):
    """ABSORBED[CELLS,NFREQ] -> EMITTED[CELLS,NFREQ or 1].  Returns (EMITTED, kernel_seconds)."""
    CELLS, NFREQ = ABSORBED.shape
    if NFREQ != sol["NFREQ"]:
        raise ValueError("absorbed file has %d frequencies, solver %d" % (NFREQ, sol["NFREQ"]))
    ABSORBED = np.array(ABSORBED, np.float32)        # A2E.py:184-185: clip the last channel
    ABSORBED[:, NFREQ - 1] = np.clip(ABSORBED[:, NFREQ - 1], 0.0, 0.2 * ABSORBED[:, NFREQ - 2])
    EMITTED = np.zeros((CELLS, 1 if IFREQ >= 0 else NFREQ), np.float32)
    tker = 0.0
    # The stochastically heated sizes with the cells resident in device memory (soc_a2e_resident_*): absorptions up once, every size a
    # launch over all cells that adds its emission to a sum on the device, the sum down once -- instead of NSIZE round trips of every
    # batch.  The sum is added in the order of the sizes, as the reference's host does (A2E.py:596-600), and the equilibrium sizes
    # follow it (they come last in the loop: isize >= NSTOCH), so the result is the same to the bit.
    nstoch = min(int(NSTOCH), len(sol["sizes"]), sol["NSIZE"])
    resident = False
    if nstoch > 0 and hasattr(engine, "a2e_resident_begin"):
        try:
            engine.a2e_resident_begin(CELLS, NFREQ)
            resident = True
        except Exception as err:                                  # not enough device memory: batches as before
            if verbose:
                print("    a2e: cells not resident (%s)" % err)
    if resident:
        try:
            t0 = time.time()
            for icell in range(0, CELLS, batch):
                engine.a2e_resident_upload(icell, ABSORBED[icell:min(icell + batch, CELLS), :])
            for isize in range(nstoch):
                engine.a2e_set_size(sol["NE"], NFREQ, sol["sizes"][isize], a2e_absorption_fraction(sol, isize))
                engine.a2e_resident_solve()
                if verbose:
                    print("    isize = %d   stochastic heating" % isize)
            for icell in range(0, CELLS, batch):
                b = min(icell + batch, CELLS)
                emit = engine.a2e_resident_download(icell, b - icell)
                if IFREQ >= 0:
                    EMITTED[icell:b, 0] += emit[:, IFREQ]
                else:
                    EMITTED[icell:b, :] += emit
            tker += time.time() - t0
        finally:
            engine.a2e_resident_end()
    for isize in range(sol["NSIZE"]):
        AF = a2e_absorption_fraction(sol, isize)
        if isize >= NSTOCH or isize >= len(sol["sizes"]):
            if sol["S_FRAC"][isize] < 1.0e-30:
                continue
            Emin, kE, oplgkE, TTT, KABS = eq_table(sol, isize)
            scale = sol["GD"] * sol["S_FRAC"][isize]
            for icell in range(0, CELLS, batch):
                b = min(icell + batch, CELLS)
                tmp = np.asarray(ABSORBED[icell:b, :] * AF, np.float32)
                t0 = time.time()
                T, emit = engine.a2e_eqtemp(icell, CELLS, NIP, FACTOR, kE, oplgkE, Emin, sol["FREQ"], KABS, TTT, tmp)
                tker += time.time() - t0
                if IFREQ >= 0:
                    EMITTED[icell:b, 0] += emit[:, IFREQ] * scale
                else:
                    EMITTED[icell:b, :] += emit * scale
            continue
        if resident:
            continue
        engine.a2e_set_size(sol["NE"], NFREQ, sol["sizes"][isize], AF)
        for icell in range(0, CELLS, batch):
            b = min(icell + batch, CELLS)
            t0 = time.time()
            emit = engine.a2e_solve(ABSORBED[icell:b, :])
            tker += time.time() - t0
            if IFREQ >= 0:
                EMITTED[icell:b, 0] += emit[:, IFREQ]
            else:
                EMITTED[icell:b, :] += emit
        if verbose:
            print("    isize = %d   stochastic heating" % isize)
    return EMITTED, tker


def cell_range(CELLS, rank, world):
    """cells [c0, c1) of rank `rank`: the cells of a grid are independent in A2E, so N GPUs split them"""
    per = (CELLS + world - 1) // world
    return min(rank * per, CELLS), min((rank + 1) * per, CELLS)


def run_sharded(engine_factory, solver, absorbed, emitted, NSTOCH=999, IFREQ=-1, comm=None, verbose=True):
    """The whole program for one rank of `comm` (or alone): memory-map the absorbed file, solve this rank's cells,
    write them into this rank's part of the emitted file.  No collective on the data path -- the ranks only wait
    for rank 0 to have created the file, and for each other at the end.  Returns (cells solved, kernel seconds)."""
    rank, world = (comm.rank, comm.world) if comm else (0, 1)
    sol = files.read_solver(solver)
    dims = np.fromfile(absorbed, np.int32, 2)
    CELLS, NFREQ = int(dims[0]), int(dims[1])
    ABS = np.memmap(absorbed, dtype=np.float32, mode='r', offset=8, shape=(CELLS, NFREQ))
    nout = 1 if IFREQ >= 0 else NFREQ
    if rank == 0:
        with open(emitted, 'wb') as fp:
            np.asarray([CELLS, nout], np.int32).tofile(fp)
            fp.truncate(8 + 4 * CELLS * nout)
    if comm:
        comm.barrier()
    c0, c1 = cell_range(CELLS, rank, world)
    tker = 0.0
    if c1 > c0:
        eng = engine_factory()
        try:
            EM, tker = run(eng, sol, ABS[c0:c1, :], NSTOCH, IFREQ, verbose=verbose and rank == 0)
        finally:
            if hasattr(eng, "close"):
                eng.close()
        out = np.memmap(emitted, dtype=np.float32, mode='r+', offset=8, shape=(CELLS, nout))
        out[c0:c1, :] = EM
        out.flush()
        del out
    if comm:
        comm.barrier()
    return c1 - c0, tker


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 4:
        print("Usage:  python -m soc_amd.a2e  solver absorbed emitted [GPU [NSTOCH [IFREQ]]]")
        print("        (N GPUs: python -m torch.distributed.run --nproc-per-node N -m soc_amd.a2e ...)")
        return 1
    from .lib import Engine
    from .dist import Comm
    NSTOCH = int(argv[5]) if len(argv) > 5 else 999
    IFREQ = int(argv[6]) if len(argv) > 6 else -1
    t0 = time.time()
    comm = Comm()
    n, tker = run_sharded(lambda: Engine(comm.local_rank), argv[1], argv[2], argv[3], NSTOCH, IFREQ,
                          comm if comm.world > 1 else None)
    DT = time.time() - t0
    if comm.rank == 0:
        print('@@  a2e %.3f SECONDS   (solver calls %.3f s on rank 0, %d ranks)' % (DT, tker, comm.world))
        print('  %4d  -- %.3e SECONDS PER CELL  -- %8.3f CELLS PER SECOND' % (n * comm.world, DT / max(n * comm.world, 1), n * comm.world / DT))
    comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
