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


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 4:
        print("Usage:  python -m soc_amd.a2e  solver absorbed emitted [GPU [NSTOCH [IFREQ]]]")
        return 1
    from .lib import Engine
    NSTOCH = int(argv[5]) if len(argv) > 5 else 999
    IFREQ = int(argv[6]) if len(argv) > 6 else -1
    t0 = time.time()
    sol = files.read_solver(argv[1])
    ABSORBED = files.read_absorbed(argv[2])
    eng = Engine(0)
    try:
        EMITTED, tker = run(eng, sol, ABSORBED, NSTOCH, IFREQ)
    finally:
        eng.close()
    files.write_emitted(argv[3], EMITTED)
    DT = time.time() - t0
    print('@@  a2e %.3f SECONDS   (solver calls %.3f s)' % (DT, tker))
    print('  %4d  -- %.3e SECONDS PER CELL  -- %8.3f CELLS PER SECOND' % (len(EMITTED), DT / len(EMITTED), len(EMITTED) / DT))
    return 0


if __name__ == "__main__":
    sys.exit(main())
