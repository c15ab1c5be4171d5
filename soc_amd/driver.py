#!/usr/bin/env python3
"""driver -- the three stages of a SOC run with several dust components, in one process and in memory:

    python -m soc_amd.driver soc.ini [--keep-files]

Counterpart of ``ASOC_driver.py soc.ini`` + ``A2E_MABU.py`` (reference ASOC_driver.py:200-473, A2E_MABU.py:236-342,
700-1180): the reference chains three programs with os.system and hands the absorptions (CELLS x NFREQ floats: 5-17 GB
at config 3) and the emission over through files.  Here

  1. the radiative-transfer run (soc_amd.asoc.AbsorptionRun with the simple dusts, absorptions kept per frequency,
     `nomap`, `nosolve`: what rt_simple.ini asks for, ASOC_driver.py:230-252) returns FABSORBED[CELLS, NFREQ];
  2. for every dust component the absorptions are split in proportion to cross section x abundance
     (split_absorbed, kernel_A2E_MABU_aux.c:3-23; RABS as A2E_MABU.py:245-342) and the emission is solved -- an
     equilibrium component with soc_eqsolver (SolveEquilibriumDust, A2E_MABU.py:436-640 -> kernel_eqsolver.c), a
     stochastically heated one with soc_amd.a2e.run on its <dust>.solver file (A2E.py) -- and summed weighted by the
     abundances (A2E_MABU.py:1128-1140);
  3. the maps are written from that emission array (AbsorptionRun.write_maps: maps.ini of ASOC_driver.py:447-473).

Same ini keys, same dust / solver / abundance / cloud files.  `emitted` is written (it is a product); the
`absorbed` file only with --keep-files.  The neural-network and library shortcuts (nnmake, nnsolve, libabs ...),
polarisation (aalg) and cosmic-ray heating are not part of this path and are refused.  With several ranks the
first stage shards work items (soc_amd.dist), the second the cells; rank 0 writes.
"""
import os
import sys
import time

import numpy as np
from scipy.interpolate import interp1d

from . import a2e, files
from .asoc import AbsorptionRun, UnsupportedOption
from .ini import User
from .launch import FACTOR

NE_EQ = 30000                    # A2E_MABU.py:478


def dust_kind(name):
    """'gsetdust' (stochastically heated: needs <name>.solver) or 'eqdust' (ASOC_driver.py:66-68)"""
    with open(name) as fp:
        return fp.readline().split()[0]


def simple_name(name):
    """the dust file the transfer run uses for a gsetdust (ASOC_driver.py:247-248: prefix gs_ dropped, _simple added)"""
    d, b = os.path.split(name)
    return os.path.join(d, '%s_simple.dust' % b.replace('.dust', '').replace('gs_', ''))


def solver_name(name):
    """ASOC_driver.py:199-201 / A2E_MABU.py:264"""
    d, b = os.path.split(name)
    b = b.replace('.dust', '')
    if b.startswith('gs_'):
        b = b[3:]
    return os.path.join(d, b + '.solver')


def planck_safe(f, T):
    """A2E_MABU.py PlanckSafe: 2h f^3 / c^2 / (exp(hf/kT) - 1), overflow-safe"""
    H_K, H_CC = 4.79924335e-11, 7.372496678e-48
    return 2.0 * H_CC * f * f * f / (np.exp(np.clip(H_K * f / T, -100.0, 100.0)) - 1.0)


def eq_dust_table(dust):
    """E -> T table of an equilibrium dust (A2E_MABU.py:470-490): FREQ, KABS per unit density, Emin, kE, oplgkE, TTT[NE]"""
    with open(dust) as fp:
        lines = fp.readlines()
    gd, gr = float(lines[1].split()[0]), float(lines[2].split()[0])
    d = np.loadtxt(dust, skiprows=4, ndmin=2)
    FREQ = np.asarray(d[:, 0], np.float32)
    KABS = np.asarray(d[:, 2] * gd * np.pi * gr ** 2.0, np.float32)
    TSTEP = 1600.0 / NE_EQ
    TT = 1.0 + TSTEP * np.arange(NE_EQ)
    F64 = np.asarray(FREQ, np.float64)
    DF = FREQ[2:] - FREQ[:-2]
    Eout = np.zeros(NE_EQ, np.float64)
    # vectorised over the temperatures (the reference loops): same sums, frequency by frequency
    B = KABS[None, :] * planck_safe(F64[None, :], TT[:, None])
    res = B[:, 0] * (FREQ[1] - FREQ[0]) + B[:, -1] * (FREQ[-1] - FREQ[-2]) + np.sum(B[:, 1:-1] * DF[None, :], axis=1)
    Eout[:] = (4.0 * np.pi * FACTOR) * 0.5 * res
    Emin, Emax = Eout[0], Eout[NE_EQ - 1] * 0.9999
    kE = (Emax / Emin) ** (1.0 / (NE_EQ - 1.0))
    oplgkE = 1.0 / np.log10(kE)
    TTT = np.asarray(interp1d(Eout, TT)(Emin * kE ** np.arange(NE_EQ)), np.float32)
    return FREQ, KABS, Emin, kE, oplgkE, TTT


def relative_cross_sections(dusts, kinds):
    """RABS[NFREQ, NDUST] (A2E_MABU.py:245-342): absorption cross section per unit density of every component --
    from the dust file (eqdust) or summed over the sizes of the solver file -- normalised per frequency, float64"""
    cols, FREQ = [], None
    for name, kind in zip(dusts, kinds):
        if kind == 'eqdust':
            with open(name) as fp:
                lines = fp.readlines()
            gd, radius = float(lines[1].split()[0]), float(lines[2].split()[0])
            d = np.loadtxt(name, skiprows=4, ndmin=2)
            FREQ = d[:, 0]
            cols.append(np.pi * radius ** 2.0 * gd * d[:, 2])
        else:
            sol = files.read_solver(solver_name(name))
            FREQ = np.asarray(sol["FREQ"], np.float64)
            cols.append(np.sum(np.asarray(sol["SK_ABS"], np.float64), axis=0))
    RABS = np.clip(np.asarray(cols, np.float64).T, 1.0e-40, 1.0e30)
    RABS /= (1.0e-40 + RABS.sum(axis=1))[:, None]
    return np.clip(RABS, 1.0e-30, 1.0), FREQ


def split_absorbed(ABSORBED, RABS, ABU, idust):
    """kernel_A2E_MABU_aux.c:3-23: OUT[c, f] = IN[c, f] * RABS[f, idust] / sum_j ABU[c, j] * RABS[f, j], with the kernel's
    types: the denominator accumulates in float (each product formed in double), the quotient is taken in double"""
    cells, nfreq = ABSORBED.shape
    ndust = RABS.shape[1]
    den = np.zeros((cells, nfreq), np.float32)
    for j in range(ndust):
        den = (den.astype(np.float64) + ABU[:, j:j + 1].astype(np.float64) * RABS[None, :, j]).astype(np.float32)
    return (ABSORBED.astype(np.float64) * RABS[None, :, idust] / den.astype(np.float64)).astype(np.float32)


class Pipeline:
    """soc.ini -> maps.  engine: soc_amd.lib.Engine (or an object with its methods); comm: soc_amd.dist.Comm"""

    def __init__(self, ini, engine, comm=None, verbose=None):
        self.comm = comm
        self.rank = comm.rank if comm else 0
        self.world = comm.world if comm else 1
        self.eng = engine
        U = User(ini)
        self.refuse(U)
        self.U = U
        # dust list as the user wrote it, and the simple dusts the transfer run works with (ASOC_driver.py:240-250)
        self.dusts = list(U.file_optical)
        self.kinds = [dust_kind(d) for d in self.dusts]
        U.file_optical = [d if k != 'gsetdust' else simple_name(d) for d, k in zip(self.dusts, self.kinds)]
        self.want_maps = not U.NOMAP
        self.want_solve = True
        # what stage 2 needs, checked before any GPU work (the transfer run of a large model takes minutes): a solver file per
        # stochastically heated dust -- soc_amd.a2e_pre writes them (ASOC_driver.py:196-228 calls A2E_pre.py there) -- and no dust
        # re-emission iterations, which this in-memory pipeline does not do
        if U.ITERATIONS > 0 and U.CLPAC > 0:
            raise UnsupportedOption("cellpackets (dust re-emission iterations) inside the pipeline: run soc_amd.asoc per iteration")
        for d, k in zip(self.dusts, self.kinds):
            if k == 'gsetdust' and not os.path.exists(solver_name(d)):
                raise FileNotFoundError("%s: the solver file of %s is missing; write it with python -m soc_amd.a2e_pre %s <frequency file> %s"
                                        % (solver_name(d), d, d, solver_name(d)))
        U.NOABSORBED, U.NOMAP, U.NOSOLVE = 0, 1, 1            # rt_simple.ini: absorptions per frequency, nomap, nosolve
        self.verbose = U.VERBOSE if verbose is None else verbose
        self.timers = {}

    @staticmethod
    def refuse(U):
        bad = [k for k in ('nnmake', 'nnsolve', 'nnabs', 'nnemit', 'nnthin', 'absthin', 'libabs', 'libmaps', 'aalg', 'crheating')
               if k in U.KEYS]
        if bad:
            raise UnsupportedOption("ini options outside the in-memory pipeline (neural-network / library shortcuts, "
                                    "polarisation, cosmic-ray heating): " + ", ".join(bad))

    def log(self, *a):
        if self.verbose and self.rank == 0:
            print(*a)

    # ---- stage 2 ----------------------------------------------------------------------------------------------
    def solve_emission(self, FABSORBED, ABU):
        """FABSORBED[CELLS, NFREQ] as the absorbed file holds it (scaled, files.scale_absorbed) -> EMITTED[CELLS, NFREQ]"""
        CELLS, NFREQ = FABSORBED.shape
        NDUST = len(self.dusts)
        RABS, FREQ = relative_cross_sections(self.dusts, self.kinds)
        if RABS.shape[0] != NFREQ:
            raise ValueError("the dusts have %d frequencies, the absorptions %d" % (RABS.shape[0], NFREQ))
        c0, c1 = a2e.cell_range(CELLS, self.rank, self.world)
        EMITTED = np.zeros((CELLS, NFREQ), np.float32)
        for idust in range(NDUST):
            t0 = time.time()
            part = split_absorbed(FABSORBED[c0:c1], RABS, ABU[c0:c1], idust)
            if self.kinds[idust] == 'eqdust':
                Fq, KABS, Emin, kE, oplgkE, TTT = eq_dust_table(self.dusts[idust])
                em = np.zeros((c1 - c0, NFREQ), np.float32)
                B = 32768                                          # A2E_MABU.py:493 (any batch gives the same cells)
                for a in range(0, c1 - c0, B):
                    b = min(a + B, c1 - c0)
                    _, em[a:b] = self.eng.eqsolver(c0 + a, CELLS, NE_EQ, FACTOR, kE, oplgkE, Emin, Fq, KABS, TTT, part[a:b])
            else:
                sol = files.read_solver(solver_name(self.dusts[idust]))
                em, _ = a2e.run(self.eng, sol, part, verbose=False)
            EMITTED[c0:c1] += em * ABU[c0:c1, idust:idust + 1]     # A2E_MABU.py:1128-1140
            self.log("  dust %d/%d %-24s %s  %.2f s" % (idust + 1, NDUST, self.dusts[idust], self.kinds[idust], time.time() - t0))
        if self.comm and self.world > 1:                           # every rank solved its cells: put the array together
            for f in range(NFREQ):
                EMITTED[:, f] = self.comm.all_reduce_host(np.ascontiguousarray(EMITTED[:, f]))
        return EMITTED

    # ---- the three stages -----------------------------------------------------------------------------------------
    def run(self, keep_files=False):
        U = self.U
        t0 = time.time()
        rt = AbsorptionRun(U, self.eng, self.comm, verbose=self.verbose)
        rt.write_packet_info()
        rt.setup_engine()
        CTABS, FABSORBED = rt.simulate_constant_sources()
        if U.ITERATIONS > 0 and rt.CLPAC > 0:
            raise UnsupportedOption("cellpackets (dust re-emission iterations) inside the pipeline: run soc_amd.asoc per iteration")
        files.scale_absorbed(FABSORBED, rt.cloud, U.GL, U.NNNLIMIT)
        self.timers["transfer"] = time.time() - t0
        # abundances: the columns of the transfer run (ASOC_aux.py read_abundances), ones where no file is given
        CELLS = rt.cloud.CELLS
        ABU = np.ones((CELLS, len(self.dusts)), np.float32)
        if rt.ABU is not None:
            ABU = np.asarray(rt.ABU, np.float32).reshape(CELLS, -1) if not U.SINGLE_ABU else \
                np.stack([np.ravel(rt.ABU), 1.0 - np.ravel(rt.ABU)], axis=1).astype(np.float32)
        t0 = time.time()
        EMITTED = self.solve_emission(FABSORBED, ABU)
        self.timers["emission"] = time.time() - t0
        if self.rank == 0:
            if len(U.file_emitted) > 0:
                files.write_emitted(U.file_emitted, EMITTED)
            if keep_files and len(U.file_absorbed) > 0:
                files.write_absorbed(U.file_absorbed, FABSORBED)
        t0 = time.time()
        if self.want_maps:
            U.NOMAP = 0
            rt.write_maps(EMITTED)
        self.timers["maps"] = time.time() - t0
        self.log("@@ driver: transfer %.2f s, emission %.2f s, maps %.2f s" % (self.timers["transfer"], self.timers["emission"], self.timers["maps"]))
        return CTABS, FABSORBED, EMITTED


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        print("Usage:  python -m soc_amd.driver soc.ini [--keep-files]")
        return 1
    from .dist import Comm
    comm = Comm()                      # (imports torch first when there are several ranks: see lib.load_library)
    from .lib import Engine
    eng = Engine(comm.local_rank)
    try:
        Pipeline(argv[1], eng, comm).run(keep_files="--keep-files" in argv)
    finally:
        eng.close()
        comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
