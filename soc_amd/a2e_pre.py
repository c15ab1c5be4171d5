"""Solver files for the stochastic-heating solver: what the reference's A2E_pre.py writes (A2E_pre.py:31-296).

    python -m soc_amd.a2e_pre <gs-dustname> <frequencyfile> <solver-file> [NE]

The per-size arrays that A2E_pre.py gets from its OpenCL kernels -- integration weights Iw with their index arrays L1, L2
(PrepareIntegrationWeightsTrapezoid) and the cooling rates Tdown (PrepareTdown) -- come from the HIP library
(``Engine.a2e_pre`` -> ``soc_a2e_pre``); the rest (temperature grid, SKABS, EA, Ibeg, file layout) is host arithmetic, as there.

The dust model is any object with the members A2E_pre.py uses of DustLib's ``GSETDust``: NSIZE, SIZE_A, GRAIN_DENSITY,
CRT_SFRAC (grain fractions x GRAIN_DENSITY), TMIN, TMAX, SKabs_Int(isize, freq), T2E(isize, T), E2T(isize, E).  The
command line reads a GSET dust file with DustLib itself (it must be importable: DustLib.py is numpy/scipy only and is not
part of this package); ``AnalyticDust`` is a small closed-form model for tests and synthetic runs -- the reference ships no
GSET dust files.
"""
import sys

import numpy as np

from . import launch

PLANCK = 6.62606957e-27          # ASOC_aux.py:28 (A2E_pre.py computes Ef = PLANCK*FREQ with it)
C_LIGHT = 2.99792458e10
H_K = 4.79924335e-11
BOLTZMANN = 1.3806488e-16


class AnalyticDust:
    """Spherical grains with Q_abs = min(1, (a / a0) (nu / nu0)^beta) and a Debye-like enthalpy
    E(T) = N_atoms k T_D f(T / T_D), f(x) = x^4 / (x^3 + 1) (E ~ T^4 at low, ~ T at high temperature),
    N_atoms = (4/3 pi a^3) rho / (mu m_H).  Sizes follow a power law dn/da ~ a^-3.5 on a logarithmic grid."""

    def __init__(self, NSIZE=4, amin=4.0e-8, amax=3.0e-6, grain_density=1.0e-10, beta=1.7, nu0=3.0e15, a0=1.0e-5,
                 rho=2.5, mu=20.0, TD=450.0, tmin=4.0, tmax=(2500.0, 40.0)):
        self.NSIZE = int(NSIZE)
        self.SIZE_A = np.logspace(np.log10(amin), np.log10(amax), self.NSIZE)
        w = self.SIZE_A ** -2.5                                   # dn/dlog a
        self.GRAIN_DENSITY = float(grain_density)
        self.CRT_SFRAC = self.GRAIN_DENSITY * w / w.sum()
        self.beta, self.nu0, self.a0 = beta, nu0, a0
        self.TD = TD
        self.NATOM = (4.0 / 3.0) * np.pi * self.SIZE_A ** 3 * rho / (mu * 1.6726e-24)
        self.TMIN = np.full(self.NSIZE, float(tmin))
        # small grains reach high temperatures after a single photon, big ones stay near equilibrium
        self.TMAX = np.exp(np.linspace(np.log(tmax[0]), np.log(tmax[1]), self.NSIZE)) if self.NSIZE > 1 else np.asarray([tmax[0]])
        self._Tgrid = np.logspace(-1, 4.3, 4000)

    def _enthalpy(self, isize, T):
        x = np.asarray(T, np.float64) / self.TD
        return self.NATOM[isize] * BOLTZMANN * self.TD * x ** 4 / (x ** 3 + 1.0)

    def T2E(self, isize, T):
        return self._enthalpy(isize, T)

    def E2T(self, isize, E):
        return np.interp(np.asarray(E, np.float64), self._enthalpy(isize, self._Tgrid), self._Tgrid)

    def SKabs(self, isize, freq):
        """pi a^2 Q_abs of one grain"""
        a = self.SIZE_A[isize]
        Q = np.minimum(1.0, (a / self.a0) * (np.asarray(freq, np.float64) / self.nu0) ** self.beta)
        return np.pi * a * a * Q

    def SKabs_Int(self, isize, freq):
        """pi a^2 Q_abs x CRT_SFRAC (grain density included), as DustLib's GSETDust.SKabs_Int"""
        return self.SKabs(isize, freq) * self.CRT_SFRAC[isize]


def planck_intensity(f, T):
    """A2E_pre.py:59-62"""
    with np.errstate(over='ignore'):                      # exp -> inf far in the Wien tail: 0 emission, as there
        return (2.0 * PLANCK * (f / C_LIGHT) ** 2.0 * f) / (np.exp(H_K * f / T) - 1.0)


def make_solver(dust, FREQ, NE, engine):
    """The content of a solver file (dict in the layout of ``synth.write_solver`` / ``files.read_solver``) for a dust model on
    the frequency grid FREQ with NE enthalpy bins per size: A2E_pre.py:31-290 with its kernels replaced by engine.a2e_pre."""
    FREQ = np.asarray(FREQ, np.float32)
    NFREQ, NSIZE, NEPO = len(FREQ), int(dust.NSIZE), NE + 1
    Ef = np.asarray(PLANCK * FREQ, np.float32)                                  # :38, :156
    CRT_SFRAC = np.clip(np.asarray(dust.CRT_SFRAC, np.float64), 1.0e-25, 1.0e30)  # :50
    SKABS = np.zeros((NSIZE, NFREQ), np.float64)
    for isize in range(NSIZE):
        SKABS[isize, :] = dust.SKabs_Int(isize, FREQ)                          # :77-79
    FACTOR_f, _ = launch.kernel_literals(1.0)                                   # -D FACTOR=%.4ef (:134)
    sizes, tgrid = [], []
    EA = np.zeros((NFREQ, NE), np.float32)
    Ibeg = np.zeros(NFREQ, np.int32)
    for isize in range(NSIZE):
        TMIN, TMAX = float(dust.TMIN[isize]), float(dust.TMAX[isize])
        T = TMIN + (TMAX - TMIN) * (np.arange(NEPO) / (NEPO - 1.0)) ** 2.0     # :206-207
        E = np.asarray(dust.T2E(isize, T), np.float64)
        SK1 = np.asarray(SKABS[isize, :] / CRT_SFRAC[isize], np.float32)        # per grain (:220)
        E_f, T_f = np.asarray(E, np.float32), np.asarray(T, np.float32)
        tgrid.append((T_f, E_f))
        k = engine.a2e_pre(FREQ, Ef, SK1, E_f, T_f, FACTOR_f)                  # :233-256
        TC = dust.E2T(isize, 0.5 * (E[0:NE] + E[1:]))                          # temperatures at the bin centres (:271)
        for iE in range(NE):
            EA[:, iE] = SKABS[isize, :] * (planck_intensity(np.asarray(FREQ, np.float64), TC[iE]) / (PLANCK * FREQ))
        EA *= launch.FACTOR * 4.0 * np.pi
        for ifreq in range(NFREQ):                                              # :277-281
            startind = 1
            while (0.5 * (E[startind - 1] + E[startind]) < Ef[ifreq]) and (startind < NEPO - 1):
                startind += 1
            Ibeg[ifreq] = startind
        sizes.append(dict(Iw=k["Iw"], L1=k["L1"], L2=k["L2"], Tdown=k["Tdown"], EA=np.array(EA, np.float32).reshape(-1),
                          Ibeg=Ibeg.copy()))
        EA = np.zeros((NFREQ, NE), np.float32)
    return dict(NFREQ=NFREQ, FREQ=FREQ, GD=np.float32(dust.GRAIN_DENSITY), NSIZE=NSIZE,
                SIZE_A=np.asarray(dust.SIZE_A, np.float32), S_FRAC=np.asarray(CRT_SFRAC / dust.GRAIN_DENSITY, np.float32),
                NE=NE, SK_ABS=np.asarray(SKABS, np.float32), sizes=sizes, tgrid=tgrid)


def write_tgrid(filename, sol):
    """<solver>.tgrid: int32 NSIZE, NE+1; per size float32 T[NE+1], E[NE+1] (A2E_pre.py:196-199, :225-227)"""
    with open(filename, "wb") as fp:
        np.asarray([sol["NSIZE"], sol["NE"] + 1], np.int32).tofile(fp)
        for T, E in sol["tgrid"]:
            np.asarray(T, np.float32).tofile(fp)
            np.asarray(E, np.float32).tofile(fp)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 3:
        print("Usage:   python -m soc_amd.a2e_pre <gs-dustname> <frequencyfile> <solver-data-file> [NE]")
        return 1
    try:
        from DustLib import GSETDust                       # the reference's dust tooling, if the user has it on PYTHONPATH
    except ImportError:
        print("reading a GSET dust file needs the reference's DustLib.py on PYTHONPATH (or build the model in Python: "
              "soc_amd.a2e_pre.make_solver takes any object with GSETDust's members)")
        return 2
    from . import synth
    from .lib import Engine
    dust = GSETDust(argv[0])
    FREQ = np.asarray(np.loadtxt(argv[1]), np.float32)
    NE = int(argv[3]) if len(argv) > 3 else 256
    eng = Engine(0)
    try:
        sol = make_solver(dust, FREQ, NE, eng)
    finally:
        eng.close()
    synth.write_solver(argv[2], sol)
    write_tgrid('%s.tgrid' % argv[2].replace('.solver', ''), sol)
    return 0


if __name__ == "__main__":
    sys.exit(main())
