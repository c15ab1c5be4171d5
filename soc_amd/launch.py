"""Launch arithmetic of the absorption run: packet counts, launch sizes, packet weights,
integration weights and seeds.  These host formulas define the results as much as the
kernels do; each function cites the ASOC.py lines it restates.
"""
import math

import numpy as np

# constants, ASOC_aux.py:28-54 and ASOC.py:81,86
FACTOR = 1.0e20
C_LIGHT = 2.99792458e10
PLANCK = 6.62606957e-27
H_K = 4.79924335e-11
PARSEC = 3.08567758e18
SEED0 = 0.8150982470475214
SEED1 = 0.1393378751427912
ADHOC = 1.0
GLOBAL_0 = 32768          # reference default for point-source / cell-emission launches
LOCAL_GPU = 32


def um2f(um):
    """wavelength [um] -> frequency [Hz] (ASOC_aux.py:67-68)"""
    return C_LIGHT / (1.0e-4 * um)


def Fix(n, l):
    """smallest integer >= n divisible by l (ASOC_aux.py:1504-1506)"""
    return int(int(math.floor((n + l - 1) / l)) * l)


def packet_counts(BGPAC, PSPAC, CLPAC, DFPAC, AREA, CELLS, LOCAL=LOCAL_GPU, USE_EMWEIGHT=0):
    """Rounded packet counts (ASOC.py:234-250).  Returns dict(PSPAC, BGPAC, CLPAC, DFPAC)."""
    PS = Fix(PSPAC, LOCAL)
    BG = Fix(Fix(BGPAC, AREA), LOCAL)
    DF = 0
    if CLPAC < 1:
        USE_EMWEIGHT = 0
    if USE_EMWEIGHT > 0:
        CL = Fix(CLPAC, LOCAL)
        if DFPAC > 0:
            DF = Fix(DFPAC, LOCAL)
    else:
        CL = Fix(Fix(CLPAC, CELLS), LOCAL)
        if DFPAC > 0:
            DF = Fix(Fix(DFPAC, CELLS), LOCAL)
    return dict(PSPAC=PS, BGPAC=BG, CLPAC=CL, DFPAC=DF)


def ps_launch(PSPAC, NO_PS, GL, GLOBAL=GLOBAL_0):
    """Point sources (ASOC.py:1031-1045).  Returns dict(GLOBAL, BATCH, PACKETS, WPS)."""
    BATCH = int(max([1, PSPAC / GLOBAL]))
    per_source = GLOBAL * BATCH
    WPS = 1.0 / (PLANCK * per_source * ((GL * PARSEC) ** 2.0))
    return dict(GLOBAL=GLOBAL, BATCH=BATCH * NO_PS, PACKETS=per_source * NO_PS, WPS=WPS)


def bg_launch(BGPAC, AREA):
    """Isotropic background (ASOC.py:1061-1066): 8 work items per surface element.
    Returns dict(GLOBAL, BATCH, PACKETS, WBG)."""
    BATCH = max([1, int(round(BGPAC / (8 * AREA)))])
    PACKETS = int(8 * AREA * BATCH)
    WBG = np.pi / (PLANCK * 8 * BATCH)
    GLOBAL = Fix(int(8 * AREA), 64)
    return dict(GLOBAL=GLOBAL, BATCH=BATCH, PACKETS=PACKETS, WBG=WBG)


def hpbg_launch(BGPAC, NX, NY, NZ):
    """Healpix background (ASOC.py:1050-1059): 100 packets per work item, no systematic
    traversal of surface elements.  Work items with id >= 8*AREA return at once in the kernel
    (kernel_ASOC.c:874) while the weight counts them -- kept as in the reference.
    Returns dict(GLOBAL, BATCH, PACKETS, WBG)."""
    BATCH = 100
    GLOBAL = Fix(BGPAC / BATCH, 64)
    PACKETS = GLOBAL * BATCH
    WBG = np.pi / PLANCK
    WBG /= (GLOBAL * BATCH) / (2 * (NX * NY + NX * NZ + NY * NZ))
    return dict(GLOBAL=GLOBAL, BATCH=BATCH, PACKETS=PACKETS, WBG=WBG)


def hpbg_sca_launch(BGPAC, NX, NY, NZ):
    """Healpix background of the scattering run (ASOCS.py:480-497): one packet per work item, aimed at a
    sphere of radius Rout that contains the cloud; the kernel rejects those that miss.
    Returns dict(GLOBAL, BATCH, PACKETS, WBG)."""
    BATCH = 1
    GLOBAL = Fix(int(BGPAC / BATCH), 64)
    PACKETS = GLOBAL * BATCH
    Rout = 0.5 * math.sqrt(NX * NX + NY * NY + NZ * NZ)
    WBG = np.pi * 4.0 * np.pi * Rout ** 2.0 / (PLANCK * PACKETS)
    return dict(GLOBAL=GLOBAL, BATCH=BATCH, PACKETS=PACKETS, WBG=WBG)


def roi_launch(ROIPAC, NELEM, ROI_NSIDE, LOCAL=LOCAL_GPU):
    """Packets of a loaded ROI record (ASOC.py:1094-1105): 100 work items per surface element, each sends a whole
    number of Healpix maps.  PACKETS carries the number of surface elements.  Returns dict(GLOBAL, BATCH, PACKETS)."""
    npix = 12 * ROI_NSIDE * ROI_NSIDE
    BATCH = max([1, int(ROIPAC / (100.0 * npix * NELEM))]) * npix
    return dict(GLOBAL=Fix(100 * NELEM, LOCAL), BATCH=BATCH, PACKETS=NELEM)


def cl_launch(PAC, CELLS, GLOBAL=GLOBAL_0):
    """Cell emission: diffuse (ASOC.py:1086-1090) or dust re-emission (ASOC.py:1640).
    Returns dict(GLOBAL, BATCH, PACKETS)."""
    return dict(GLOBAL=GLOBAL, BATCH=int(PAC / CELLS), PACKETS=PAC)


def trapezoid_weight(FFREQ, IFREQ):
    """FF = FREQ * (trapezoid interval), kernel argument TW (ASOC.py:1219-1223)."""
    NFREQ = len(FFREQ)
    FF = float(FFREQ[IFREQ])
    if IFREQ == 0:
        FF *= 0.5 * (float(FFREQ[1]) - float(FFREQ[0]))
    elif IFREQ == NFREQ - 1:
        FF *= 0.5 * (float(FFREQ[NFREQ - 1]) - float(FFREQ[NFREQ - 2]))
    else:
        FF *= 0.5 * (float(FFREQ[IFREQ + 1]) - float(FFREQ[IFREQ - 1]))
    return FF


def launch_seed(SEED, IFREQ, DEVICES=1, ID=0):
    """seed = fmod(SEED+SEED0+(DEVICES*IFREQ+ID)*SEED1, 1.0) (ASOC.py:1247); the reference's
    per-device term is what the weak-scaling multi-GPU mode uses (one replica per rank)."""
    return math.fmod(SEED + SEED0 + (DEVICES * IFREQ + ID) * SEED1, 1.0)


def shard_range(GLOBAL, rank, world):
    """Work-item range of `rank` when one logical launch is split over `world` GPUs
    (SURVEY.md 8(e)): contiguous, multiples of 64 except possibly the last."""
    per = Fix(int(math.ceil(GLOBAL / world)), 64)
    first = min(GLOBAL, rank * per)
    last = min(GLOBAL, first + per)
    return first, last - first


def shard_launches(globals_, weights, rank, world):
    """Split a SEQUENCE of launches (the frequencies of the source blocks of a run, ASOC.py:1028-1545) over `world` GPUs:
    rank r executes a contiguous share of the sequence -- whole launches, at full population, and a work-item range of
    the launch at either end of its share -- so that every rank simulates the same number of packets (`weights`: packets
    per launch).  Returns [(first, count)] per launch (count 0: not this rank's).  Cut points depend on the sequence only,
    so all ranks agree; boundaries are multiples of 64 work items.  With one launch this is shard_range."""
    n = len(globals_)
    cum = [0.0]
    for w in weights:
        cum.append(cum[-1] + float(w))
    total = cum[-1]
    out = []
    for i in range(n):
        G = int(globals_[i])
        a, b = cum[i], cum[i + 1]

        def cut(k):
            x = total * k / world
            if x <= a or b <= a:
                return 0
            if x >= b:
                return G
            return min(G, Fix(int(G * (x - a) / (b - a)), 64))
        first, last = cut(rank), (G if rank == world - 1 else cut(rank + 1))
        out.append((first, max(0, last - first)))
    return out


H_K_D = 4.79924335e-11                 # ASOC_aux.py:36
H_CC20 = 7.372496678e-28                # ASOC_aux.py:40
NE_TEMPERATURE = 30000                  # ASOC.py:641


def planck_safe(f, T):
    """ASOC_aux.py:60-62"""
    return 2.0e-20 * ((H_CC20 * f) * f) * f / (np.exp(np.clip(H_K_D * f / T, -100, +100)) - 1.0)


def kernel_literals(GL):
    """The float literals the reference compiles into its kernels: -D FACTOR=%.4ef, -D LENGTH=%.5ef
    with LENGTH = GL*PARSEC (ASOC.py:345-348).  Returns (FACTOR, LENGTH) as float32."""
    return np.float32(float("%.4e" % FACTOR)), np.float32(float("%.5e" % (GL * PARSEC)))


def temperature_table(FFREQ, FABS, GL, NE=NE_TEMPERATURE):
    """E <-> T mapping of the equilibrium-temperature solve (ASOC.py:643-689): energies emitted by a
    grain at T = 1 + i*1600/NE K (trapezoid integral of FABS*B_nu, with the FACTOR scaling), inverted on a
    logarithmic energy grid E[i] = Emin*kE^i by linear interpolation.  Returns (Emin, kE, TTT float32[NE])."""
    FFREQ = np.asarray(FFREQ, np.float64)
    FABS = np.asarray(FABS, np.float64)
    TSTEP = 1600.0 / NE
    TT = 1.0 + TSTEP * np.arange(NE)
    DF = FFREQ[2:] - FFREQ[:(-2)]
    Eout = np.zeros(NE, np.float64)
    for i in range(NE):
        TMP = FABS * planck_safe(FFREQ, TT[i])
        res = TMP[0] * (FFREQ[1] - FFREQ[0]) + TMP[-1] * (FFREQ[-1] - FFREQ[-2])
        res += np.sum(TMP[1:(-1)] * DF)
        Eout[i] = (4.0 * np.pi * FACTOR / (GL * PARSEC)) * 0.5 * res
    Emin, Emax = Eout[0], Eout[NE - 1] * 0.9999
    kE = (Emax / Emin) ** (1.0 / (NE - 1.0))
    TTT = np.asarray(np.interp(Emin * kE ** np.arange(NE), Eout, TT), np.float32)
    return Emin, kE, TTT


def solve_temperature_host(EABS, cloud, Emin, kE, TTT, GL, beta=None, empty_below=1.0e-10):
    """The reference's host temperature solve (ASOC.py:2042-2073), vectorised: the default without the key `CLT`
    and the only path that takes the escape probability beta of an ALI run.  Its interpolation weight
    wi = (Emin kE^(iE+1) - E) / (Emin kE^(iE+1) - kE^iE) is kept as written (the device kernel divides by
    Emin kE^iE (kE-1) instead); cells with density < 1e-10 (parents, empty cells) get 0.  The `MPT` variant
    (ASOC.py:2076-2119) is the same formula with density <= 0 as the test for an empty cell: empty_below=0."""
    NE = len(TTT)
    oplgkE = 1.0 / math.log10(kE)
    scale = (6.62607e-27 * FACTOR) / (GL * PARSEC)
    T = np.zeros(cloud.CELLS, np.float32)
    TT = np.asarray(TTT, np.float64)
    for level in range(cloud.LEVELS):
        a, b = int(cloud.OFF[level]), int(cloud.OFF[level] + cloud.LCELLS[level])
        dens = np.asarray(cloud.DENS[a:b], np.float64)
        ok = (dens >= empty_below) if empty_below > 0.0 else (dens > 0.0)
        Ein = (scale / ADHOC) * np.asarray(EABS[a:b], np.float64)[ok] * (8.0 ** level) / dens[ok]
        if beta is not None:
            Ein = Ein / np.asarray(beta[a:b], np.float64)[ok]
        with np.errstate(divide='ignore', invalid='ignore'):
            iE = np.clip(np.floor(oplgkE * np.log10(Ein / Emin)), 0, NE - 2)
        iE = np.nan_to_num(iE, nan=0.0).astype(np.int64)
        wi = (Emin * kE ** (iE + 1) - Ein) / (Emin * kE ** (iE + 1) - kE ** iE)
        t = np.zeros(b - a, np.float64)
        t[ok] = wi * TT[iE] + (1.0 - wi) * TT[iE + 1]
        T[a:b] = t
    return T


def emission_host(FREQ, FABS, TNEW, GL):
    """The reference's host emission (ASOC.py:2199-2232, without the key `CLE`): EMITTED[CELLS, nfreq] =
    FACTOR 4 pi / (h f) FABS B_f(TNEW) / (GL pc), float32; cells with TNEW = 0 (parents) get the value of the
    clipped exponent exp(80), as there."""
    T = np.asarray(TNEW, np.float64)
    out = np.zeros((T.size, len(FREQ)), np.float32)
    with np.errstate(divide='ignore', over='ignore', invalid='ignore'):
        for k in range(len(FREQ)):
            f = float(FREQ[k])
            B = 2.0e-20 * ((H_CC20 * f) * f) * f / (np.exp(np.clip(H_K_D * f / T, -80, +80)) - 1.0)
            out[:, k] = ((FACTOR * 4.0 * np.pi / (PLANCK * f)) * float(FABS[k]) * B) / (GL * PARSEC)
    return out


def mirror_mask(MIRROR):
    """`mirror xXyYzZ` -> bit mask of reflecting faces (ASOC.py:319-321)"""
    MIRROR = MIRROR or ""
    return (1 * ('x' in MIRROR) + 2 * ('X' in MIRROR) + 4 * ('y' in MIRROR) + 8 * ('Y' in MIRROR)
            + 16 * ('z' in MIRROR) + 32 * ('Z' in MIRROR))


def set_observer_directions(OBS_THETA, OBS_PHI):
    """Observer directions and image axes (ASOC_aux.py:1129-1183): for every (theta, phi)
    [rad] the unit vector towards the observer ODIR and the map axes RA (increases to the
    right) and DE, as rows of a rotation of (x, y, z).  Components of ODIR with magnitude
    below 1e-5 are set to 1e-5, as in the reference.  Returns NDIR and three [NDIR,4]
    float32 arrays (OpenCL float3 = 16 bytes)."""
    th = list(OBS_THETA) if len(OBS_THETA) else [0.5 * math.pi]
    ph = list(OBS_PHI) if len(OBS_PHI) else [0.0]
    NDIR = len(th)
    ODIR = np.zeros((NDIR, 4), np.float32)
    RA = np.zeros((NDIR, 4), np.float32)
    DE = np.zeros((NDIR, 4), np.float32)
    for i in range(NDIR):
        b, a = 0.5 * math.pi - th[i], ph[i]
        R = np.zeros((3, 3), np.float32)
        R[0, :] = [math.cos(a) * math.cos(b), -math.sin(a), -math.cos(a) * math.sin(b)]
        R[1, :] = [math.sin(a) * math.cos(b), math.cos(a), -math.sin(a) * math.sin(b)]
        R[2, :] = [math.sin(b), 0.0, math.cos(b)]
        ODIR[i, :3] = np.matmul(R, [1, 0, 0])
        RA[i, :3] = np.matmul(R, [0, 1, 0])
        DE[i, :3] = np.matmul(R, [0, 0, 1])
        for k in range(3):
            if abs(ODIR[i, k]) < 1.0e-5:
                ODIR[i, k] = 1.0e-5
    return NDIR, ODIR, RA, DE
