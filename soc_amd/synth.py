"""Synthetic clouds in SOC's cloud-file layout.

The hierarchy encoding follows the reference reader (ASOC_aux.py:716-803) and the
kernels' use of it (kernel_ASOC_aux.c:156-160, 268-273):
  * one float32 vector per level, concatenated; OFF[l] = first cell of level l;
  * value > 0  -> density of a leaf cell;
  * value <= 0 -> parent: ``-value`` reinterpreted as int32 is the index, WITHIN THE NEXT
    LEVEL, of the first of its 8 children; children are contiguous, octant = x + 2y + 4z.
"""
import numpy as np


def I2F(i):
    """int32 index -> float32 with the same bits (ASOC_aux.py:14-16)."""
    return np.asarray(i, np.int32).view(np.float32)


def F2I(x):
    """float32 -> int32 with the same bits (ASOC_aux.py:18-20)."""
    return np.asarray(x, np.float32).view(np.int32)


class Cloud:
    """Density hierarchy: NX, NY, NZ, LEVELS, LCELLS[LEVELS], OFF[LEVELS], DENS[CELLS]."""

    def __init__(self, NX, NY, NZ, levels):
        self.NX, self.NY, self.NZ = int(NX), int(NY), int(NZ)
        self.H = [np.ascontiguousarray(h, np.float32) for h in levels]
        self.LEVELS = len(self.H)
        self.LCELLS = np.asarray([len(h) for h in self.H], np.int32)
        self.OFF = np.zeros(self.LEVELS, np.int32)
        self.OFF[1:] = np.cumsum(self.LCELLS)[:-1]
        self.CELLS = int(self.LCELLS.sum())
        self.DENS = np.concatenate(self.H).astype(np.float32)
        self.AREA = 2 * (self.NX * self.NY + self.NY * self.NZ + self.NZ * self.NX)

    def write(self, filename):
        """Write the cloud file: int32 NX,NY,NZ,LEVELS,CELLS; per level int32 LCELLS + float32[LCELLS]."""
        with open(filename, "wb") as fp:
            np.asarray([self.NX, self.NY, self.NZ, self.LEVELS, self.CELLS], np.int32).tofile(fp)
            for h in self.H:
                np.asarray([len(h)], np.int32).tofile(fp)
                h.tofile(fp)

    def leaf_mask(self):
        return self.DENS > 0.0

    def level_of_cells(self):
        lev = np.zeros(self.CELLS, np.int32)
        for l in range(self.LEVELS):
            lev[self.OFF[l]:self.OFF[l] + self.LCELLS[l]] = l
        return lev


def cartesian_cloud(N, seed=1234, uniform=None, NY=None, NZ=None):
    """N^3 (or N x NY x NZ) single-level cloud.  Lognormal density exp(N(0,1))*1e3 clipped to
    [1, 1e5] or a uniform value."""
    NY = N if NY is None else NY
    NZ = N if NZ is None else NZ
    if uniform is not None:
        d = np.full(N * NY * NZ, uniform, np.float32)
    else:
        rng = np.random.default_rng(seed)
        d = np.clip(np.exp(rng.standard_normal(N * NY * NZ)) * 1.0e3, 1.0, 1.0e5).astype(np.float32)
    return Cloud(N, NY, NZ, [d])


def octree_cloud(N, levels=4, frac=0.10, seed=1234, sigma=0.3, uniform=None):
    """N^3 root grid; on every level the densest ``frac`` of the leaf cells are refined into
    8 children (density = parent x lognormal(sigma), renormalised to conserve mass), for
    ``levels`` hierarchy levels in total (levels=4 <=> 3 refinement levels)."""
    rng = np.random.default_rng(seed)
    if uniform is not None:
        d0 = np.full(N * N * N, uniform, np.float64)
    else:
        d0 = np.clip(np.exp(rng.standard_normal(N * N * N)) * 1.0e3, 1.0, 1.0e5)
    H = [d0]
    for l in range(levels - 1):
        cur = H[l]
        n_ref = int(round(frac * len(cur)))
        if n_ref < 1:
            break
        if uniform is not None:
            parents = np.sort(rng.choice(len(cur), n_ref, replace=False))
        else:
            parents = np.sort(np.argpartition(cur, len(cur) - n_ref)[len(cur) - n_ref:])
        pd = cur[parents]
        if uniform is not None:
            kids = np.repeat(pd[:, None], 8, axis=1)
        else:
            w = np.exp(sigma * rng.standard_normal((n_ref, 8)))
            w *= 8.0 / w.sum(axis=1, keepdims=True)
            kids = pd[:, None] * w
        H.append(kids.reshape(-1))
        first_child = (8 * np.arange(n_ref)).astype(np.int32)
        link = -I2F(first_child).astype(np.float32)         # -0.0 for index 0: still <= 0
        cur = cur.astype(np.float32)
        cur[parents] = link
        H[l] = cur
    H = [np.asarray(h, np.float32) for h in H]
    # densities must stay > 0 after the float32 cast
    for h in H:
        leaf = h > 0
        h[leaf] = np.maximum(h[leaf], 1.0e-6)
    return Cloud(N, N, N, H)


def kat_octree():
    """The 4^3-root, 3-level tree of SURVEY.md 8(c): LCELLS=[64,8,8]; leaf value = 1 + global
    index; root cell 21 and level-1 cell 7 are parents of the first octet of the next level."""
    l0 = (1.0 + np.arange(64)).astype(np.float32)
    l1 = (1.0 + 64 + np.arange(8)).astype(np.float32)
    l2 = (1.0 + 72 + np.arange(8)).astype(np.float32)
    l0[21] = -I2F(0)
    l1[7] = -I2F(0)
    return Cloud(4, 4, 4, [l0, l1, l2])


def hg_scattering_table(g, bins=2500):
    """DSC/CSC rows for a Henyey-Greenstein phase function with asymmetry g.

    DSC: phase function on a linear cos(theta) grid -1..+1; CSC: cos(theta) at equidistant
    cumulative probability with CSC[0]=+1 ... CSC[bins-1]=-1 (layout: ASOC_aux.py:619-647;
    the recipe follows what DustLib.write_eqdust_dsc documents, DustLib.py:2532-2566)."""
    mu = np.linspace(-1.0, 1.0, bins)
    g = float(g)
    dsc = (1.0 - g * g) / (4.0 * np.pi * (1.0 + g * g - 2.0 * g * mu) ** 1.5)
    # inverse cumulative of HG: P(mu' < mu)
    u = np.linspace(0.0, 1.0, bins)           # cumulative probability measured from mu=+1
    if abs(g) < 1.0e-5:
        csc = 1.0 - 2.0 * u
    else:
        # probability of cos > mu:  (1-g^2)/(2g) * (1/(1-g) - 1/sqrt(1+g^2-2g mu))
        s = 1.0 / (1.0 - g) - 2.0 * g * u / (1.0 - g * g)
        csc = (1.0 + g * g - 1.0 / (s * s)) / (2.0 * g)
    return dsc.astype(np.float32), np.clip(csc, -1.0, 1.0).astype(np.float32)


# ---------------------------------------------------------------------------------------
# synthetic stochastic-heating solver data (input of A2E, SURVEY.md 5.4 "solver" layout)
# ---------------------------------------------------------------------------------------

def synth_solver(NFREQ=50, NE=128, NSIZE=3, seed=5):
    """A synthetic ``*.solver`` data set in the layout A2E_pre.py writes (A2E_pre.py:180-290) and
    A2E.py reads (A2E.py:116-127, 354-370).  The reference ships no GSET dust files, so the
    arrays are built from a simple enthalpy-grid model: transition l->u is fed by the
    frequencies whose photon energy matches E_u - E_l within the bin width; cooling goes
    u -> u-1; emission per enthalpy bin is Planck-like.  Returns a dict of numpy arrays."""
    rng = np.random.default_rng(seed)
    h = 6.62606957e-27
    FREQ = np.logspace(np.log10(1.5e11), np.log10(2.0e15), NFREQ).astype(np.float32)
    SIZE_A = np.logspace(-7.5, -6.0, NSIZE).astype(np.float32)
    S_FRAC = (np.ones(NSIZE) / NSIZE).astype(np.float32)
    GD = np.float32(1.0e-10)
    SK_ABS = (1e-22 * (FREQ[None, :] / 1e13) ** 1.5 * (SIZE_A[:, None] / 1e-7) ** 2 / NSIZE).astype(np.float32)
    sizes = []
    for isize in range(NSIZE):
        E = h * 1.0e11 * (h * 4.0e15 / (h * 1.0e11)) ** (np.arange(NE) / (NE - 1.0))     # enthalpy bins [erg]
        W = np.gradient(E)
        L1 = np.ones((NE, NE), np.int32)
        L2 = np.zeros((NE, NE), np.int32)
        Iw = []
        hnu = h * FREQ.astype(np.float64)
        for l in range(NE - 1):
            for u in range(l + 1, NE):
                dE = E[u] - E[l]
                m = np.nonzero((hnu >= dE - 0.75 * W[u]) & (hnu <= dE + 0.75 * W[u]))[0]
                if len(m) > 0:
                    L1[l, u], L2[l, u] = m[0], m[-1]
                    w = rng.uniform(0.5, 1.5, len(m)) * 1.0e-3 / len(m)
                    if rng.random() < 0.02:
                        w[0] = -w[0]                    # the reference clamps negative sums with max(I, 0)
                    Iw.extend(w)
        Tdown = (1.0e-2 * (1.0 + np.arange(NE)) ** 1.5 * rng.uniform(0.9, 1.1, NE)).astype(np.float32)
        T_of_bin = 5.0 + 600.0 * (np.arange(NE) / (NE - 1.0)) ** 2
        x = np.clip(4.79924335e-11 * FREQ[:, None].astype(np.float64) / T_of_bin[None, :], 0, 80)
        EA = (1e-30 * FREQ[:, None].astype(np.float64) ** 2 / (np.exp(x) - 1.0 + 1e-30)
              * SK_ABS[isize][:, None] * 1e20).astype(np.float32)                       # [NFREQ, NE]
        Ibeg = np.zeros(NFREQ, np.int32)
        for j in range(NFREQ):
            nz = np.nonzero(EA[j] > 1e-30 * EA[j].max())[0]
            Ibeg[j] = nz[0] if len(nz) else NE - 1
        sizes.append(dict(Iw=np.asarray(Iw, np.float32), L1=L1.reshape(-1), L2=L2.reshape(-1), Tdown=Tdown,
                          EA=EA.reshape(-1), Ibeg=Ibeg))
    return dict(NFREQ=NFREQ, FREQ=FREQ, GD=GD, NSIZE=NSIZE, SIZE_A=SIZE_A, S_FRAC=S_FRAC, NE=NE, SK_ABS=SK_ABS,
                sizes=sizes)


def write_solver(filename, sol):
    """int32 NFREQ; f32 FREQ; f32 GD; int32 NSIZE; f32 SIZE_A; f32 S_FRAC; int32 NE; f32 SK_ABS[NSIZE,NFREQ];
    per size: int32 noIw; f32 Iw; int32 L1[NE*NE]; int32 L2[NE*NE]; f32 Tdown[NE]; f32 EA[NFREQ*NE]; int32 Ibeg[NFREQ]"""
    with open(filename, "wb") as fp:
        np.asarray([sol["NFREQ"]], np.int32).tofile(fp)
        np.asarray(sol["FREQ"], np.float32).tofile(fp)
        np.asarray([sol["GD"]], np.float32).tofile(fp)
        np.asarray([sol["NSIZE"]], np.int32).tofile(fp)
        np.asarray(sol["SIZE_A"], np.float32).tofile(fp)
        np.asarray(sol["S_FRAC"], np.float32).tofile(fp)
        np.asarray([sol["NE"]], np.int32).tofile(fp)
        np.asarray(sol["SK_ABS"], np.float32).tofile(fp)
        for s in sol["sizes"]:
            np.asarray([len(s["Iw"])], np.int32).tofile(fp)
            np.asarray(s["Iw"], np.float32).tofile(fp)
            np.asarray(s["L1"], np.int32).tofile(fp)
            np.asarray(s["L2"], np.int32).tofile(fp)
            np.asarray(s["Tdown"], np.float32).tofile(fp)
            np.asarray(s["EA"], np.float32).tofile(fp)
            np.asarray(s["Ibeg"], np.int32).tofile(fp)


def a2e_absorption_fraction(sol, isize):
    """AF of A2E.py:338-341: share of the absorptions taken by one size, per grain."""
    K_ABS = np.sum(sol["SK_ABS"], axis=0)
    S_FRAC = np.clip(sol["S_FRAC"], 1.0e-32, 1.0e30)
    AF = np.asarray(sol["SK_ABS"][isize, :], np.float64) / np.asarray(K_ABS, np.float64)
    AF /= S_FRAC[isize] * sol["GD"]
    return np.asarray(np.clip(AF, 1.0e-32, 1.0e+100), np.float32)
