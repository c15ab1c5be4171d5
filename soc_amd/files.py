"""Readers and writers of SOC's on-disk formats: all little-endian raw
int32/float32, no padding.  Each function cites the reference code that defines the layout."""
import os

import numpy as np

from .launch import FACTOR, PARSEC
from .synth import Cloud


class FileError(ValueError):
    pass


def read_cloud(filename, kdensity=1.0, max_levels=999):
    """Cloud file -> Cloud (ASOC_aux.py:716-803).  int32 NX,NY,NZ,LEVELS,CELLS; per level
    int32 LCELLS + float32[LCELLS]; leaf densities are scaled by ``kdensity`` and clipped to
    [1e-6, 1e20] when kdensity != 1 (ASOC_aux.py:779-781); links (<= 0) are left untouched."""
    with open(filename, 'rb') as fp:
        hdr = np.fromfile(fp, np.int32, 5)
        if hdr.size != 5:
            raise FileError("%s: truncated cloud header" % filename)
        NX, NY, NZ, LEVELS, CELLS = [int(v) for v in hdr]
        if LEVELS > max_levels:
            # keyword `levels`: the hierarchy is cut, the new cloud written beside the old one and used (ASOC_aux.py:748-761)
            # Every rank of a torch.distributed run comes here at the same time: each writes the (identical) cut cloud under a name
            # of its own and moves it into place -- os.replace is atomic, so no reader ever sees a truncated file.
            newname = '%s.MAX%d' % (filename, max_levels)
            tmpname = '%s.tmp.%d' % (newname, os.getpid())
            try:
                cut_levels(filename, tmpname, max_levels - 1)
                os.replace(tmpname, newname)
            finally:
                if os.path.exists(tmpname):
                    os.remove(tmpname)
            return read_cloud(newname, kdensity, max_levels)
        H = []
        for level in range(LEVELS):
            n = np.fromfile(fp, np.int32, 1)
            if n.size != 1 or n[0] < 0:
                break
            d = np.fromfile(fp, np.float32, int(n[0]))
            if d.size != n[0]:
                raise FileError("%s: level %d truncated" % (filename, level))
            if kdensity != 1.0:
                m = d > 0.0
                d[m] = np.clip(np.float32(kdensity) * d[m], 1.0e-6, 1e20)
            H.append(d)
    c = Cloud(NX, NY, NZ, H)
    if c.CELLS != CELLS:
        raise FileError("%s: header says %d cells, levels hold %d" % (filename, CELLS, c.CELLS))
    return c


def cut_levels(infile, outfile, maxlevel):
    """OT_cut_levels (ASOC_aux.py:651-713) with the AverageParent kernel (kernel_OT_tools.c:5-24) in numpy: levels above
    ``maxlevel`` (0, 1, ...) are dropped, from the bottom up every parent (value <= 1e-9: a link) becomes a leaf with the
    mean of its eight children -- the float32 sum in child order, divided by 8, as the kernel adds them."""
    with open(infile, 'rb') as fp:
        NX, NY, NZ, LEVELS, CELLS = [int(v) for v in np.fromfile(fp, np.int32, 5)]
        H = []
        for _ in range(LEVELS):
            n = int(np.fromfile(fp, np.int32, 1)[0])
            d = np.fromfile(fp, np.float32, n)
            if d.size != n:
                raise FileError("%s: truncated" % infile)
            H.append(d)
    maxlevel = min(LEVELS - 1, int(maxlevel))
    for i in range(LEVELS - 2, maxlevel - 1, -1):
        P, C = H[i], H[i + 1]
        par = np.nonzero(~(P > np.float32(1.0e-9)))[0]
        first = (-P[par]).view(np.int32).astype(np.int64)
        if par.size and (first.min() < 0 or first.max() + 8 > C.size):
            raise FileError("%s: level %d holds a link outside level %d" % (infile, i, i + 1))
        f = np.zeros(par.size, np.float32)
        for k in range(8):
            f = f + C[first + k]
        P[par] = f / np.float32(8.0)
    with open(outfile, 'wb') as fp:
        np.asarray([NX, NY, NZ, maxlevel + 1, sum(len(h) for h in H[:maxlevel + 1])], np.int32).tofile(fp)
        for i in range(maxlevel + 1):
            np.asarray([len(H[i])], np.int32).tofile(fp)
            np.asarray(H[i], np.float32).tofile(fp)


def read_dust(filenames, GL):
    """Dust files -> FFREQ, [G], [ABS], [SCA] per dust (ASOC_aux.py:557-596): text, line 2
    grain density, line 3 grain size [cm], line 4 NFREQ, rows ``freq g Qabs Qsca``;
    Q * GRAIN_DENSITY*pi*a^2*GL*PARSEC = optical depth per unit density per root cell."""
    FFREQ, AFG, AFABS, AFSCA = None, [], [], []
    for fn in filenames:
        with open(fn) as fp:
            lines = fp.readlines()
        gd = float(lines[1].split()[0])
        gs = float(lines[2].split()[0])
        coeff = gd * np.pi * gs ** 2.0 * GL * PARSEC
        d = np.loadtxt(fn, skiprows=4, ndmin=2)
        f = np.asarray(d[:, 0], np.float32)
        if FFREQ is not None and len(f) != len(FFREQ):
            raise FileError("dusts must have the same frequency grid")
        FFREQ = f
        AFG.append(np.asarray(d[:, 1], np.float32))
        AFABS.append(np.asarray(d[:, 2] * coeff, np.float32))
        AFSCA.append(np.asarray(d[:, 3] * coeff, np.float32))
    return FFREQ, AFG, AFABS, AFSCA


def read_scattering_functions(filenames, NFREQ, BINS):
    """dsc file(s) -> FDSC[ndust,NFREQ,BINS], FCSC[ndust,NFREQ,BINS] (ASOC_aux.py:619-647)."""
    nd = len(filenames)
    FDSC = np.zeros((nd, NFREQ, BINS), np.float32)
    FCSC = np.zeros((nd, NFREQ, BINS), np.float32)
    for i, fn in enumerate(filenames):
        a = np.fromfile(fn, np.float32)
        if a.size != 2 * NFREQ * BINS:
            raise FileError("%s holds %d floats, expected 2*%d*%d" % (fn, a.size, NFREQ, BINS))
        FDSC[i] = a[:NFREQ * BINS].reshape(NFREQ, BINS)
        FCSC[i] = a[NFREQ * BINS:].reshape(NFREQ, BINS)
    return FDSC, FCSC


def write_scattering_functions(filename, DSC, CSC):
    with open(filename, 'wb') as fp:
        np.asarray(DSC, np.float32).tofile(fp)
        np.asarray(CSC, np.float32).tofile(fp)


def read_background_intensity(filename, NFREQ, scale=1.0):
    """float32 I_nu[NFREQ] (ASOC_aux.py:1081-1102)"""
    a = np.fromfile(filename, np.float32, NFREQ)
    if a.size != NFREQ:
        raise FileError("background intensity for %d frequencies, optical data for %d" % (a.size, NFREQ))
    return a * np.float32(scale)


def read_source_luminosities(filenames, NFREQ, scaling):
    """float32 L_nu[NFREQ] per source (ASOC_aux.py:1107-1124)"""
    LPS = np.zeros((len(filenames), NFREQ), np.float32)
    for i, fn in enumerate(filenames):
        a = np.fromfile(fn, np.float32, NFREQ)
        if a.size != NFREQ:
            raise FileError("source %d: intensity for %d frequencies, optical data for %d" % (i, a.size, NFREQ))
        LPS[i] = a * scaling[i]
    return LPS


def read_abundances(filenames, cells):
    """ABU[cells, ndust] or None when every dust has constant abundance (ASOC_aux.py:600-615)."""
    if not any(f[0] != '#' for f in filenames):
        return None
    ABU = np.ones((cells, len(filenames)), np.float32)
    for i, fn in enumerate(filenames):
        if fn[0] != '#':
            ABU[:, i] = np.fromfile(fn, np.float32, cells)
    return ABU


def mmap_diffuserad(filename, CELLS):
    """diffuse emission file: int32 CELLS,NFREQ'; float32 [CELLS,NFREQ'] photons/Hz/cm3
    (ASOC_aux.py:839-863)"""
    dims = np.fromfile(filename, np.int32, 2)
    if dims[0] != CELLS:
        raise FileError("diffuse field has %d cells but the cloud has %d" % (dims[0], CELLS))
    return np.memmap(filename, dtype='float32', mode='r', offset=8, shape=(int(dims[0]), int(dims[1])))


def write_diffuserad(filename, data):
    data = np.asarray(data, np.float32)
    with open(filename, 'wb') as fp:
        np.asarray(data.shape, np.int32).tofile(fp)
        data.tofile(fp)


def scale_absorbed(FABSORBED, cloud, GL, nnnlimit=0.0, absthin=1):
    """Final scaling of the per-frequency absorptions before they are written
    (ASOC.py:2793-2809): x FACTOR*8^level/(GL*PARSEC)/DENS; parents (and cells with density
    <= nnnlimit) are marked -1e20.  In place; returns the array.  absthin > 1: FABSORBED holds every absthin-th cell
    (`nnmake`, ASOC.py:2815-2836)."""
    if absthin > 1:
        ind_all = np.arange(0, cloud.CELLS, absthin)
        i0 = 0
        for level in range(cloud.LEVELS):
            a, b = int(cloud.OFF[level]), int(cloud.OFF[level] + cloud.LCELLS[level])
            coeff = (8.0 ** level) * (FACTOR / (GL * PARSEC))
            ind = ind_all[(ind_all >= a) & (ind_all < b)]
            n = len(ind)
            with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
                FABSORBED[i0:i0 + n, :] *= (coeff / cloud.DENS[ind].reshape(n, 1)).astype(np.float32)
            m = np.nonzero(cloud.DENS[ind] <= nnnlimit)[0]
            FABSORBED[i0 + m, :] = -1.0e20
            i0 += n
        return FABSORBED
    for level in range(cloud.LEVELS):
        a, b = int(cloud.OFF[level]), int(cloud.OFF[level] + cloud.LCELLS[level])
        coeff = (8.0 ** level) * (FACTOR / (GL * PARSEC))
        with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
            FABSORBED[a:b, :] *= (coeff / cloud.DENS[a:b].reshape(b - a, 1)).astype(np.float32)
        m = np.nonzero(cloud.DENS[a:b] <= nnnlimit)[0]
        FABSORBED[a + m, :] = -1.0e20
    return FABSORBED


def write_absorbed(filename, FABSORBED):
    """absorbed file: int32 CELLS,NFREQ; float32 [CELLS,NFREQ] (ASOC.py:2866-2875)"""
    FABSORBED = np.asarray(FABSORBED, np.float32)
    with open(filename, 'wb') as fp:
        np.asarray(FABSORBED.shape, np.int32).tofile(fp)
        FABSORBED.tofile(fp)


def create_absorbed(filename, rows, nfreq):
    """An absorbed file of the right size with its header, columns to be filled by write_absorbed_columns (several ranks, each the
    frequencies it simulated: no collective for the per-frequency absorptions)."""
    with open(filename, 'wb') as fp:
        np.asarray([rows, nfreq], np.int32).tofile(fp)
        fp.truncate(8 + 4 * int(rows) * int(nfreq))


def write_absorbed_columns(filename, FABSORBED, columns):
    """Write the given frequency columns of FABSORBED[CELLS, NFREQ] into an existing absorbed file."""
    dims = np.fromfile(filename, np.int32, 2)
    if tuple(int(x) for x in dims) != tuple(FABSORBED.shape):
        raise FileError("%s: holds %s, the run has %s" % (filename, tuple(dims), FABSORBED.shape))
    mm = np.memmap(filename, np.float32, mode='r+', offset=8, shape=FABSORBED.shape)
    for f in columns:
        mm[:, f] = FABSORBED[:, f]
    mm.flush()
    del mm


def read_absorbed(filename):
    dims = np.fromfile(filename, np.int32, 2)
    return np.fromfile(filename, np.float32, offset=8).reshape(int(dims[0]), int(dims[1]))


def analyse_external_point_sources(NX, NY, NZ, PSPOS, NO_PS, PS_METHOD):
    """Visible cloud sides per external point source -> XPS_NSIDE[NO_PS], XPS_SIDE[3*NO_PS],
    XPS_AREA[3*NO_PS] (ASOC_aux.py:1538-1632).  As in the reference, the "areas" are 1/nside
    (true projected areas are not computed there either) and for PS_METHOD 5 XPS_AREA[3*i]
    holds the cosine of the cone that contains the cloud, measured from the axis of the LAST
    illuminated side found."""
    XPS_NSIDE = np.zeros(max(NO_PS, 1), np.int32)
    XPS_SIDE = np.zeros(3 * max(NO_PS, 1), np.int32)
    XPS_AREA = np.zeros(3 * max(NO_PS, 1), np.float32)
    axis = np.zeros(3, np.float32)
    for i in range(NO_PS):
        p = PSPOS[i]
        if (0.0 <= p[0] <= NX) and (0.0 <= p[1] <= NY) and (0.0 <= p[2] <= NZ):
            continue
        no = 0
        for cond, side, ax in ((p[0] > NX, 0, (-1.0, 0.0, 0.0)), (p[0] < 0.0, 1, (1.0, 0.0, 0.0)),
                               (p[1] > NY, 2, (0.0, -1.0, 0.0)), (p[1] < 0.0, 3, (0.0, 1.0, 0.0)),
                               (p[2] > NZ, 4, (0.0, 0.0, -1.0)), (p[2] < 0.0, 5, (0.0, 0.0, 1.0))):
            if cond:
                XPS_SIDE[3 * i + no] = side
                XPS_AREA[3 * i + no] = 1.0
                axis = np.asarray(ax, np.float32)
                no += 1
        XPS_NSIDE[i] = no
        XPS_AREA[3 * i:3 * i + 3] /= no
    if PS_METHOD == 5:
        for i in range(NO_PS):
            cos_theta = 0.5 * np.pi
            for ii in range(8):
                vec = np.zeros(3, np.float32)
                vec[0] = NX * (ii % 2 == 0) - PSPOS[i][0]
                vec[1] = NY * ((ii / 2) % 2 == 0) - PSPOS[i][1]
                vec[2] = NZ * ((ii / 4) % 2 == 0) - PSPOS[i][2]
                tmp = abs(float(np.dot(axis, vec))) / float(np.linalg.norm(vec))
                cos_theta = min(cos_theta, tmp)
            XPS_AREA[3 * i] = cos_theta
    return XPS_NSIDE, XPS_SIDE, XPS_AREA


def read_solver(filename):
    """``*.solver`` file -> dict (layout: A2E_pre.py:180-290, read as A2E.py:116-127, 354-370 does)."""
    with open(filename, 'rb') as fp:
        NFREQ = int(np.fromfile(fp, np.int32, 1)[0])
        FREQ = np.fromfile(fp, np.float32, NFREQ)
        GD = np.fromfile(fp, np.float32, 1)[0]
        NSIZE = int(np.fromfile(fp, np.int32, 1)[0])
        SIZE_A = np.fromfile(fp, np.float32, NSIZE)
        S_FRAC = np.clip(np.fromfile(fp, np.float32, NSIZE), 1.0e-32, 1.0e30)
        NE = int(np.fromfile(fp, np.int32, 1)[0])
        SK_ABS = np.fromfile(fp, np.float32, NSIZE * NFREQ).reshape(NSIZE, NFREQ)
        sizes = []
        for isize in range(NSIZE):
            n = np.fromfile(fp, np.int32, 1)
            if n.size != 1:
                break                      # files written for equilibrium sizes only stop early
            noIw = int(n[0])
            s = dict(Iw=np.fromfile(fp, np.float32, noIw), L1=np.fromfile(fp, np.int32, NE * NE),
                     L2=np.fromfile(fp, np.int32, NE * NE), Tdown=np.fromfile(fp, np.float32, NE),
                     EA=np.fromfile(fp, np.float32, NE * NFREQ), Ibeg=np.fromfile(fp, np.int32, NFREQ))
            if s["Ibeg"].size != NFREQ:
                raise FileError("%s: size %d truncated" % (filename, isize))
            sizes.append(s)
    return dict(NFREQ=NFREQ, FREQ=FREQ, GD=GD, NSIZE=NSIZE, SIZE_A=SIZE_A, S_FRAC=S_FRAC, NE=NE, SK_ABS=SK_ABS, sizes=sizes)


def write_emitted(filename, EMITTED):
    """emitted file: int32 CELLS,NFREQ; float32 [CELLS,NFREQ] (A2E.py:151-156)"""
    write_absorbed(filename, EMITTED)


def mmap_emitted(filename, CELLS, REMIT_NFREQ):
    """Read-only view of an existing emitted file as EMITTED[CELLS, REMIT_NFREQ]
    (ASOC_aux.py:869-935); the scattering run only reads it."""
    dims = np.fromfile(filename, np.int32, 2)
    if dims[0] != CELLS or dims[1] != REMIT_NFREQ:
        raise FileError("%s holds %d cells x %d frequencies, the run needs %d x %d" % (
            filename, dims[0], dims[1], CELLS, REMIT_NFREQ))
    return np.memmap(filename, dtype='float32', mode='r', offset=8, shape=(int(CELLS), int(REMIT_NFREQ)))


def write_outcoming(filename, FFREQ, OUTCOMING):
    """outcoming.socs (ASOCS.py:409-416): int32 NPIX.y, NPIX.x, NFREQ; float32 FFREQ[NFREQ];
    float32 OUTCOMING[NFREQ, NDIR, NPIX.y, NPIX.x]"""
    OUTCOMING = np.asarray(OUTCOMING, np.float32)
    with open(filename, 'wb') as fp:
        np.asarray([OUTCOMING.shape[2], OUTCOMING.shape[3], OUTCOMING.shape[0]], np.int32).tofile(fp)
        np.asarray(FFREQ, np.float32).tofile(fp)
        OUTCOMING.tofile(fp)


def read_outcoming(filename, NDIR):
    """-> FFREQ, OUTCOMING[NFREQ, NDIR, NPIX.y, NPIX.x]"""
    ny, nx, nfreq = np.fromfile(filename, np.int32, 3)
    FFREQ = np.fromfile(filename, np.float32, nfreq, offset=12)
    data = np.fromfile(filename, np.float32, offset=12 + 4 * int(nfreq))
    return FFREQ, data.reshape(int(nfreq), NDIR, int(ny), int(nx))


def hpbg_for_frequency(sky, scale, weighted, clip_low=1.0e-3, skip_empty=True):
    """Device arrays of the Healpix background for one frequency (ASOC.py:1196-1214): sky[49152]
    (file units x user scaling) -> photons per package BG = scale*sky, and with `weighted`
    the cumulative pixel probability HPBGP (pixels chosen in proportion to their clipped
    intensity, the packet weight corrected by HPBGW).  Returns (BG, HPBGP or None), or None
    for an empty sky in weighted mode.  ASOCS.py:598-611 clips at 1e-2 instead of 1e-3 and has
    no empty-sky test (clip_low, skip_empty)."""
    sky = np.asarray(sky, np.float32)
    if not weighted:
        return np.asarray(np.float32(scale) * sky, np.float32), None
    tmp = np.asarray(sky, np.float64)
    if skip_empty and np.max(tmp) < 1.0e-40:
        return None
    tmp = tmp / np.mean(tmp)
    tmp = np.clip(tmp, clip_low, 1.0e4)
    tmp /= np.sum(tmp)
    HPBGW = (1.0 / 49152.0) / tmp
    HPBGP = np.cumsum(tmp)
    HPBGP[-1] = 1.00001
    return np.asarray(scale * sky * HPBGW, np.float32), np.asarray(HPBGP, np.float32)


def write_outcoming_healpix(filename, NSIDE, FFREQ, OUTCOMING):
    """outcoming.socs for a Healpix map (ASOCS.py:418-426): int32 NSIDE, NFREQ; float32 FFREQ[NFREQ];
    float32 OUTCOMING[NFREQ, 12*NSIDE^2]"""
    OUTCOMING = np.asarray(OUTCOMING, np.float32)
    with open(filename, 'wb') as fp:
        np.asarray([NSIDE, OUTCOMING.shape[0]], np.int32).tofile(fp)
        np.asarray(FFREQ, np.float32).tofile(fp)
        OUTCOMING.tofile(fp)


def write_temperature(filename, cloud, TNEW):
    """temperature file: the layout of the cloud file (ASOC.py:2125-2133): int32 NX,NY,NZ,LEVELS,CELLS;
    per level int32 LCELLS[level] + float32 T of its cells"""
    TNEW = np.asarray(TNEW, np.float32)
    with open(filename, 'wb') as fp:
        np.asarray([cloud.NX, cloud.NY, cloud.NZ, cloud.LEVELS, cloud.CELLS], np.int32).tofile(fp)
        for level in range(cloud.LEVELS):
            a, b = int(cloud.OFF[level]), int(cloud.OFF[level] + cloud.LCELLS[level])
            np.asarray([cloud.LCELLS[level]], np.int32).tofile(fp)
            TNEW[a:b].tofile(fp)


def read_temperature(filename, cloud):
    """temperature file (read_otfile, ASOC_aux.py:1420-1445) of this cloud -> T[CELLS].  A file that lacks the deepest
    level is refused (the reference copies parent values down, ASOC.py:721-731)."""
    with open(filename, 'rb') as fp:
        nx, ny, nz, levels, cells = (int(v) for v in np.fromfile(fp, np.int32, 5))
        if (nx, ny, nz, levels, cells) != (cloud.NX, cloud.NY, cloud.NZ, cloud.LEVELS, cloud.CELLS):
            raise FileError("%s: temperatures of a %dx%dx%d cloud with %d levels and %d cells, the model has %dx%dx%d, %d, %d" % (
                filename, nx, ny, nz, levels, cells, cloud.NX, cloud.NY, cloud.NZ, cloud.LEVELS, cloud.CELLS))
        T = np.zeros(cells, np.float32)
        for level in range(levels):
            n = int(np.fromfile(fp, np.int32, 1)[0])
            if n != int(cloud.LCELLS[level]):
                raise FileError("%s: level %d holds %d cells, the model %d" % (filename, level, n, cloud.LCELLS[level]))
            a = int(cloud.OFF[level])
            T[a:a + n] = np.fromfile(fp, np.float32, n)
    return T


# ---------------------------------------------------------------------------------------
# region-of-interest records (roisave / roiload): int32 (nx, ny, nz, nside, nfreq) + float32 [nfreq, nelem * 12*nside^2]
# with nelem = nx*ny + ny*nz + nz*nx surface elements (ASOC.py:909-944)
# ---------------------------------------------------------------------------------------

def roi_elements(n):
    return int(n[0]) * int(n[1]) + int(n[1]) * int(n[2]) + int(n[2]) * int(n[0])


def open_roi_load(filename, ROI_NSIDE, NFREQ):
    """memory map of a record to load.  Returns (DIM[3], data[NFREQ, nelem*12*nside^2])."""
    hdr = np.fromfile(filename, np.int32, 5)
    if hdr.size != 5:
        raise ValueError("ROI file %s: short header" % filename)
    if hdr[3] != ROI_NSIDE:
        raise ValueError("ROI file %s has nside %d, ini-file has %d" % (filename, hdr[3], ROI_NSIDE))
    if hdr[4] != NFREQ:
        raise ValueError("ROI file %s has %d, current run %d frequencies" % (filename, hdr[4], NFREQ))
    n = roi_elements(hdr[:3]) * 12 * ROI_NSIDE * ROI_NSIDE
    return np.asarray(hdr[:3], np.int32), np.memmap(filename, dtype=np.float32, mode='r', offset=20, shape=(NFREQ, n))


def create_roi_save(filename, ROI, ROI_STEP, ROI_NSIDE, NFREQ):
    """header + zeroed record on disk, as a writable memory map [NFREQ, nelem*12*nside^2] (ASOC.py:927-940)"""
    n = [(int(ROI[2 * i + 1]) - int(ROI[2 * i]) + 1) * int(ROI_STEP) for i in range(3)]
    np.asarray(n + [ROI_NSIDE, NFREQ], np.int32).tofile(filename)
    npix = roi_elements(n) * 12 * ROI_NSIDE * ROI_NSIDE
    with open(filename, "ab") as fp:
        fp.truncate(20 + 4 * NFREQ * npix)
    m = np.memmap(filename, dtype=np.float32, mode='r+', offset=20, shape=(NFREQ, npix))
    m[:, :] = 0.0
    return m


# ---- intensity file (saveint / dustem keys; ASOC.py:990-1000, :2733-2757) ----------------------------------------
def create_intensity_file(path, CELLS, NFREQ, vectors):
    """Memory-mapped float32 body of the intensity file, zeroed: [CELLS, NFREQ] behind an 8-byte header (saveint 1) or
    [CELLS, NFREQ, 4] = (I, Ix, Iy, Iz) behind a 12-byte header (saveint 2).  The header is written by
    finish_intensity_file, as in the reference."""
    shape, off = ((CELLS, NFREQ, 4), 12) if vectors else ((CELLS, NFREQ), 8)
    m = np.memmap(path, dtype=np.float32, mode="w+", shape=shape, offset=off)
    m[...] = 0.0
    return m


def finish_intensity_file(path, INTENSITY, CELLS, NFREQ, vectors):
    """saveint 2: Ix, Iy, Iz become fractions of the total intensity (ASOC.py:2736-2738); then the header"""
    if vectors:
        for k in (1, 2, 3):
            INTENSITY[:, :, k] /= (INTENSITY[:, :, 0] + 1.0e-33)
    INTENSITY.flush()
    del INTENSITY
    with open(path, "r+b") as fp:
        np.asarray([CELLS, NFREQ, 4] if vectors else [CELLS, NFREQ], np.int32).tofile(fp)


def read_intensity(path):
    """-> [CELLS, NFREQ] or [CELLS, NFREQ, 4] of the file saveint writes"""
    with open(path, "rb") as fp:
        cells, nfreq = (int(v) for v in np.fromfile(fp, np.int32, 2))
        rest = os.path.getsize(path) - 8
        if rest == 4 * cells * nfreq:
            return np.fromfile(fp, np.float32).reshape(cells, nfreq)
        four = int(np.fromfile(fp, np.int32, 1)[0])
        if four != 4 or rest - 4 != 16 * cells * nfreq:
            raise FileError("%s: not an intensity file" % path)
        return np.fromfile(fp, np.float32).reshape(cells, nfreq, 4)


# ---- FITS images (fits key; the reference builds them with astropy.io.fits through ASOC_aux.MakeFits, :1723-1790) ----
def _fits_card(key, value, comment=""):
    if isinstance(value, bool):
        v = "%20s" % ("T" if value else "F")
    elif isinstance(value, (int, np.integer)):
        v = "%20d" % int(value)
    elif isinstance(value, (float, np.floating)):
        t = repr(float(value)).upper()
        if "E" not in t and "." not in t and "INF" not in t and "NAN" not in t:
            t += ".0"
        v = "%20s" % t
    else:
        v = "'%-8s'" % str(value).replace("'", "''")
        v = "%-20s" % v
    card = "%-8s= %s" % (key, v)
    if comment:
        card += " / " + comment
    return card[:80].ljust(80)


def write_fits(path, data, lon, lat, pix, freq=(), galactic=False):
    """One primary HDU with the header MakeFits gives its images: data[n, m] (or [nchn, n, m] with `freq`), float32
    big-endian; tangent projection centred on (lon, lat) [deg], pixel `pix` [rad].  Written by hand (80-character cards,
    2880-byte blocks) -- the image has no astropy."""
    a = np.asarray(data, np.float32)
    cube = a.ndim == 3
    n, m = a.shape[-2], a.shape[-1]
    cards = [_fits_card("SIMPLE", True, "conforms to FITS standard"), _fits_card("BITPIX", -32, "array data type"),
             _fits_card("NAXIS", 3 if cube else 2, "number of array dimensions"), _fits_card("NAXIS1", m), _fits_card("NAXIS2", n)]
    if cube:
        cards.append(_fits_card("NAXIS3", a.shape[0]))
    cards += [_fits_card("EXTEND", True),
              _fits_card("CRVAL1", float(lon)), _fits_card("CRVAL2", float(lat)),
              _fits_card("CDELT1", -float(pix) * 180.0 / np.pi), _fits_card("CDELT2", float(pix) * 180.0 / np.pi),
              _fits_card("CRPIX1", 0.5 * (m + 1) + 0.5), _fits_card("CRPIX2", 0.5 * (n + 1) + 0.5)]
    if galactic:
        cards += [_fits_card("CTYPE1", "GLON-TAN"), _fits_card("CTYPE2", "GLAT-TAN"), _fits_card("COORDSYS", "GALACTIC")]
    else:
        cards += [_fits_card("CTYPE1", "RA---TAN"), _fits_card("CTYPE2", "DEC--TAN"), _fits_card("COORDSYS", "EQUATORIAL"),
                  _fits_card("EQUINOX", 2000.0)]
    if cube:
        cards += [_fits_card("CRPIX3", 1), _fits_card("CRVAL3", 0.0), _fits_card("CDELT3", 1), _fits_card("CTYPE3", "channel")]
        for i, f in enumerate(freq):
            cards.append(("COMMENT F[ %3d ] = %.4e" % (i, float(f))).ljust(80))
    cards.append("END".ljust(80))
    head = "".join(cards)
    head += " " * (-len(head) % 2880)
    body = a.astype(">f4").tobytes()
    body += b"\0" * (-len(body) % 2880)
    with open(path, "wb") as fp:
        fp.write(head.encode("ascii"))
        fp.write(body)


def read_fits(path):
    """-> (header dict, data) of a file write_fits wrote (primary HDU, BITPIX -32)"""
    with open(path, "rb") as fp:
        raw = fp.read()
    hdr, pos, done = {}, 0, False
    comments = []
    while not done:
        block = raw[pos:pos + 2880].decode("ascii")
        pos += 2880
        for i in range(0, 2880, 80):
            card = block[i:i + 80]
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if key == "COMMENT":
                comments.append(card[8:].strip())
            elif card[8:10] == "= ":
                v = card[10:].split(" / ")[0].strip()
                if v.startswith("'"):
                    hdr[key] = v.strip("'").strip()
                elif v in ("T", "F"):
                    hdr[key] = (v == "T")
                else:
                    hdr[key] = float(v) if any(c in v for c in ".EN") else int(v)
    if hdr["BITPIX"] != -32:
        raise FileError("%s: BITPIX %s" % (path, hdr["BITPIX"]))
    shape = [hdr["NAXIS%d" % k] for k in range(hdr["NAXIS"], 0, -1)]
    n = int(np.prod(shape))
    data = np.frombuffer(raw, ">f4", n, pos).astype(np.float32).reshape(shape)
    hdr["COMMENT"] = comments
    return hdr, data
