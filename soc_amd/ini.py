"""ini-file interface of the absorption / scattering runs.

Keeps the keyword set, the prefix matching and the defaults of the reference parser
(``User.__init__``, ASOC_aux.py:79-553; keyword table in SURVEY.md Appendix A) so that an
existing ``my.ini`` drives this engine unchanged.  The parser is written fresh: a table of
(prefix, minimum argument count, handler) replaces the chain of ``key.find(...)==0`` tests;
what it must reproduce is the matching rule -- the FIRST token is compared by PREFIX,
case-sensitively (argument-less switches are lower-cased first), every handler whose prefix
matches fires, and the raw tokens of every line are kept in ``KEYS``.
"""
import math

import numpy as np

from .launch import um2f

MAXPS = 4000
D2R = 0.0174532925


class IniError(ValueError):
    pass


class User:
    """Run parameters.  Attribute names are the reference's (ASOC_aux.py:84-236)."""

    def __init__(self, filename=None, text=None):
        self._defaults()
        if filename is not None:
            with open(filename) as fp:
                text = fp.read()
        if text is not None:
            for line in text.splitlines():
                self._parse_line(line)
        # DFPAC==CLPAC or DFPAC>0 when CLPAC==0 (ASOC_aux.py:534)
        if self.CLPAC > 0:
            self.DFPAC = self.CLPAC

    # -------------------------------------------------------------------------------
    def _defaults(self):
        s = self
        s.file_cloud = ''
        s.file_diffuse = ''
        s.file_background = ''
        s.file_constant_load = ''
        s.file_constant_save = ''
        s.file_external_mask = ''
        s.file_optical = []
        s.file_scafunc = []
        s.file_abundance = []
        s.file_hpbg = ''
        s.HPBG_WEIGHTED = False
        s.file_absorbed = 'default.absorbed'
        s.file_emitted = 'soc.emitted'
        s.file_sourcemap = ''
        s.file_temperature = ''
        s.file_savetau = ''
        s.file_pssavetau = ''
        s.file_scattering = 'scattering'
        s.kernel_defs = ''
        s.GL = 0.0
        s.MAP_DX = 1.0
        s.KDENSITY = 1.0
        s.DISTANCE = 0.0
        s.ITERATIONS = 1
        s.STEP_WEIGHT = [-1, 0, 0]
        s.DIR_WEIGHT = [-1, 0, 0]
        s.NPIX = (10, 10)
        s.FAST_MAP = -1
        s.REMIT_F = [0.0, 1e30]
        s.SIM_F = [1.0e8, 1.0e17]
        s.LEVEL_THRESHOLD = 0
        s.INTOBS = (-1e12, 0.0, 0.0)
        s.MAPCENTRE = (-1e12, 0.0, 0.0)
        s.DEVICES = 'c'
        s.sDEVICE = ''
        s.FISSION = 0
        s.DSC_BINS = 0
        s.LOCAL = -1
        s.GLOBAL = -1
        s.BATCH = 30
        s.OBS_THETA = []
        s.OBS_PHI = []
        s.PSPAC = 0
        s.PS_METHOD = 0
        s.BGPAC = 0
        s.CLPAC = 0
        s.DFPAC = 0
        s.NO_PS = 0
        s.file_pointsource = []
        s.PS_SCALING = np.ones(MAXPS, np.float32)
        s.PSPOS = np.zeros((MAXPS, 4), np.float32)      # cl float3 = 4 floats
        s.PSPOS[:, 0] = -1e10
        s.DO_SPLIT = 0
        s.POLMAP = 0
        s.POLSIM = 0
        s.BFILES = []
        s.POLSTAT = 0
        s.NOSOLVE = 0
        s.LOAD_TEMPERATURE = 0
        s.NOMAP = 0
        s.NOABSORBED = 0
        s.SAVE_INTENSITY = 0
        s.SAVE_INTENSITY_FILE = 'ISRF.DAT'
        s.USE_EMWEIGHT = 0
        s.EMWEIGHT_SKIP = 3
        s.EMWEIGHT_LIM = [0.0, 1e10, 0.0]
        s.p0 = 0.2
        s.MAXLOS = 1e10
        s.MINLOS = -1.0
        s.Y_SHEAR = 0.0
        s.INTERPOLATE = 0
        s.SEED = math.pi / 4.0
        s.MAP_FREQ = [1.0e6, 1e18]
        s.SINGLE_MAP_FREQ = np.asarray([], np.float32)
        s.SOLVE_ON_DEVICE = 0
        s.FFS = 1
        s.BG_METHOD = 0
        s.WITH_ALI = 0
        s.WITH_REFERENCE = 0
        s.scale_background = 1.0
        s.LEVELS = 999
        s.KEYS = {}
        s.PLATFORM = -1
        s.IDEVICE = 0
        s.K_DIFFUSE = 1.0
        s.SINGLE_ABU = 0
        s.OPT_IS_HALF = 0
        s.POL_RHO_WEIGHT = 0
        s.savetau_freq = []
        s.pssavetau_freq = -1.0
        s.ROI = np.zeros(6, np.int32)
        s.ROI_STEP = 0
        s.ROI_MAP = 0
        s.ROI_NSIDE = 16
        s.WITH_ROI_SAVE = 0
        s.WITH_ROI_LOAD = 0
        s.ROI_LOAD_SCALE = 1.0
        s.FILE_ROI_SAVE = ''
        s.FILE_ROI_LOAD = ''
        s.ROIPAC = 0
        s.OUT_NSIDE = 128
        s.MAP_INTERPOLATION = 0
        s.FITS = 0
        s.FITS_PREFIX = 'map'
        s.FITS_RA = 0.0
        s.FITS_DE = 0.0
        s.MIRROR = ''
        s.VERBOSE = 1
        s.MMAP_ABSORBED = 0
        s.MMAP_EMITTED = 0
        s.CR_HEATING = 0.0
        s.ABSTHIN = -1
        s.NNNLIMIT = 0.0
        s.DUST_FILE = ''
        s.ALIGN_DAT = ''
        # filled in by the readers
        s.AREA = 0.0
        s.AXY = s.AXZ = s.AYZ = 0.0
        s.NFREQ = -1
        s.FFREQ, s.FABS, s.FSCA = [], [], []
        s.CELLS = 0

    # -------------------------------------------------------------------------------
    def _parse_line(self, line):
        s = line.split('#')[0].split()
        if len(s) < 1:
            return
        key0 = s[0]
        if key0 == 'DEFS':
            body = line[4:]
            self.kernel_defs = body[:body.index('#')] if '#' in body else body
        if key0.find('mapum') == 0:
            f = [um2f(float(x)) for x in s[1:]]
            self.SINGLE_MAP_FREQ = np.sort(np.concatenate((self.SINGLE_MAP_FREQ, np.asarray(f, np.float32))))
        if key0 == 'singleabu':
            self.SINGLE_ABU = 1
        if key0 == 'optishalf':
            self.OPT_IS_HALF = 1
        self.KEYS[key0] = s[1:]

        # switches without arguments: lower-cased prefix match (ASOC_aux.py:273-285)
        low = key0.lower()
        for prefix, attrs in _FLAGS:
            if low.find(prefix) == 0:
                for a, v in attrs:
                    setattr(self, a, v)
        if low.find('savetau') == 0 and len(s) > 2:
            self.file_savetau = s[1]
            for x in s[2:]:
                self.savetau_freq.append(0.0 if float(x) < 0.0 else um2f(float(x)))
        if low.find('pssavetau') == 0:
            self.file_pssavetau = s[1]
            self.pssavetau_freq = um2f(float(s[2]))
        if low.find('fits') == 0:
            self.FITS = 1
            if len(s) >= 3:
                self.FITS_RA, self.FITS_DE = float(s[1]), float(s[2])
                if len(s) >= 4:
                    self.FITS_PREFIX = s[3]
        if low.find('mirror') == 0 and len(s) > 1:
            self.MIRROR = s[1]

        # keywords with arguments: case-sensitive prefix match on the first token
        for prefix, nargs, handler in _KEYWORDS:
            if len(s) > nargs and key0.find(prefix) == 0:
                handler(self, s)
        if key0 == 'roi' and len(s) >= 7:
            self.ROI = np.asarray([int(v) for v in s[1:7]], np.int32)

    # -------------------------------------------------------------------------------
    def Validate(self):
        """ASOC_aux.py:539-550"""
        ok = True
        if len(self.file_cloud) < 1:
            print("*** Cloud model not definied: keyword cloud")
            ok = False
        if (self.CLPAC < 1) and (self.WITH_ALI > 0):
            print("*** WARNING:  CLPAC=0 and WITH_ALI=%d -> WITH_ALI=0" % self.WITH_ALI)
            self.WITH_ALI = 0
        if self.PSPAC < 1:
            self.NO_PS = 0
        return ok


# ---- handlers ------------------------------------------------------------------------

def _set(attr, conv, idx=1):
    def h(u, s):
        setattr(u, attr, conv(s[idx]))
    return h


def _device(u, s):
    u.DEVICES = s[1].lower()
    if len(s) > 2:
        u.sDEVICE = s[2]


def _platform(u, s):
    u.PLATFORM = int(s[1])
    if len(s) > 2:
        try:
            u.IDEVICE = int(s[2])
        except ValueError:
            u.IDEVICE = 0


def _diffuse(u, s):
    u.file_diffuse = s[1]
    if len(s) > 2:
        u.K_DIFFUSE = float(s[2])


def _optical(u, s):
    u.file_optical.append(s[1])
    if len(s) > 2 and s[2][0:1] != '#':
        u.file_abundance.append(s[2])
    else:
        u.file_abundance.append('#')


def _background(u, s):
    u.file_background = s[1]
    if len(s) > 2:
        u.scale_background = float(s[2])


def _hpbg(u, s):
    u.file_hpbg = s[1]
    if len(s) > 2:
        u.scale_background = float(s[2])
    if len(s) > 3:
        u.HPBG_WEIGHTED = int(s[3])


def _saveint(u, s):
    u.SAVE_INTENSITY = int(s[1])
    if len(s) > 2:
        u.SAVE_INTENSITY_FILE = s[2]


def _emwei(u, s):
    u.USE_EMWEIGHT = int(s[1])
    if len(s) > 3:
        u.EMWEIGHT_LIM = [float(s[2]), float(s[3]), 0.0]
        if len(s) > 4:
            u.EMWEIGHT_LIM[2] = float(s[4])
            if len(s) > 5:
                u.EMWEIGHT_SKIP = int(s[5])


def _dsc(u, s):
    u.file_scafunc.append(s[1])
    if len(u.file_scafunc) == 1:
        u.DSC_BINS = int(s[2])
    elif u.DSC_BINS != int(s[2]):
        raise IniError("scattering functions: number of bins must be the same for all dusts")


def _direct(u, s):
    if len(u.OBS_THETA) >= 10:
        raise IniError("cannot have more than 10 directions")
    u.OBS_THETA.append(float(s[1]) * D2R)
    u.OBS_PHI.append(float(s[2]) * D2R)


def _roisave(u, s):
    u.WITH_ROI_SAVE, u.FILE_ROI_SAVE, u.ROI_STEP = 1, s[1], int(s[2])


def _roiload(u, s):
    u.WITH_ROI_LOAD, u.FILE_ROI_LOAD, u.ROI_LOAD_SCALE = 1, s[1], float(s[2])


def _polsim(u, s):
    u.POLSIM, u.BFILES = 1, [s[1], s[2], s[3]]


def _polmap(u, s):
    u.POLMAP, u.BFILES = 1, [s[1], s[2], s[3]]
    if len(s) == 5:
        u.MAXLOS = float(s[4])
    if len(s) > 5:
        u.MINLOS, u.MAXLOS = float(s[4]), float(s[5])


def _mapping(u, s):
    u.NPIX = (int(s[1]), int(s[2]))
    u.MAP_DX = float(s[3])
    if len(s) > 4:
        try:
            u.FAST_MAP = int(s[4])
        except ValueError:
            pass


def _mapview(u, s):
    u.OBS_THETA = [float(s[1]) * math.pi / 180.0]
    u.OBS_PHI = [float(s[2]) * math.pi / 180.0]
    if len(s) >= 5:
        u.NPIX = (int(s[3]), int(s[4]))
        if len(s) >= 6:
            u.MAP_DX = float(s[5])
            if len(s) >= 9:
                u.MAPCENTRE = (float(s[6]), float(s[7]), float(s[8]))


def _pointsource(u, s):
    if u.NO_PS >= MAXPS:
        raise IniError("reached maximum number of point sources = %d" % MAXPS)
    u.PSPOS[u.NO_PS, 0:3] = [float(s[1]), float(s[2]), float(s[3])]
    u.file_pointsource.append(s[4])
    if len(s) > 5 and s[5] != '#':
        u.PS_SCALING[u.NO_PS] = float(s[5])
    u.NO_PS += 1


_FLAGS = [
    ('nosolve', [('NOSOLVE', 1)]),
    ('loadtemp', [('LOAD_TEMPERATURE', 1)]),
    ('nomap', [('NOMAP', 1)]),
    ('noabs', [('NOABSORBED', 1)]),
    ('dustem', [('NOABSORBED', 1), ('SAVE_INTENSITY', 1)]),
    ('solveondev', [('SOLVE_ON_DEVICE', 1)]),
    ('xemonhost', [('XEM_ON_HOST', 1)]),
    ('polrhoweight', [('POL_RHO_WEIGHT', 1)]),
    ('roimap', [('ROI_MAP', 1)]),
]

# (prefix, minimum number of arguments, handler); order follows ASOC_aux.py:313-525
_KEYWORDS = [
    ('device', 1, _device),
    ('fission', 1, _set('FISSION', int)),
    ('verbose', 1, _set('VERBOSE', int)),
    ('mmapabs', 1, _set('MMAP_ABSORBED', int)),
    ('mmapemit', 1, _set('MMAP_EMITTED', int)),
    ('sourcemap', 1, _set('file_sourcemap', str)),
    ('tempera', 1, _set('file_temperature', str)),
    ('cloud', 1, _set('file_cloud', str)),
    ('absorb', 1, _set('file_absorbed', str)),
    ('scatter', 1, _set('file_scattering', str)),
    ('emit', 1, _set('file_emitted', str)),
    ('split', 1, _set('DO_SPLIT', int)),
    ('mapint', 1, _set('MAP_INTERPOLATION', int)),
    ('polstat', 1, _set('POLSTAT', int)),
    ('absthin', 1, _set('ABSTHIN', int)),
    ('nnnlimit', 1, _set('NNNLIMIT', float)),
    ('dustfile', 1, _set('DUST_FILE', str)),
    ('radiusalign', 1, _set('ALIGN_DAT', str)),
    ('platform', 1, _platform),
    ('diffus', 1, _diffuse),
    ('optic', 1, _optical),
    ('externalm', 1, _set('file_external_mask', str)),
    ('backg', 1, _background),
    ('hpbg', 1, _hpbg),
    ('polred', 1, _set('file_polred', str)),
    ('cload', 1, _set('file_constant_load', str)),
    ('csave', 1, _set('file_constant_save', str)),
    ('iterations', 1, _set('ITERATIONS', int)),
    ('threshold', 1, _set('LEVEL_THRESHOLD', int)),
    ('gridlen', 1, _set('GL', float)),
    ('p0', 1, _set('p0', float)),
    ('distance', 1, _set('DISTANCE', float)),
    ('bgpac', 1, _set('BGPAC', lambda a: int(float(a)))),
    ('pspac', 1, _set('PSPAC', lambda a: int(float(a)))),
    ('psmetho', 1, _set('PS_METHOD', int)),
    ('cellpac', 1, _set('CLPAC', lambda a: int(round(float(a))))),
    ('roipac', 1, _set('ROIPAC', lambda a: int(round(float(a))))),
    ('roinside', 1, _set('ROI_NSIDE', lambda a: int(round(float(a))))),
    ('diffpac', 1, _set('DFPAC', int)),
    ('seed', 1, _set('SEED', lambda a: float(np.clip(float(a), -1.0, 1.0)))),
    ('dens', 1, _set('KDENSITY', float)),
    ('CR_HEATING', 1, _set('CR_HEATING', float)),
    ('batch', 1, _set('BATCH', int)),
    ('local', 1, _set('LOCAL', int)),
    ('global', 1, _set('GLOBAL', int)),
    ('forcedfirst', 1, _set('FFS', int)),
    ('ffs', 1, _set('FFS', int)),
    ('bgmethod', 1, _set('BG_METHOD', int)),
    ('ali', 1, _set('WITH_ALI', int)),
    ('reference', 1, _set('WITH_REFERENCE', int)),
    ('saveint', 1, _saveint),
    ('levels', 1, _set('LEVELS', int)),
    ('yshear', 1, _set('Y_SHEAR', float)),
    ('interpol', 1, _set('INTERPOLATE', float)),
    ('outnside', 1, _set('OUT_NSIDE', int)),
    ('emwei', 1, _emwei),
    # two arguments
    ('remit', 2, lambda u, s: setattr(u, 'REMIT_F', [um2f(float(s[2])), um2f(float(s[1]))])),
    ('simum', 2, lambda u, s: setattr(u, 'SIM_F', [um2f(float(s[2])), um2f(float(s[1]))])),
    ('dsc', 2, _dsc),
    ('direwei', 2, lambda u, s: setattr(u, 'DIR_WEIGHT', [int(s[1]), float(s[2])])),
    ('direct', 2, _direct),
    ('wavelen', 2, lambda u, s: setattr(u, 'MAP_FREQ', [um2f(float(s[2])), um2f(float(s[1]))])),
    ('roisave', 2, _roisave),
    ('roiload', 2, _roiload),
    # three arguments
    ('polsim', 3, _polsim),
    ('polmap', 3, _polmap),
    ('perspec', 3, lambda u, s: setattr(u, 'INTOBS', (float(s[1]), float(s[2]), float(s[3])))),
    ('stepwei', 3, lambda u, s: setattr(u, 'STEP_WEIGHT', [int(s[1]), float(s[2]), float(s[3])])),
    ('mapping', 3, _mapping),
    ('mapcent', 3, lambda u, s: setattr(u, 'MAPCENTRE', (float(s[1]), float(s[2]), float(s[3])))),
    ('mapview', 3, _mapview),
    # four arguments
    ('pointsou', 4, _pointsource),
]
