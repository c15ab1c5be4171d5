#!/usr/bin/env python3
"""asoc -- the absorption run of SOC on MI355X:  python -m soc_amd.asoc my.ini

Drop-in for the photon-packet part of ``ASOC.py <ini>`` (reference ASOC.py:77-1560): reads
the same ini file and the same cloud / dust / dsc / background / point-source / diffuse
files, simulates the constant radiation sources frequency by frequency on the HIP engine
and writes the same products of that stage: ``packet.info``, the ``absorbed`` file
(unless ``noabsorbed``) and the ``csave`` file of frequency-integrated absorptions.
Temperature solve, emission and maps are downstream of the tallies and
not part of this engine; with ``noabsorbed`` the integrated absorptions are written to
``<prefix>.ctabs`` so a downstream solver can pick them up.

The host loop is the reference's (source block II -> frequency IFREQ, ASOC.py:1028-1545);
what runs inside is new: geometry and feature switches are run-time arguments of one
ahead-of-time compiled library (no per-run kernel build), and with several ranks each
launch is split by logical work-item id with one all-reduce of the tallies
(soc_amd/dist.py).
"""
import sys
import time

import numpy as np

from . import files, launch
from .ini import User
from .launch import PLANCK, PARSEC


class UnsupportedOption(RuntimeError):
    pass


def _check_supported(USER, NDUST, WITH_MSF):
    bad = []
    if USER.DO_SPLIT:
        bad.append("split")
    if int(USER.STEP_WEIGHT[2]) > 2:
        bad.append("stepweight with a third argument > 2 (the kernel then uses an uninitialised free path, kernel_ASOC.c:516-535)")
    if USER.DIR_WEIGHT[0] > 0:
        bad.append("direweight (-D DIR_WEIGHT > 0 does not compile in the reference: pweight, pind undeclared, kernel_ASOC.c:770-775)")
    if USER.PS_METHOD == 3:
        bad.append("psmethod 3 (does not compile in the reference either)")
    if USER.WITH_REFERENCE and USER.SAVE_INTENSITY > 0:
        bad.append("saveint with the reference field (ASOC.py:994 asserts against it: the tallies hold differences)")
    if 'SUBITERATIONS' in USER.KEYS:
        bad.append("SUBITERATIONS (sub-iterations of the reference field, ASOC.py:2261-2720; there the branch starts from "
                   "`TOLD = 0.0*TNEW` with TNEW = None unless `loadtemp` is given, ASOC.py:700, :2282)")
    # keys the parser knows (soc_amd/ini.py keeps the reference's keyword set) whose effect is not built: refused, so that
    # an ini file using them stops here instead of finishing with products missing or different
    if 'nnmake' in USER.KEYS and USER.ABSTHIN > 1 and USER.MMAP_ABSORBED > 0:
        bad.append("nnmake with absthin and mmapabs (the reference adds thinned rows to a full-size memory map there and stops, ASOC.py:623-630, :1496)")
    if 'nnmake' in USER.KEYS and USER.ABSTHIN > 1 and USER.WITH_REFERENCE and not USER.NOABSORBED:
        bad.append("nnmake with absthin and the reference field (not built)")
    if USER.POLMAP or USER.POLSIM or len(USER.BFILES) > 0 or len(getattr(USER, "file_polred", "")) > 0:
        bad.append("polmap / polred / magnetic-field files (polarisation maps)")
    if USER.FAST_MAP >= 2:
        bad.append("mapping with a fourth argument >= 2 (FAST_MAP 2..998: kernel_ASOC_map_X.c, all frequencies per launch -- the reference's "
                   "own branch stops at ASOC.py:3553, a list compared with a float; >= 999: one map per hierarchy level, kernel_ASOC_map_H.c)")
    if USER.MAP_INTERPOLATION > 2 or USER.MAP_INTERPOLATION < 0:
        bad.append("mapint other than 0, 1, 2 (kernel_ASOC_map.c:656-810 knows those: a larger value leaves Adens, Aemit ... unset there)")
    if len(USER.kernel_defs.strip()) > 0:
        bad.append("DEFS (extra -D options for the OpenCL compiler)")
    # accepted without effect, because they have none in the reference either: `interpolate` and `yshear` reach only the
    # per-level map kernel (kernel_ASOC_map_H.c, FAST_MAP >= 999: refused above), `externalmask` only the SUBITERATIONS
    # branch (refused above), `sourcemap` is parsed and never read (ASOC_aux.py:322), `bgmethod` is a -D that no kernel tests
    # (`loadtemp` with iterations > 0 has no effect in the reference: the temperatures read are replaced before any use --
    # the block that would use them with ALI, ASOC.py:2062-2071, is switched off there -- so it has none here)
    if bad:
        raise UnsupportedOption("ini options not supported by this engine: " + ", ".join(bad))


class AbsorptionRun:
    """The constant-source part of an ASOC run.  ``engine`` is a soc_amd.lib.Engine (or an
    object with the same methods); ``comm`` a soc_amd.dist.Comm."""

    def __init__(self, USER, engine, comm=None, verbose=None, workdir=".", shard="items"):
        """shard (several ranks): "items" -- every launch is split by work-item ranges (identical packets and events per rank, one
        all-reduce of the per-cell buffer per frequency when absorptions are saved); "launches" -- the launches themselves are
        dealt out: a run that keeps the per-frequency absorptions gives every frequency to ONE rank, which owns that column of
        the absorbed file (no collective for INT at all; TABS is reduced once); a TABS-only run gives every rank a contiguous
        share of the launch sequence (launch.shard_launches).  Same packets and streams either way."""
        self.U = USER
        self.eng = engine
        self.comm = comm
        if shard not in ("items", "launches"):
            raise ValueError("shard: 'items' or 'launches'")
        self.shard = shard
        self.rank = comm.rank if comm else 0
        self.world = comm.world if comm else 1
        self.verbose = USER.VERBOSE if verbose is None else verbose
        self.timers = dict(Tkernel=0.0, Tpush=0.0, Tpull=0.0)
        self.packets = 0
        self._load_inputs()

    def log(self, *a):
        if self.verbose and self.rank == 0:
            print(*a)

    # ---------------------------------------------------------------------------------
    def _load_inputs(self):
        U = self.U
        if not U.Validate():
            raise ValueError("check the ini file: no cloud defined")
        if U.GL <= 0.0:
            raise ValueError("gridlength must be given")
        self.FFREQ, self.AFG, self.AFABS, self.AFSCA = files.read_dust(U.file_optical, U.GL)
        self.NFREQ = U.NFREQ = len(self.FFREQ)
        self.NDUST = len(self.AFABS)
        if self.NFREQ < 2:
            raise ValueError("the dust file needs >= 2 frequencies (trapezoid weights, ASOC.py:1220); "
                             "restrict the simulated range with `simum` instead")
        self.FDSC, self.FCSC = files.read_scattering_functions(U.file_scafunc, self.NFREQ, U.DSC_BINS)
        self.WITH_MSF = WITH_MSF = len(self.FDSC) > 1
        _check_supported(U, self.NDUST, WITH_MSF)
        self.IBG = files.read_background_intensity(U.file_background, self.NFREQ, U.scale_background) \
            if U.BGPAC > 0 else []
        self.LPS = files.read_source_luminosities(U.file_pointsource[:U.NO_PS], self.NFREQ, U.PS_SCALING) \
            if U.NO_PS > 0 else []
        self.HPBG = []
        if len(U.file_hpbg) > 2:                                   # ASOC.py:291-297, NSIDE 64 fixed
            self.HPBG = np.fromfile(U.file_hpbg, np.float32).reshape(self.NFREQ, 49152) * np.float32(U.scale_background)
        self.cloud = files.read_cloud(U.file_cloud, U.KDENSITY, U.LEVELS)
        c = self.cloud
        U.AREA, U.CELLS = float(c.AREA), c.CELLS
        self.log("NX %d, NY %d, NZ %d LEVELS %d, CELLS %d" % (c.NX, c.NY, c.NZ, c.LEVELS, c.CELLS))
        self.ABU = files.read_abundances(U.file_abundance, c.CELLS)
        self.WITH_ABU = self.ABU is not None
        if self.WITH_ABU and U.SINGLE_ABU:
            if self.NDUST != 2:
                raise ValueError("singleabu assumes exactly two dust components")
            if self.WITH_MSF:
                raise ValueError("singleabu cannot be used with several scattering functions (ASOC.py:159-161)")
            self.ABU = np.ravel(self.ABU[:, 0])
        if self.WITH_MSF and not self.WITH_ABU:
            raise ValueError("cannot have multiple scattering functions without multiple dusts with variable abundances (ASOC.py:168-170)")
        if self.WITH_MSF and len(self.FDSC) != self.NDUST:
            raise ValueError("%d dsc files for %d dust species" % (len(self.FDSC), self.NDUST))
        self.DIFFUSERAD = files.mmap_diffuserad(U.file_diffuse, c.CELLS) if len(U.file_diffuse) > 0 else []

        # LOCAL only enters through the rounding of packet counts (ASOC.py:221-227)
        LOCAL = 8 if 'c' in U.DEVICES else 32
        if 'local' in U.KEYS:
            LOCAL = int(U.KEYS['local'][0])
        self.LOCAL = LOCAL
        pc = launch.packet_counts(U.BGPAC, U.PSPAC, U.CLPAC, U.DFPAC, int(U.AREA), c.CELLS, LOCAL, U.USE_EMWEIGHT)
        self.PSPAC, self.BGPAC, self.CLPAC, self.DFPAC = pc["PSPAC"], pc["BGPAC"], pc["CLPAC"], pc["DFPAC"]
        if U.ITERATIONS < 1:
            U.NOABSORBED = 1
        self.log('PACKETS: PSPAC %d   BGPAC %d  CLPAC %d  DFPAC %d' % (self.PSPAC, self.BGPAC, self.CLPAC, self.DFPAC))
        self.XPS = files.analyse_external_point_sources(c.NX, c.NY, c.NZ, U.PSPOS, int(U.NO_PS), int(U.PS_METHOD))
        # launch size for point-source / cell-emission launches: reference default 32768
        # (ASOC.py:86), `global` keyword overrides (more work items fill an MI355X better)
        self.GLOBAL_0 = U.GLOBAL if U.GLOBAL > 0 else launch.GLOBAL_0
        self.with_int = 2 if U.SAVE_INTENSITY == 2 else int((U.SAVE_INTENSITY == 1) or (not U.NOABSORBED))
        if U.SAVE_INTENSITY > 0 and self.WITH_ABU:
            # ASOC.py:1499-1515 divides by ABS, which the abundance branch never sets (:1146-1165 fill OPT only)
            raise UnsupportedOption("saveint with an abundance file (the reference scales the intensity by 1/ABS = 1/0 there)")
        self.INTENSITY = None

    def write_packet_info(self, path="packet.info"):
        """int32 [BGPAC, PSPAC, DFPAC, CLPAC] (ASOC.py:251)"""
        if self.rank == 0:
            np.asarray([self.BGPAC, self.PSPAC, self.DFPAC, self.CLPAC], np.int32).tofile(path)

    # ---------------------------------------------------------------------------------
    def setup_engine(self):
        e, c, U = self.eng, self.cloud, self.U
        e.set_cloud(c)
        e.set_features(with_int=self.with_int, ps_method=U.PS_METHOD, use_emweight=min(max(U.USE_EMWEIGHT, 0), 2))
        e.set_mirror(launch.mirror_mask(U.MIRROR))
        # `stepweight a b c`: the reference hands the kernels -D SW_A=int(a) -D SW_B=b -D STEP_WEIGHT=int(c), each float
        # written with %.3e (ASOC.py:348,357) -- the drop-in passes the same values
        sw = int(U.STEP_WEIGHT[2])
        if sw > 0 or hasattr(e, "set_step_weight"):
            e.set_step_weight(sw, float("%.3e" % int(U.STEP_WEIGHT[0])), float("%.3e" % U.STEP_WEIGHT[1]))
        if U.CR_HEATING > 0 or hasattr(e, "set_cr_heating"):
            # -D CR_HEATING=%d -D CR_HEATING_RATE=%.3ef with (USER.CR_HEATING>0), USER.CR_HEATING (ASOC.py:352,362): device solve only,
            # as in the reference (its host loop, used with ALI, has no such term)
            e.set_cr_heating(float("%.3e" % U.CR_HEATING) if U.CR_HEATING > 0 else 0.0)
        if U.ROI_MAP or hasattr(e, "set_map_roi"):
            e.set_map_roi(U.ROI if U.ROI_MAP else None)              # -D ROI_MAP (ASOC.py:345,354; :3126-3133)
        if U.LEVEL_THRESHOLD > 0 or hasattr(e, "set_map_threshold"):
            e.set_map_threshold(max(0, int(U.LEVEL_THRESHOLD)))      # -D LEVEL_THRESHOLD (ASOC.py:349,359)
        if U.MAP_INTERPOLATION > 0 or hasattr(e, "set_map_interpolation"):
            e.set_map_interpolation(int(U.MAP_INTERPOLATION))        # -D MAP_INTERPOLATION (ini key mapint; ASOC.py:352,362)
        if self.WITH_ABU:
            if U.OPT_IS_HALF or hasattr(e, "set_opt_half"):
                e.set_opt_half(bool(U.OPT_IS_HALF))            # OPT as fp16 (ASOC.py:1158-1159)
            e.set_abundances(self.ABU, single=bool(U.SINGLE_ABU))
        if self.comm:
            self.comm.attach(e, c.CELLS)

    def _optical_for(self, IFREQ):
        """scalar ABS,SCA summed over species, or OPT[CELLS,2] with abundances (ASOC.py:1146-1175)"""
        e = self.eng
        if self.WITH_ABU:
            # OPT = sum over species of ABU * (AFABS, AFSCA) is built on the device from the abundances uploaded
            # once (setup_engine); the reference uploads 8*CELLS bytes per frequency (ASOC.py:1146-1160,1177)
            e.set_optical_abu([a[IFREQ] for a in self.AFABS], [a[IFREQ] for a in self.AFSCA])
            ABS = np.float32(sum(a[IFREQ] for a in self.AFABS))
            SCA = np.float32(sum(a[IFREQ] for a in self.AFSCA))
        else:
            ABS, SCA = np.float32(0.0), np.float32(0.0)
            for idust in range(self.NDUST):
                ABS += self.AFABS[idust][IFREQ]
                SCA += self.AFSCA[idust][IFREQ]
            e.set_opt(None)
        e.set_optical(ABS, SCA)
        return ABS, SCA

    def _scatter_tables_for(self, IFREQ):
        """DSC, CSC of the frequency; one pair per species with WITH_MSF (ASOC.py:1234-1243)"""
        if self.WITH_MSF:
            self.eng.set_scatter_tables(self.FDSC[:, IFREQ, :], self.FCSC[:, IFREQ, :])
        else:
            self.eng.set_scatter_table(self.FDSC[0, IFREQ, :], self.FCSC[0, IFREQ, :])

    def _save_intensity(self, IFREQ, FREQ, ABS, TMP):
        """saveint 1|2: the per-frequency INT tally (already summed over ranks) becomes the mean intensity of the cells,
        INTENSITY += (h*f/ABS) * 8^level * INT / n; saveint 2 adds the vector sums INTX, INTY, INTZ the same way
        (ASOC.py:1499-1515, :1895-1908)"""
        U, c, e = self.U, self.cloud, self.eng
        comps = [TMP]
        if U.SAVE_INTENSITY == 2:
            for which in (3, 4, 5):
                v = e.read_tally(which)
                if self.comm and self.world > 1:
                    v = self.comm.all_reduce_host(v)
                comps.append(v)
        if self.rank != 0:
            return
        if self.INTENSITY is None:
            self.INTENSITY = files.create_intensity_file(U.SAVE_INTENSITY_FILE, c.CELLS, self.NFREQ, U.SAVE_INTENSITY == 2)
        with np.errstate(divide="ignore", invalid="ignore"):
            for icomp, v in enumerate(comps):
                for level in range(c.LEVELS):
                    # float32 throughout, as numpy evaluates KDEV*(PLANCK*FREQ/ABS)*(8.0**level) with ABS a float32 array
                    coeff = np.float32(launch.PLANCK * FREQ) / np.float32(ABS) * np.float32(8.0 ** level)
                    a, b = int(c.OFF[level]), int(c.OFF[level] + c.LCELLS[level])
                    if U.SAVE_INTENSITY == 2:
                        self.INTENSITY[a:b, IFREQ, icomp] += coeff * v[a:b] / c.DENS[a:b]
                    else:
                        self.INTENSITY[a:b, IFREQ] += coeff * v[a:b] / c.DENS[a:b]

    def _constant_launch(self, II):
        """Launch shape of source block II (ASOC.py:1036-1110), or None when the block is not simulated."""
        U, c = self.U, self.cloud
        if II == 0:
            if (self.PSPAC < 1) or (U.NO_PS < 1):
                return None
            return launch.ps_launch(self.PSPAC, U.NO_PS, U.GL, self.GLOBAL_0)
        if II == 1:
            if self.BGPAC < 1:
                return None
            return launch.hpbg_launch(self.BGPAC, c.NX, c.NY, c.NZ) if len(self.HPBG) > 0 else launch.bg_launch(self.BGPAC, int(U.AREA))
        if II == 2:
            if len(self.DIFFUSERAD) < 1 or self.DFPAC < 1:
                return None
            return launch.cl_launch(self.DFPAC, c.CELLS, self.GLOBAL_0)
        if U.ROIPAC < 1 or self.ROI_LOAD is None:
            return None
        return launch.roi_launch(U.ROIPAC, files.roi_elements(self.ROI_DIM), U.ROI_NSIDE)

    def _launch_shares(self, by_frequency):
        """shard == "launches": {(II, IFREQ): (first, count)} for this rank over the sequence of launches of the constant sources.
        by_frequency: whole launches, the k-th simulated frequency to rank k % world (the rank then owns that frequency's INT)."""
        U = self.U
        seq = []
        for II in range(4):
            L = self._constant_launch(II)
            if L is None:
                continue
            for IFREQ in range(self.NFREQ):
                FREQ = float(self.FFREQ[IFREQ])
                if (FREQ < U.SIM_F[0]) or (FREQ > U.SIM_F[1]):
                    continue
                seq.append((II, IFREQ, L))
        if by_frequency:
            sim = sorted({f for _, f, _ in seq})
            owner = {f: k % self.world for k, f in enumerate(sim)}
            return {(II, f): ((0, L["GLOBAL"]) if owner[f] == self.rank else (0, 0)) for II, f, L in seq}, owner
        parts = launch.shard_launches([L["GLOBAL"] for _, _, L in seq], [L["PACKETS"] for _, _, L in seq], self.rank, self.world)
        return {(II, f): pc for (II, f, _), pc in zip(seq, parts)}, None

    def simulate_constant_sources(self):
        """for II in (point sources, background, diffuse): for IFREQ: launch (ASOC.py:1028-1545).
        Returns CTABS[CELLS] and FABSORBED[CELLS,NFREQ] (or None with noabsorbed)."""
        U, e, c = self.U, self.eng, self.cloud
        CELLS, NFREQ, FFREQ = c.CELLS, self.NFREQ, self.FFREQ
        CTABS = np.zeros(CELLS, np.float32)
        # `nnmake` with `absthin N`: the absorptions of every N-th cell only (ASOC.py:100-105, :632-638)
        thin = U.ABSTHIN if (U.ABSTHIN > 1 and 'nnmake' in U.KEYS) else 1
        self.absthin = thin
        FABSORBED = None if U.NOABSORBED else np.zeros(((CELLS + thin - 1) // thin, NFREQ), np.float32)
        if len(U.file_constant_load) > 0:
            self.log("=== CLOAD => %s" % U.file_constant_load)
            return np.fromfile(U.file_constant_load, np.float32, CELLS), FABSORBED
        rng = np.random.default_rng()
        DEVICES, ID, KDEV = 1, 0, 1.0                 # sharded launches reproduce ONE device (ASOC.py:179-181)
        # region of interest (ASOC.py:909-944): record what enters ROI / send in what an enclosing run recorded
        self.ROI_SAVE = self.ROI_LOAD = None
        if U.WITH_ROI_LOAD:
            self.ROI_DIM, self.ROI_LOAD = files.open_roi_load(U.FILE_ROI_LOAD, U.ROI_NSIDE, NFREQ)
        if U.WITH_ROI_SAVE:
            n = e.set_roi_save(U.ROI, U.ROI_STEP, U.ROI_NSIDE)
            self.ROI_SAVE = files.create_roi_save(U.FILE_ROI_SAVE, U.ROI, U.ROI_STEP, U.ROI_NSIDE, NFREQ) if self.rank == 0 \
                else np.zeros((NFREQ, n), np.float32)
        # TABS-only runs (noabsorbed): nothing is read back per frequency and CTABS is the sum over the source blocks, so ALL
        # launches of the constant sources are handed to the engine as one batch: on hierarchies walked brick-locally the
        # point-source, background and diffuse launches of all frequencies share brick sweeps (include/soc_hip.h:
        # soc_batch_begin; up to 128 launches per sweep), elsewhere the engine starts a new sweep where the kind changes.
        one_batch = (not self.with_int) and (not U.WITH_ROI_SAVE) and hasattr(e, "batch_begin") and U.ITERATIONS >= 1
        # several ranks, shard == "launches": the launches themselves are dealt out (see __init__).  Runs that keep the per-frequency
        # absorptions: a frequency belongs to one rank (not with the intensity file, region-of-interest records or emission iterations,
        # which need every frequency on every rank -- those keep the work-item split)
        shares, self.freq_owner = None, None
        if self.shard == "launches" and self.comm and self.world > 1 and U.ITERATIONS >= 1:
            own_freq = self.with_int
            if (not own_freq) or (FABSORBED is not None and U.SAVE_INTENSITY == 0 and (not U.WITH_ROI_SAVE) and self.CLPAC < 1 and thin == 1):
                shares, self.freq_owner = self._launch_shares(by_frequency=own_freq)
        owned = self.freq_owner is not None
        # Runs that keep the per-frequency absorptions on a hierarchy: frequency by frequency, the source blocks of a frequency as one
        # batch that tallies into one INT array (soc_batch_begin_shared_int), so that its point-source, background and diffuse
        # launches share brick sweeps and brick queues.  (Cartesian grids: the per-launch INT batches below; the intensity file,
        # region-of-interest records and a Healpix sky keep the block-by-block loop.)
        if (self.with_int and FABSORBED is not None and c.LEVELS > 1 and U.SAVE_INTENSITY == 0 and not U.WITH_ROI_SAVE
                and not U.WITH_ROI_LOAD and len(self.HPBG) == 0 and hasattr(e, "batch_begin_shared_int") and U.ITERATIONS >= 1
                and (shares is None or owned)):
            return self._simulate_by_frequency(CTABS, FABSORBED, shares, owned, rng)
        if one_batch:
            e.zero(0)
            e.batch_begin(0)
        for II in range(4):
            if U.ITERATIONS < 1:
                continue
            WPS = WBG = 0.0
            if II == 0:
                if (self.PSPAC < 1) or (U.NO_PS < 1):
                    continue
                L = launch.ps_launch(self.PSPAC, U.NO_PS, U.GL, self.GLOBAL_0)
                WPS = L["WPS"]
                self.log("=== PS  GLOBAL %d x BATCH %d = %d" % (L["GLOBAL"], L["BATCH"], L["PACKETS"]))
            elif II == 1:
                if self.BGPAC < 1:
                    continue
                L = launch.hpbg_launch(self.BGPAC, c.NX, c.NY, c.NZ) if len(self.HPBG) > 0 else \
                    launch.bg_launch(self.BGPAC, int(U.AREA))
                WBG = L["WBG"]
                self.log("=== BG: BGPAC %d, BATCH %d, GLOBAL %d" % (L["PACKETS"], L["BATCH"], L["GLOBAL"]))
            elif II == 2:
                if len(self.DIFFUSERAD) < 1 or self.DFPAC < 1:
                    continue
                L = launch.cl_launch(self.DFPAC, CELLS, self.GLOBAL_0)
                self.log("=== DFPAC %d, GLOBAL %d, BATCH %d" % (self.DFPAC, L["GLOBAL"], L["BATCH"]))
            else:
                if U.ROIPAC < 1 or self.ROI_LOAD is None:
                    continue
                L = launch.roi_launch(U.ROIPAC, files.roi_elements(self.ROI_DIM), U.ROI_NSIDE)
                self.log("=== ROI: GLOBAL %d, BATCH %d, elements %d" % (L["GLOBAL"], L["BATCH"], L["PACKETS"]))
            first, count = self.comm.shard(L["GLOBAL"]) if self.comm else (0, L["GLOBAL"])
            if not one_batch:
                e.zero(0)
            # TABS-only runs (noabsorbed): nothing is read back per frequency, so consecutive frequencies
            # are handed to the engine together and share brick sweeps (include/soc_hip.h: soc_batch_begin)
            deferred = (not one_batch) and (not self.with_int) and self.ROI_SAVE is None and hasattr(e, "batch_begin")   # the engine decides per launch
            # runs that keep the per-frequency absorptions, Cartesian grids: up to 16 frequencies per batch, every launch
            # with its own INT tally, read after the batch -- and summed over the ranks then, on its way to the host array.
            # (Hierarchies: one launch at a time; with `global` large enough each is a brick sweep of its own, the INT
            # tally in LDS beside TABS -- DESIGN.md.)
            int_batched = (self.with_int and FABSORBED is not None and self.ROI_SAVE is None and II != 3 and c.LEVELS == 1
                           and U.SAVE_INTENSITY == 0 and hasattr(e, "batch_begin_int"))
            group = []

            def end_group():
                e.batch_end()
                for k, f in enumerate(group):
                    arr = e.batch_read_int(k)
                    if self.comm and self.world > 1 and not owned:
                        arr = self.comm.all_reduce_host(arr)
                    FABSORBED[:, f] += arr[0::self.absthin]
                del group[:]
            if deferred:
                e.batch_begin(0)
            if int_batched:
                e.batch_begin_int(16)
            for IFREQ in range(NFREQ):
                FREQ = float(FFREQ[IFREQ])
                if (FREQ < U.SIM_F[0]) or (FREQ > U.SIM_F[1]):
                    continue
                t0 = time.time()
                ABS, SCA = self._optical_for(IFREQ)
                if self.with_int and not int_batched:
                    e.zero(1)
                PS = (self.LPS[:, IFREQ] * np.float32(WPS)) / np.float32(FREQ) if II == 0 else np.zeros(1, np.float32)
                BG = np.float32(float(self.IBG[IFREQ]) * WBG / FREQ) if len(self.IBG) == NFREQ else np.float32(0.0)
                FF = np.float32(launch.trapezoid_weight(FFREQ, IFREQ))
                self._scatter_tables_for(IFREQ)
                if U.SEED > 0:
                    seed = launch.launch_seed(U.SEED, IFREQ, DEVICES, ID)
                else:
                    seed = float(rng.random())
                    if self.comm and self.world > 1:      # every rank must use the same streams
                        seed = self._bcast_seed(seed)
                if shares is not None:
                    first, count = shares.get((II, IFREQ), (0, 0))
                    if count == 0:
                        continue                          # another rank's launch
                if II == 2:
                    dr_ind = IFREQ + (self.DIFFUSERAD.shape[1] - NFREQ)
                    if dr_ind < 0 or dr_ind >= self.DIFFUSERAD.shape[1]:
                        continue
                    EMIT = np.zeros(CELLS, np.float32)
                    for level in range(c.LEVELS):
                        coeff = U.GL * PARSEC / (8.0 ** level) * U.K_DIFFUSE
                        a, b = int(c.OFF[level]), int(c.OFF[level] + c.LCELLS[level])
                        EMIT[a:b] = self.DIFFUSERAD[a:b, dr_ind] * coeff
                    e.set_emission(EMIT, None)
                if II == 3:
                    # scale in again the dependence on the grid length (ASOC.py:1419-1421)
                    e.set_roi_load(self.ROI_DIM, U.ROI_NSIDE,
                                   np.asarray(self.ROI_LOAD[IFREQ, :] * U.ROI_LOAD_SCALE / (U.GL * U.GL), np.float32))
                if self.ROI_SAVE is not None:
                    e.roi_zero()                                   # per frequency (ASOC.py:1301-1302)
                hp = (II == 1) and len(self.HPBG) > 0
                if hp:
                    sky = files.hpbg_for_frequency(self.HPBG[IFREQ], WBG / FREQ, U.HPBG_WEIGHTED)
                    if sky is None:
                        continue                                   # empty sky (ASOC.py:1200)
                    e.set_hpbg(*sky)
                self.timers["Tpush"] += time.time() - t0
                t0 = time.time()
                if II == 2:
                    e.sim_cl(II, L["PACKETS"], L["BATCH"], seed, FF, L["GLOBAL"], gid_first=first, gid_count=count)
                elif hp:
                    e.sim_hp(L["PACKETS"], L["BATCH"], seed, FF, L["GLOBAL"], gid_first=first, gid_count=count)
                else:
                    e.sim_pb(II, L["PACKETS"], L["BATCH"], seed, BG, FF,
                             PSPOS=U.PSPOS[:max(U.NO_PS, 1), :3], PS=PS, XPS=self.XPS,
                             GLOBAL=L["GLOBAL"], gid_first=first, gid_count=count)
                if self.with_int and self.comm and not int_batched and not owned:
                    self.comm.all_reduce_tally(e, 1)      # one all-reduce of the per-cell buffer per frequency
                if int_batched:
                    group.append(IFREQ)
                    if len(group) == 16:
                        end_group()
                        e.batch_begin_int(16)
                elif not (deferred or one_batch):
                    e.sync()
                self.timers["Tkernel"] += time.time() - t0
                self.packets += L["PACKETS"]
                t0 = time.time()
                if (FABSORBED is not None or U.SAVE_INTENSITY > 0) and not int_batched:
                    TMP = e.read_tally(1)
                    if FABSORBED is not None:
                        FABSORBED[:, IFREQ] += TMP[0::self.absthin]
                    if U.SAVE_INTENSITY > 0:
                        self._save_intensity(IFREQ, FREQ, ABS, TMP)
                if self.ROI_SAVE is not None:
                    # += : point sources, background and a loaded record all pass here; GL^2 scales away the
                    # dependence on the current grid length (ASOC.py:1466-1475)
                    rec = e.roi_read()
                    if self.comm:
                        rec = self.comm.all_reduce_host(rec)
                    self.ROI_SAVE[IFREQ, :] += rec * np.float32(U.GL * U.GL)
                self.timers["Tpull"] += time.time() - t0
                if self.verbose and self.rank == 0:
                    print("  FREQ %3d/%3d  %10.3e   BG %12.4e  PS %12.4e   TW %10.3e" % (
                        IFREQ + 1, NFREQ, FREQ, BG, PS[0], FF))
            if deferred:
                t0 = time.time()
                e.batch_end()
                e.sync()
                self.timers["Tkernel"] += time.time() - t0
            if int_batched:
                t0 = time.time()
                end_group()
                self.timers["Tkernel"] += time.time() - t0
            if one_batch:
                continue                                   # the tally is read once, after the last block
            if self.comm:
                self.comm.all_reduce_tally(e, 0)          # TABS: integrated over frequency on the device
            t0 = time.time()
            CTABS += e.read_tally(0)
            self.timers["Tpull"] += time.time() - t0
            self.log("******  CONSTANT   %10s   CTABS -> %12.4e" % (['PS', 'BG', 'DE', 'ROI'][II], float(np.mean(CTABS))))
        if one_batch:
            t0 = time.time()
            e.batch_end()
            e.sync()
            self.timers["Tkernel"] += time.time() - t0
            if self.comm:
                self.comm.all_reduce_tally(e, 0)          # ONE all-reduce of TABS for all source blocks
            t0 = time.time()
            CTABS += e.read_tally(0)
            self.timers["Tpull"] += time.time() - t0
            self.log("******  CONSTANT   all source blocks   CTABS -> %12.4e" % float(np.mean(CTABS)))
        if self.ROI_LOAD is not None:
            e.set_roi_load(None, 0, None)
        if isinstance(self.ROI_SAVE, np.memmap):
            self.ROI_SAVE.flush()
        return CTABS, FABSORBED

    # Frequencies of an absorbed-file run on a hierarchy that share a sweep (an INT tally and, for cell emission, a copy of the emission each).
    # Measured on config 3 (bench.py --workload C3INT --int-groups 4): 2.00e8 packets/s against 2.09e8 with one frequency per sweep -- the
    # brick queues are per frequency (a workgroup's LDS tallies belong to one INT array), so more frequencies mean more queues of the same
    # length, not longer ones; the default stays 1.
    FREQS_PER_SWEEP = 1

    def _simulate_by_frequency(self, CTABS, FABSORBED, shares, owned, rng):
        """for IFREQ: for II in (point sources, background, diffuse): launch -- the loop of ASOC.py:1028-1545 with the frequency
        outside, for runs that keep the per-frequency absorptions: the launches of one frequency are one batch with one INT
        tally.  TABS integrates over everything on the device and is read once (CTABS is the sum over the blocks anyway).
        Where the engine has them, groups: up to FREQS_PER_SWEEP frequencies are deferred into one sweep, each with its own INT tally
        (soc_batch_begin_int_groups / soc_batch_next_int) -- that many times the packets per brick and pass -- and read after it."""
        U, e, c = self.U, self.eng, self.cloud
        CELLS, NFREQ, FFREQ = c.CELLS, self.NFREQ, self.FFREQ
        grouped = hasattr(e, "batch_begin_int_groups")
        pend = []

        def end_sweep():
            t0 = time.time()
            e.batch_end()
            e.sync()
            self.timers["Tkernel"] += time.time() - t0
            t0 = time.time()
            for k, f in enumerate(pend):
                arr = e.batch_read_int(k)
                if self.comm and self.world > 1 and not owned:
                    arr = self.comm.all_reduce_host(arr)      # the per-cell buffer of one frequency, summed over the ranks
                FABSORBED[:, f] += arr[0::self.absthin]
            self.timers["Tpull"] += time.time() - t0
            del pend[:]
        blocks = [(II, self._constant_launch(II)) for II in range(3)]
        blocks = [(II, L) for II, L in blocks if L is not None]
        for II, L in blocks:
            self.log("=== %s  GLOBAL %d x BATCH %d = %d" % (['PS', 'BG', 'DE'][II], L["GLOBAL"], L["BATCH"], L["PACKETS"]))
        e.zero(0)
        for IFREQ in range(NFREQ):
            FREQ = float(FFREQ[IFREQ])
            if (FREQ < U.SIM_F[0]) or (FREQ > U.SIM_F[1]):
                continue
            t0 = time.time()
            ABS, SCA = self._optical_for(IFREQ)
            self._scatter_tables_for(IFREQ)
            FF = np.float32(launch.trapezoid_weight(FFREQ, IFREQ))
            if U.SEED > 0:
                seed = launch.launch_seed(U.SEED, IFREQ, 1, 0)
            else:
                seed = float(rng.random())
                if self.comm and self.world > 1:          # every rank must use the same streams
                    seed = self._bcast_seed(seed)
            mine = [(II, L) + ((0, L["GLOBAL"]) if shares is None else shares.get((II, IFREQ), (0, 0))) for II, L in blocks]
            if shares is None and self.comm:
                mine = [(II, L) + self.comm.shard(L["GLOBAL"]) for II, L in blocks]
            mine = [m for m in mine if m[3] > 0]
            if not mine:
                continue                                  # another rank's frequency
            if grouped:
                if not pend:
                    e.batch_begin_int_groups(self.FREQS_PER_SWEEP)
                e.batch_next_int()
            else:
                e.zero(1)
                e.batch_begin_shared_int(len(mine))
            self.timers["Tpush"] += time.time() - t0
            for II, L, first, count in mine:
                t0 = time.time()
                if II == 2:
                    dr_ind = IFREQ + (self.DIFFUSERAD.shape[1] - NFREQ)
                    if dr_ind < 0 or dr_ind >= self.DIFFUSERAD.shape[1]:
                        continue
                    EMIT = np.zeros(CELLS, np.float32)
                    for level in range(c.LEVELS):
                        coeff = U.GL * PARSEC / (8.0 ** level) * U.K_DIFFUSE
                        a, b = int(c.OFF[level]), int(c.OFF[level] + c.LCELLS[level])
                        EMIT[a:b] = self.DIFFUSERAD[a:b, dr_ind] * coeff
                    e.set_emission(EMIT, None)
                self.timers["Tpush"] += time.time() - t0
                t0 = time.time()
                if II == 2:
                    e.sim_cl(II, L["PACKETS"], L["BATCH"], seed, FF, L["GLOBAL"], gid_first=first, gid_count=count)
                else:
                    PS = (self.LPS[:, IFREQ] * np.float32(L["WPS"])) / np.float32(FREQ) if II == 0 else np.zeros(1, np.float32)
                    BG = np.float32(float(self.IBG[IFREQ]) * L["WBG"] / FREQ) if (II == 1 and len(self.IBG) == NFREQ) else np.float32(0.0)
                    e.sim_pb(II, L["PACKETS"], L["BATCH"], seed, BG, FF, PSPOS=U.PSPOS[:max(U.NO_PS, 1), :3], PS=PS, XPS=self.XPS,
                             GLOBAL=L["GLOBAL"], gid_first=first, gid_count=count)
                self.timers["Tkernel"] += time.time() - t0
                self.packets += L["PACKETS"]
            if grouped:
                pend.append(IFREQ)
                if len(pend) >= self.FREQS_PER_SWEEP:
                    end_sweep()
            else:
                t0 = time.time()
                e.batch_end()
                e.sync()
                if self.comm and not owned:
                    self.comm.all_reduce_tally(e, 1)          # one all-reduce of the per-cell buffer per frequency
                self.timers["Tkernel"] += time.time() - t0
                t0 = time.time()
                FABSORBED[:, IFREQ] += e.read_tally(1)[0::self.absthin]
                self.timers["Tpull"] += time.time() - t0
            if self.verbose and self.rank == 0:
                print("  FREQ %3d/%3d  %10.3e   TW %10.3e" % (IFREQ + 1, NFREQ, FREQ, FF))
        if pend:
            end_sweep()
        if self.comm:
            self.comm.all_reduce_tally(e, 0)              # TABS: integrated over frequency and source blocks on the device
        t0 = time.time()
        CTABS += e.read_tally(0)
        self.timers["Tpull"] += time.time() - t0
        self.log("******  CONSTANT   all source blocks   CTABS -> %12.4e" % float(np.mean(CTABS)))
        return CTABS, FABSORBED

    # ---------------------------------------------------------------------------------
    def emission_iterations(self, CTABS, FABSORBED):
        """Simulation <-> temperature cycles (ASOC.py:1593-2260, the paths without reference field and
        ALI): per iteration the dust emission of the previous one is simulated with SimRAM_CL
        (`cellpackets`), the integrated absorptions TABS + CTABS give the equilibrium temperature of
        every cell and that the emission at every frequency -- EqTemperature and Emission on the device
        with the keys `CLT` / `CLE` in the ini, else the reference's host formulas (its host temperature loop uses
        a different interpolation weight, ASOC.py:2060).  Writes the temperature and emitted files.
        Returns (TNEW or None, EMITTED[CELLS, REMIT_NFREQ])."""
        U, e, c = self.U, self.eng, self.cloud
        CELLS, NFREQ, FFREQ = c.CELLS, self.NFREQ, self.FFREQ
        m = np.nonzero((FFREQ >= U.REMIT_F[0]) & (FFREQ <= U.REMIT_F[1]))[0]
        I1, I2 = int(m[0]), int(m[-1])
        solve = (not U.NOSOLVE) and bool(U.NOABSORBED)
        if solve and self.NDUST > 1:
            raise ValueError("temperatures can be solved here for a single dust component only (ASOC.py:260-263)")
        if (I2 - I1 + 1 < NFREQ) and U.ITERATIONS > 0 and self.CLPAC > 0:
            raise ValueError("remit cannot restrict the frequencies when cell emission is simulated (ASOC.py:209-211)")
        EMITTED = None
        try:
            EMITTED = np.array(files.mmap_emitted(U.file_emitted, CELLS, I2 - I1 + 1))
        except (OSError, files.FileError, ValueError):
            EMITTED = np.zeros((CELLS, I2 - I1 + 1), np.float32)
        TNEW = None
        if solve:
            Emin, kE, TTT = launch.temperature_table(FFREQ, self.AFABS[0], U.GL)
            FACTOR_f, LENGTH_f = launch.kernel_literals(U.GL)
        EMWEI = np.ones(CELLS, np.float32) * np.float32(self.CLPAC / CELLS) if U.USE_EMWEIGHT > 0 else None
        EMPAC = None
        ali = bool(U.WITH_ALI)
        if ali:
            e.set_ali(1)
        beta = None
        hostrng = np.random.default_rng(int(U.SEED * 2 ** 31) if U.SEED > 0 else None)
        EMIT = np.zeros(CELLS, np.float32)
        # reference field (`reference` key, ASOC.py:796-812, :1571-1586): the packets of an iteration carry the CHANGE of
        # the emission since the previous one, EMITTED - OEMITTED; the absorptions the previous emission caused, OTABS,
        # are added back on the host.  Both are damped by k = iteration/ITERATIONS at the start of an iteration, so the
        # first one simulates the full field.  reference = AABB continues a run of AA iterations at iteration BB.
        ref = int(U.WITH_REFERENCE)
        OEMITTED = OTABS = OXAB = OXEM = None
        if ref > 0 and self.CLPAC > 0:
            if I2 - I1 + 1 != NFREQ:
                raise ValueError("the reference field needs emission at every simulated frequency (ASOC.py:797)")
            OEMITTED = np.zeros((CELLS, NFREQ), np.float32)
            OTABS = np.zeros(CELLS, np.float32)
            if ref > 1 and ref % 100 > 0:
                OEMITTED[:, :] = np.fromfile('OEMITTED.save', np.float32).reshape(CELLS, NFREQ)
                OTABS[:] = np.fromfile('OTABS.save', np.float32)
            if ali:
                OXAB, OXEM = np.zeros(CELLS, np.float32), np.zeros(CELLS, np.float32)
        for iteration in range(U.ITERATIONS):
            self.log("ITERATION %d/%d" % (iteration + 1, U.ITERATIONS))
            e.zero(0)
            XEM = np.full(CELLS, 1.0e-32, np.float64) if ali else None      # ASOC.py:1606
            if OEMITTED is not None:                                         # ASOC.py:1607-1632
                k = iteration / float(U.ITERATIONS) if ref == 1 else (iteration + ref % 100) / float(ref // 100)
                OEMITTED *= np.float32(k)
                OTABS *= np.float32(k)
            if self.CLPAC > 0:
                GLOBAL, BATCH = self.GLOBAL_0, max(1, int(self.CLPAC / CELLS))
                first, count = self.comm.shard(GLOBAL) if self.comm else (0, GLOBAL)
                skip = U.EMWEIGHT_SKIP - 1
                # TABS-only iterations: the frequencies are handed to the engine together; with `global` raised to
                # about the number of cells they share brick sweeps (include/soc_hip.h: soc_batch_begin, soc_sim_cl)
                deferred = (not self.with_int) and (not ali) and U.USE_EMWEIGHT < 2 and hasattr(e, "batch_begin")
                if deferred:
                    e.batch_begin(0)
                for IFREQ in range(NFREQ):
                    FREQ = float(FFREQ[IFREQ])
                    if self.with_int:
                        e.zero(1)
                    if (FREQ < U.SIM_F[0]) or (FREQ > U.SIM_F[1]):
                        continue
                    t0 = time.time()
                    ABS_f, _ = self._optical_for(IFREQ)
                    FF = np.float32(launch.trapezoid_weight(FFREQ, IFREQ))
                    self._scatter_tables_for(IFREQ)
                    if IFREQ < I1 or IFREQ > I2:
                        continue
                    if OEMITTED is not None:                       # ASOC.py:1728-1735
                        EMIT[:] = EMITTED[:, IFREQ - I1] - OEMITTED[:, IFREQ - I1]
                        OEMITTED[:, IFREQ - I1] = EMITTED[:, IFREQ - I1]
                    else:
                        EMIT[:] = EMITTED[:, IFREQ - I1]
                    for level in range(c.LEVELS):
                        coeff = U.GL * PARSEC / (8.0 ** level) / launch.FACTOR
                        a, b = int(c.OFF[level]), int(c.OFF[level] + c.LCELLS[level])
                        EMIT[a:b] *= coeff * c.DENS[a:b]
                    EMIT[c.DENS < 1.0e-10] = 0.0
                    if ali:
                        XEM += EMIT * np.float64(FF)               # integral of the emitted energy (ASOC.py:1741)
                    if U.USE_EMWEIGHT > 0:                         # ASOC.py:1745-1771
                        skip += 1
                        if skip % U.EMWEIGHT_SKIP == 0:
                            tmp = np.asarray(EMITTED[:, IFREQ - I1], np.float64).copy()
                            tmp[~np.isfinite(tmp)] = 0.0
                            tmp[:] = self.CLPAC * tmp / (np.sum(tmp) + 1.0e-65)
                            EMWEI[:] = np.clip(tmp, U.EMWEIGHT_LIM[0], U.EMWEIGHT_LIM[1])
                            EMWEI[hostrng.random(CELLS) > EMWEI] = 0.0
                            if U.EMWEIGHT_LIM[2] > 0.0:
                                EMWEI[EMWEI < U.EMWEIGHT_LIM[2]] = 0.0
                            if U.USE_EMWEIGHT == 2:                # packets per cell in multiples of 100 (ASOC.py:1773-1780)
                                EMPAC = np.asarray(100 * np.round(tmp / 100), np.int32)
                                EMWEI[:] = 1.0 / (EMPAC + 1e-10)
                    e.set_emission(EMIT, EMWEI)
                    if U.SEED > 0:
                        seed = float(np.fmod(U.SEED + IFREQ * launch.SEED1, 1.0))      # ASOC.py:1807 (no SEED0 here)
                    else:
                        seed = float(hostrng.random())
                        if self.comm and self.world > 1:
                            seed = self._bcast_seed(seed)
                    self.timers["Tpush"] += time.time() - t0
                    t0 = time.time()
                    if U.USE_EMWEIGHT == 2 and EMPAC is not None:
                        # the host lists the cells that still owe packets, 100 per cell and launch (ASOC.py:1811-1840)
                        EMDONE = np.zeros(CELLS, np.int32)
                        EMINDEX = np.zeros(CELLS, np.int32)
                        f2, c2 = self.comm.shard(8192) if self.comm else (0, 8192)
                        while True:
                            mm = np.nonzero(EMDONE < EMPAC)[0]
                            if len(mm) < 1:
                                break
                            EMINDEX[:len(mm)] = mm
                            EMINDEX[len(mm):] = -1
                            e.set_emindex(EMINDEX)
                            e.sim_cl(2, self.CLPAC, BATCH, seed, FF, 8192, gid_first=f2, gid_count=c2)
                            EMDONE[mm] += 100
                            self.packets += 100 * len(mm)
                    else:
                        e.sim_cl(2, self.CLPAC, BATCH, seed, FF, GLOBAL, gid_first=first, gid_count=count)
                    if self.with_int and self.comm:
                        self.comm.all_reduce_tally(e, 1)
                    if not deferred:
                        e.sync()
                    self.timers["Tkernel"] += time.time() - t0
                    self.packets += CELLS * BATCH
                    if iteration == U.ITERATIONS - 1 and (FABSORBED is not None or U.SAVE_INTENSITY > 0):
                        TMP = e.read_tally(1)
                        if FABSORBED is not None:
                            FABSORBED[:, IFREQ] += TMP[0::self.absthin]
                        if U.SAVE_INTENSITY > 0:                   # ASOC.py:1885-1908
                            self._save_intensity(IFREQ, FREQ, ABS_f, TMP)
                if deferred:
                    t0 = time.time()
                    e.batch_end()
                    e.sync()
                    self.timers["Tkernel"] += time.time() - t0
                if self.comm:
                    self.comm.all_reduce_tally(e, 0)
                    if ali:
                        self.comm.all_reduce_tally(e, 2)
                if OEMITTED is not None:
                    # the device holds the absorptions of EMITTED - OEMITTED: add what OEMITTED caused (ASOC.py:1965-1975)
                    EABS = e.read_tally(0) + OTABS
                    OTABS[:] = EABS
                    EABS = EABS + CTABS
                    if ali:                                        # ASOC.py:1925-1936
                        OXAB += e.read_tally(2)
                        OXEM += np.asarray(XEM, np.float32)
                        beta = (OXEM - OXAB) / OXEM
                else:
                    EABS = e.read_tally(0) + CTABS
                    if ali:
                        beta = (XEM - e.read_tally(2)) / XEM       # escape probability (ASOC.py:1939-1942)
            else:
                EABS = np.array(CTABS, np.float32)
            if solve:
                t0 = time.time()
                # the ini keys pick the solver as in the reference: `CLT` (without ALI) = the EqTemperature kernel,
                # otherwise its host loop, the only one that knows beta (ASOC.py:2027, :2042; `MPT` = the same
                # formula on several processes); `CLE` = the Emission kernel, otherwise the host formula (:2159, :2199)
                if ('CLT' in U.KEYS) and not ali:
                    TNEW = e.solve_temperature(launch.ADHOC, kE, Emin, TTT, FACTOR_f, LENGTH_f, EABS)
                else:
                    TNEW = launch.solve_temperature_host(EABS, c, Emin, kE, TTT, U.GL, beta if ali else None,
                                                         empty_below=0.0 if 'MPT' in U.KEYS else 1.0e-10)
                if CELLS < 1e8:                                    # ASOC.py:2122-2136
                    TNEW[~np.isfinite(TNEW)] = 10.0
                    mok = c.DENS > 1.0e-8
                    TNEW[mok] = np.clip(TNEW[mok], 3.0, 1600.0)
                if 'CLE' in U.KEYS:
                    e.set_temperature(TNEW)                        # ASOC.py:2160: the host's TNEW goes to the device
                    EMITTED[:, :] = e.emission(FFREQ[I1:I2 + 1], self.AFABS[0][I1:I2 + 1], FACTOR_f, LENGTH_f)
                else:
                    if I1 > 0:
                        raise ValueError("the host emission loop indexes EMITTED with the frequency index (ASOC.py:2215, :2226): "
                                         "with `remit` cutting the low frequencies it fails in the reference; add `CLE`")
                    if 'MPE' in U.KEYS:
                        TNEW[TNEW < 3.0] = 10.0                    # ASOC.py:2211
                    EMITTED[:, :] = launch.emission_host(FFREQ[I1:I2 + 1], self.AFABS[0][I1:I2 + 1], TNEW, U.GL)
                self.timers["Tsolve"] = self.timers.get("Tsolve", 0.0) + time.time() - t0
        if self.rank == 0 and solve and U.ITERATIONS > 0:
            if len(U.file_temperature) > 0:
                files.write_temperature(U.file_temperature, c, TNEW)
            files.write_emitted(U.file_emitted, EMITTED)
            if OEMITTED is not None and ref > 1:                   # for the run that continues this one (ASOC.py:2251-2253)
                OEMITTED.tofile('OEMITTED.save')
                OTABS.tofile('OTABS.save')
        return TNEW, EMITTED

    def emission_from_temperature_file(self):
        """`loadtemp` with `iterations 0` (ASOC.py:700-764): the emission of an equilibrium dust from a stored temperature
        file, through the Emission kernel; written to the `emitted` file and used for the maps"""
        U, e, c = self.U, self.eng, self.cloud
        if self.NDUST > 1:
            raise ValueError("loadtemp recomputes the emission of a single equilibrium dust (ASOC.py:757-760 uses AFABS[0])")
        m = np.nonzero((self.FFREQ >= U.REMIT_F[0]) & (self.FFREQ <= U.REMIT_F[1]))[0]
        I1, I2 = int(m[0]), int(m[-1])
        TNEW = files.read_temperature(U.file_temperature, c)
        FACTOR_f, LENGTH_f = launch.kernel_literals(U.GL)
        e.set_temperature(TNEW)
        EMITTED = np.asarray(e.emission(self.FFREQ[I1:I2 + 1], self.AFABS[0][I1:I2 + 1], FACTOR_f, LENGTH_f), np.float32)
        if self.rank == 0 and len(U.file_emitted) > 0:
            files.write_emitted(U.file_emitted, EMITTED)
        return TNEW, EMITTED

    # ---------------------------------------------------------------------------------
    def write_maps(self, EMITTED):
        """Surface-brightness maps from the emission (ASOC.py:2924-3177, the plain `Mapping` path): for every
        direction map_dir_XX.bin = int32 NPIX.x, NPIX.y + one float32 [NPIX.y, NPIX.x] image [Jy/sr] per selected
        frequency.  `perspective` gives the longitude x latitude image seen from that position.  Optical-depth
        images for `savetau` frequencies are written as <file>_tau_<um>.bin, the column density (`savetau file -1`) as
        <file>_colden.fits; `fits` with `mapum` gives one FITS image per direction and frequency instead.  NPIX.y < 0:
        write_healpix_maps.  Map interpolation, ROI maps and polarisation maps are refused (_check_supported)."""
        U, e, c = self.U, self.eng, self.cloud
        if U.NPIX[1] == 0:
            self.log("mapping with NPIX.y == 0: neither the flat (NPIX.y > 0, ASOC.py:2924) nor the Healpix branch (NPIX.y < 0, :3185)")
            return
        if U.NPIX[1] < 0:
            return self.write_healpix_maps(EMITTED)
        NFREQ, FFREQ = self.NFREQ, self.FFREQ
        m = np.nonzero((FFREQ >= U.REMIT_F[0]) & (FFREQ <= U.REMIT_F[1]))[0]
        I1, I2 = int(m[0]), int(m[-1])
        NDIR, ODIR, RA, DE = launch.set_observer_directions(U.OBS_THETA, U.OBS_PHI)
        centre = U.MAPCENTRE if U.MAPCENTRE[0] > -1e7 else (0.5 * c.NX, 0.5 * c.NY, 0.5 * c.NZ)   # ASOC_aux.py:791-793
        KK = (1.0e23 / launch.FACTOR) * PLANCK / (4.0 * np.pi) * (U.GL * PARSEC)                 # ASOC.py:2997-2998
        _, LENGTH_f = launch.kernel_literals(U.GL)
        singles = np.asarray(getattr(U, "SINGLE_MAP_FREQ", []), np.float64)
        savetau = np.asarray(getattr(U, "savetau_freq", []), np.float64)
        # `fits` together with `mapum`: one FITS file per direction and selected frequency instead of map_dir_XX.bin
        # (ASOC.py:2977-2996); the header is MakeFits' (files.write_fits), the pixel GL*MAP_DX over the distance (1 kpc unless given)
        using_fits = (getattr(U, "FITS", 0) > 0) and (len(singles) > 0)
        pix = U.GL * U.MAP_DX / (U.DISTANCE if U.DISTANCE > 0.0 else 1000.0)
        fps = []
        if self.rank == 0 and not using_fits:
            for idir in range(NDIR):
                fp = open("map_dir_%02d.bin" % idir, "wb")
                np.asarray([U.NPIX[0], U.NPIX[1]], np.int32).tofile(fp)
                fps.append(fp)
        first_freq = True
        for IFREQ in range(NFREQ):
            FREQ = float(FFREQ[IFREQ])
            save_spe = (IFREQ >= I1) and (IFREQ <= I2)
            if (FREQ < U.MAP_FREQ[0]) or (FREQ > U.MAP_FREQ[1]):
                continue
            save_tau, save_colden = 0, 0
            if len(savetau) > 0:                                   # ASOC.py:3048-3058
                if np.min(np.abs((savetau - FREQ) / FREQ)) < 0.001:
                    save_tau = 1
                if (save_tau == 0) and first_freq and (np.min(savetau) <= 0.0):
                    save_colden = 1                                # `savetau file -1`: column density, with the first mapped frequency
            first_freq = False
            if len(singles) > 0 and np.min(np.abs(FREQ - singles)) / FREQ > 0.005:
                save_spe = False
            if not save_spe and not save_tau and not save_colden:
                continue
            ABS, SCA = self._optical_for(IFREQ)
            EMIT = np.asarray(KK * FREQ * EMITTED[:, IFREQ - I1], np.float32) if save_spe else np.zeros(c.CELLS, np.float32)
            um = launch.C_LIGHT / FREQ * 1.0e4
            ums = '%.0f' % um if um > 20.0 else ('%.1f' % um if um > 2.0 else '%.2f' % um)
            for idir in range(NDIR):
                MAP, TAU = e.map(EMIT, ODIR[idir], RA[idir], DE[idir], U.NPIX, U.MAP_DX, centre, ABS, SCA,
                                 INTOBS=U.INTOBS, save_colden=save_colden, LENGTH=LENGTH_f)
                if self.rank != 0:
                    continue
                suffix = '_dir%d' % idir if NDIR > 1 else ''
                tail = '' if NDIR == 1 else '_%03d' % idir
                if save_spe:
                    if using_fits:                                  # :3143-3148
                        files.write_fits("%s_%s%s.fits" % (U.FITS_PREFIX, ums, tail), MAP, U.FITS_RA, U.FITS_DE, pix)
                    else:
                        np.asarray(MAP, np.float32).tofile(fps[idir])
                if save_colden:                                     # always a FITS image in the reference (:3152-3159)
                    files.write_fits('%s_colden%s%s.fits' % (U.file_savetau, suffix, tail), TAU, U.FITS_RA, U.FITS_DE, pix)
                if save_tau:                                        # :3160-3171
                    name = '%s_tau_%s%s%s' % (U.file_savetau, ums, suffix, tail)
                    if using_fits:
                        files.write_fits(name + '.fits', TAU, U.FITS_RA, U.FITS_DE, pix)
                    else:
                        np.asarray(TAU, np.float32).tofile(name + '.bin')
        for fp in fps:
            fp.close()

    def write_ps_tau(self):
        """`pssavetau file um`: for every observer direction <file>_<idir>.dat with one line per point source -- its index,
        the column density [cm-2 per unit density] and the optical depth towards the observer at the frequency of the grid
        closest to `um` (ASOC.py:3576-3645, PSTau)"""
        U, e = self.U, self.eng
        IFREQ = int(np.argmin(np.abs(self.FFREQ - U.pssavetau_freq)))
        ABS, SCA = self._optical_for(IFREQ)
        NDIR, ODIR, RA, DE = launch.set_observer_directions(U.OBS_THETA, U.OBS_PHI)
        _, LENGTH_f = launch.kernel_literals(U.GL)
        for idir in range(NDIR):
            col, tau = e.ps_tau(U.PSPOS[:U.NO_PS, :3], ODIR[idir], ABS, SCA, LENGTH_f)
            if self.rank == 0:
                with open("%s_%d.dat" % (U.file_pssavetau, idir), "w") as fp:
                    for i in range(U.NO_PS):
                        fp.write('%6d  %12.4e  %12.4e\n' % (i, col[i], tau[i]))

    def write_healpix_maps(self, EMITTED):
        """`mapping NSIDE -1 dx`: all-sky map of the emission seen from `perspective` (HealpixMapping, kernel_ASOC_map.c),
        file layout of ASOC.py:3185-3320: map_dir_00_H.bin = int32 [NPIX.x, NPIX.y], int32 [frequencies, LEVELS], then one
        float32 [12*NSIDE^2] map [Jy/sr] per frequency of the emitted range inside `wavelength`.
        The reference itself stops in this branch with a NameError (SAVE_COLDEN is never assigned, :3291/:3297) after
        writing the two headers; this writes the file its loop describes, with SAVE_COLDEN = 0 (no column-density file:
        its savetau tests compare a list with a float, :3303-3306)."""
        U, e, c = self.U, self.eng, self.cloud
        FFREQ = self.FFREQ
        m = np.nonzero((FFREQ >= U.REMIT_F[0]) & (FFREQ <= U.REMIT_F[1]))[0]
        I1, I2 = int(m[0]), int(m[-1])
        NSIDE = int(U.NPIX[0])
        _, ODIR, RA, DE = launch.set_observer_directions(U.OBS_THETA, U.OBS_PHI)
        centre = U.MAPCENTRE if U.MAPCENTRE[0] > -1e7 else (0.5 * c.NX, 0.5 * c.NY, 0.5 * c.NZ)
        KK = (1.0e23 / launch.FACTOR) * PLANCK / (4.0 * np.pi) * (U.GL * PARSEC)
        _, LENGTH_f = launch.kernel_literals(U.GL)
        sel = [i for i in range(I1, I2 + 1) if U.MAP_FREQ[0] <= float(FFREQ[i]) <= U.MAP_FREQ[1]]
        fp = None
        if self.rank == 0:
            fp = open("map_dir_%02d_H.bin" % 0, "wb")              # NDIR = 1 for Healpix maps (ASOC.py:2917)
            np.asarray([U.NPIX[0], U.NPIX[1]], np.int32).tofile(fp)
            np.asarray([len(sel), c.LEVELS], np.int32).tofile(fp)
        for IFREQ in sel:
            FREQ = float(FFREQ[IFREQ])
            ABS, SCA = self._optical_for(IFREQ)
            EMIT = np.asarray(EMITTED[:, IFREQ - I1] * np.float32(KK) * np.float32(FREQ), np.float32)    # :3283
            MAP, _ = e.map(EMIT, ODIR[0], RA[0], DE[0], U.NPIX, U.MAP_DX, centre, ABS, SCA, INTOBS=U.INTOBS, save_colden=0,
                           LENGTH=LENGTH_f, healpix=NSIDE)
            if fp:
                np.asarray(MAP, np.float32).tofile(fp)
        if fp:
            fp.close()

    def _bcast_seed(self, seed):
        t = self.comm.torch.tensor([seed], dtype=self.comm.torch.float64,
                                   device="cuda" if self.comm.backend == "nccl" else "cpu")
        self.comm.dist.broadcast(t, 0)
        return float(t[0])

    # ---------------------------------------------------------------------------------
    def run(self):
        t00 = time.time()
        self.write_packet_info()
        self.setup_engine()
        CTABS, FABSORBED = self.simulate_constant_sources()
        U = self.U
        self.TNEW, self.EMITTED = None, None
        if U.ITERATIONS > 0 and (self.CLPAC > 0 or ((not U.NOSOLVE) and U.NOABSORBED)) and hasattr(self.eng, "solve_temperature"):
            self.TNEW, self.EMITTED = self.emission_iterations(CTABS, FABSORBED)
        elif U.LOAD_TEMPERATURE and U.ITERATIONS < 1 and hasattr(self.eng, "emission"):
            self.TNEW, self.EMITTED = self.emission_from_temperature_file()
        if (not U.NOMAP) and self.EMITTED is not None and hasattr(self.eng, "map"):
            self.write_maps(self.EMITTED)
        if U.NO_PS > 0 and U.pssavetau_freq > 0.0 and U.NPIX[1] > 0 and hasattr(self.eng, "ps_tau"):
            self.write_ps_tau()
        if self.rank == 0 and self.INTENSITY is not None:          # ASOC.py:2733-2757
            files.finish_intensity_file(U.SAVE_INTENSITY_FILE, self.INTENSITY, self.cloud.CELLS, self.NFREQ, U.SAVE_INTENSITY == 2)
            self.INTENSITY = None
        if self.rank == 0:
            if len(U.file_constant_save) > 0:
                CTABS.tofile(U.file_constant_save)                 # ASOC.py:1547-1549
            if FABSORBED is not None and getattr(self, "freq_owner", None) is None:
                files.scale_absorbed(FABSORBED, self.cloud, U.GL, U.NNNLIMIT, self.absthin)
                files.write_absorbed(U.file_absorbed, FABSORBED)   # ASOC.py:2866-2875
            elif FABSORBED is None:
                prefix = U.KEYS.get('prefix', ['soc'])[0] if U.KEYS.get('prefix') else 'soc'
                CTABS.tofile(prefix + ".ctabs")
        if FABSORBED is not None and getattr(self, "freq_owner", None) is not None:
            # every rank holds the columns of the frequencies it simulated and writes them itself: no collective (as a2e.run_sharded)
            files.scale_absorbed(FABSORBED, self.cloud, U.GL, U.NNNLIMIT, self.absthin)
            if self.rank == 0:
                files.create_absorbed(U.file_absorbed, FABSORBED.shape[0], FABSORBED.shape[1])
            self.comm.barrier()
            mine = [f for f, r in self.freq_owner.items() if r == self.rank]
            if self.rank == 0:
                mine += [f for f in range(FABSORBED.shape[1]) if f not in self.freq_owner]      # frequencies outside `simum`: nobody's
            files.write_absorbed_columns(U.file_absorbed, FABSORBED, mine)
            self.comm.barrier()
        wall = time.time() - t00
        if self.rank == 0 and self.verbose:
            print("Tkernel %.3f  Tpush %.3f  Tpull %.3f" % (self.timers["Tkernel"], self.timers["Tpush"], self.timers["Tpull"]))
            if self.timers["Tkernel"] > 0:
                print("%.4e photon packets / s (simulation section, %d GPU%s)" % (
                    self.packets / self.timers["Tkernel"], self.world, "s" if self.world > 1 else ""))
            print("@@ asoc %.2f seconds WC" % wall)
        return CTABS, FABSORBED


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        print("Usage:  python -m soc_amd.asoc ini-file")
        return 1
    from .dist import Comm
    from .lib import Engine
    USER = User(argv[1])
    comm = Comm()
    eng = Engine(comm.local_rank)
    try:
        AbsorptionRun(USER, eng, comm).run()
    finally:
        eng.close()
        comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
