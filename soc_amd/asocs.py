#!/usr/bin/env python3
"""asocs -- images of scattered light on MI355X:  python -m soc_amd.asocs my.ini

Drop-in for ``ASOCS.py <ini>`` (reference ASOCS.py:1-930): reads the same ini file and input
files, simulates point sources (SimRAM_PS), the isotropic background (SimRAM_PB), a diffuse
field and the dust emission (SimRAM_CL) frequency by frequency with peel-off towards the
``direction`` observers, and writes ``outcoming.socs`` = surface brightness
OUTCOMING[NFREQ, NDIR, NPIX.y, NPIX.x] in Jy/sr (ASOCS.py:887-899) with the reference's header.

The host loop (source block II -> frequency; then the CLPAC loop) and every launch formula
are the reference's; the kernels are soc_amd/csrc/soc_sca.hip.  With several ranks each
launch is split by logical work-item id and the image is summed with one all-reduce.
``perspective x y z`` switches to one Healpix map (``outnside``) seen from that position,
``hpbg`` to the Healpix background (sca SimRAM_HP).  Not covered (refused with a clear
message): ROI files, several scattering functions.
"""
import sys
import time

import numpy as np

from . import files, launch
from .asoc import AbsorptionRun, UnsupportedOption
from .ini import User
from .launch import PLANCK, PARSEC, Fix


class ScatteringRun(AbsorptionRun):
    def _load_inputs(self):
        super()._load_inputs()
        U, c = self.U, self.cloud
        if U.ROIPAC > 0:
            raise UnsupportedOption("roiload in the scattering run")
        if self.WITH_ABU and U.SINGLE_ABU:
            raise UnsupportedOption("singleabu in the scattering run (ASOCS.py does not know the key: it reads ABU[CELLS, NDUST], :51-55)")
        if len(U.file_hpbg) > 2:                                   # ASOCS.py:104-107: no user scaling here
            self.HPBG = np.fromfile(U.file_hpbg, np.float32).reshape(self.NFREQ, 49152)
        self.healpix = U.INTOBS[0] > -10000.0                      # ASOCS.py:44-48: one Healpix map seen from INTOBS
        if self.healpix:
            self.NDIR = -int(U.OUT_NSIDE)
        else:
            self.NDIR, self.ODIR, self.RA, self.DE = launch.set_observer_directions(U.OBS_THETA, U.OBS_PHI)
        if U.MAPCENTRE[0] < -1e7:                                  # ASOC_aux.py:791-793
            U.MAPCENTRE = (0.5 * c.NX, 0.5 * c.NY, 0.5 * c.NZ)
        m = np.nonzero((self.FFREQ >= U.REMIT_F[0]) & (self.FFREQ <= U.REMIT_F[1]))[0]
        if len(m) < 1:
            raise ValueError("remit selects no frequencies")
        self.REMIT_I1, self.REMIT_I2, self.REMIT_NFREQ = int(m[0]), int(m[-1]), len(m)
        # launch sizes and packet counts, ASOCS.py:79-98
        LOCAL = 8 if 'c' in U.DEVICES else 32
        if 'local' in U.KEYS:
            LOCAL = int(U.KEYS['local'][0])
        G0 = 65536
        if 'global' in U.KEYS:
            G0 = int(U.KEYS['global'][0])
        self.LOCAL, self.GLOBAL_0 = LOCAL, Fix(G0, 32 * LOCAL)
        self.PSPAC = Fix(U.PSPAC, LOCAL)
        self.BGPAC = Fix(Fix(U.BGPAC, int(U.AREA)), LOCAL)
        if U.USE_EMWEIGHT > 0:
            self.CLPAC, self.DFPAC = Fix(U.CLPAC, LOCAL), Fix(U.DFPAC, LOCAL)
        else:
            self.CLPAC, self.DFPAC = Fix(Fix(U.CLPAC, c.CELLS), LOCAL), Fix(Fix(U.DFPAC, c.CELLS), LOCAL)
        self.EMITTED = []
        if self.CLPAC > 0:
            self.EMITTED = files.mmap_emitted(U.file_emitted, c.CELLS, self.REMIT_NFREQ)
        self.EMWEI = None
        if U.USE_EMWEIGHT > 0:
            pac = max(self.CLPAC, self.DFPAC)
            self.EMWEI = np.ones(c.CELLS, np.float32) * np.float32(pac / c.CELLS)
        self._skip = 2
        self._hostrng = np.random.default_rng(int(U.SEED * 2 ** 31) if U.SEED > 0 else None)

    # ---------------------------------------------------------------------------------
    def setup_engine(self):
        e, c, U = self.eng, self.cloud, self.U
        e.set_cloud(c)
        e.set_features(with_int=0, ps_method=U.PS_METHOD, use_emweight=min(max(U.USE_EMWEIGHT, 0), 1))
        e.set_mirror(launch.mirror_mask(U.MIRROR))
        if self.WITH_ABU:
            if U.OPT_IS_HALF or hasattr(e, "set_opt_half"):
                e.set_opt_half(bool(U.OPT_IS_HALF))            # ASOCS.py:524-525
            e.set_abundances(self.ABU)                         # once; OPT per frequency on the device (ASOCS.py:519-523)
        if self.healpix:
            e.sca_set_healpix(U.OUT_NSIDE, U.INTOBS, U.FFS)
            npix = 12 * U.OUT_NSIDE * U.OUT_NSIDE
        else:
            e.sca_set_view(self.ODIR, self.RA, self.DE, U.NPIX, U.MAP_DX, U.MAPCENTRE, U.FFS)
            npix = self.NDIR * U.NPIX[0] * U.NPIX[1]
        # The frequencies of a source block are ONE batch with an image each (soc_sca_batch_images): the launches that can run as rays on
        # brick-local hierarchies are deferred and share sweeps (many more rays per brick and pass than one launch has), the others run
        # at once into their frequency's image.  Several ranks then add the images up once per block, on the host.
        self.batched = hasattr(e, "sca_batch_images")
        if self.comm and not self.batched:
            self.comm.attach_image(e, npix)

    def _update_emwei(self, IFREQ):
        """Packets per cell from the emission, every third frequency (ASOCS.py:546-595)."""
        U = self.U
        self._skip += 1
        if self._skip != 3:
            return False
        self._skip = 0
        tmp = np.asarray(self.EMITTED[:, IFREQ - self.REMIT_I1], np.float64).copy() if len(self.EMITTED) else \
            np.ones(self.cloud.CELLS, np.float64)
        tmp[~np.isfinite(tmp)] = 0.0
        tmp[:] = self.CLPAC * tmp / (np.sum(tmp) + 1.0e-65)
        self.EMWEI[:] = np.clip(tmp, U.EMWEIGHT_LIM[0], U.EMWEIGHT_LIM[1])
        self.EMWEI[self._hostrng.random(self.cloud.CELLS) > self.EMWEI] = 0.0        # Russian roulette
        if U.EMWEIGHT_LIM[2] > 0.0:
            self.EMWEI[self.EMWEI < U.EMWEIGHT_LIM[2]] = 0.0
        return True

    def _seed(self, IFREQ, with_seed0):
        U = self.U
        if U.SEED > 0:
            if with_seed0:
                return launch.launch_seed(U.SEED, IFREQ, 1, 0)                       # ASOCS.py:634
            return float(np.fmod(U.SEED + IFREQ * launch.SEED1, 1.0))               # ASOCS.py:856
        seed = float(self._hostrng.random())
        if self.comm and self.world > 1:
            seed = self._bcast_seed(seed)
        return seed

    def _collect(self, OUTCOMING, IFREQ):
        if self.batched:
            self._launched.append(IFREQ)                          # read after the batch (_end_block)
            return
        t0 = time.time()
        OUT = self.comm.all_reduce_image(self.eng) if self.comm else self.eng.sca_read_out()
        OUTCOMING[IFREQ] += OUT
        self.timers["Tpull"] += time.time() - t0

    def _begin_block(self, max_launches=0):
        """max_launches: cell-emission launches keep a copy of the emission each (8 B per cell), so their batches are shorter"""
        self._launched = []
        if self.batched:
            self.eng.batch_begin(max_launches)
            self.eng.sca_batch_images(self.NFREQ)

    def _begin_frequency(self, IFREQ):
        if self.batched:
            self.eng.sca_batch_select(IFREQ)
        else:
            self.eng.sca_zero()

    def _end_block(self, OUTCOMING):
        if not self.batched:
            return
        t0 = time.time()
        self.eng.batch_end()
        self.eng.sync()
        self.timers["Tkernel"] += time.time() - t0
        t0 = time.time()
        if self._launched:
            imgs = np.stack([self.eng.sca_batch_read(k) for k in self._launched])
            if self.comm and self.world > 1:
                imgs = self.comm.all_reduce_host(imgs)
            for i, k in enumerate(self._launched):
                OUTCOMING[k] += imgs[i].reshape(OUTCOMING[k].shape)
        self.eng.sca_batch_images(0)
        self.timers["Tpull"] += time.time() - t0

    # ---------------------------------------------------------------------------------
    def simulate(self):
        """-> OUTCOMING[NFREQ, NDIR, NPIX.y, NPIX.x], photons per pixel before the final scaling"""
        U, e, c = self.U, self.eng, self.cloud
        CELLS, NFREQ, FFREQ = c.CELLS, self.NFREQ, self.FFREQ
        OUTCOMING = np.zeros((NFREQ, 12 * U.OUT_NSIDE * U.OUT_NSIDE), np.float32) if self.healpix else \
            np.zeros((NFREQ, self.NDIR, U.NPIX[1], U.NPIX[0]), np.float32)
        EMIT = np.zeros(CELLS, np.float32)
        for II in range(3):                                        # ASOCS.py:428-723
            WPS = WBG = 0.0
            if II == 0:
                if (self.PSPAC < 1) or (U.NO_PS < 1):
                    continue
                L = launch.ps_launch(self.PSPAC, U.NO_PS, U.GL, self.GLOBAL_0)
                WPS = L["WPS"]
            elif II == 1:
                if self.BGPAC < 1:
                    continue
                L = launch.hpbg_sca_launch(self.BGPAC, c.NX, c.NY, c.NZ) if len(self.HPBG) > 0 else \
                    launch.bg_launch(self.BGPAC, int(U.AREA))
                WBG = L["WBG"]
            else:
                if len(self.DIFFUSERAD) < 1 or self.DFPAC < 1:
                    continue
                L = dict(GLOBAL=self.GLOBAL_0, BATCH=int(self.DFPAC / CELLS), PACKETS=self.DFPAC)
            self.log("=== II=%d  GLOBAL %d, BATCH %d, PACKETS %d" % (II, L["GLOBAL"], L["BATCH"], L["PACKETS"]))
            first, count = self.comm.shard(L["GLOBAL"]) if self.comm else (0, L["GLOBAL"])
            self._skip = 2
            self._begin_block(16 if II == 2 else 0)
            for IFREQ in range(NFREQ):
                FREQ = float(FFREQ[IFREQ])
                if (FREQ < U.SIM_F[0]) or (FREQ > U.SIM_F[1]):
                    continue
                t0 = time.time()
                self._begin_frequency(IFREQ)
                self._optical_for(IFREQ)
                BG = np.float32(float(self.IBG[IFREQ]) * WBG / FREQ) if len(self.IBG) == NFREQ else np.float32(0.0)
                PS = (self.LPS[:, IFREQ] * np.float32(WPS)) / np.float32(FREQ) if II == 0 else np.zeros(1, np.float32)
                if U.USE_EMWEIGHT > 0:
                    self._update_emwei(IFREQ)
                self._scatter_tables_for(IFREQ)
                seed = self._seed(IFREQ, True)
                if II == 2:
                    if IFREQ >= self.DIFFUSERAD.shape[1]:
                        continue
                    for level in range(c.LEVELS):
                        coeff = U.GL * PARSEC / (8.0 ** level) * U.K_DIFFUSE
                        a, b = int(c.OFF[level]), int(c.OFF[level] + c.LCELLS[level])
                        EMIT[a:b] = self.DIFFUSERAD[a:b, IFREQ] * coeff
                    EMIT[c.DENS < 1.0e-10] = 0.0
                    e.set_emission(EMIT, self.EMWEI)
                hp = (II == 1) and len(self.HPBG) > 0
                if hp:
                    sky = files.hpbg_for_frequency(self.HPBG[IFREQ], WBG / FREQ, U.HPBG_WEIGHTED, clip_low=1.0e-2, skip_empty=False)
                    e.set_hpbg(*sky)
                self.timers["Tpush"] += time.time() - t0
                t0 = time.time()
                if hp:
                    e.sca_sim_hp(L["PACKETS"], L["BATCH"], seed, L["GLOBAL"], gid_first=first, gid_count=count)
                elif II == 0:
                    e.sca_sim_ps(L["PACKETS"], L["BATCH"], seed, BG, U.PSPOS[:U.NO_PS, :3], PS, XPS=self.XPS,
                                 GLOBAL=L["GLOBAL"], gid_first=first, gid_count=count)
                elif II == 1:
                    e.sca_sim_pb(II, L["PACKETS"], L["BATCH"], seed, BG, GLOBAL=L["GLOBAL"], gid_first=first, gid_count=count)
                else:
                    e.sca_sim_cl(II, L["PACKETS"], L["BATCH"], seed, L["GLOBAL"], gid_first=first, gid_count=count)
                if not self.batched:
                    e.sync()
                self.timers["Tkernel"] += time.time() - t0
                self.packets += L["PACKETS"]
                self._collect(OUTCOMING, IFREQ)
                if self.verbose and self.rank == 0:
                    print("  FREQ %3d/%3d  %10.3e --  BG %10.3e  PS %10.3e" % (IFREQ + 1, NFREQ, FREQ, BG, PS[0]))
            self._end_block(OUTCOMING)

        # dust emission from the emitted file, ASOCS.py:733-881
        if self.CLPAC > 0:
            GLOBAL, BATCH = self.GLOBAL_0, max(1, int(self.CLPAC / CELLS))
            first, count = self.comm.shard(GLOBAL) if self.comm else (0, GLOBAL)
            self.log("=== CLPAC %d, GLOBAL %d, BATCH %d" % (CELLS * BATCH, GLOBAL, BATCH))
            self._skip = 2
            self._begin_block(16)
            for IFREQ in range(NFREQ):
                FREQ = float(FFREQ[IFREQ])
                self._begin_frequency(IFREQ)
                if (FREQ < U.SIM_F[0]) or (FREQ > U.SIM_F[1]):
                    continue
                if IFREQ < self.REMIT_I1 or IFREQ > self.REMIT_I2:
                    continue
                t0 = time.time()
                self._optical_for(IFREQ)
                self._scatter_tables_for(IFREQ)
                EMIT[:] = self.EMITTED[:, IFREQ - self.REMIT_I1]
                for level in range(c.LEVELS):
                    coeff = 1.0e-20 * U.GL * PARSEC / (8.0 ** level)
                    a, b = int(c.OFF[level]), int(c.OFF[level] + c.LCELLS[level])
                    EMIT[a:b] *= coeff * c.DENS[a:b]
                EMIT[c.DENS < 1.0e-10] = 0.0
                if U.USE_EMWEIGHT > 0:
                    self._update_emwei(IFREQ)
                e.set_emission(EMIT, self.EMWEI)
                seed = self._seed(IFREQ, False)
                self.timers["Tpush"] += time.time() - t0
                t0 = time.time()
                e.sca_sim_cl(2, self.CLPAC, BATCH, seed, GLOBAL, gid_first=first, gid_count=count)
                if not self.batched:
                    e.sync()
                self.timers["Tkernel"] += time.time() - t0
                self.packets += CELLS * BATCH
                self._collect(OUTCOMING, IFREQ)
            self._end_block(OUTCOMING)
        return OUTCOMING

    def run(self):
        t00 = time.time()
        self.write_packet_info()
        self.setup_engine()
        OUTCOMING = self.simulate()
        U = self.U
        for IFREQ in range(self.NFREQ):                            # ASOCS.py:887-897: photons -> Jy/sr
            if self.healpix:
                k = float(self.FFREQ[IFREQ]) * 1.0e23 * PLANCK / (4.0 * np.pi / (12.0 * self.NDIR * self.NDIR))
            else:
                k = float(self.FFREQ[IFREQ]) * 1.0e23 * PLANCK / (U.MAP_DX * U.MAP_DX)
            OUTCOMING[IFREQ] *= k
        if self.rank == 0:
            if self.healpix:
                files.write_outcoming_healpix("outcoming.socs", U.OUT_NSIDE, self.FFREQ, OUTCOMING)
            else:
                files.write_outcoming("outcoming.socs", self.FFREQ, OUTCOMING)
                if U.FITS > 0 and self.NDIR == 1:                  # ASOCS.py:882-892: the cube of the one direction, frequencies as comments
                    pix = U.GL * U.MAP_DX / (U.DISTANCE if U.DISTANCE > 0.0 else 1000.0)
                    files.write_fits('%s.fits' % U.file_scattering, OUTCOMING[:, 0], U.FITS_RA, U.FITS_DE, pix, freq=self.FFREQ)
        if self.rank == 0 and self.verbose:
            print("Tkernel %.3f  Tpush %.3f  Tpull %.3f" % (self.timers["Tkernel"], self.timers["Tpush"], self.timers["Tpull"]))
            if self.timers["Tkernel"] > 0:
                print("%.4e photon packets / s (simulation section, %d GPU%s)" % (
                    self.packets / self.timers["Tkernel"], self.world, "s" if self.world > 1 else ""))
            print("@@ asocs %.2f seconds WC" % (time.time() - t00))
        return OUTCOMING


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        print("Usage:  python -m soc_amd.asocs ini-file")
        return 1
    from .dist import Comm
    from .lib import Engine
    USER = User(argv[1])
    comm = Comm()
    eng = Engine(comm.local_rank)
    try:
        ScatteringRun(USER, eng, comm).run()
    finally:
        eng.close()
        comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
