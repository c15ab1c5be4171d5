"""One process per GPU: work-item sharding of a logical launch and the tally all-reduce.

Packets are independent, RNG streams are a function of the logical work-item id and tallies
are additive, so the only exchange step of the path is a sum of the
per-cell absorption buffers: one ``all_reduce`` per frequency for INT (when absorptions are
saved per frequency) and one per source block for TABS.  On GPUs the reduction runs in place
on the engine's tally memory through RCCL (torch.distributed backend "nccl"); the "gloo"
backend is used on CPU-only hosts (tests).
"""
import os

import numpy as np

from .launch import shard_range


class Comm:
    """Rank/world bookkeeping + tally reduction.  world == 1 needs no torch at all."""

    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = None
        self._tensors = {}
        self._stream = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.torch, self.dist = torch, dist
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            self.backend = backend
            if not dist.is_initialized():
                if backend == "nccl":
                    torch.cuda.set_device(self.local_rank)
                    dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                            device_id=torch.device("cuda", self.local_rank))
                else:
                    dist.init_process_group("gloo", rank=self.rank, world_size=self.world)

    def shard(self, GLOBAL):
        """(first, count) of the logical work items this rank executes."""
        return shard_range(GLOBAL, self.rank, self.world)

    def _engine_stream(self, engine):
        """The engine runs on a stream torch created, and that stream is torch's current one: c10d orders its
        collective behind the current stream and the current stream behind the collective, so tally kernels,
        all-reduce and read-back run in program order.  (torch's default stream has the null handle, which
        soc_set_stream reads as "the engine's own stream" -- a stream no collective is ordered with.)"""
        t = self.torch
        if self._stream is None:
            self._stream = t.cuda.Stream()
            t.cuda.set_stream(self._stream)
        engine.set_stream(self._stream.cuda_stream)

    def attach(self, engine, cells):
        """RCCL path: make the engine's tallies torch tensors so they are reduced in place, on a stream
        shared with torch (see _engine_stream)."""
        if self.world > 1 and self.backend == "nccl":
            t = self.torch
            self._engine_stream(engine)
            for which in (0, 1):
                buf = t.zeros(cells, dtype=t.float32, device="cuda")
                engine.bind_tally(which, buf.data_ptr(), cells)
                self._tensors[which] = buf

    def all_reduce_tally(self, engine, which):
        """Sum tally `which` over all ranks (result on every rank)."""
        if self.world == 1:
            return
        if self.backend == "nccl" and which in self._tensors:
            self.dist.all_reduce(self._tensors[which])
        elif self.backend == "nccl":                       # a tally that is not bound to a tensor (XAB): via the host
            arr = self.torch.from_numpy(np.ascontiguousarray(engine.read_tally(which))).cuda()
            self.dist.all_reduce(arr)
            engine.write_tally(which, arr.cpu().numpy())
        else:
            arr = self.torch.from_numpy(np.ascontiguousarray(engine.read_tally(which)))
            self.dist.all_reduce(arr)
            engine.write_tally(which, arr.numpy())

    def all_reduce_host(self, arr):
        """Sum a host array over all ranks (small per-frequency records: the ROI record)."""
        if self.world == 1:
            return arr
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        if self.backend == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def attach_image(self, engine, npix):
        """Same for the scattered-light image OUT[NDIR*NPIX_Y*NPIX_X]."""
        if self.world > 1 and self.backend == "nccl":
            t = self.torch
            self._engine_stream(engine)
            buf = t.zeros(npix, dtype=t.float32, device="cuda")
            engine.sca_bind_out(buf.data_ptr())
            self._tensors["out"] = buf

    def all_reduce_image(self, engine):
        """Sum the image over all ranks; returns it as a host array [NDIR, NPIX_Y, NPIX_X]."""
        if self.world == 1:
            return engine.sca_read_out()
        if self.backend == "nccl":
            self.dist.all_reduce(self._tensors["out"])
            return engine.sca_read_out()
        arr = self.torch.from_numpy(np.ascontiguousarray(engine.sca_read_out()))
        self.dist.all_reduce(arr)
        return arr.numpy()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.destroy_process_group()
