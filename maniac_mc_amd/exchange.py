"""The path's one exchange step: independent GCMC replicas never communicate
while they run; once per block every rank contributes its replicas' molecule-count histogram and a
few running sums (trials, accepted moves, energy sums) and all ranks receive the per-rank table.

torch.distributed is used as plumbing only: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests.  The message is <= 40 KB per rank (int64[NB_MAX_MOLECULE + 1] + 8 doubles), i.e.
latency-bound; one all_gather per block, no other collective on the data path.
"""
from __future__ import annotations

import numpy as np


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def barrier():
    d = _dist()
    if d:
        d.barrier()


def gather_block_stats(sums, histogram=None):
    """All-gather per-rank running sums (float64[k]) and an optional histogram (int64[nbins]).

    Returns (sums_by_rank [world, k], hist_by_rank [world, nbins] or None).  Deterministic: the
    table is ordered by rank, reductions are left to the caller."""
    import torch
    d = _dist()
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    hist = None if histogram is None else np.ascontiguousarray(histogram, dtype=np.int64)
    if d is None:
        return sums[None, :].copy(), (None if hist is None else hist[None, :].copy())
    dev = torch.device("cuda", torch.cuda.current_device()) if d.get_backend() == "nccl" else torch.device("cpu")
    n = d.get_world_size()
    ts = torch.from_numpy(sums).to(dev)
    out_s = [torch.empty_like(ts) for _ in range(n)]
    d.all_gather(out_s, ts)
    sums_by_rank = torch.stack(out_s).cpu().numpy()
    hist_by_rank = None
    if hist is not None:
        th = torch.from_numpy(hist).to(dev)
        out_h = [torch.empty_like(th) for _ in range(n)]
        d.all_gather(out_h, th)
        hist_by_rank = torch.stack(out_h).cpu().numpy()
    return sums_by_rank, hist_by_rank


def max_over_ranks(value: float) -> float:
    import torch
    d = _dist()
    if d is None:
        return float(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if d.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    d.all_reduce(t, op=d.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(value: float) -> float:
    return -max_over_ranks(-float(value))


def molecule_count_histogram(counts, nbins):
    """Histogram of the replicas' current molecule counts (the uptake histogram of an isotherm point)."""
    c = np.clip(np.asarray(counts, dtype=np.int64), 0, nbins - 1)
    return np.bincount(c, minlength=nbins).astype(np.int64, copy=False)


class CAbiComm:
    """The same exchange behind the C ABI (include/maniac_gpu.h: mgpu_comm_create / mgpu_allgather_block_stats -- RCCL's
    ncclAllGather called from libmaniac_hip.so, no torch involved): what a Fortran host links against.  One rank needs no
    id and never initialises RCCL; with more ranks, rank 0 makes the id (CAbiComm.unique_id()) and the host passes it
    round (bench.py --exchange c-abi broadcasts it over its torch.distributed group)."""

    ID_BYTES = 128

    def __init__(self, device=0, rank=0, world=1, unique_id=None):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self.L = _lib.lib()
        self.h = C.c_void_p()
        self.rank, self.world = int(rank), int(world)
        if unique_id is not None and len(unique_id) != self.ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes CAbiComm.unique_id() returned")
        buf = (C.c_char * self.ID_BYTES).from_buffer_copy(bytes(unique_id)) if unique_id is not None else None
        _lib.check(self.L.mgpu_comm_create(C.byref(self.h), C.c_int(device), C.c_int(rank), C.c_int(world), buf))

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = (C.c_char * CAbiComm.ID_BYTES)()
        _lib.check(_lib.lib().mgpu_comm_unique_id(buf))
        return bytes(buf)

    def gather_block_stats(self, sums, histogram=None):
        """Same contract as the module-level gather_block_stats."""
        C = self._C
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        hist = np.zeros(0, dtype=np.int64) if histogram is None else np.ascontiguousarray(histogram, dtype=np.int64)
        out_s = np.zeros((self.world, sums.shape[0]), dtype=np.float64)
        out_h = np.zeros((self.world, hist.shape[0]), dtype=np.int64)
        dp, lp = C.POINTER(C.c_double), C.POINTER(C.c_longlong)
        self._lib.check(self.L.mgpu_allgather_block_stats(self.h, C.c_int(sums.shape[0]), sums.ctypes.data_as(dp), C.c_int(hist.shape[0]),
                                                          hist.ctypes.data_as(lp), out_s.ctypes.data_as(dp), out_h.ctypes.data_as(lp)))
        return out_s, (None if histogram is None else out_h)

    def close(self):
        if self.h:
            self.L.mgpu_comm_destroy(self.h)
            self.h = self._C.c_void_p()
