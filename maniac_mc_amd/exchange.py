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


def molecule_count_histogram(counts, nbins):
    """Histogram of the replicas' current molecule counts (the uptake histogram of an isotherm point)."""
    h = np.zeros(nbins, dtype=np.int64)
    c = np.clip(np.asarray(counts, dtype=np.int64), 0, nbins - 1)
    np.add.at(h, c, 1)
    return h
