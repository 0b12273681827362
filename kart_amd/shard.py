"""Read sharding across ranks (one process per GPU) and the path's only collective.

Kart's hot path has no exchange step: reads are independent, the index is
replicated per GPU, and the only cross-rank quantity is the run summary
{total reads, unmapped, paired, distance} of reference src/Mapping.cpp:730-741 -- one small
all-reduce (RCCL on the GPU box, gloo in the CPU tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous range [lo, hi) of whole units (4000-read chunks or read pairs) owned by `rank`;
    the first n_units % world ranks take one extra unit."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_counters(counters, device=None):
    """Sum a list of int counters over all ranks (no-op when torch.distributed is not initialised)."""
    t = torch.tensor(list(counters), dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def max_over_ranks(value, device=None):
    """Element-wise maximum over all ranks of a float or a list of floats (the per-step wall times of a run: a step lasts as
    long as its slowest rank)."""
    scalar = not isinstance(value, (list, tuple))
    t = torch.tensor([value] if scalar else list(value), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0].item()) if scalar else [float(x) for x in t.tolist()]
