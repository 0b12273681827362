"""Read sharding across ranks (one process per GPU) and the path's only collective.

Kart's hot path has no exchange step: reads are independent, the index is
replicated per GPU, and the only cross-rank quantity is the run summary
{total reads, unmapped, paired, distance} of reference src/Mapping.cpp:730-741 -- one small
all-reduce (RCCL on the GPU box, gloo in the CPU tests)."""
from __future__ import annotations

import os
import shutil

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


def gather_rows(values, device=None):
    """Every rank's list of floats, as a list of rows in rank order, on every rank (one all-gather of a small tensor; a single
    process gets its own row)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        rows = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, t)
        return [[float(x) for x in r.tolist()] for r in rows]
    return [[float(x) for x in t.tolist()]]


# ---- a sharded mapping run, rank by rank ---------------------------------------------------------------------------------
# One process per GPU maps the contiguous range of whole 4000-read chunks that falls into its byte slice of the input files
# (kart_amd/csrc/host/detail/shard.inc; the reference's counterpart is the chunk fan-out to its worker threads,
# src/Mapping.cpp:504-512, 716-717, and iPaired / iDistance behind EstDistance, :533-540).  What the ranks share is a small
# file every one of them maps -- the rendezvous -- named on the command line of each rank's run; no read data is exchanged.


def shard_arguments(rank: int, world: int, rendezvous: str, parts: bool = True):
    """The options that make one mapping run (`kart-amd ...`, `Session.map([...])`) rank `rank` of `world`: `-shard r/N -rendezvous
    <file>` and, with `parts`, `-parts` (every rank writes its own `<out>.<r>`; without it all ranks write one file, each at the
    offset behind the earlier ranks' text).  Nothing for a single process."""
    if world <= 1:
        return []
    if not 0 <= rank < world:
        raise ValueError("rank %d of %d" % (rank, world))
    a = ["-shard", "%d/%d" % (rank, world), "-rendezvous", rendezvous]
    if parts:
        a.append("-parts")
    return a


def part_files(out: str, world: int, parts: bool = True):
    """The files a run with output `out` leaves: `<out>.0 .. <out>.<world-1>` with `parts` on several ranks, else `out`."""
    return [out + ".%d" % q for q in range(world)] if (parts and world > 1) else [out]


def concatenate_parts(out: str, world: int, remove: bool = True):
    """`cat <out>.0 ... > <out>`: the parts of a `-parts` run in rank order ARE the single-process file (rank 0's part starts with
    the header).  A single process wrote `out` itself (shard_arguments adds no `-parts` then): nothing to do.  The text goes to a
    temporary file next to `out` and takes its name at the end, so `out` is never one of the sources and a failure leaves the
    parts where they are.  Returns `out`."""
    if world <= 1:
        return out
    parts = [out + ".%d" % q for q in range(world)]
    tmp = out + ".cat.%d" % os.getpid()
    try:
        with open(tmp, "wb") as dst:
            for f in parts:
                with open(f, "rb") as src:
                    shutil.copyfileobj(src, dst, 1 << 24)
        os.replace(tmp, out)
    except BaseException:
        try:
            os.remove(tmp)
        except OSError:
            pass
        raise
    if remove:
        for f in parts:
            os.remove(f)
    return out


def remove_rendezvous(path: str):
    """A rendezvous file serves ONE run (its flags are never reset): remove it before the ranks start the next one."""
    try:
        os.remove(path)
    except OSError:
        pass


def map_shard(run, arguments, out: str, rank: int, world: int, rendezvous: str, parts: bool = True):
    """Rank `rank`'s share of one mapping run: `run(arguments + ["-o", out] + shard_arguments(...))`, where `run` is
    `Session.map` (the product: libkart_host.so in this process, one GPU) or anything that takes the same argument list.  All
    ranks must call it for the same run; it returns whatever `run` returns (the rank's own counters -- sum them with
    allreduce_counters).  The ranks need no barrier before it beyond the rendezvous file not existing yet."""
    return run(list(arguments) + ["-o", out] + shard_arguments(rank, world, rendezvous, parts))
