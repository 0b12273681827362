"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- read sharding with no data-path collective,
counter all-reduce, max-over-ranks timing -- exercised with real processes."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kart_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_units, rank, world)
    mine = list(range(lo, hi))
    # per-rank "mapping": counters derived from the units this rank owns
    counters = [len(mine), sum(u % 7 == 0 for u in mine), 2 * sum(u % 3 == 0 for u in mine), sum(mine)]
    total = shard.allreduce_counters(counters)
    tmax = shard.max_over_ranks(1.0 + rank)
    # per-step wall times of K timed steps: a step lasts as long as its slowest rank (bench.py sums these maxima)
    steps = shard.max_over_ranks([1.0 + rank, 3.0 - rank, 2.0])
    dist.barrier()
    out.put((rank, lo, hi, total, tmax, steps))
    dist.destroy_process_group()


def test_two_rank_sharding_and_counter_allreduce():
    world, n_units = 2, 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards tile the input exactly, in order
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_units
    want = [n_units, sum(u % 7 == 0 for u in range(n_units)), 2 * sum(u % 3 == 0 for u in range(n_units)), sum(range(n_units))]
    for _, _, _, total, tmax, steps in res:
        assert total == want          # every rank sees the whole-job counters
        assert tmax == 2.0            # max over ranks
        assert steps == [2.0, 3.0, 2.0]   # element-wise: the per-step maxima


def test_shard_range_properties():
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
