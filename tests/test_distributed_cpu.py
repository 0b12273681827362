"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- read sharding with no data-path collective,
counter all-reduce, max-over-ranks timing -- exercised with real processes."""
import gzip
import os
import socket
import subprocess

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


def _map_worker(rank, world, port, binary, prefix, f1, f2, out, rdv, parts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def run(arguments):            # what Session.map is on the GPU box: one mapping run of this rank (here the CPU backend's CLI)
        r = subprocess.run([binary] + arguments, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode()[-400:]
        return r.stdout.decode()

    dist.barrier()
    log = shard.map_shard(run, ["-silent", "-t", "3", "-i", prefix, "-f", f1, "-f2", f2], out, rank, world, rdv, parts)
    mine = 0
    for f in shard.part_files(out, world, parts) if parts else []:
        if f.endswith(".%d" % rank):
            mine = sum(1 for ln in open(f, "rb") if not ln.startswith(b"@"))
    total = shard.allreduce_counters([mine])
    dist.barrier()
    q.put((rank, total[0], log[-200:]))
    dist.destroy_process_group()


def test_two_ranks_map_one_library_like_bench(tmp_path):
    """bench.py --gpus 2 on CPU: two torch.distributed ranks (gloo), each ONE mapping run with `-shard r/2 -rendezvous <file>`
    (shard.map_shard; the CPU backend's CLI stands in for Session.map), no data-path collective, the record counts all-reduced;
    the parts in rank order -- and the one shared file -- are the golden SAM of the unmodified reference."""
    from conftest import GOLDEN, ROOT, SMALL_PREFIX
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_backend")], stdout=subprocess.DEVNULL)
    binary = os.path.join(ROOT, "tests", "_build", "kart-host-oracle")
    files = []
    for name in ("pe_1.fq", "pe_2.fq"):
        dst = str(tmp_path / name)
        with gzip.open(os.path.join(GOLDEN, "sam", name + ".gz")) as fi, open(dst, "wb") as fo:
            fo.write(fi.read())
        files.append(dst)
    want = gzip.open(os.path.join(GOLDEN, "sam", "pe.sam.gz")).read()
    n_records = sum(1 for ln in want.split(b"\n") if ln and not ln.startswith(b"@"))
    world = 2
    ctx = mp.get_context("spawn")
    for parts in (True, False):
        out, rdv = str(tmp_path / ("out_%d.sam" % parts)), str(tmp_path / ("rdv_%d" % parts))
        shard.remove_rendezvous(rdv)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_map_worker, args=(r, world, port, binary, SMALL_PREFIX, files[0], files[1], out, rdv, parts, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=240) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        if parts:
            assert [os.path.exists(f) for f in shard.part_files(out, world)] == [True, True]
            assert all(total == n_records for _, total, _ in res)      # every rank sees the whole job's record count
            shard.concatenate_parts(out, world)
            assert not os.path.exists(out + ".0")
        assert open(out, "rb").read() == want


def test_shard_arguments():
    assert shard.shard_arguments(0, 1, "/x/rdv") == []
    assert shard.shard_arguments(1, 4, "/x/rdv") == ["-shard", "1/4", "-rendezvous", "/x/rdv", "-parts"]
    assert shard.shard_arguments(3, 4, "/x/rdv", parts=False) == ["-shard", "3/4", "-rendezvous", "/x/rdv"]
    assert shard.part_files("/o.sam", 3) == ["/o.sam.0", "/o.sam.1", "/o.sam.2"]
    assert shard.part_files("/o.sam", 3, parts=False) == ["/o.sam"] and shard.part_files("/o.sam", 1) == ["/o.sam"]
    import pytest
    with pytest.raises(ValueError):
        shard.shard_arguments(4, 4, "/x/rdv")


def test_shard_range_properties():
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_concatenate_parts_leaves_a_single_process_file_alone(tmp_path):
    """with one process the run wrote `out` itself; concatenate_parts(out, 1) used to truncate it, copy the empty file
    onto itself and remove it.  With several parts the text is assembled beside `out` and renamed, and `out` is never a source."""
    out = str(tmp_path / "a.sam")
    open(out, "wb").write(b"@PG\tone process\nr1\t4\n")
    assert shard.concatenate_parts(out, 1) == out
    assert open(out, "rb").read() == b"@PG\tone process\nr1\t4\n"
    assert shard.part_files(out, 1) == [out] and shard.shard_arguments(0, 1, "rv") == []
    for q, text in enumerate((b"@PG\thead\nr1\t4\n", b"r2\t4\n", b"r3\t4\n")):
        open(out + ".%d" % q, "wb").write(text)
    assert shard.concatenate_parts(out, 3) == out               # an older `out` is replaced, not appended to
    assert open(out, "rb").read() == b"@PG\thead\nr1\t4\nr2\t4\nr3\t4\n"
    assert sorted(os.listdir(tmp_path)) == ["a.sam"]
    open(out + ".0", "wb").write(b"x\n")                        # a missing part: an error, `out` untouched, no temporary left
    try:
        shard.concatenate_parts(out, 2)
        raise AssertionError("a missing part must raise")
    except FileNotFoundError:
        pass
    assert open(out, "rb").read() == b"@PG\thead\nr1\t4\nr2\t4\nr3\t4\n" and sorted(os.listdir(tmp_path)) == ["a.sam", "a.sam.0"]
