"""CPU: bench.py's sizing under the GPU box's host shape -- 3.17 TB of RAM behind a 300 GiB cgroup limit, /dev/shm nominally 1.5 TB
(profiles/r04a_box.txt) -- for 1, 2, 4 and 8 ranks: the files of a step (FASTQ + two outputs, in tmpfs = RAM charged to the cgroup) plus
what the rank processes hold themselves must stay inside the cgroup, whatever the tmpfs or MemAvailable say.  Round 3 sized the job
from the tmpfs figure and lost the box; no 8-GPU node was ever available to try the N = 8 sizing on, so it is pinned here."""
import builtins
import io

import pytest

import bench

GIB = 1 << 30
MEMINFO = "MemTotal:       3170000000 kB\nMemFree:        3000000000 kB\nMemAvailable:   3100000000 kB\nShmem:            1000000 kB\n"


@pytest.fixture
def box(monkeypatch):
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/proc/meminfo":
            return io.StringIO(MEMINFO)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(bench, "_read_int", lambda p: {"/sys/fs/cgroup/memory.max": 300 * GIB, "/sys/fs/cgroup/memory.current": 6 * GIB}.get(p))
    monkeypatch.setattr(bench.shutil, "disk_usage", lambda p: type("du", (), {"free": 1500 * GIB, "total": 1500 * GIB, "used": 0})())
    return bench.host_memory()


def test_usable_memory_is_the_cgroups_headroom(box):
    assert box["usable"] == 294 * GIB and box["cgroup_max"] == 300 * GIB          # not MemAvailable (3.1 TB), not the tmpfs (1.5 TB)


@pytest.mark.parametrize("world,lanes", [(1, 8), (2, 8), (4, 4), (8, 4)])
def test_the_job_and_its_rank_processes_fit_the_cgroup(box, world, lanes):
    n_pairs = bench.pick_pairs(box, True, world, lanes)
    files = bench.job_bytes(n_pairs)
    ranks = world * bench.rank_bytes(lanes, True)
    assert files <= bench.MEM_SHARE * box["usable"]
    assert files + ranks <= bench.PROCESS_SHARE * box["usable"], (n_pairs, files / GIB, ranks / GIB)
    assert n_pairs >= 40_000_000                                   # still configs[2]'s size class at every N (100 M reads: 50 M pairs)
    if world <= 4:
        assert n_pairs == 50_000_000
    # the side legs (configs[4], configs[3], configs[1]) are sized from the same share, one input + one output at a time
    assert 2 * 10_000_000 * bench.REC_BYTES + 2 * 10_000_000 * 500 <= bench.MEM_SHARE * box["usable"]


def test_a_small_host_gets_a_small_job(monkeypatch):
    mem = {"usable": 48 * GIB, "shm_free": 30 * GIB}
    n = bench.pick_pairs(mem, True, 1, 8)
    assert 1_000_000 <= n < 50_000_000 and bench.job_bytes(n) <= 0.4 * 48 * GIB
