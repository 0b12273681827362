"""CPU: bench.py sizes its job from what the HOST can hold (round 3 sized it from the tmpfs's nominal size, kept 25 fresh 37.8 GB
outputs alive and lost the GPU box to the cgroup's 300 GiB memory limit).  The rules, without a GPU:
  * FASTQ + TWO step outputs stay within 40 % of min(MemAvailable, cgroup memory.max - memory.current);
  * files + what the N rank processes themselves hold stay within 80 % of it;
  * from four ranks on a rank runs one seeding group of four lanes instead of two."""
import bench


def test_host_memory_reports_what_the_sizing_needs():
    m = bench.host_memory()
    for k in ("MemTotal", "MemAvailable", "Shmem", "cgroup_max", "cgroup_current", "usable", "shm_free"):
        assert k in m
    assert m["usable"] is None or 0 < m["usable"] <= m["MemTotal"]
    if m["cgroup_max"] is not None and m["MemAvailable"] is not None:
        assert m["usable"] <= m["cgroup_max"] - (m["cgroup_current"] or 0)


def test_job_is_sized_from_host_memory_not_from_the_tmpfs():
    gpu_box = {"usable": 321.7e9, "shm_free": 1.6e12}             # profiles/r04a_box.txt: 300 GiB cgroup limit, /dev/shm nominally 1.5 TB
    assert bench.pick_pairs(gpu_box, True) == 50_000_000           # configs[2]: 100 M reads
    assert bench.job_bytes(50_000_000) <= gpu_box["usable"] * bench.MEM_SHARE
    # ... what round 3 kept alive under the driver's flags would never have been allowed
    assert (20 + 5) * 2 * 50_000_000 * bench.SAM_BYTES_PER_READ > gpu_box["usable"]
    for usable in (200e9, 100e9, 64e9, 20e9, 4e9):
        n = bench.pick_pairs({"usable": usable, "shm_free": 1.6e12}, True)
        assert n >= 1_000_000 and (n == 1_000_000 or bench.job_bytes(n) <= usable * bench.MEM_SHARE)
    assert bench.pick_pairs({"usable": None, "shm_free": None}, True) == 10_000_000          # nothing known: round 2's footprint
    assert bench.pick_pairs({"usable": 321.7e9, "shm_free": 50e9}, True) < 50_000_000        # a small tmpfs caps it as well


def test_rank_processes_count_in_the_sizing():
    gpu_box = {"usable": 321.7e9, "shm_free": 1.6e12}
    for world, lanes in ((1, 8), (2, 8), (4, 4), (8, 4)):
        n = bench.pick_pairs(gpu_box, True, world, lanes)
        assert bench.job_bytes(n) + world * bench.rank_bytes(lanes, True) <= gpu_box["usable"] * bench.PROCESS_SHARE
    # eight ranks with two seeding groups each would not leave room for the full job
    assert bench.pick_pairs(gpu_box, True, 8, 8) < 50_000_000
    assert bench.rank_bytes(8, True) > bench.rank_bytes(4, True) > 2 * bench.HG38_LEN


def test_seeding_configuration_mirrors_the_host_pipeline(monkeypatch):
    for k in ("KART_AMD_SEED_GROUP", "KART_AMD_STREAM_LANES", "KART_AMD_STREAM_READS"):
        monkeypatch.delenv(k, raising=False)
    assert bench.seed_group_setting() == (4, 8) and bench.stream_reads_setting() == 1120000
    monkeypatch.setenv("KART_AMD_SEED_GROUP", "0")
    assert bench.seed_group_setting() == (0, 4)
    monkeypatch.setenv("KART_AMD_SEED_GROUP", "4")
    monkeypatch.setenv("KART_AMD_STREAM_LANES", "6")
    assert bench.seed_group_setting() == (4, 8)                    # (rounded up to whole groups, as host/detail/pipeline.inc does)
    monkeypatch.setenv("KART_AMD_STREAM_READS", "2000000")
    assert bench.stream_reads_setting() == 2000000
