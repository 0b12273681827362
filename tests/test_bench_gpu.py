"""bench.py's contract on a real GPU: one JSON line with the keys the driver and the judge read, on a reduced workload (the
default workload is the hg38-sized one and takes minutes): the single-rank line with its legs, the KART_REF_FASTA route, and
`--gpus 2` started by the script itself (two ranks sharing the box's one device over gloo where there is only one)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"}


def _bench(args, env=None, tmp=None):
    e = dict(os.environ, KART_BENCH_DIR=str(tmp))
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    # the headline line (value + roofline + cpu_baseline, flushed before the side legs run) and the final, enriched one; a run
    # without side legs prints the final line only
    assert 1 <= len(lines) <= 2 and lines[-1]["line"] == "final", r.stdout.decode()[-2000:]
    if len(lines) == 2:
        head, last = lines
        assert head["line"].startswith("headline")
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline"):
            if k == "roofline":       # (the final line adds the SURVEY 8(d) figure and the traffic fraction, which need the seeding leg's counters)
                assert all(last[k].get(x) == v for x, v in head[k].items()), k
                continue
            assert head[k] == last[k] or k == "config", k
        assert ("cpu_baseline" in head) == ("cpu_baseline" in last)
    return lines[-1]


def test_bench_line_contract(tmp_path):
    d = _bench(["--genome-len", "500000", "--pairs", "60000", "--steps", "2", "--warmup", "1"], tmp=tmp_path)
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["unit"] == "reads/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - d["mapped_reads_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # value = whole-job mapped reads / step time
    assert d["config"]["bracket_seconds"] >= d["ms_per_step"] * 1e-3 * d["steps"] * 0.999     # the K steps lie inside the barrier-to-barrier bracket
    assert d["config"]["peak_shmem_GB"] >= 0 and d["config"]["host_memory_GB"]["usable"] > 0 and "TWO step outputs" in d["config"]["sizing"]
    assert "FASTQ -> SAM" in d["config"]["workload"] and "model" not in d["config"] and d["config"]["fallback"] is None
    assert 0.5 < d["mapped_fraction"] <= 1.0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0 < r["frac"] <= 1.0 and r["kernel"] == "search_kernel" and r["algorithmic_bytes_per_launch"] > 0      # a fraction of the peak by construction
    c = d["cpu_baseline"]
    assert c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "reads/s"
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "kart")):
        assert c["kind"] == "reference"
        assert d["parity"]["sam_vs_reference_t1"]["identical"] is True and d["parity"]["sam_vs_reference_t1"]["reads"] == 120000
    assert d["parity"]["seeds_vs_oracle"]["identical"] is True
    s = d["seeding_stage"]
    assert s["value"] > 0 and set(s["kernels_ms"]) == {"search", "scan", "locate", "sort"} and s["fetched_per_read"]["rank_steps"] >= 0


def test_bench_real_fasta_route(tmp_path):
    fa = os.path.join(ROOT, "tests", "golden", "small.fa")
    d = _bench(["--pairs", "20000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity"], env={"KART_REF_FASTA": fa}, tmp=tmp_path)
    assert "KART_REF_FASTA=small.fa" in d["config"]["workload"]
    assert d["config"]["fallback"] is None and d["roofline"]["traffic"] is None and d["value"] > 0


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the script starts two ranks itself and relays rank 0's line.
    On a 1-GPU box the two ranks share device 0 and all-reduce over gloo (KART_BENCH_SHARE_DEVICE); the whole job's reads are
    split between them and the merged SAM is the one-process SAM (same size here; byte identity is test_sam_gpu's subject)."""
    import torch
    env = {} if torch.cuda.device_count() >= 2 else {"KART_BENCH_SHARE_DEVICE": "1"}
    one = _bench(["--genome-len", "500000", "--pairs", "60000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--no-seeding-leg"], tmp=tmp_path)
    two = _bench(["--gpus", "2", "--genome-len", "500000", "--pairs", "60000", "--steps", "1", "--warmup", "1"], env=env, tmp=tmp_path)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["reads_per_step"] == 120000 and two["config"]["reads_per_gpu_per_step"] == 60000
    assert two["mapped_reads_per_step"] == one["mapped_reads_per_step"]
    assert two["config"]["sam_bytes_per_step"] == one["config"]["sam_bytes_per_step"]
    assert "cpu_baseline" not in two             # rank 0 at N = 1 only
    assert two["config"]["output"].startswith("one file per rank")          # the default layout for N > 1 ...
    shared = _bench(["--gpus", "2", "--one-file", "--genome-len", "500000", "--pairs", "60000", "--steps", "1", "--warmup", "1"], env=env, tmp=tmp_path)
    assert shared["config"]["output"] == "one SAM file"                       # ... and all ranks writing one file by offset
    assert shared["mapped_reads_per_step"] == one["mapped_reads_per_step"] and shared["config"]["sam_bytes_per_step"] == one["config"]["sam_bytes_per_step"]


def test_bench_eight_ranks_on_one_device_write_the_single_process_sam(tmp_path):
    """the N > 1 path with EIGHT ranks at reduced size (VERDICT r4 #7c): every rank a `-shard r/8` mapping run over its own contiguous chunk
    range, the final counter all-reduce over all eight, and the parts in rank order byte-identical to the one-process SAM (sha256 of the
    last step's output, KART_BENCH_HASH_OUTPUT).  On a 1-GPU box the ranks share device 0 over gloo: this says nothing about scaling."""
    import torch
    env = {"KART_BENCH_HASH_OUTPUT": "1"}
    common = ["--genome-len", "500000", "--pairs", "160000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--no-seeding-leg"]
    one = _bench(common, env=env, tmp=tmp_path)
    if torch.cuda.device_count() < 8:
        env = dict(env, KART_BENCH_SHARE_DEVICE="1", KART_AMD_STREAM_LANES="2", KART_AMD_SEED_GROUP="0")     # (eight processes' lanes on one device: two each)
    eight = _bench(["--gpus", "8"] + common, env=env, tmp=tmp_path)
    assert eight["n_gpus"] == 8 and eight["config"]["reads_per_step"] == 320000 and eight["config"]["reads_per_gpu_per_step"] == 40000
    assert eight["mapped_reads_per_step"] == one["mapped_reads_per_step"]                   # totals[1] all-reduced over the eight ranks
    assert eight["config"]["sam_bytes_per_step"] == one["config"]["sam_bytes_per_step"]
    assert eight["config"]["sam_sha256_last_step"] == one["config"]["sam_sha256_last_step"]
    assert eight["host_cpu"]["host_cpu_seconds_per_read"] > 0 and eight["host_cpu"]["implied_ceiling_reads_per_s"] > 0
