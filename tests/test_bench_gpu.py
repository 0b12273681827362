"""bench.py's contract on a real GPU: one JSON line with the keys the driver and the judge read, on a reduced
workload (the default workload is the hg38-sized one and takes minutes), and the KART_REF_FASTA route."""
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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract(tmp_path):
    d = _bench(["--genome-len", "500000", "--pairs", "50000", "--steps", "2", "--warmup", "1", "--no-e2e"], tmp=tmp_path)
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "reads/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 100000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "reads/s"
    if c["kind"] == "reference":      # oracle/_ref travelled: the reference's own object code is the baseline, the port rides along
        assert c["port"]["kind"] == "port" and c["port"]["value"] > 0
    assert d["config"]["parity_sample"].startswith("ok") and "workload" in d["config"] and "model" not in d["config"]


def test_bench_real_fasta_route(tmp_path):
    fa = os.path.join(ROOT, "tests", "golden", "small.fa")
    d = _bench(["--pairs", "20000", "--steps", "1", "--warmup", "1", "--no-e2e", "--no-cpu-baseline"], env={"KART_REF_FASTA": fa}, tmp=tmp_path)
    assert "KART_REF_FASTA=small.fa" in d["config"]["workload"] and d["config"]["parity_sample"].startswith("ok")
    assert d["config"]["fallback"] is None and d["roofline"]["traffic"] is None
