"""GPU: the code paths an hg38-sized index takes, on an index large enough to behave like one: a 50 Mbp repeat-rich
synthetic genome (100 M symbols, MinSeedLength 14, q-mer table q = 13, 24-bit+ ranks), built by the GPU index writer, 64-bit
kernel variants forced (KG_FORCE_U64), 1 M reads -- seeds (and candidates of a prefix) bit-identical to the CPU oracle.
Runs tools/parity_large.py in a child process (the kernel-variant switches are read once per process)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env,sa,reads", [({"KG_FORCE_U64": "1"}, "full", 1_000_000), ({}, "full", 400_000), ({"KG_FORCE_U64": "1"}, "sampled", 200_000),
                                          ({"KG_FORCE_U64": "1", "KG_NO_PLANES3": "1"}, "full", 200_000),      # at most double steps
                                          ({"KG_FORCE_U64": "1", "KG_NO_PLANES2": "1"}, "full", 200_000)])     # single steps only
def test_50mbp_index_gpu_equals_oracle(env, sa, reads, tmp_path_factory):
    wd = str(tmp_path_factory.getbasetemp() / "large50")          # the three cases share one index
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_large.py"), "--genome-len", "50000000", "--reads", str(reads), "--sa", sa,
                        "--workdir", wd], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
    assert r.returncode == 0, (r.stdout.decode()[-600:], r.stderr.decode()[-1200:])
    res = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert res["seeds_identical"] and res["candidates_identical"] and res["reads"] == reads and res["seeds"] > reads
    assert res["force_u64"] == ("KG_FORCE_U64" in env)
    # the search took the steps the index allows: triples where the three-step planes are resident, doubles without them
    st = res["steps"]
    if "KG_NO_PLANES2" in env:
        assert st["double_steps"] == 0 and st["rank_steps"] > 0
    elif "KG_NO_PLANES3" in env:
        assert st["double_steps"] > 0 and st["triple_steps"] == 0
    else:
        assert st["triple_steps"] > 0
