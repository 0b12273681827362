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


@pytest.mark.parametrize("flags", [[], ["-m"]])
def test_50mbp_index_sam_equals_live_reference(flags, tmp_path_factory, tmp_path):
    """the alignment stage's large-coordinate paths (64-bit kernels forced, 24 contigs, positions beyond 2^25) through the whole
    product: FASTQ -> SAM of 100 k pairs on the 50 Mbp index, byte-identical to the reference's -t 1 (with -m up to the FLAGs the
    reference never assigns, which are found from the reference alone)"""
    from test_host_pipeline import UNSET_FLAG, assert_sam_equals_reference_with_its_own_mask, reference_sam_and_never_assigned_flags
    kart_ref = os.path.join(ROOT, "oracle", "_ref", "kart")
    kart_amd = os.path.join(ROOT, "kart_amd", "bin", "kart-amd")
    assert os.path.exists(kart_ref), "oracle/_ref/kart did not travel to the GPU box"
    wd = str(tmp_path_factory.getbasetemp() / "large50")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "large_sam_inputs.py"), "--genome-len", "50000000", "--pairs", "100000", "--workdir", wd, "--err", "0.015"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-1200:]
    inp = json.loads(r.stdout.decode().strip().splitlines()[-1])
    args = ["-i", inp["prefix"], "-f", inp["f1"], "-f2", inp["f2"]] + flags
    ref_lines, never = reference_sam_and_never_assigned_flags(kart_ref, args, str(tmp_path))
    out = str(tmp_path / "amd.sam")
    for env in ({"KG_FORCE_U64": "1"}, {"KG_FORCE_U64": "1", "KART_AMD_NO_STREAM": "1"}):          # through the device stream, and through the host's reader / printer
        p = subprocess.run([kart_amd, "-silent", "-t", "8", "-o", out] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           env=dict(os.environ, KART_AMD_UNSET_FLAG=str(UNSET_FLAG), KART_AMD_VERBOSE="1", **env))
        assert p.returncode == 0, p.stdout.decode()[-800:]
        masked = assert_sam_equals_reference_with_its_own_mask(ref_lines, never, open(out, "rb").read())
        assert masked == 0 or flags == ["-m"]
        log = p.stdout.decode()
        dev = [l for l in log.splitlines() if l.startswith("device report:") and "decided on the device" in l][0]
        assert int(dev.split()[2]) > 190000, dev
