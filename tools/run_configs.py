#!/usr/bin/env python3
"""BASELINE.json configs[3] and configs[4] at a stated, large size on the hg38-sized synthetic index of bench.py (GPU box):
    [3] -pacbio : N x 7000 bp single-end reads at 15 % (substitution) error
    [4] -m      : N pairs of 150 bp at 2 % error + 0.1 % haplotype substitutions, multi-hit output
each with: kart-amd on the full set, the unmodified reference at -t <host quota> on the full set, and byte identity of kart-amd
with the reference's -t 1 on a prefix (FLAG masked for -m exactly where the reference never assigns it, KART_AMD_UNSET_FLAG).
MEASUREMENT TOOL.  usage: python tools/run_configs.py [pacbio_reads] [multihit_pairs]   -> JSON on stdout"""
import json, os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_long = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
n_mh = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
L = bench.HG38_LEN
dev = torch.device("cuda", 0)
wd = bench.pick_workdir(80 << 30)
prefix = os.path.join(wd, "synth_v2_%d" % L)
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "1000000", "--leg", "seeding", "--seed-steps", "1"], stdout=subprocess.DEVNULL)   # builds + caches the index
codes = bench.make_large_codes(L, 3, dev)
exe, ref = os.path.join(ROOT, "kart_amd", "bin", "kart-amd"), os.path.join(ROOT, "oracle", "_ref", "kart")
if os.environ.get("RUN_CONFIGS_NO_REF"):
    ref = "/nonexistent"          # A/B runs of kart-amd alone
cores = bench.effective_cores()
UNSET = 1 << 20
res = {"genome_len": L, "host_cpu_quota": cores}


def write_long_reads(path, n, read_len, err, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev)
    with open(path, "wb") as fh:
        for s in range(0, n, 20000):
            m = min(20000, n - s)
            pos = bench.DECOY_LEN + (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (L - read_len - 1)).long()
            r = codes[pos[:, None] + ar]
            flip = torch.rand(m, generator=g, device=dev) < 0.5
            r = torch.where(flip[:, None], (3 - r).flip(1), r)
            e = torch.rand(r.shape, generator=g, device=dev) < err
            r = torch.where(e, (r + torch.randint(1, 4, r.shape, generator=g, device=dev, dtype=torch.uint8)) & 3, r)
            txt = acgt[r.long()].cpu().numpy()
            q = b"5" * read_len
            for i in range(m):
                fh.write(b"@L%d\n" % (s + i) + txt[i].tobytes() + b"\n+\n" + q + b"\n")


def run(cmd, env=None):
    t = time.perf_counter()
    err = open(os.environ["RUN_CONFIGS_STDERR"], "ab") if os.environ.get("RUN_CONFIGS_STDERR") else subprocess.DEVNULL
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=err, env=dict(os.environ, KART_AMD_VERBOSE="1", **(env or {})))
    dt = time.perf_counter() - t
    out = {"rc": r.returncode, "process_seconds": round(dt, 2)}
    for line in r.stdout.decode().splitlines():
        if line.startswith("mapping seconds"):
            out["mapping_seconds"] = float(line.split(":")[1])
        if line.startswith(("stage seconds", "worker thread-seconds", "device report", "All the", "chunks re-mapped", "fragment pairs", "cpu seconds")):
            out.setdefault("log", []).append(line.strip()[:400])
    return out


def head(src, dst, n_lines):
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        for _ in range(n_lines):
            l = fi.readline()
            if not l:
                break
            fo.write(l)


def same_up_to_unset_flags(a, b):
    la, lb = a.split(b"\n"), b.split(b"\n")
    if len(la) != len(lb):
        return False, 0
    masked = 0
    for x, y in zip(la, lb):
        if x == y:
            continue
        fx, fy = x.split(b"\t"), y.split(b"\t")
        if len(fy) < 2 or int(fy[1]) != UNSET or fx[:1] + fx[2:] != fy[:1] + fy[2:]:
            return False, masked
        masked += 1
    return True, masked


# ---- configs[3]: -pacbio ----------------------------------------------------------------------------------------------------------
def config3():
    fq = os.path.join(wd, "cfg3_long.fq")
    write_long_reads(fq, n_long, 7000, 0.15, seed=31)
    c3 = {"reads": n_long, "read_len": 7000, "error": 0.15}
    a = run([exe, "-silent", "-i", prefix, "-f", fq, "-pacbio", "-t", str(cores), "-o", os.path.join(wd, "cfg3_amd.sam")])
    c3["kart_amd"] = dict(a, reads_per_s_mapping_phase=round(n_long / a["mapping_seconds"]) if a.get("mapping_seconds") else None, reads_per_s_process=round(n_long / a["process_seconds"]))
    if os.path.exists(ref):
        b = run([ref, "-silent", "-i", prefix, "-f", fq, "-pacbio", "-t", str(cores), "-o", os.path.join(wd, "cfg3_ref.sam")])
        c3["reference_t%d" % cores] = dict(b, reads_per_s_process=round(n_long / b["process_seconds"]))
        c3["same_records_as_reference"] = sorted(open(os.path.join(wd, "cfg3_amd.sam"), "rb").read().split(b"\n")) == sorted(open(os.path.join(wd, "cfg3_ref.sam"), "rb").read().split(b"\n"))
        k = min(n_long, 3000)
        pq = os.path.join(wd, "cfg3_prefix.fq")
        head(fq, pq, 4 * k)
        run([exe, "-silent", "-i", prefix, "-f", pq, "-pacbio", "-t", str(cores), "-o", os.path.join(wd, "cfg3_p_amd.sam")])
        r1 = run([ref, "-silent", "-i", prefix, "-f", pq, "-pacbio", "-t", "1", "-o", os.path.join(wd, "cfg3_p_ref.sam")])
        c3["prefix_identity_vs_reference_t1"] = {"reads": k, "identical": open(os.path.join(wd, "cfg3_p_amd.sam"), "rb").read() == open(os.path.join(wd, "cfg3_p_ref.sam"), "rb").read(),
                                                 "reference_t1_process_seconds": r1["process_seconds"]}
    res["configs[3] -pacbio"] = c3
    if os.environ.get("RUN_CONFIGS_KEEP_INPUTS"):          # (profiling: the same command again under rocprofv3)
        c3["command"] = [exe, "-silent", "-i", prefix, "-f", fq, "-pacbio", "-t", str(cores), "-o", os.path.join(wd, "cfg3_amd.sam")]
        return
    for f in ("cfg3_long.fq", "cfg3_amd.sam", "cfg3_ref.sam", "cfg3_prefix.fq", "cfg3_p_amd.sam", "cfg3_p_ref.sam"):
        try: os.remove(os.path.join(wd, f))
        except OSError: pass


# ---- configs[4]: -m, 2 % error ---------------------------------------------------------------------------------------------------
def config4():
    f1, f2 = os.path.join(wd, "cfg4_1.fq"), os.path.join(wd, "cfg4_2.fq")
    bench.write_fastq_pairs(codes, n_mh, 41, f1, f2, dev, err=0.02)
    c4 = {"reads": 2 * n_mh, "error": 0.02}
    a = run([exe, "-silent", "-i", prefix, "-f", f1, "-f2", f2, "-m", "-t", str(cores), "-o", os.path.join(wd, "cfg4_amd.sam")])
    c4["kart_amd"] = dict(a, reads_per_s_mapping_phase=round(2 * n_mh / a["mapping_seconds"]) if a.get("mapping_seconds") else None, reads_per_s_process=round(2 * n_mh / a["process_seconds"]))
    # the same reads without -m (same device report, one record per read)
    a2 = run([exe, "-silent", "-i", prefix, "-f", f1, "-f2", f2, "-t", str(cores), "-o", os.path.join(wd, "cfg4_amd_nom.sam")])
    c4["kart_amd_without_m"] = dict(a2, reads_per_s_mapping_phase=round(2 * n_mh / a2["mapping_seconds"]) if a2.get("mapping_seconds") else None)
    if os.path.exists(ref):
        b = run([ref, "-silent", "-i", prefix, "-f", f1, "-f2", f2, "-m", "-t", str(cores), "-o", os.path.join(wd, "cfg4_ref.sam")])
        c4["reference_t%d" % cores] = dict(b, reads_per_s_process=round(2 * n_mh / b["process_seconds"]))
        k = min(n_mh, 100_000)
        p1, p2 = os.path.join(wd, "cfg4_p1.fq"), os.path.join(wd, "cfg4_p2.fq")
        head(f1, p1, 4 * k); head(f2, p2, 4 * k)
        ident = {}
        for tag, flags in (("with_m", ["-m"]), ("without_m", [])):
            run([exe, "-silent", "-i", prefix, "-f", p1, "-f2", p2, "-t", str(cores), "-o", os.path.join(wd, "cfg4_p_amd.sam")] + flags, env={"KART_AMD_UNSET_FLAG": str(UNSET)})
            run([ref, "-silent", "-i", prefix, "-f", p1, "-f2", p2, "-t", "1", "-o", os.path.join(wd, "cfg4_p_ref.sam")] + flags)
            ok, masked = same_up_to_unset_flags(open(os.path.join(wd, "cfg4_p_ref.sam"), "rb").read(), open(os.path.join(wd, "cfg4_p_amd.sam"), "rb").read())
            ident[tag] = {"reads": 2 * k, "identical_up_to_never_assigned_flags": ok, "records_with_never_assigned_flag": masked}
            # the device report against the host's implementation of the same reference code, record by record
            chk = subprocess.run([exe, "-silent", "-i", prefix, "-f", p1, "-f2", p2, "-t", str(cores), "-o", os.path.join(wd, "cfg4_p_amd.sam")] + flags, stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, env=dict(os.environ, KART_AMD_VERBOSE="1", KART_AMD_CHECK_ALIGN="1", KART_AMD_UNSET_FLAG=str(UNSET)))
            ident[tag]["check_align"] = [l for l in chk.stdout.decode().splitlines() if l.startswith("CHECK_ALIGN")]
        c4["prefix_identity_vs_reference_t1"] = ident
    res["configs[4] -m"] = c4
    for f in ("cfg4_1.fq", "cfg4_2.fq", "cfg4_amd.sam", "cfg4_amd_nom.sam", "cfg4_ref.sam", "cfg4_p1.fq", "cfg4_p2.fq", "cfg4_p_amd.sam", "cfg4_p_ref.sam"):
        try: os.remove(os.path.join(wd, f))
        except OSError: pass

if n_long > 0:
    config3()
if n_mh > 0:
    config4()
print(json.dumps(res))
