#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native Kart hot path on synthetic 150 bp paired-end reads.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = the configuration BASELINE.json's metric is quoted on, 150 bp PE on hg38 (configs[2]): there is no
network for the real FASTA, so a seeded hg38-SIZED synthetic genome stands in (3.1 Gbp, 45 % of it mutated copies
of a 300 bp and a 6 kb repeat family, behind a 2 kb decoy contig; generated on the device).  Its FM-index
(2L = 6.2 G symbols) is built here by kart_amd.index_build on the GPU (~35 s) and loaded with the full suffix
array (61 GB in HBM); 10 M read pairs (20 M reads of 150 bp, 1 % substitution errors + 0.1 % haplotype
substitutions) are generated directly in HBM.  If the large index cannot be built on the machine the run falls
back to configs[1] (E. coli-sized, 4,639,675 bp) and says so in config.fallback; `--genome-len` selects a size.
One "step" = one pass of the GPU hot path over one batch of 20 M resident reads:
kg_seed_batch_device = search (FM-index backward search) + scan + locate (SA recovery) + sort.
Every rank owns one GPU with a replicated index and its own read shard (weak scaling, no data-path
collective); RCCL is used only for the final counter all-reduce.

The JSON line carries `roofline` for the dominant kernel (search_kernel; HIP events recorded on
the launch stream by the library) and `cpu_baseline` (the CPU oracle port of the same step, all
host cores, on a bounded sample; rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GENOME_LEN = 4_639_675            # configs[1]: E. coli-sized
HG38_LEN = 3_100_000_000          # configs[2]: hg38-sized (synthetic stand-in: there is no network for the real FASTA)
DECOY_LEN = 2_000
READ_LEN = 150
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_genome(seed, length=GENOME_LEN):
    from kart_amd import synth
    return synth.make_genome([("decoy", DECOY_LEN), ("chrE", length)], seed=seed, gc=0.508)


def contig_table(length):
    """(name, offset, len) of the sequence contigs behind the decoy: one contig up to 300 Mbp, else 24 equal
    "chromosomes" (contig lengths are 32-bit in the BWA/Kart index format, reference src/structure.h:44-50)."""
    if length < 300_000_000:
        return [("chrE", DECOY_LEN, length)]
    n = 24
    part = length // n
    out, off = [], DECOY_LEN
    for i in range(n):
        ln = part if i < n - 1 else length - part * (n - 1)
        out.append(("chr%d" % (i + 1), off, ln))
        off += ln
    return out


def make_large_codes(length, seed, dev, repeat_frac=0.45):
    """hg38-like synthetic genome for the large experiments, generated on the device: uniform random bases with
    `repeat_frac` of the positions overwritten by mutated copies of a short (300 bp) and a long (6 kb) repeat
    family (10 % / 5 % divergence), so that multi-hit seeds and freq > 50 drops occur.
    Returns uint8 codes [DECOY_LEN + length] (the decoy contig is plain random)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    total = DECOY_LEN + length
    codes = torch.randint(0, 4, (total,), generator=g, device=dev, dtype=torch.uint8)
    for fam_len, share, div in ((300, 0.6, 0.10), (6000, 0.4, 0.05)):
        fam = torch.randint(0, 4, (fam_len,), generator=g, device=dev, dtype=torch.uint8)
        n_copies = int(length * repeat_frac * share / fam_len)
        per = max(1, (1 << 26) // fam_len)
        ar = torch.arange(fam_len, device=dev)
        # copies of one family sit on distinct slots of a fam_len grid, so the scatter below never writes a
        # position twice (overlapping writes would make the genome depend on the write order)
        import math
        n_slots = length // fam_len
        stride = int(n_slots * 0.6180339887) | 1
        while math.gcd(stride, n_slots) != 1:
            stride += 2
        slots = (torch.arange(n_copies, device=dev, dtype=torch.int64) * stride + 12345) % n_slots   # a permutation prefix: distinct slots
        for a in range(0, n_copies, per):
            m = min(per, n_copies - a)
            starts = DECOY_LEN + slots[a:a + m] * fam_len
            vals = fam.repeat(m, 1)
            mut = torch.rand(vals.shape, generator=g, device=dev) < div
            vals = torch.where(mut, torch.randint(0, 4, vals.shape, generator=g, device=dev, dtype=torch.uint8), vals)
            codes[(starts[:, None] + ar).reshape(-1)] = vals.reshape(-1)
    return codes


def gen_reads_device(genome_codes, n_pairs, seed, err, dev):
    """(enc uint8 [2*n_pairs*150], offsets int64) on the device, reads as the mapper sees them:
    mate 1 as sequenced, mate 2 reverse-complemented (reference src/GetData.cpp:125-135)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    L = genome_codes.numel()
    enc = torch.empty(2 * n_pairs * READ_LEN, dtype=torch.uint8, device=dev)
    view = enc.view(n_pairs, 2, READ_LEN)
    ar = torch.arange(READ_LEN, device=dev)
    chunk = 1 << 20
    for s in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - s)
        frag = torch.clamp((torch.randn(m, generator=g, device=dev) * 50 + 500).round().long(), min=READ_LEN)
        pos = DECOY_LEN + (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (L - DECOY_LEN - frag)).long()
        left = genome_codes[pos[:, None] + ar]                       # forward strand, fragment start
        right_fwd = genome_codes[(pos + frag - READ_LEN)[:, None] + ar]  # forward strand, fragment end
        flip = torch.rand(m, generator=g, device=dev) < 0.5
        # pair orientation: read 1 forward / read 2 (after the mapper's revcomp) forward, or both on the reverse strand
        rc = lambda x: (3 - x).flip(1)
        r1 = torch.where(flip[:, None], rc(right_fwd), left)
        r2 = torch.where(flip[:, None], rc(left), right_fwd)
        both = torch.stack([r1, r2], 1)
        e = torch.rand(both.shape, generator=g, device=dev) < err
        bump = torch.randint(1, 4, both.shape, generator=g, device=dev, dtype=torch.uint8)
        both = torch.where(e, (both + bump) & 3, both)
        view[s:s + m] = both
    offsets = torch.arange(0, 2 * n_pairs + 1, device=dev, dtype=torch.int64) * READ_LEN
    return enc, offsets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=10_000_000, help="read pairs per GPU per step")
    ap.add_argument("--sa", choices=["sampled", "full"], default="full")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the bounded FASTQ->SAM leg")
    ap.add_argument("--bucketed", action="store_true", default=None, help="force the bucketed (large-N) suffix-array builder")
    ap.add_argument("--repeat-frac", type=float, default=0.45, help="EXPERIMENT: share of the large synthetic genome covered by the two repeat families (default 0.45 = configs[2])")
    ap.add_argument("--genome-len", type=int, default=None,
                    help="synthetic genome length; default = hg38-sized (configs[2], the size BASELINE.json's metric is quoted on), "
                         "falling back to configs[1] (4,639,675) if the large index cannot be built on this machine")
    args = ap.parse_args()
    fallback_note = None
    if args.genome_len is None:
        args.genome_len = HG38_LEN
        try:
            return run(args, None)
        except Exception as exc:   # large path failed (disk, memory, ...): measure configs[1] instead and say so
            import traceback
            traceback.print_exc()
            fallback_note = "hg38-sized workload failed on this machine (%s: %s); fell back to configs[1]" % (type(exc).__name__, str(exc)[:200])
            args.genome_len = GENOME_LEN
            os.environ.pop("KART_REF_FASTA", None)
            try:
                import torch.distributed as dist
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
    return run(args, fallback_note)


def run(args, fallback_note):

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # debug aid for 1-GPU boxes: KART_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and uses gloo, so the
    # multi-rank control flow can be exercised where only one GPU exists (never set by the driver)
    share = os.environ.get("KART_BENCH_SHARE_DEVICE") == "1"
    if share:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    from kart_amd import api, index_build, shard, synth

    # ---- index (built once by rank 0, replicated per GPU) ------------------------------------------
    workdir = os.environ.get("KART_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "kart_bench_%d" % os.getuid())
    os.makedirs(workdir, exist_ok=True)
    prefix = os.path.join(workdir, "ecoli_like" if args.genome_len == GENOME_LEN else "synth_v2_%d%s%s" % (args.genome_len, "_b" if args.bucketed else "", "" if args.repeat_frac == 0.45 else "_r%g" % args.repeat_frac))   # v2: 24 contigs above 300 Mbp
    t_idx = time.time()
    large = args.genome_len >= 300_000_000
    ref_fa = os.environ.get("KART_REF_FASTA")
    if ref_fa and not os.path.exists(ref_fa):
        raise RuntimeError("KART_REF_FASTA=%s does not exist" % ref_fa)
    if ref_fa:
        # a real reference (hg38 when the box has it): index it with this repository's writer, simulate
        # the reads from its forward strand on the device
        fwd, anns, ambs = index_build.pack_contigs(index_build.read_fasta(ref_fa))
        args.genome_len = int(len(fwd))
        large = True
        prefix = os.path.join(workdir, "ref_%s_%d" % (os.path.basename(ref_fa).replace(".", "_"), os.path.getsize(ref_fa)))
    have_index = all(os.path.exists(prefix + e) for e in (".bwt", ".sa", ".pac", ".ann", ".amb"))
    if ref_fa:
        codes = torch.from_numpy(fwd).to(dev)
        genome = None
        if rank == 0 and not have_index:
            index_build.build_index_from_codes(fwd, anns, ambs, prefix + ".tmp", device=str(dev), bucketed=args.bucketed, verbose=True)
            for e in (".bwt", ".sa", ".pac", ".ann", ".amb"):
                os.replace(prefix + ".tmp" + e, prefix + e)
            torch.cuda.empty_cache()
        del fwd
    elif large:
        # large experiments: hg38-like codes made on the device, no FASTA round trip
        codes = make_large_codes(args.genome_len, seed=3, dev=dev, repeat_frac=args.repeat_frac)
        genome = None
        if rank == 0 and not have_index:
            anns = [("decoy", "(null)", 0, DECOY_LEN, 0)] + [(nm, "(null)", off, ln, 0) for nm, off, ln in contig_table(args.genome_len)]
            index_build.build_index_from_codes(codes.cpu().numpy(), anns, [], prefix + ".tmp", device=str(dev), bucketed=args.bucketed, verbose=True)
            for e in (".bwt", ".sa", ".pac", ".ann", ".amb"):
                os.replace(prefix + ".tmp" + e, prefix + e)
            torch.cuda.empty_cache()
    else:
        genome = make_genome(seed=2, length=args.genome_len)
        if rank == 0 and not have_index:
            fa = prefix + ".fa"
            synth.write_fasta(fa, genome)
            index_build.build_index(fa, prefix + ".tmp", device=str(dev))
            for e in (".bwt", ".sa", ".pac", ".ann", ".amb"):
                os.replace(prefix + ".tmp" + e, prefix + e)
        codes = torch.from_numpy(np.concatenate([synth.encode(genome["decoy"]), synth.encode(genome["chrE"])])).to(dev)
    if world > 1:
        dist.barrier()
    t_build = time.time() - t_idx
    ix = api.Index(prefix, local, api.KG_SA_FULL if args.sa == "full" else api.KG_SA_SAMPLED)
    t_idx = time.time() - t_idx

    # ---- resident inputs ----------------------------------------------------------------------------
    n_pairs = args.pairs
    n_reads = 2 * n_pairs
    n_bases = n_reads * READ_LEN
    batches = [gen_reads_device(codes, n_pairs, seed=1000 + 17 * rank + b, err=0.011, dev=dev) for b in range(2)]
    seed_cap = (12 if large else 6) * n_reads + 1024
    d_seed_off = torch.empty(n_reads + 1, dtype=torch.int64, device=dev)
    d_seeds = torch.empty(seed_cap * 16, dtype=torch.uint8, device=dev)
    ws = api.Workspace(ix, n_reads, n_bases)
    ws.set_profiling(True)
    stream = torch.cuda.current_stream(dev).cuda_stream
    mode = api.KG_MODE_FAST

    def step(b):
        enc, off = batches[b % 2]
        ws.seed_batch_device(enc.data_ptr(), off.data_ptr(), n_reads, n_bases, d_seed_off.data_ptr(), d_seeds.data_ptr(),
                             seed_cap, mode, stream=stream)

    # ---- parity spot check on this very input (not timed) ------------------------------------------
    parity = "skipped"
    orc = None
    if rank == 0:
        from oracle import oracle as O
        step(0)
        torch.cuda.synchronize(dev)
        assert ws.overflow() == 0, "seed buffer too small"
        k = 4000
        enc_h = batches[0][0][: k * READ_LEN].cpu().numpy()
        off_h = np.arange(k + 1, dtype=np.int64) * READ_LEN
        orc = O.Oracle(prefix)
        so_o, s_o = orc.seed_batch(enc_h, off_h, 0, threads=min(8, effective_cores()))
        so_g = d_seed_off[: k + 1].cpu().numpy()
        s_g = d_seeds[: int(so_g[k]) * 16].cpu().numpy().view(api.SEED_DT)
        assert (so_g == so_o).all() and (s_g == s_o.astype(api.SEED_DT)).all(), "GPU seeds differ from the oracle"
        parity = "ok (%d reads bit-identical to the oracle)" % k

    # ---- timed region ---------------------------------------------------------------------------------
    for w in range(args.warmup):
        step(w)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s)
        # event queries happen after the step has been enqueued; they synchronise on the step's last
        # event, which is inside the timed region anyway (steps are serialised on one stream)
        kernel_ms.append(ws.kernel_ms())
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    cnt = ws.counters()
    cdev = None if share else dev
    totals = shard.allreduce_counters([n_reads * args.steps, int(cnt.seeds)], device=cdev)   # the path's only collective
    elapsed = shard.max_over_ranks(elapsed, device=cdev)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    kms = np.array(kernel_ms)                                  # (steps, 4): search, scan, locate, sort
    search_ms = float(kms[:, 0].mean())
    c = cnt.as_dict()
    search_bytes = 64 * (c["lf1"] + 2 * c["lf2"]) + c["bases"]            # search kernel's share of bytes_seed
    locate_bytes = 64 * c["inv"] + 8 * c["sa"] + 16 * c["seeds"]
    achieved = search_bytes / (search_ms * 1e-3) / 1e9
    value = float(totals[0]) / elapsed
    traffic, traffic_src = measured_traffic(n_reads, args)
    line = {
        "metric": "mapped reads/sec (whole node), 150 bp PE",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": ("KART_REF_FASTA=%s (%d bp, real FASTA, ambiguous bases replaced as the index does)" % (os.path.basename(ref_fa), args.genome_len) if ref_fa else
                                "configs[1]: E. coli-like 4.64 Mbp" if args.genome_len == GENOME_LEN else
                                "configs[2]: hg38-sized (3.1 Gbp, 45 pct repeat families)" if args.genome_len == HG38_LEN and args.repeat_frac == 0.45 else
                                "EXPERIMENT: hg38-sized (3.1 Gbp), %g pct repeat families" % (100 * args.repeat_frac) if args.genome_len == HG38_LEN else
                                "EXPERIMENT: %d bp%s" % (args.genome_len, " hg38-like (45 pct repeats)" if large else "")) + (" genome, " if ref_fa else " synthetic genome, ") + "%d x 150 bp PE reads per GPU per step, "
                               "1%% substitution errors + 0.1%% haplotype substitutions; step = seeding hot path "
                               "(BWT search + SA locate + sort) on HBM-resident reads" % n_reads,
                   "reads_per_gpu_per_step": n_reads, "sa_mode": args.sa, "index_bytes": int(ix.info.device_bytes),
                   "index_residency": ("index (%.1f MB) fits the 256 MiB Infinity Cache; reads stream from HBM" if ix.info.device_bytes < 256e6
                                       else "index (%.1f MB) exceeds the 256 MiB Infinity Cache: rank gathers are HBM accesses") % (ix.info.device_bytes / 1e6),
                   "parallelism": "read-sharded x%d, index replicated" % world,
                   "fallback": fallback_note,
                   "index_build_s": round(t_build, 2), "index_build_plus_load_s": round(t_idx, 2), "parity_sample": parity},
        "roofline": {"bound": "hbm", "kernel": "search_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": search_bytes, "avg_launch_ms": search_ms,
                     "traffic_GBps": (traffic / (search_ms * 1e-3) / 1e9) if traffic else None,
                     "note": "achieved = the reference algorithm's block reads (SURVEY 8d: 64 B per LF step, two for a two-block step, + read bases) "
                             "per second.  Since the single-suffix searches finish by text comparison the kernel no longer performs most of those "
                             "reads, so achieved can exceed the HBM peak; traffic/traffic_GBps are what the kernel really moved (PMC)."},
        "kernels_ms": {"search": search_ms, "scan": float(kms[:, 1].mean()), "locate": float(kms[:, 2].mean()), "sort": float(kms[:, 3].mean())},
        "bytes_seed_per_read": (search_bytes + locate_bytes) / n_reads,
        "work_per_read": {k2: v / n_reads for k2, v in c.items()},
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["nw_kernels"] = nw_leg(ix, dev)
        except Exception as exc:      # a side measurement must never cost the line
            line["nw_kernels"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:120])}
        line["cpu_baseline"] = cpu_baseline(orc, batches[0][0], READ_LEN, prefix, workdir)
    if world == 1 and not args.no_e2e:
        del d_seeds, d_seed_off, batches
        if orc is not None:
            orc.close()
        ws.close()
        ix.close()                      # the CLI loads its own copy of the index
        torch.cuda.empty_cache()
        line["end_to_end"] = end_to_end(prefix, genome, workdir, n_pairs=250_000 if genome is None else 500_000, codes=codes if genome is None else None)
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def effective_cores():
    """CPUs this process can actually use: the cgroup CPU quota when there is one (the GPU boxes expose 256
    logical CPUs under a 16-core quota), else the affinity mask / CPU count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def measured_traffic(n_reads, args):
    """HBM/fabric bytes per search_kernel launch from the committed PMC passes of this exact command
    (rocprofv3 --pmc, separate passes; profiles/*_pmc_summary.json, corrected for gfx950 as
    MI355X_MICROARCH.md prescribes).  bench.py cannot run the profiler on itself, so the figure is
    only reported when the workload matches the profiled one; otherwise null."""
    best = None
    if getattr(args, "repeat_frac", 0.45) != 0.45 or os.environ.get("KART_REF_FASTA"):
        return None, None          # the profiled workload is the default one
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if f.endswith("_pmc_summary.json"):
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", f))).get("_search_traffic")
            except Exception:
                t = None
            if t and t.get("reads_per_launch") == n_reads and t.get("genome_len", GENOME_LEN) == args.genome_len:
                best = (t["traffic_bytes_per_launch"], "profiles/" + f)
    return best if best else (None, None)


def write_fastq_from_codes(codes, n_pairs, seed, f1, f2, dev):
    """FASTQ pair files from the device read generator (large genomes: no host-side copy of the genome exists)."""
    enc, _ = gen_reads_device(codes, n_pairs, seed=seed, err=0.011, dev=dev)
    arr = enc.view(n_pairs, 2, READ_LEN).cpu().numpy()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.frombuffer(b"TGCA", dtype=np.uint8)
    r1 = acgt[arr[:, 0, :]]
    r2 = comp[arr[:, 1, ::-1]]        # the generator holds mate 2 as the mapper does (reverse-complemented): undo that
    qual = b"5" * READ_LEN
    for path, rr, mate in ((f1, r1, 1), (f2, r2, 2)):
        with open(path, "wb") as fh:
            for i in range(n_pairs):
                fh.write(b"@r%d\t/%d\n" % (i, mate) + rr[i].tobytes() + b"\n+\n" + qual + b"\n")


def nw_leg(ix, dev):
    """Gap-closing kernels on device-resident fragment batches (not `value`): pairs/s and GCUPS (DP cells per second) per size
    class -- integer DP is bound by VALU/LDS throughput, not by memory.  1-8 bases is the 97 % case of 150 bp reads."""
    rng = np.random.default_rng(0)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    for name, n, lo, hi in (("1-8", 4_000_000, 1, 9), ("9-32", 1_000_000, 9, 33), ("33-128", 100_000, 33, 129), ("300-1100", 2_000, 300, 1101)):
        m = rng.integers(lo, hi, size=n)
        k = np.clip(m + rng.integers(-3, 4, size=n), lo, hi - 1)
        o1 = np.zeros(n + 1, np.int64); o2 = np.zeros(n + 1, np.int64)
        np.cumsum(m, out=o1[1:]); np.cumsum(k, out=o2[1:])
        f1 = lut[torch.randint(0, 4, (int(o1[-1]) + 16,), device=dev).long()]
        f2 = lut[torch.randint(0, 4, (int(o2[-1]) + 16,), device=dev).long()]
        d1, d2 = torch.from_numpy(o1).to(dev), torch.from_numpy(o2).to(dev)
        ops = torch.empty(int(o1[-1] + o2[-1]) + 16, dtype=torch.uint8, device=dev)
        ln = torch.empty(n, dtype=torch.int32, device=dev)

        def call():
            rc = ix.lib.kg_nw_batch_device(ix.h, f1.data_ptr(), d1.data_ptr(), f2.data_ptr(), d2.data_ptr(), n, int(max(m.max(), k.max())), ops.data_ptr(), ln.data_ptr(), stream)
            assert rc == 0, ix.lib.kg_last_error()
        call(); torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(3):
            call()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t) / 3
        out[name] = {"pairs_per_s": round(n / dt), "GCUPS": round(float((m.astype(np.float64) * k).sum()) / dt / 1e9, 1)}
        del f1, f2, ops, ln
    return out


def end_to_end(prefix, genome, workdir, n_pairs=500_000, codes=None):
    """Bounded FASTQ -> SAM leg (not `value`): kart_amd/bin/kart-amd (host pipeline + the same kernels) on 1 M reads of
    the same genome, and -- when the unmodified reference binary travelled with the snapshot -- oracle/_ref/kart on
    the same files at -t 1 (the byte-identity check) and -t 32 (its best setting on this host class)."""
    import subprocess
    from kart_amd import synth
    exe = os.path.join(ROOT, "kart_amd", "bin", "kart-amd")
    ref = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(exe):
        return None
    f1, f2 = os.path.join(workdir, "e2e_1.fq"), os.path.join(workdir, "e2e_2.fq")
    if codes is not None:
        write_fastq_from_codes(codes, n_pairs, 5, f1, f2, codes.device)
    else:
        names, r1, r2 = synth.simulate_pairs(genome, n_pairs, seed=5, err=0.01)
        synth.write_fastq(f1, names, r1, mate=1)
        synth.write_fastq(f2, names, r2, mate=2)
    common = ["-silent", "-i", prefix, "-f", f1, "-f2", f2]
    out = {"reads": 2 * n_pairs, "unit": "reads/s"}

    def run(cmd):
        t = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, KART_AMD_VERBOSE="1"))
        dt = time.perf_counter() - t
        ms = [float(l.split(":")[1]) for l in r.stdout.decode().splitlines() if l.startswith("mapping seconds")]
        return r.returncode, dt, (ms[0] if ms else None)

    threads = effective_cores()          # more workers than the CPU quota only adds context switches
    rc, dt, ms = run([exe] + common + ["-t", str(threads), "-o", os.path.join(workdir, "e2e_amd.sam")])
    if rc != 0:
        return {"error": "kart-amd failed"}
    out["kart_amd"] = {"threads": threads, "process_seconds": round(dt, 3), "mapping_seconds": ms,
                       "reads_per_s_mapping_phase": round(2 * n_pairs / ms) if ms else None, "reads_per_s_process": round(2 * n_pairs / dt)}
    if os.path.exists(ref):
        rc1, dt1, _ = run([ref] + common + ["-t", "1", "-o", os.path.join(workdir, "e2e_ref1.sam")])
        rcn, dtn, _ = run([ref] + common + ["-t", str(threads), "-o", os.path.join(workdir, "e2e_refn.sam")])
        if rc1 == 0 and rcn == 0:
            out["reference_kart"] = {"t1_reads_per_s": round(2 * n_pairs / dt1), "t%d_reads_per_s" % threads: round(2 * n_pairs / dtn)}
            out["sam_identical_to_reference_t1"] = open(os.path.join(workdir, "e2e_amd.sam"), "rb").read() == open(os.path.join(workdir, "e2e_ref1.sam"), "rb").read()
    # a larger sample for steady-state rates (the small one above is dominated by the batch ramp and the index load; it is
    # small because the reference's -t 1 run, the identity check, maps ~10 k reads/s); reference at -t <threads> only
    big = 4 * n_pairs if codes is not None else 8 * n_pairs
    g1, g2 = os.path.join(workdir, "e2e_big_1.fq"), os.path.join(workdir, "e2e_big_2.fq")
    try:
        if codes is not None:
            write_fastq_from_codes(codes, big, 6, g1, g2, codes.device)
        else:
            names, r1, r2 = synth.simulate_pairs(genome, big, seed=6, err=0.01)
            synth.write_fastq(g1, names, r1, mate=1)
            synth.write_fastq(g2, names, r2, mate=2)
        bigc = ["-silent", "-i", prefix, "-f", g1, "-f2", g2]
        rc, dt, ms = run([exe] + bigc + ["-t", str(threads), "-o", os.path.join(workdir, "e2e_big_amd.sam")])
        if rc == 0:
            out["steady_state"] = {"reads": 2 * big, "kart_amd": {"process_seconds": round(dt, 3), "mapping_seconds": ms,
                                                                  "reads_per_s_mapping_phase": round(2 * big / ms) if ms else None, "reads_per_s_process": round(2 * big / dt)}}
            if os.path.exists(ref):
                rcn, dtn, _ = run([ref] + bigc + ["-t", str(threads), "-o", os.path.join(workdir, "e2e_big_ref.sam")])
                if rcn == 0:
                    out["steady_state"]["reference_kart_t%d" % threads] = {"process_seconds": round(dtn, 3), "reads_per_s_process": round(2 * big / dtn)}
    except Exception as exc:      # the extra sample must never cost the line
        out["steady_state"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:120])}
    for f in (f1, f2, g1, g2, "e2e_amd.sam", "e2e_ref1.sam", "e2e_refn.sam", "e2e_big_amd.sam", "e2e_big_ref.sam"):
        try:
            os.remove(f if os.path.isabs(f) else os.path.join(workdir, f))
        except OSError:
            pass
    return out


_REF_CHILD = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
prefix, npy, threads = sys.argv[2], sys.argv[3], int(sys.argv[4])
enc = np.load(npy)
n = len(enc) // 150
off = np.arange(n + 1, dtype=np.int64) * 150
sh = O.RefShim(prefix, threads=threads)
k = min(n, 20000)
t = time.perf_counter(); sh.seed_batch_count(enc[: k * 150], off[: k + 1], 0, threads); rate = k / (time.perf_counter() - t)
m = int(min(n, max(k, rate * 12)))
t = time.perf_counter(); seeds = sh.seed_batch_count(enc[: m * 150], off[: m + 1], 0, threads); dt = time.perf_counter() - t
print("REFJSON " + json.dumps({"reads": m, "seconds": dt, "seeds": seeds}))
"""


def reference_baseline(prefix, enc_host, cores, workdir):
    """The reference's own object code (oracle/_ref/libkartref_shim.so: IdentifySeedPairs_FastMode -> BWT_Search -> bwt_sa, built
    from the sources where they lie) on the same reads, `cores` threads, in a child process (it keeps its index in
    process globals and exits on errors).  None when oracle/_ref did not travel."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libkartref_shim.so")):
        return None
    npy = os.path.join(workdir, "cpu_baseline_sample.npy")
    np.save(npy, enc_host)
    try:
        r = subprocess.run([sys.executable, "-c", _REF_CHILD, ROOT, prefix, npy, str(cores)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
        for line in r.stdout.decode().splitlines():
            if line.startswith("REFJSON "):
                return json.loads(line[8:])
    except Exception:
        pass
    finally:
        try:
            os.remove(npy)
        except OSError:
            pass
    return None


def cpu_baseline(orc, enc_dev, read_len, prefix=None, workdir=None):
    """The same step (seeding incl. SA locate and sort) on the host cores, on a bounded sample of the same reads (sized
    for roughly 10-20 s of CPU work): the reference's own object code when oracle/_ref travelled with the repository
    ("reference"), and this repository's CPU restatement of it ("port")."""
    cores = effective_cores()
    k = 20000
    enc = enc_dev[: k * read_len].cpu().numpy()
    off = np.arange(k + 1, dtype=np.int64) * read_len
    t = time.perf_counter()
    orc.seed_batch(enc, off, 0, threads=cores)
    rate = k / (time.perf_counter() - t)
    n = int(min(enc_dev.numel() // read_len, max(k, rate * 12)))
    enc = enc_dev[: n * read_len].cpu().numpy()
    off = np.arange(n + 1, dtype=np.int64) * read_len
    t = time.perf_counter()
    orc.seed_batch(enc, off, 0, threads=cores)
    dt = time.perf_counter() - t
    port = {"value": n / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "%d reads of the same batch, oracle/liboracle.so seed_batch (FM search + SA locate + sort), %d threads = the cgroup CPU quota "
                      "(%d logical CPUs visible), %.1f s" % (n, cores, os.cpu_count() or 1, dt)}
    ref = reference_baseline(prefix, enc, cores, workdir) if prefix and workdir else None
    if not ref:
        return port
    return {"value": ref["reads"] / ref["seconds"], "unit": "reads/s", "cores": cores, "kind": "reference",
            "sample": "%d reads of the same batch through the reference's own IdentifySeedPairs_FastMode (oracle/_ref/libkartref_shim.so, compiled from "
                      "the reference sources), %d threads = the cgroup CPU quota (%d logical CPUs visible), %.1f s" % (ref["reads"], cores, os.cpu_count() or 1, ref["seconds"]),
            "port": port}


if __name__ == "__main__":
    main()
