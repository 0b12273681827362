#!/usr/bin/env python3
"""bench.py -- FASTQ -> SAM throughput of the MI355X-native Kart hot path on synthetic 150 bp paired-end reads.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: this script starts the N ranks itself -- torch.distributed.run children, before anything
     here touches a GPU -- and relays rank 0's line; under torchrun it is one of the ranks)

Metric (BASELINE.json): mapped reads/sec of the whole job, 150 bp paired-end on hg38 (configs[2]).  There is no network for
the real FASTA, so a seeded hg38-SIZED synthetic genome stands in (3.1 Gbp in 24 contigs behind a 2 kb decoy, 45 % of it
mutated copies of a 300 bp and a 6 kb repeat family, generated on the device); its FM-index (2L = 6.2 G symbols) is built by
kart_amd.index_build on the GPU and loaded with the full suffix array, the 2-bit text, the 4^16-entry q-mer table and the two- /
three-step rank planes (168 GB of HBM per GPU; --sa compact: 67 GB).  configs[2] names 100 M reads: 50 M read pairs after wgsim's
model (two haplotypes with substitutions and indels, recurrent 1 % sequencing errors: benchkit/reads.py) are written as two FASTQ files (33 GB) when the HOST can hold them and two outputs
(see "Footprint"), otherwise fewer; `--pairs` fixes the number.  `KART_REF_FASTA=<fa>` benchmarks a real reference instead;
`--genome-len` selects another synthetic size; if the large index cannot be built on the machine a single-rank run falls back
to configs[1] and says so in config.fallback.

One "step" = one complete mapping run over those files -- the reference's Mapping() (src/Mapping.cpp:639-742): FASTQ parsed,
seeds found (FM-index search + SA locate + sort), chained, paired / rescued, gaps closed (NW), SAM text written to a FRESH output
file -- through the host library (include/kart_host.h: index resident across runs, as the metric excludes the index load).
`value` = mapped reads of the K timed steps / the sum of their wall times (every step bracketed by torch.cuda.synchronize, max
over ranks per step); the whole loop is bracketed by a barrier + synchronize on both sides (config.bracket_seconds).  With N ranks
the SAME reads are split into N contiguous chunk ranges, one process + one GPU + one index replica each (strong scaling), the
output byte-identical to one process mapping everything; RCCL carries only the final counter all-reduce.

Footprint (round 3 lost its GPU box here): the files live in /dev/shm, i.e. in RAM charged to this process group.  The job is sized
from min(MemAvailable, cgroup memory.max - memory.current) -- not from the tmpfs's nominal size -- so that FASTQ + TWO step
outputs stay within 40 % of it; at most two step outputs ever exist (all but the newest are removed between the steps, outside the
per-step timers), they are gone before the side legs write their own files, and the observed Shmem high-water mark is reported
(config.peak_shmem_GB).  With N ranks the figure is the same (the FASTQ is written once, the N parts add up to one output); each
rank adds its host copy of the reference (2 B per base: 6.2 GB) and ~4 GB of page-locked lane buffers.

The JSON line is printed TWICE: the headline (value, roofline from the timed region, cpu_baseline) as soon as those exist, flushed,
and the same line enriched by the side legs at the very end ("line": "final") -- a leg that dies late cannot cost the measurement.
It carries: `roofline` for the dominant GPU kernel, taken FROM THE TIMED REGION (search_kernel: bytes the IMPLEMENTED search needs --
exported by the kernel itself, kg_workspace_traffic -- summed over the launches of the timed steps, over the sum of their HIP-event
durations on the lanes' streams, against the 8 TB/s HBM peak; `traffic` from the committed PMC pass of this command when there is
one), `cpu_baseline` (the unmodified reference binary at -t <host quota> on a bounded prefix of the same files, same box, same run),
then `parity` (SAM byte identity with the reference's -t 1 on a 0.3 M-read prefix, and GPU seeds == CPU oracle on a random 200 k-read
sample; the 0.5 M-read / -m / -pacbio identities on this index are tests/test_hg38_gpu.py), `other_configs` (configs[4] -m and
configs[3] -pacbio through the same session), `seeding_stage` (the GPU seeding step alone on HBM-resident reads in ONE launch of 20 M
reads -- a sub-field with its own roofline) and `nw_kernels`.  Rank 0 at N = 1 only for everything but `value`.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
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


# the read generators (wgsim's model: two haplotypes with substitutions AND indels, recurrent sequencing errors, its names and quality
# character; records of varying width) live in benchkit/reads.py
from benchkit.reads import (MAX_REC_BYTES, copy_records, gen_reads_device, release_haplotypes, split_records, write_fastq_pairs,   # noqa: E402,F401
                            write_long_reads)

REC_BYTES = MAX_REC_BYTES          # sizing bound only: a prefix of such a file is a number of records (copy_records), not a byte range


def _read_int(path):
    try:
        t = open(path).read().strip()
        return None if t == "max" else int(t)
    except (OSError, ValueError):
        return None


def host_memory():
    """What the machine can still hold in RAM, in bytes: min(MemAvailable, the cgroup's memory.max - memory.current).  Files in
    /dev/shm are RAM (shmem) charged to the cgroup of whoever touches the pages first; the tmpfs's own size limit (what
    shutil.disk_usage / fstatvfs report) is NEITHER of these and is only a further cap.  Round 3 sized the job from the tmpfs
    figure, kept 25 fresh 37.8 GB outputs and lost the box."""
    mi = {}
    try:
        for l in open("/proc/meminfo"):
            k, v = l.split(":", 1)
            mi[k] = int(v.split()[0]) * 1024
    except OSError:
        pass
    avail = mi.get("MemAvailable")
    out = {"MemTotal": mi.get("MemTotal"), "MemAvailable": avail, "Shmem": mi.get("Shmem")}
    cg_max = cg_cur = None
    for base in ("/sys/fs/cgroup", ):
        cg_max, cg_cur = _read_int(base + "/memory.max"), _read_int(base + "/memory.current")
        if cg_max is None and cg_cur is None:          # cgroup v1
            cg_max, cg_cur = _read_int(base + "/memory/memory.limit_in_bytes"), _read_int(base + "/memory/memory.usage_in_bytes")
    if cg_max is not None and cg_max >= (1 << 60):
        cg_max = None                                  # (v1's "no limit")
    out["cgroup_max"], out["cgroup_current"] = cg_max, cg_cur
    cands = [x for x in (avail, (cg_max - (cg_cur or 0)) if cg_max is not None else None) if x is not None]
    out["usable"] = min(cands) if cands else None
    try:
        du = shutil.disk_usage("/dev/shm")
        out["shm_free"] = du.free
    except OSError:
        out["shm_free"] = None
    return out


def shmem_now():
    try:
        for l in open("/proc/meminfo"):
            if l.startswith("Shmem:"):
                return int(l.split()[1]) * 1024
    except OSError:
        pass
    return 0


MEM_SHARE = 0.40                 # the job's files (FASTQ + two step outputs + the side legs' files) may take this much of host_memory()["usable"]
SAM_BYTES_PER_READ = 400         # ~378 measured; bound used for sizing


def job_bytes(n_pairs):
    """tmpfs bytes alive at the same time while the steps run: the two FASTQ files + at most TWO step outputs"""
    return 2 * n_pairs * REC_BYTES + 2 * (2 * n_pairs) * SAM_BYTES_PER_READ


PROCESS_SHARE = 0.80             # files + what the N rank processes themselves hold (below) may take this much of it


def rank_bytes(lanes, large):
    """host memory of one rank process besides the files: the host copy of the reference (2 B per base: forward + reverse strand), the
    stream lanes' page-locked buffers (~1.8 GB each at 1 M-read batches: two text windows, the SAM text, records, candidates), and
    python + torch + the HIP runtime"""
    return (2 * HG38_LEN if large else 2 * GENOME_LEN) + lanes * int((1800 << 20) * max(1.0, stream_reads_setting() / 1120000.0)) + (4 << 30)


def pick_pairs(mem, large, world=1, lanes=8):
    """read pairs per step: configs[2]'s 50 M (100 M reads) if FASTQ + two outputs stay within MEM_SHARE of what the host can
    hold AND files + the rank processes' own memory within PROCESS_SHARE, else the largest multiple of 5 M that does (at least 1 M)"""
    full = 50_000_000 if large else 10_000_000
    usable = mem.get("usable")
    if usable is None:
        return min(full, 10_000_000)
    budget = min(usable * MEM_SHARE, usable * PROCESS_SHARE - world * rank_bytes(lanes, large))
    if mem.get("shm_free") is not None:
        budget = min(budget, mem["shm_free"] * 0.8)
    n = full
    while n > 1_000_000 and job_bytes(n) > budget:
        n -= 5_000_000 if n > 5_000_000 else 1_000_000
    return max(n, 1_000_000)


def pick_workdir(need_bytes):
    """KART_BENCH_DIR, else /dev/shm when the HOST can hold the files (they stay in memory: no disk in the timed region), else the temp dir"""
    d = os.environ.get("KART_BENCH_DIR")
    if d:
        return d
    mem = host_memory()
    try:
        if (mem["usable"] or 0) * MEM_SHARE > need_bytes and shutil.disk_usage("/dev/shm").free > need_bytes:
            return os.path.join("/dev/shm", "kart_bench_%d" % os.getuid())
    except OSError:
        pass
    return os.path.join(tempfile.gettempdir(), "kart_bench_%d" % os.getuid())


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=None, help="read pairs of the whole job (split over the ranks); default 50 M = the 100 M reads of configs[2] "
                                                             "when the work directory holds them and their SAM, else 10 M")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[3] / configs[4] sub-lines")
    ap.add_argument("--parts", action="store_true", help="N > 1: one output file per rank (kart-amd -parts; their concatenation is the single-process SAM) -- the default for N > 1")
    ap.add_argument("--one-file", action="store_true", help="N > 1: all ranks write into ONE shared SAM file by offset (the ranks then meet at that file's page-cache locks: ~20 GB/s from one L3 domain, less from several)")
    ap.add_argument("--leg", choices=["all", "seeding"], default="all", help="seeding: only the GPU seeding step on resident reads (what the rocprofv3 passes profile)")
    ap.add_argument("--seed-steps", type=int, default=5, help="timed launches of the seeding-stage leg")
    ap.add_argument("--sa", choices=["auto", "sampled", "full", "compact", "wide", "dense4", "dense8"], default="auto",
                    help="suffix array placement (seeding-stage leg and, via KART_AMD_SA, the mapping runs); auto = the product's default: full below 2^32 text symbols; above, wide (5-byte suffix array, full q-mer table, "
                         "triple planes: ~150 GB for a human-sized index) where the device has the room, else compact (67 GB); dense4 / dense8 = the smaller index")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpu-pipeline", action="store_true", help="skip the secondary leg whose text goes to /dev/null (gpu_pipeline_value)")
    ap.add_argument("--no-parity", action="store_true", help="skip the reference -t 1 identity leg and the oracle sample")
    ap.add_argument("--no-seeding-leg", action="store_true")
    ap.add_argument("--bucketed", action="store_true", default=None, help="force the bucketed (large-N) suffix-array builder")
    ap.add_argument("--repeat-frac", type=float, default=0.45, help="EXPERIMENT: share of the large synthetic genome covered by the two repeat families (default 0.45 = configs[2])")
    ap.add_argument("--genome-len", type=int, default=None,
                    help="synthetic genome length; default = hg38-sized (configs[2], the size BASELINE.json's metric is quoted on), "
                         "falling back to configs[1] (4,639,675) if the large index cannot be built on this machine (single rank only)")
    ap.add_argument("--threads", type=int, default=None, help="worker threads per rank (default: the host CPU quota / ranks)")
    args = ap.parse_args()
    args.parts = not args.one_file          # (N > 1) a part per rank unless asked otherwise; `kart-amd -gpu a,b,..` itself defaults to one file

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # no launcher around us: start the ranks as children BEFORE anything here initialises a GPU, relay rank 0's line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, stdout=subprocess.PIPE)
        out = r.stdout.decode()
        lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
        sys.stdout.write(lines[-1] + "\n" if lines else out)
        return r.returncode if r.returncode else (0 if lines else 1)
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))

    fallback_note = None
    if args.genome_len is None:
        args.genome_len = HG38_LEN
        if world > 1:
            return run(args, None)         # several ranks: no per-rank fallback (the ranks could not agree on it); a failure ends the job
        try:
            return run(args, None)
        except Exception as exc:   # large path failed (disk, memory, ...): measure configs[1] instead and say so
            import traceback
            traceback.print_exc()
            fallback_note = "hg38-sized workload failed on this machine (%s: %s); fell back to configs[1]" % (type(exc).__name__, str(exc)[:200])
            args.genome_len = GENOME_LEN
            os.environ.pop("KART_REF_FASTA", None)
    return run(args, fallback_note)


def prepare_index(args, dev, rank, workdir, barrier):
    """(prefix, codes on the device, build seconds).  Rank 0 builds (or finds) the index files; everybody regenerates the
    genome codes on its own device (seeded, identical)."""
    from kart_amd import index_build, synth
    prefix = os.path.join(workdir, "ecoli_like" if args.genome_len == GENOME_LEN else "synth_v2_%d%s%s" % (args.genome_len, "_b" if args.bucketed else "", "" if args.repeat_frac == 0.45 else "_r%g" % args.repeat_frac))
    t0 = time.time()
    large = args.genome_len >= 300_000_000
    ref_fa = os.environ.get("KART_REF_FASTA")
    if ref_fa and not os.path.exists(ref_fa):
        raise RuntimeError("KART_REF_FASTA=%s does not exist" % ref_fa)
    exts = (".bwt", ".sa", ".pac", ".ann", ".amb")
    if ref_fa:
        fwd, anns, ambs = index_build.pack_contigs(index_build.read_fasta(ref_fa))
        args.genome_len = int(len(fwd))
        prefix = os.path.join(workdir, "ref_%s_%d" % (os.path.basename(ref_fa).replace(".", "_"), os.path.getsize(ref_fa)))
        codes = torch.from_numpy(fwd).to(dev)
        if rank == 0 and not all(os.path.exists(prefix + e) for e in exts):
            index_build.build_index_from_codes(fwd, anns, ambs, prefix + ".tmp", device=str(dev), bucketed=args.bucketed, verbose=True)
            for e in exts:
                os.replace(prefix + ".tmp" + e, prefix + e)
        del fwd
    elif large or args.genome_len != GENOME_LEN:
        codes = make_large_codes(args.genome_len, seed=3, dev=dev, repeat_frac=args.repeat_frac)
        if rank == 0 and not all(os.path.exists(prefix + e) for e in exts):
            anns = [("decoy", "(null)", 0, DECOY_LEN, 0)] + [(nm, "(null)", off, ln, 0) for nm, off, ln in contig_table(args.genome_len)]
            index_build.build_index_from_codes(codes.cpu().numpy(), anns, [], prefix + ".tmp", device=str(dev), bucketed=args.bucketed, verbose=True)
            for e in exts:
                os.replace(prefix + ".tmp" + e, prefix + e)
    else:
        genome = make_genome(seed=2, length=args.genome_len)
        if rank == 0 and not all(os.path.exists(prefix + e) for e in exts):
            fa = prefix + ".fa"
            synth.write_fasta(fa, genome)
            index_build.build_index(fa, prefix + ".tmp", device=str(dev))
            for e in exts:
                os.replace(prefix + ".tmp" + e, prefix + e)
        codes = torch.from_numpy(np.concatenate([synth.encode(genome["decoy"]), synth.encode(genome["chrE"])])).to(dev)
    torch.cuda.empty_cache()
    barrier()
    return prefix, codes, time.time() - t0


def workload_name(args):
    ref_fa = os.environ.get("KART_REF_FASTA")
    if ref_fa:
        return "KART_REF_FASTA=%s (%d bp, real FASTA, ambiguous bases replaced as the index does)" % (os.path.basename(ref_fa), args.genome_len)
    if args.genome_len == GENOME_LEN:
        return "configs[1]: E. coli-like 4.64 Mbp synthetic genome"
    if args.genome_len == HG38_LEN and args.repeat_frac == 0.45:
        return "configs[2]: hg38-sized (3.1 Gbp, 45 pct repeat families) synthetic genome"
    if args.genome_len == HG38_LEN:
        return "EXPERIMENT: hg38-sized (3.1 Gbp) synthetic genome, %g pct repeat families" % (100 * args.repeat_frac)
    return "EXPERIMENT: %d bp synthetic genome, hg38-like (45 pct repeats)" % args.genome_len


def run(args, fallback_note):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # debug aid for 1-GPU boxes: KART_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and uses gloo, so the
    # multi-rank control flow can be exercised where only one GPU exists (never set by the driver)
    share = os.environ.get("KART_BENCH_SHARE_DEVICE") == "1"
    if share:
        local = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    barrier()
    from kart_amd import api, shard

    mem0 = host_memory()
    if world >= 4 and "KART_AMD_STREAM_LANES" not in os.environ and "KART_AMD_SEED_GROUP" not in os.environ:
        # N >= 4 ranks share one host: one seeding group of four lanes per rank instead of two (every rank maps 1/N of the reads, and
        # eight ranks' 64 lanes would pin ~115 GB of host memory beside the files)
        os.environ["KART_AMD_STREAM_LANES"] = "4"
        os.environ["KART_AMD_SEED_GROUP"] = "4"
    if args.pairs is None:
        # configs[2] names 100 M reads: 32 GB of FASTQ and 38 GB of SAM per step, page-cache resident -- when the HOST can hold
        # FASTQ + two outputs within MEM_SHARE of min(MemAvailable, cgroup headroom); otherwise fewer
        pick = [pick_pairs(mem0, args.genome_len >= 300_000_000, world, seed_group_setting()[1])]
        if world > 1:
            from kart_amd import shard as _sh
            pick = [-_sh.max_over_ranks(-float(pick[0]), device=None if share else dev)]      # the smallest pick of any rank, for all
        args.pairs = int(pick[0])
    n_pairs = args.pairs
    n_reads = 2 * n_pairs
    workdir = pick_workdir(2 * n_pairs * REC_BYTES + n_reads * 450 + (12 << 30))
    os.makedirs(workdir, exist_ok=True)
    prefix, codes, t_build = prepare_index(args, dev, rank, workdir, barrier)
    large = args.genome_len >= 300_000_000

    if args.leg == "seeding":
        if rank == 0:
            line = seeding_leg(args, api, prefix, codes, dev, n_reads_leg=min(n_reads, 20_000_000), oracle_sample=0)
            print(json.dumps({"metric": "seeding stage only (profiling leg)", "seeding_stage": line}))
        if world > 1:
            dist.destroy_process_group()
        return 0

    # ---- the input files (rank 0 writes them; every rank maps its chunk range of the same two files) ----------------
    f1, f2 = os.path.join(workdir, "bench_1.fq"), os.path.join(workdir, "bench_2.fq")
    t0 = time.time()
    read_stats = {}
    if rank == 0:
        read_stats = write_fastq_pairs(codes, n_pairs, 5, f1, f2, dev, err=0.01)
        release_haplotypes()          # (6 GB of HBM; the side legs derive them again -- same seed, same haplotypes)
    t_fastq = time.time() - t0
    barrier()
    if rank != 0:
        del codes
    torch.cuda.empty_cache()

    # ---- the session: index resident on this rank's GPU -----------------------------------------------------------------
    cores = effective_cores()
    threads = args.threads or max(2, cores // world)
    t0 = time.time()
    if args.sa != "auto":
        os.environ["KART_AMD_SA"] = args.sa          # (read by the host library when it loads the index)
    resolved_sa(args, dev)
    sess = api.HostSession(prefix, local, threads)
    t_load = time.time() - t0
    # At most TWO step outputs exist at any time: step k writes a fresh file (a mapping run creates its output -- reusing a
    # file's pages would be cheaper than what the metric times) and step k-2's file is removed BETWEEN the steps, outside the
    # per-step timers.  Round 3 kept all of them (25 x 37.8 GB of shmem under the driver's flags) and lost the box.
    outs = []
    peak_shmem = [shmem_now()]

    def out_files(out):
        return shard.part_files(out, world, args.parts)

    def drop(out):
        if rank == 0:
            for f in out_files(out):
                try:
                    os.remove(f)
                except OSError:
                    pass

    def step(tag):
        out = os.path.join(workdir, "bench_out_%s.sam" % tag)
        outs.append(out)
        # this rank's share of the run: -shard r/N -rendezvous <file> [-parts] behind the run's own options (kart_amd/shard.py)
        return shard.map_shard(sess.map, ["-silent", "-f", f1, "-f2", f2], out, rank, world, os.path.join(workdir, "rdv_%s" % tag), args.parts)

    def between_steps():
        """not timed: note the shmem high-water mark, remove all but the newest output, line the ranks up for the next step"""
        peak_shmem[0] = max(peak_shmem[0], shmem_now())
        barrier()
        while len(outs) > 1:
            drop(outs.pop(0))
        clean_rendezvous()
        barrier()

    def clean_rendezvous():
        if rank == 0:
            for f in os.listdir(workdir):
                if f.startswith("rdv_"):
                    os.remove(os.path.join(workdir, f))

    if rank == 0:
        for f in os.listdir(workdir):                  # (left behind by a run that died)
            if f.startswith("bench_out_"):
                os.remove(os.path.join(workdir, f))
    clean_rendezvous()
    barrier()
    # (the warm-up runs also sum their SAM text on the device -- KG_STREAM_CHECKSUM, ~12 ms per run, untimed: the gpu_pipeline leg below,
    #  whose text never reaches a file, must have made the same text)
    warm_stats = []
    os.environ["KG_STREAM_CHECKSUM"] = "1"
    for w in range(args.warmup):
        warm_stats.append(step("w%d" % w))
        between_steps()
    del os.environ["KG_STREAM_CHECKSUM"]
    torch.cuda.synchronize(dev)
    barrier()
    t0 = time.perf_counter()
    stats, step_wall = [], []
    import resource
    cpu_steps = 0.0                                   # user + system CPU seconds of this rank's process inside the timed steps (the library's threads are its own)
    for s_ in range(args.steps):
        torch.cuda.synchronize(dev)
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        ts = time.perf_counter()
        stats.append(step("s%d" % s_))
        torch.cuda.synchronize(dev)
        step_wall.append(time.perf_counter() - ts)
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu_steps += (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
        if s_ + 1 < args.steps:
            between_steps()
    torch.cuda.synchronize(dev)
    barrier()
    bracket = time.perf_counter() - t0
    peak_shmem[0] = max(peak_shmem[0], shmem_now())
    cdev = None if share else dev
    reads_mapped_here = sum(int(st.total_reads - st.unmapped) for st in stats)
    reads_here = sum(int(st.total_reads) for st in stats)
    totals = shard.allreduce_counters([reads_here, reads_mapped_here, sum(int(st.respeculated) for st in stats)], device=cdev)   # the path's only collective
    step_max = shard.max_over_ranks(step_wall, device=cdev)       # a step lasts as long as its slowest rank
    cpu_all = shard.allreduce_counters([int(cpu_steps * 1e6)], device=cdev)[0] / 1e6     # (microseconds: the counters travel as integers)
    elapsed = float(sum(step_max))                                # the K timed steps; what lies between them (removing old outputs) is not a mapping run
    bracket = shard.max_over_ranks(bracket, device=cdev)
    clean_rendezvous()
    # ---- the ranks one by one: what each one's device, copy engines and lane threads did per step (rank order) ------------------------
    K_ = float(args.steps)
    lanes_here = max(1, int(stats[-1].lanes))
    rank_rows = shard.gather_rows([
        sum(float(st.stage_ms[i]) for st in stats for i in range(5)) / K_,          # kernels of all lanes, summed (lanes overlap)
        sum(float(st.stage_ms[5]) for st in stats) / K_,                             # copies back to the host
        sum(float(st.lane_seconds[0]) for st in stats) / K_ / lanes_here * 1e3,      # per lane: held back by the writer
        sum(float(st.lane_seconds[4]) for st in stats) / K_ / lanes_here * 1e3,      # per lane: inside the device call
        sum(float(st.lane_seconds[5]) for st in stats) / K_ / lanes_here * 1e3,      # per lane: the reads handed back
        sum(float(st.lane_seconds[1]) for st in stats) / K_ / lanes_here * 1e3,      # per lane: reading + uploading its input block
        sum(float(st.lane_seconds[2]) for st in stats) / K_ / lanes_here * 1e3,      # per lane: waiting for the batch before to be parsed
        sum(float(st.lane_seconds[3]) for st in stats) / K_ / lanes_here * 1e3,      # per lane: the parse call
        sum(float(st.map_seconds) for st in stats) / K_ * 1e3,
        cpu_steps / K_, float(lanes_here), sum(float(st.total_reads) for st in stats) / K_], device=cdev)
    # ---- secondary: the same steps with the text left in the lanes' page-locked buffers (output to /dev/null, KART_AMD_OUTPUT_NULL) and
    #      summed on the device instead: what the GPU pipeline + copy engines deliver when the host's page cache is out of the way.  Never the headline.
    pipe = None
    if not args.no_gpu_pipeline:
        os.environ["KART_AMD_OUTPUT_NULL"] = "1"; os.environ["KG_STREAM_CHECKSUM"] = "1"
        try:
            p_wall, p_stats = [], []
            ru_p0 = resource.getrusage(resource.RUSAGE_SELF)
            for s_ in range(min(args.steps, 3)):
                between_steps()
                torch.cuda.synchronize(dev)
                ts = time.perf_counter()
                # (not through step(): no output file comes of it, `outs` keeps naming the last real one)
                p_stats.append(shard.map_shard(sess.map, ["-silent", "-f", f1, "-f2", f2], os.path.join(workdir, "bench_null_%d.sam" % s_), rank, world, os.path.join(workdir, "rdv_p%d" % s_), True))
                torch.cuda.synchronize(dev)
                p_wall.append(time.perf_counter() - ts)
            between_steps()
        finally:
            del os.environ["KART_AMD_OUTPUT_NULL"]; del os.environ["KG_STREAM_CHECKSUM"]
        p_max = shard.max_over_ranks(p_wall, device=cdev)
        p_tot = shard.allreduce_counters([sum(int(st.total_reads - st.unmapped) for st in p_stats), int(p_stats[-1].text_checksum[0]), int(p_stats[-1].text_checksum[1]),
                                          int(warm_stats[-1].text_checksum[0]) if warm_stats else 0, int(warm_stats[-1].text_checksum[1]) if warm_stats else 0,
                                          int(p_stats[-1].text_out_bytes), int(warm_stats[-1].text_out_bytes) if warm_stats else 0], device=cdev)
        ru_p1 = resource.getrusage(resource.RUSAGE_SELF)
        lp = max(1, int(p_stats[-1].lanes))
        pipe = {"value": p_tot[0] / float(sum(p_max)), "unit": "reads/s", "steps": len(p_max), "ms_per_step": [round(x * 1e3, 1) for x in p_max],
                "rank0_cpu_seconds_per_step": round(((ru_p1.ru_utime - ru_p0.ru_utime) + (ru_p1.ru_stime - ru_p0.ru_stime)) / len(p_max), 2),
                "rank0_lane_ms_per_step": {nm: round(sum(float(st.lane_seconds[i]) for st in p_stats) / len(p_stats) / lp * 1e3, 1)
                                           for i, nm in enumerate(("held_back_by_writer", "read_and_upload", "waiting_for_the_parse_before", "parse_call", "device_call", "host_reads"))},
                "rank0_device_ms_per_step": {nm: round(sum(float(st.stage_ms[i]) for st in p_stats) / len(p_stats), 1) for i, nm in enumerate(("parse", "seed", "chain", "align", "format", "copy_out"))},
                "what": "NOT the metric: the same mapping runs with the SAM text left in the lanes' page-locked buffers (written to /dev/null) and summed on the device "
                        "(sam_checksum_kernel) -- the rate of the GPU pipeline and the copy engines with the host's copy into fresh page-cache pages out of the way; "
                        "beside `value` it tells device scaling from page-cache scaling in an N > 1 line",
                "device_text_bytes": p_tot[5], "device_text_byte_sum": p_tot[1], "device_text_lines": p_tot[2],
                "file_writing_run": ({"device_text_bytes": p_tot[6], "device_text_byte_sum": p_tot[3], "device_text_lines": p_tot[4]} if warm_stats else None),
                "same_lines_as_a_file_writing_run": (bool(p_tot[2] == p_tot[4]) if warm_stats else None),
                "note": "the sums are of the text the DEVICE made: a chunk mapped under a speculated EstDistance that did not hold is mapped again by the host and its device "
                        "text dropped -- which chunks those are depends on the lanes' timing, so byte sums of two runs agree to a few chunks' worth (the files themselves are "
                        "identical: the identity legs compare those); the line count -- one line per read the device decided -- does not depend on it"}
    if rank != 0:
        sess.close()
        if world > 1:
            dist.destroy_process_group()
        return 0

    assert totals[0] == n_reads * args.steps, "the ranks together mapped %d reads per step, expected %d" % (totals[0] // max(1, args.steps), n_reads)
    value = float(totals[1]) / elapsed           # MAPPED reads of the whole job per second (BASELINE.json's metric)
    sw = np.sort(np.array(step_max))
    # the dominant kernel inside the timed region: every search_kernel launch of the timed steps (HIP events on the lanes' streams)
    sk_ms = sum(float(st.search_kernel_ms) for st in stats)
    sk_n = sum(int(st.search_kernel_launches) for st in stats)
    sk_bytes = sum(float(st.search_useful_bytes) for st in stats)
    stage_names = ("parse", "seed", "chain", "align", "format", "copy_out")
    stage_ms = {nm: sum(float(st.stage_ms[i]) for st in stats) / args.steps for i, nm in enumerate(stage_names)}
    line = {
        "metric": "mapped reads/sec (whole node), 150 bp PE",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload_name(args) + ", %d x 150 bp paired-end reads, wgsim model incl. indels (two haplotypes: mutation rate 0.001, 15 %% of them indels extended with 0.3; "
                               "recurrent substitution errors -e 0.01; wgsim's names, records of varying width) in two FASTQ files; "
                               "step = one FASTQ -> SAM mapping run of the whole job (parse, FM-index seeding, chaining, pairing / rescue, NW gap closing, SAM written), "
                               "index resident" % n_reads,
                   "reads_per_step": n_reads, "reads_per_gpu_per_step": n_reads // world,
                   "read_model": {"pairs_with_indel_share": read_stats.get("pairs_with_indel", 0) / max(1, n_pairs), "reads_with_indel_share": read_stats.get("reads_with_indel", 0) / max(1, n_reads),
                                  "reads_with_N": read_stats.get("reads_with_N", 0), "fastq_bytes": read_stats.get("fastq_bytes"),
                                  "source": "benchkit/reads.py (wgsim/wgsim.c:99-102,104-166,243-391 restated on the device)"}, "threads_per_rank": threads, "host_cpu_quota": cores,
                   "parallelism": "%d process(es), one GPU + index replica each, contiguous chunk ranges of the same input, SAM merged by file offset" % world,
                   "files": "page-cache resident (%s)" % workdir,
                   "seed_group": seed_group_setting()[0], "stream_lanes": seed_group_setting()[1], "stream_reads": stream_reads_setting(),
                   "sa_mode": resolved_sa(args),
                   "seeding": ("ONE search launch per round over the parsed batches of %d stream lanes (%d lanes in flight, 1.12 M-read batches)" % seed_group_setting()) if seed_group_setting()[0]
                              else "every stream lane seeds its own 1.12 M-read batch (%d lanes)" % seed_group_setting()[1],
                   "sam_bytes_per_step": sum(os.path.getsize(f) for f in out_files(outs[-1])),
                   "host_memory_GB": {k: (round(v / 1e9, 1) if isinstance(v, (int, float)) else v) for k, v in mem0.items()},
                   "sizing": "reads per step chosen so that FASTQ + TWO step outputs (%.1f GB) stay within %d %% of min(MemAvailable, cgroup headroom) = %s GB; every step writes a "
                             "fresh output, all but the newest are removed between the steps (outside the per-step timers)" % (job_bytes(n_pairs) / 1e9, int(MEM_SHARE * 100), "%.1f" % (mem0["usable"] / 1e9) if mem0.get("usable") else "?"),
                   "rank_processes_GB": round(world * rank_bytes(seed_group_setting()[1], large) / 1e9, 1),
                   "peak_shmem_GB": round(peak_shmem[0] / 1e9, 1), "shmem_before_GB": round((mem0.get("Shmem") or 0) / 1e9, 1),
                   "timing": "value = mapped reads of the K timed steps / the sum of their wall times (each step bracketed by torch.cuda.synchronize, max over ranks per step); "
                             "bracket_seconds = barrier-to-barrier wall of the same K steps including the removal of old outputs between them",
                   "bracket_seconds": round(bracket, 3),
                   "output": "one file per rank (-parts)" if (args.parts and world > 1) else "one SAM file",
                   "fallback": fallback_note, "index_build_s": round(t_build, 2), "index_load_s": round(t_load, 2), "fastq_write_s": round(t_fastq, 2)},
        "mapped_reads_per_step": totals[1] // args.steps, "mapped_fraction": totals[1] / max(1, totals[0]),
        "chunks_remapped_per_step": totals[2] / args.steps,
        "rank0_step_seconds": [round(x, 3) for x in step_wall], "rank0_map_seconds": [round(st.map_seconds, 3) for st in stats],
        "reads_per_sec_total": float(totals[0]) / elapsed,
        "step_seconds": {"min": float(sw[0]), "p10": float(np.percentile(sw, 10)), "median": float(np.median(sw)), "p90": float(np.percentile(sw, 90)), "max": float(sw[-1]),
                         "mean": float(sw.mean())},
        "device_ms_per_step": dict(stage_ms, what="rank 0: HIP-event time of the stages of all batches of a step, summed over the lanes (stages of different lanes overlap); "
                                                  "reads through the device stream per step: %d" % (sum(int(st.stream_reads) for st in stats) // args.steps)),
    }
    if sk_n > 0 and sk_ms > 0:
        achieved = sk_bytes / (sk_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(0, args, tag="timed")
        line["roofline"] = {"bound": "hbm", "kernel": "search_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": traffic, "traffic_source": traffic_src,
                            "source": "the timed region: %d launches in %d steps on rank 0, %.0f reads per launch on average" % (sk_n, args.steps, reads_here / max(1, sk_n)),
                            "launches": sk_n, "avg_launch_ms": sk_ms / sk_n, "algorithmic_bytes_per_launch": sk_bytes / sk_n,
                            "algorithmic_bytes_per_read": sk_bytes / max(1, reads_here), "search_kernel_ms_per_step": sk_ms / args.steps,
                            "note": "achieved = bytes the IMPLEMENTED search needs (kg_workspace_traffic's formula, summed over the launches) / the sum of the launches' HIP-event "
                                    "durations (the ramp-up launches of a step -- batches of 4 k .. 256 k reads while EstDistance settles -- included).  Other lanes' kernels "
                                    "share the device, so a launch's duration includes what they took from it."}

    # ---- the host side of the path: what a read costs in CPU time bounds the whole node whatever the number of GPUs (the ranks share the host) ----
    cpu_per_read = cpu_all / max(1, totals[0])
    line["host_cpu"] = {"cpu_seconds_per_step_all_ranks": cpu_all / args.steps, "host_cpu_seconds_per_read": cpu_per_read, "host_cpu_quota": cores,
                        "implied_ceiling_reads_per_s": (cores / cpu_per_read) if cpu_per_read > 0 else None,
                        "note": "user + system CPU seconds of the rank processes inside the timed steps (reading + uploading the FASTQ text, the handed-back reads, the output copies and "
                                "their page faults) per read; quota / that = the rate at which this host saturates however many GPUs feed it -- the bound on 1 -> N scaling on this "
                                "host shape (unmeasured beyond the GPUs this run had)"}
    # ---- the ranks one by one, and the secondary rate without the host's output copy: what makes an N > 1 line readable ----------------
    names = ("device_kernels_ms_per_step_summed_over_lanes", "copy_engine_ms_per_step", "lane_held_back_by_writer_ms_per_step", "lane_in_device_call_ms_per_step",
             "lane_host_reads_ms_per_step", "lane_read_and_upload_ms_per_step", "lane_waiting_for_the_parse_before_ms_per_step", "lane_parse_call_ms_per_step",
             "map_ms_per_step", "cpu_seconds_per_step", "lanes", "reads_per_step")
    line["per_rank"] = {"what": "rank order; per timed step.  device_kernels: HIP-event time of the stages parse .. format, summed over the rank's lanes (lanes overlap on the device, so "
                                "it can exceed the step); copy_engine: the copies back to the host; lane_*: averages over the rank's lane threads -- held back by the writer = waiting "
                                "until the host has copied the lane's previous text into the output file's pages (page-cache side), in the device call = kernels + copies of "
                                "the lane's batch (device side)",
                        "ranks": [{n: (round(v, 3) if n != "reads_per_step" else int(v)) for n, v in zip(names, row)} for row in rank_rows]}
    if pipe is not None:
        line["gpu_pipeline_value"] = pipe
    # ---- every kernel of the timed region with a share of the step: HIP events around each launch on the lanes' streams (kg_stream_timing_t::
    # kernel_ms), against the bytes the kernel has to touch (its inputs and outputs as laid out in HBM) ----------------
    line["kernels"] = kernel_entries(stats, args.steps, n_reads, sk_ms / args.steps if sk_n else 0.0, args)
    if os.environ.get("KART_BENCH_HASH_OUTPUT") == "1" and outs:
        # test aid: the last step's SAM as one stream -- the ranks' parts in rank order, or the one shared file -- so that a test can hold an
        # N-rank run to the single-process bytes
        import hashlib
        h = hashlib.sha256()
        for f in out_files(outs[-1]):
            with open(f, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    h.update(blk)
        line["config"]["sam_sha256_last_step"] = h.hexdigest()
    # the step outputs go first: the side legs below write their own files and must never add to them
    while outs:
        drop(outs.pop(0))
    emitted = [False]

    def emit(final=True):
        """the JSON line, flushed; called as soon as value + roofline + cpu_baseline exist and again, enriched, at the very end
        (a side leg that dies late must not cost the line; a reader that wants ONE line takes the last)"""
        line["line"] = "final" if final else "headline (an enriched copy follows when the side legs have run)"
        sys.stdout.write(json.dumps(line) + "\n")
        sys.stdout.flush()
        emitted[0] = True

    # the alignment stage (pairing / rescue / plan / NW / finish / records: everything kg_align_batch does) has a roofline entry of its own:
    # HIP-event time of the stage per step against the bytes it must touch -- read characters, candidates and their seeds in, the
    # per-candidate report (78 B) and one record per read (112 B) out; the 2-bit text it compares with is ~1/4 B per read base.
    n_cand = sum(float(st.candidates) for st in stats) / args.steps
    n_cseed = sum(float(st.candidate_seeds) for st in stats) / args.steps
    if stage_ms["align"] > 0 and n_cand > 0:
        aln_bytes = n_reads * (READ_LEN + READ_LEN / 4 + 112) + n_cand * (32 + 78) + n_cseed * 16
        aln_gbs = aln_bytes / (stage_ms["align"] * 1e-3) / 1e9
        line["alignment_stage"] = {"bound": "hbm", "kernels": "aln_trivial, aln_pair, aln_rescue, aln_bin, aln_plan_fast, aln_plan, aln_partition, nw_*, aln_finish, aln_final",
                                   "ms_per_step": stage_ms["align"], "candidates_per_read": n_cand / n_reads, "candidate_seeds_per_read": n_cseed / n_reads,
                                   "algorithmic_bytes_per_step": aln_bytes, "algorithmic_bytes_per_read": aln_bytes / n_reads,
                                   "achieved": aln_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": aln_gbs / HBM_PEAK_GBS,
                                   "note": "HIP-event time of the stage on the lanes' streams (a group's lanes take turns at it, so it is mostly the stage alone on the device) "
                                           "against the bytes it has to move: dependent small gathers per candidate, not streaming -- the wait share per kernel "
                                           "(SQ_WAIT_ANY / SQ_WAVE_CYCLES) is in the round's PMC summary under profiles/"}

    if world == 1 and not args.no_cpu_baseline:
        try:
            line["host_output"] = host_output_entry(workdir, line["config"]["sam_bytes_per_step"], elapsed / args.steps, threads)
        except Exception as exc:      # a side measurement must never cost the line
            line["host_output"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:160])}
    if world == 1:
        # ---- CPU baseline + parity on a prefix of the very files that were timed -------------------------------------------
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = reference_legs(args, sess, prefix, workdir, f1, f2, n_pairs, threads, cores, want="baseline").get("cpu_baseline")
            except Exception as exc:      # a side measurement must never cost the line
                line["cpu_baseline_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
        if not (args.no_seeding_leg and args.no_parity and (args.no_other_configs or not large)):
            emit(final=False)
        if not args.no_parity:
            try:
                line["parity"] = reference_legs(args, sess, prefix, workdir, f1, f2, n_pairs, threads, cores, want="identity")
            except Exception as exc:
                line["parity"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    for f in (f1, f2):
        try:
            if not os.environ.get("KART_BENCH_KEEP_FASTQ"):          # (debug aid: the two input files stay for a run of kart-amd by hand)
                os.remove(f)
        except OSError:
            pass
    if world == 1 and large and not args.no_other_configs:
        try:
            if os.environ.get("KART_BENCH_ONLY_GZ_LEG"):             # (an experiment's short cut: the gz leg alone)
                line["other_configs"] = {"gz_input": gz_leg(sess, workdir, codes, dev, threads, cores, (host_memory().get("usable") or (64 << 30)) * MEM_SHARE)}
            else:
                line["other_configs"] = other_configs(args, sess, prefix, workdir, codes, dev, threads, cores, host_memory())
        except Exception as exc:      # a side measurement must never cost the line
            line["other_configs"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    sess.close()
    torch.cuda.empty_cache()
    if world == 1 and large and not args.no_other_configs:
        # configs[1] has an index of its own: a second session, opened once the large one has left the device (beside its 168 GB + the
        # stream's lanes a second stream does not fit: the first attempt ended the run with an allocation failure, profiles/r05s)
        try:
            mem1 = host_memory()
            line.setdefault("other_configs", {})["configs[1]"] = config1(args, workdir, dev, threads, (mem1.get("usable") or (64 << 30)) * MEM_SHARE, os.path.join(ROOT, "oracle", "_ref", "kart"))
        except Exception as exc:      # a side measurement must never cost the line
            line.setdefault("other_configs", {})["configs[1]"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    if world == 1 and not args.no_seeding_leg:
        try:
            seed = seeding_leg(args, api, prefix, codes, dev, n_reads_leg=min(n_reads, 20_000_000), oracle_sample=0 if args.no_parity else 200_000)
            line["seeding_stage"] = seed
            if "roofline" in line:
                seed["roofline_single_launch"] = seed.pop("roofline")
                survey_and_traffic_fractions(line["roofline"], seed.get("reference_algorithm_bytes_per_read"), reads_here / max(1, sk_n))
                survey_and_traffic_fractions(seed["roofline_single_launch"], seed.get("reference_algorithm_bytes_per_read"), seed.get("reads_per_launch", 0))
            else:
                line["roofline"] = seed.pop("roofline")
            if "oracle_sample" in seed:
                line.setdefault("parity", {})["seeds_vs_oracle"] = seed.pop("oracle_sample")
            if not args.no_cpu_baseline:
                line["nw_kernels"] = seed.pop("nw_kernels", None)
        except AssertionError:
            raise                          # (a parity failure is not a side matter)
        except Exception as exc:
            line["seeding_stage"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    line["config"]["peak_shmem_GB_whole_run"] = round(max(peak_shmem[0], shmem_now()) / 1e9, 1)
    emit()
    if world > 1:
        dist.destroy_process_group()
    return 0


def reference_legs(args, sess, prefix, workdir, f1, f2, n_pairs, threads, cores, want):
    """On prefixes of the timed files (their first records: benchkit.reads.copy_records), same box, same run:
      want="identity" : oracle/_ref/kart -t 1 vs this pipeline, 0.3 M reads, SAM compared byte for byte (the 0.5 M-read comparison
                        on this index is tests/test_hg38_gpu.py's);
      want="baseline" : oracle/_ref/kart -t <host quota> on a prefix sized for ~15-25 s of its mapping time."""
    ref = os.path.join(ROOT, "oracle", "_ref", "kart")
    out = {}
    if not os.path.exists(ref):
        return {"note": "oracle/_ref/kart did not travel with this snapshot: no reference legs"}

    def prefix_files(tag, pairs):
        g1, g2 = os.path.join(workdir, "%s_1.fq" % tag), os.path.join(workdir, "%s_2.fq" % tag)
        for src, dst in ((f1, g1), (f2, g2)):
            copy_records(src, dst, pairs)
        return g1, g2

    def run_ref(g1, g2, t, sam):
        t0 = time.perf_counter()
        r = subprocess.run([ref, "-silent", "-i", prefix, "-f", g1, "-f2", g2, "-o", sam, "-t", str(t)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        secs = None
        for l in r.stdout.decode().splitlines():          # "All the N paired-end reads have been processed in S seconds." (src/Mapping.cpp:730)
            if "have been processed in" in l:
                secs = int(l.split(" in ")[1].split()[0])
        return r.returncode, dt, secs

    tiny1, tiny2 = prefix_files("tiny", 2)
    rc, load_s, _ = run_ref(tiny1, tiny2, cores, os.path.join(workdir, "tiny.sam"))          # ~ the reference's index load on this box
    if want == "identity":
        p = min(n_pairs, int(os.environ.get("KART_BENCH_IDENT_PAIRS", "150000")))          # (a longer identity run: one reference thread maps ~9 k reads/s)
        g1, g2 = prefix_files("ident", p)
        sam_ref, sam_amd = os.path.join(workdir, "ident_ref.sam"), os.path.join(workdir, "ident_amd.sam")
        rc1, dt1, _ = run_ref(g1, g2, 1, sam_ref)
        st = sess.map(["-silent", "-f", g1, "-f2", g2, "-o", sam_amd])
        same = rc1 == 0 and open(sam_ref, "rb").read() == open(sam_amd, "rb").read()
        out["sam_vs_reference_t1"] = {"reads": 2 * p, "identical": bool(same), "reference_t1_reads_per_s": round(2 * p / max(1e-9, dt1 - load_s)),
                                      "this_pipeline_map_seconds": round(st.map_seconds, 3)}
        for f in (g1, g2, sam_ref, sam_amd):
            os.remove(f)
    if want == "baseline":
        # size the sample from a short probe so that the reference's mapping takes ~20 s
        probe = min(n_pairs, 100_000)
        g1, g2 = prefix_files("probe", probe)
        rc, dt, _ = run_ref(g1, g2, cores, os.path.join(workdir, "probe.sam"))
        rate = 2 * probe / max(0.2, dt - load_s)
        pairs = int(min(n_pairs, max(probe, rate * 20 / 2)))
        g1, g2 = prefix_files("base", pairs)
        sam = os.path.join(workdir, "base.sam")
        rc, dt, secs = run_ref(g1, g2, cores, sam)
        map_s = max(0.5, dt - load_s)
        st = sess.map(["-silent", "-f", g1, "-f2", g2, "-o", os.path.join(workdir, "base_amd.sam")])
        out["cpu_baseline"] = {"value": 2 * pairs / map_s, "unit": "reads/s", "cores": cores, "kind": "reference",
                               "sample": "the first %d reads of the timed FASTQ files through the unmodified reference binary (oracle/_ref/kart, compiled from the reference "
                                         "sources) at -t %d = the host CPU quota (%d logical CPUs visible): %.1f s whole process, %.1f s of it index load (measured on a "
                                         "2-pair input), its own report: %s s; this pipeline on the same sample: %.2f s" % (2 * pairs, cores, os.cpu_count() or 1, dt, load_s, secs, st.map_seconds),
                               "whole_process_seconds": round(dt, 2), "index_load_seconds": round(load_s, 2), "this_pipeline_same_sample_reads_per_s": round(2 * pairs / st.map_seconds)}
        for f in (g1, g2, sam, os.path.join(workdir, "base_amd.sam"), os.path.join(workdir, "probe.sam"), os.path.join(workdir, "probe_1.fq"), os.path.join(workdir, "probe_2.fq")):
            try:
                os.remove(f)
            except OSError:
                pass
    for f in (tiny1, tiny2, os.path.join(workdir, "tiny.sam")):
        try:
            os.remove(f)
        except OSError:
            pass
    return out


def seeding_leg(args, api, prefix, codes, dev, n_reads_leg, oracle_sample):
    """The GPU seeding step alone -- kg_seed_batch_device = pack + search (FM-index) + scan + locate + sort -- on HBM-resident
    reads: kernel times from HIP events on the launch stream, the bytes the implemented search fetches (kg_workspace_traffic),
    the reference-algorithm counters of SURVEY 8d, and (oracle_sample > 0) GPU seeds == CPU oracle on a random sample."""
    n_reads = n_reads_leg & ~1
    n_bases = n_reads * READ_LEN
    large = args.genome_len >= 300_000_000
    ix = api.Index(prefix, dev.index or 0, {"auto": api.KG_SA_AUTO, "full": api.KG_SA_FULL, "sampled": api.KG_SA_SAMPLED, "compact": api.KG_SA_FULL40, "wide": api.KG_SA_FULL40_WIDE, "dense4": api.KG_SA_DENSE4, "dense8": api.KG_SA_DENSE8}[args.sa])
    batches = [gen_reads_device(codes, n_reads // 2, seed=1000 + b, err=0.01, dev=dev) for b in range(2)]
    release_haplotypes()
    seed_cap = (12 if large else 6) * n_reads + 1024
    d_seed_off = torch.empty(n_reads + 1, dtype=torch.int64, device=dev)
    d_seeds = torch.empty(seed_cap * 16, dtype=torch.uint8, device=dev)
    ws = api.Workspace(ix, n_reads, n_bases)
    ws.set_profiling(True)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(b):
        enc, off = batches[b % 2]
        ws.seed_batch_device(enc.data_ptr(), off.data_ptr(), n_reads, n_bases, d_seed_off.data_ptr(), d_seeds.data_ptr(), seed_cap, api.KG_MODE_FAST, stream=stream)

    step(0)
    torch.cuda.synchronize(dev)
    assert ws.overflow() == 0, "seed buffer too small"
    out = {}
    if oracle_sample:
        from oracle import oracle as O
        k = min(oracle_sample, n_reads)
        rng = np.random.default_rng(12345)
        pick = np.sort(rng.choice(n_reads, size=k, replace=False))
        so_all = d_seed_off.cpu().numpy()
        enc_h = batches[0][0].view(n_reads, READ_LEN)[torch.from_numpy(pick).to(dev)].cpu().numpy().reshape(-1)
        off_h = np.arange(k + 1, dtype=np.int64) * READ_LEN
        orc = O.Oracle(prefix)
        so_o, s_o = orc.seed_batch(enc_h, off_h, 0, threads=effective_cores())
        orc.close()
        seeds_all = d_seeds[: int(so_all[n_reads]) * 16].cpu().numpy().view(api.SEED_DT)
        cnt_g = so_all[pick + 1] - so_all[pick]
        ok = bool((cnt_g == np.diff(so_o)).all())
        if ok:
            idx = np.concatenate([np.arange(so_all[r], so_all[r + 1]) for r in pick]) if k else np.zeros(0, np.int64)
            ok = bool((seeds_all[idx] == s_o.astype(api.SEED_DT)).all())
        del seeds_all
        out["oracle_sample"] = {"reads": int(k), "identical": ok, "what": "a random sample of the seeding leg's batch: GPU seeds (order included) == the CPU oracle's"}
        assert ok, "GPU seeds differ from the oracle"
    for w in range(1):
        step(w)
    torch.cuda.synchronize(dev)
    kernel_ms = []
    t0 = time.perf_counter()
    for s_ in range(args.seed_steps):
        step(s_)
        kernel_ms.append(ws.kernel_ms())
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    cnt = ws.counters()
    tr = ws.traffic()
    kms = np.array(kernel_ms)
    search_ms = float(kms[:, 0].mean())
    c = cnt.as_dict()
    useful = tr.useful_bytes(n_reads)
    min_lines = tr.min_lines(n_reads)
    achieved = useful / (search_ms * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(n_reads, args)
    ref_alg = 64 * (c["lf1"] + 2 * c["lf2"]) + c["bases"]
    out.update({
        "value": n_reads * args.seed_steps / elapsed, "unit": "reads/s", "reads_per_launch": n_reads, "launches": args.seed_steps,
        "what": "kg_seed_batch_device on HBM-resident reads: pack + search + scan + locate + sort (last round's `value`)",
        "sa_mode": SA_NAMES.get(int(ix.info.sa_mode), args.sa), "index_bytes": int(ix.info.device_bytes),
        "kernels_ms": {"search": search_ms, "scan": float(kms[:, 1].mean()), "locate": float(kms[:, 2].mean()), "sort": float(kms[:, 3].mean())},
        "fetched_per_read": {k2: v / n_reads for k2, v in tr.as_dict().items() if k2 != "sa_entry_bytes"},
        "reference_algorithm_per_read": {k2: v / n_reads for k2, v in c.items()},
        "reference_algorithm_bytes_per_read": (ref_alg + 64 * c["inv"] + 8 * c["sa"] + 16 * c["seeds"]) / n_reads,
        "roofline": {"bound": "hbm", "kernel": "search_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": useful, "algorithmic_bytes_per_read": useful / n_reads, "avg_launch_ms": search_ms,
                     "launch": "%d reads of 150 bp" % n_reads,
                     "min_128B_lines_per_read": min_lines / n_reads, "min_line_bytes_per_launch": min_lines * 128,
                     "line_bound_frac": min_lines * 128 / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic_GBps": (traffic / (search_ms * 1e-3) / 1e9) if traffic else None,
                     "traffic_over_algorithmic": (traffic / useful) if traffic else None,
                     "note": "achieved = bytes the IMPLEMENTED search needs (kg_workspace_traffic: 8 B per q-mer table entry, 2 x 16 B rank segments per single "
                             "step, header + segment (16 B each) per rank of a double / triple step on the pair / triple planes, one SA entry, 48 B per text-comparison round, packed read "
                             "words, hit records) / HIP-event time of search_kernel.  reference_algorithm_per_read: lf1 + lf2 is exact, the split only with single steps.  Every gather "
                             "costs a whole 128-byte line of HBM traffic: min_line_bytes is the line traffic this layout cannot avoid, traffic the measured one (PMC)."}})
    if not args.no_cpu_baseline and args.leg == "all":
        try:
            out["nw_kernels"] = nw_leg(ix, dev)
        except Exception as exc:      # a side measurement must never cost the line
            out["nw_kernels"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:120])}
    ws.close()
    ix.close()
    return out


def gz_leg(sess, workdir, codes, dev, threads, cores, budget):
    """the same reads as plain files, as ordinary gzip files through the several-thread reader, and through one gzread() stream per file"""
    import hashlib
    from benchkit.gz import gzip_one_member
    n_gz = int(max(100_000, min(int(os.environ.get("KART_BENCH_GZ_PAIRS", "20000000")), budget // (4 * REC_BYTES + 2 * 500))))
    f1, f2 = os.path.join(workdir, "gz_1.fq"), os.path.join(workdir, "gz_2.fq")
    write_fastq_pairs(codes, n_gz, 43, f1, f2, dev, err=0.01)
    t0 = time.perf_counter()
    sizes = [gzip_one_member(f, f + ".gz", 6, max(1, cores)) for f in (f1, f2)]
    t_pack = time.perf_counter() - t0
    sam = os.path.join(workdir, "gz.sam")

    def one(files, env):
        for k_, v_ in env.items():
            os.environ[k_] = v_
        try:
            if os.path.exists(sam):
                os.remove(sam)
            st_ = sess.map(["-silent", "-f", files[0], "-f2", files[1], "-o", sam])
        finally:
            for k_ in env:
                del os.environ[k_]
        h = hashlib.sha256()
        with open(sam, "rb") as fh:
            for blk in iter(lambda: fh.read(64 << 20), b""):
                h.update(blk)
        return (st_.total_reads - st_.unmapped) / st_.map_seconds, round(st_.map_seconds, 3), h.hexdigest()

    plain = one((f1, f2), {})
    gz = (f1 + ".gz", f2 + ".gz")
    one(gz, {})                                              # (warm-up: the general path's buffers)
    par = one(gz, {})
    ser = one(gz, {"KART_AMD_NO_PGZ": "1", "KART_AMD_NO_GZ_STREAM": "1"})
    host = one(gz, {"KART_AMD_NO_GZ_STREAM": "1"})
    result = {"workload": "%d x 150 bp paired-end reads (wgsim model -e 0.01) as two ordinary single-member gzip files (level 6, %.2f GB of text in %.2f GB; "
                                   "written by benchkit/gz.py in %.1f s), hg38-sized index" % (2 * n_gz, sum(x[0] for x in sizes) / 1e9, sum(x[1] for x in sizes) / 1e9, t_pack),
                       "value": par[0], "unit": "mapped reads/s", "map_seconds": par[1],
                       "one_gzread_stream_per_file": {"value": ser[0], "map_seconds": ser[1], "what": "KART_AMD_NO_PGZ=1 KART_AMD_NO_GZ_STREAM=1: the reference's way (gzgets(), src/GetData.cpp:145-219) and this repository's until round 5"},
                       "several_threads_into_the_hosts_gz_reader": {"value": host[0], "map_seconds": host[1], "what": "KART_AMD_NO_GZ_STREAM=1: inflated by several threads, then line index and views on the host"},
                       "plain_files": {"value": plain[0], "map_seconds": plain[1]},
                       "same_sam_bytes_all_four": bool(plain[2] == par[2] == ser[2] == host[2]), "threads": threads,
                       "what": "host/detail/pgzip.inc: block starts found by search, the unknown 32 KB in front of each chunk carried as symbols and resolved in order, "
                               "CRC-32 of the pieces combined and held against the trailer; a thread per file writes the text into a growing block that the device's "
                               "FASTQ-in / SAM-out stream reads like a mapped plain file (GzProducer, host/detail/batch_reader.inc)"}
    for f in (f1, f2, gz[0], gz[1], sam):
        if os.path.exists(f):
            os.remove(f)
    return result


def other_configs(args, sess, prefix, workdir, codes, dev, threads, cores, mem):
    """configs[4] (-m, 2.1 % error) and configs[3] (-pacbio, 7 kb reads at 15 % error) through the same session, one run each, with
    SAM identity against the reference's -t 1 on a prefix (for -m: up to the FLAGs the reference never assigns, App. B-12)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "kart")
    out = {}
    UNSET = 1 << 20

    def ref_run(flags, files, sam, t):
        t0 = time.perf_counter()
        a = [ref, "-silent", "-i", prefix, "-f", files[0]] + (["-f2", files[1]] if len(files) > 1 else []) + flags + ["-o", sam, "-t", str(t)]
        r = subprocess.run(a, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return r.returncode, time.perf_counter() - t0

    # ---- configs[4]: -m ------------------------------------------------------------------------------------------------
    budget = (mem.get("usable") or (64 << 30)) * MEM_SHARE          # each leg's files (input + ONE output at a time) stay within the same share of host memory as the steps'
    n_mh = int(max(100_000, min(50_000_000, budget // (2 * REC_BYTES + 2 * 500))))          # (configs[4]'s literal size: 100 M reads, when the host holds the files)
    f1, f2 = os.path.join(workdir, "cfg4_1.fq"), os.path.join(workdir, "cfg4_2.fq")
    mh_stats = write_fastq_pairs(codes, n_mh, 41, f1, f2, dev, err=0.02)
    sam = os.path.join(workdir, "cfg4.sam")
    sess.map(["-silent", "-f", f1, "-f2", f2, "-m", "-o", sam])          # (warm-up: the stream's buffers grow for -m's extra records)
    os.remove(sam)
    st = sess.map(["-silent", "-f", f1, "-f2", f2, "-m", "-o", sam])
    c4 = {"workload": "configs[4]: %d x 150 bp paired-end reads, wgsim model -e 0.02 -r 0.001 (haplotype indels incl.), -m (multi-hit output), hg38-sized index" % (2 * n_mh),
          "pairs_with_indel_share": mh_stats["pairs_with_indel"] / max(1, mh_stats["pairs"]),
          "value": (st.total_reads - st.unmapped) / st.map_seconds, "unit": "mapped reads/s", "map_seconds": round(st.map_seconds, 3),
          "reads_through_the_device_stream": int(st.stream_reads), "sam_bytes": os.path.getsize(sam),
          "device_ms": {nm: float(st.stage_ms[i]) for i, nm in enumerate(("parse", "seed", "chain", "align", "format", "copy_out"))},
          "kernels_ms": {nm: round(float(st.kernel_ms[i]), 2) for i, nm in enumerate(KERNEL_SLOTS) if float(st.kernel_ms[i]) > 0},
          "rescue_windows": float(st.aln_counts[4]), "pairs_decided_by_aln_trivial": float(st.aln_counts[6])}
    os.remove(sam)
    if os.path.exists(ref) and not args.no_parity:
        # 1 M reads of the leg's files against the reference's -t 1: slices of whole chunks, one reference process each (a slice starts
        # its own EstDistance history, in the reference and here alike), this pipeline on the same slices one after the other
        n_sl = max(1, min(8, cores // 2))
        per = min(int(os.environ.get("KART_BENCH_MH_IDENT_PAIRS", "500000")), n_mh) // n_sl // 2000 * 2000
        jobs = []
        for i in range(n_sl):
            p1, p2 = os.path.join(workdir, "cfg4_p%d_1.fq" % i), os.path.join(workdir, "cfg4_p%d_2.fq" % i)
            copy_records(f1, p1, per, skip_records=i * per); copy_records(f2, p2, per, skip_records=i * per)
            sr = os.path.join(workdir, "cfg4_p%d_ref.sam" % i)
            jobs.append((p1, p2, sr, subprocess.Popen([ref, "-silent", "-i", prefix, "-f", p1, "-f2", p2, "-m", "-o", sr, "-t", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        ok, masked, t0 = True, 0, time.perf_counter()
        os.environ["KART_AMD_UNSET_FLAG"] = str(UNSET)
        try:
            for i, (p1, p2, sr, proc) in enumerate(jobs):
                sa = os.path.join(workdir, "cfg4_p%d_amd.sam" % i)
                sess.map(["-silent", "-f", p1, "-f2", p2, "-m", "-o", sa])
        finally:
            del os.environ["KART_AMD_UNSET_FLAG"]
        for i, (p1, p2, sr, proc) in enumerate(jobs):
            sa = os.path.join(workdir, "cfg4_p%d_amd.sam" % i)
            rc = proc.wait()
            la, lb = open(sr, "rb").read().split(b"\n"), open(sa, "rb").read().split(b"\n")
            ok = ok and rc == 0 and len(la) == len(lb)
            for x, y in zip(la, lb) if ok else ():
                if x == y:
                    continue
                fx, fy = x.split(b"\t"), y.split(b"\t")
                if len(fy) < 2 or int(fy[1]) != UNSET or fx[:1] + fx[2:] != fy[:1] + fy[2:]:
                    ok = False
                    break
                masked += 1
            for f in (p1, p2, sa, sr):
                os.remove(f)
        c4["sam_vs_reference_t1"] = {"reads": 2 * per * n_sl, "slices": n_sl, "identical_up_to_never_assigned_flags": bool(ok), "records_with_never_assigned_flag": masked,
                                     "seconds": round(time.perf_counter() - t0, 1)}
    out["configs[4]"] = c4
    for f in (f1, f2):
        os.remove(f)

    if not os.environ.get("KART_BENCH_NO_GZ_LEG"):
        out["gz_input"] = gz_leg(sess, workdir, codes, dev, threads, cores, budget)

    # ---- configs[3]: -pacbio -------------------------------------------------------------------------------------------
    read_len = 7000
    n_long = int(max(10_000, min(2_000_000, budget // (2 * read_len + 16 + 2 * read_len + 1000))))        # (configs[3]'s literal size: 2 M reads = 28 GB of FASTQ in, ~30 GB of SAM out)
    fq = os.path.join(workdir, "cfg3_long.fq")
    long_stats = write_long_reads(codes, n_long, read_len, 31, fq, dev)
    release_haplotypes()
    sam = os.path.join(workdir, "cfg3.sam")
    sess.map(["-silent", "-f", fq, "-pacbio", "-o", sam])               # (warm-up: the two long-read workspaces, their scratch and the page-locked result arrays grow to the batch size)
    os.remove(sam)
    st = sess.map(["-silent", "-f", fq, "-pacbio", "-o", sam])
    c3 = {"workload": "configs[3]: %d x %d bp single-end reads, wgsim model -e 0.15 -r 0.001 (haplotype indels incl.), -pacbio, hg38-sized index" % (n_long, read_len),
          "reads_with_indel_share": long_stats["reads_with_indel"] / max(1, long_stats["reads"]),
          "value": (st.total_reads - st.unmapped) / st.map_seconds, "unit": "mapped reads/s", "map_seconds": round(st.map_seconds, 3), "sam_bytes": os.path.getsize(sam)}
    os.remove(sam)
    if os.path.exists(ref) and not args.no_parity:
        # the reference at -t 1 takes ~10 ms per such read: the sample is cut into slices, one reference process each (they print
        # their chunks in completion order with more threads), the slices' records in order are what one process prints for the whole sample
        k, n_sl = int(os.environ.get("KART_BENCH_LONG_IDENT", "20000")), max(1, min(8, cores // 2))
        k = min(k, n_long) // n_sl * n_sl
        pq = os.path.join(workdir, "cfg3_p.fq")
        copy_records(fq, pq, k)
        sls = [os.path.join(workdir, "cfg3_p_%d.fq" % i) for i in range(n_sl)]
        split_records(pq, sls, k // n_sl)
        sa = os.path.join(workdir, "cfg3_p_amd.sam")
        srs = [os.path.join(workdir, "cfg3_p_ref_%d.sam" % i) for i in range(n_sl)]
        t0 = time.perf_counter()
        procs = [subprocess.Popen([ref, "-silent", "-i", prefix, "-f", sl, "-pacbio", "-o", sr, "-t", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for sl, sr in zip(sls, srs)]
        sess.map(["-silent", "-f", pq, "-pacbio", "-o", sa])
        rcs = [p_.wait() for p_ in procs]
        dt = time.perf_counter() - t0
        parts = [open(sr, "rb").read() for sr in srs]
        header = b"".join(l for l in parts[0].splitlines(True) if l.startswith(b"@"))
        want = header + b"".join(b"".join(l for l in p_.splitlines(True) if not l.startswith(b"@")) for p_ in parts)
        c3["sam_vs_reference_t1"] = {"reads": k, "identical": bool(not any(rcs) and open(sa, "rb").read() == want), "reference_processes": n_sl,
                                     "reference_t1_seconds_whole_processes": round(dt, 1)}
        for f in [pq, sa] + sls + srs:
            os.remove(f)
    out["configs[3]"] = c3
    os.remove(fq)
    return out


def config1(args, workdir, dev, threads, budget, ref):
    """BASELINE.json configs[1]: E. coli-sized genome (4 639 675 bp behind the 2 kb decoy), 10 M x 150 bp pairs at 1 % error -- a second
    session beside the large one (the index takes ~0.3 GB of HBM and lives in the L2 / Infinity Cache: SURVEY 8(d) asks for L2-hit
    counters here instead of an HBM fraction; they come from the committed PMC pass of `bench.py --genome-len 4639675`)."""
    from kart_amd import api
    a1 = argparse.Namespace(genome_len=GENOME_LEN, bucketed=None, repeat_frac=0.45)
    prefix1, codes1, _ = prepare_index(a1, dev, 0, workdir, lambda: None)
    n_pairs = int(max(100_000, min(5_000_000, budget // (2 * REC_BYTES + 2 * 450))))          # (10 M reads: "10M x 150 bp PE")
    f1, f2 = os.path.join(workdir, "cfg1_1.fq"), os.path.join(workdir, "cfg1_2.fq")
    st1 = write_fastq_pairs(codes1, n_pairs, 7, f1, f2, dev, err=0.01)
    release_haplotypes()
    del codes1
    sess1 = api.HostSession(prefix1, 0, threads)
    sam = os.path.join(workdir, "cfg1.sam")
    try:
        sess1.map(["-silent", "-f", f1, "-f2", f2, "-o", sam])          # (warm-up: the stream's buffers)
        os.remove(sam)
        st = sess1.map(["-silent", "-f", f1, "-f2", f2, "-o", sam])
        c1 = {"workload": "configs[1]: E. coli-sized synthetic genome (4 639 675 bp + 2 kb decoy), %d x 150 bp paired-end reads, wgsim model -e 0.01 -r 0.001 (haplotype indels incl.)" % (2 * n_pairs),
              "pairs_with_indel_share": st1["pairs_with_indel"] / max(1, st1["pairs"]),
              "value": (st.total_reads - st.unmapped) / st.map_seconds, "unit": "mapped reads/s", "map_seconds": round(st.map_seconds, 3),
              "reads_through_the_device_stream": int(st.stream_reads), "sam_bytes": os.path.getsize(sam),
              "device_ms": {nm: float(st.stage_ms[i]) for i, nm in enumerate(("parse", "seed", "chain", "align", "format", "copy_out"))}}
        if st.search_kernel_launches > 0 and st.search_kernel_ms > 0:
            gbs = float(st.search_useful_bytes) / (float(st.search_kernel_ms) * 1e-3) / 1e9
            c1["search_kernel"] = {"ms": float(st.search_kernel_ms), "launches": int(st.search_kernel_launches), "useful_GBps": gbs,
                                   "note": "the 9 MB rank structure and the 134 MB q-mer table are L2 / Infinity-Cache resident: bytes per second against the HBM peak say nothing "
                                           "here; the L2 hit rate below is the figure SURVEY 8(d) asks for"}
        os.remove(sam)
        pdir = os.path.join(ROOT, "profiles")
        for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
            if f.endswith("_ecoli_pmc_summary.json"):
                try:
                    l2 = json.load(open(os.path.join(pdir, f))).get("_l2")
                    if l2:
                        c1["l2"] = dict(l2, source="profiles/" + f)
                except Exception:
                    pass
        if os.path.exists(ref) and not args.no_parity:
            k = min(n_pairs, 100_000)
            p1, p2 = os.path.join(workdir, "cfg1_p1.fq"), os.path.join(workdir, "cfg1_p2.fq")
            for src, dst in ((f1, p1), (f2, p2)):
                copy_records(src, dst, k)
            sa, sr = os.path.join(workdir, "cfg1_p_amd.sam"), os.path.join(workdir, "cfg1_p_ref.sam")
            sess1.map(["-silent", "-f", p1, "-f2", p2, "-o", sa])
            t0 = time.perf_counter()
            rc = subprocess.run([ref, "-silent", "-i", prefix1, "-f", p1, "-f2", p2, "-o", sr, "-t", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
            c1["sam_vs_reference_t1"] = {"reads": 2 * k, "identical": bool(rc == 0 and open(sa, "rb").read() == open(sr, "rb").read()),
                                         "reference_t1_seconds_whole_process": round(time.perf_counter() - t0, 1)}
            for f in (p1, p2, sa, sr):
                os.remove(f)
    finally:
        sess1.close()
        for f in (f1, f2):
            try:
                os.remove(f)
            except OSError:
                pass
    return c1


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


def stream_reads_setting():
    """reads per stream batch as host/mapper.hpp chooses them (KART_AMD_STREAM_READS, default 1.12 M)"""
    v = int(os.environ.get("KART_AMD_STREAM_READS", "0") or 0)
    return v if v >= 4000 else 1120000


def seed_group_setting():
    """(lanes per seeding group, stream lanes) as host/detail/pipeline.inc chooses them: KART_AMD_SEED_GROUP (default 4: ONE search launch over four
    lanes' batches), KART_AMD_STREAM_LANES (default two groups, or four independent lanes)"""
    g = int(os.environ.get("KART_AMD_SEED_GROUP", "4"))
    g = min(g, 8) if g > 1 else 0
    lanes = int(os.environ.get("KART_AMD_STREAM_LANES", "0"))
    lanes = min(lanes, 16) if lanes > 0 else (2 * g if g else 4)
    if g and lanes % g:
        lanes = (lanes + g - 1) // g * g
    return g, lanes


def resolved_sa(args, dev=None):
    """the index mode `--sa auto` becomes (kg_index_load's rule: full below 2^32 text symbols -- both strands --; above, wide -- 5-byte suffix
    array, full q-mer table, triple planes -- where the device keeps 64 GiB free behind its ~24.5 bytes per symbol, else compact).  Asked with
    `dev` right before the index is loaded (the same free-memory figure the library sees), remembered for the callers without one."""
    if args.sa != "auto":
        return args.sa
    n_sym = 2 * int(args.genome_len or HG38_LEN)
    if n_sym < 0xFFFFFFFF:
        return "full"
    if dev is not None:
        free_b, _ = torch.cuda.mem_get_info(dev)
        args.sa_resolved = "wide" if (free_b > 24.5 * n_sym + (64 << 30) and not os.environ.get("KG_AUTO_COMPACT")) else "compact"
    return getattr(args, "sa_resolved", "compact")


SA_NAMES = {0: "sampled", 1: "full", 5: "compact", 6: "wide", 4: "dense4", 8: "dense8"}
KERNEL_SLOTS = ("chain", "aln_pair", "aln_rescue", "aln_plan_fast", "aln_plan", "aln_partition", "nw", "aln_finish", "aln_final", "sam_size", "sam_format",
                "fq_parse", "fq_materialise", "locate_sort", "aln_trivial")
VALU_INT32_PEAK_TOPS = 78.0       # int32 VALU lane-operations per second of the device, in 1e12 (VERDICT r4 #6's figure; ~12 of them per DP cell)
NW_OPS_PER_CELL = 12.0


def survey_and_traffic_fractions(roof, ref_bytes_per_read, reads_per_launch):
    """SURVEY.md 8(d) defines the algorithmic bytes of seeding on the REFERENCE's algorithm (64-byte Occ blocks per LF step, 8 B per SA
    entry, 16 B per seed).  The implemented search skips ~90 % of those steps (q-mer table, pair / triple planes, text comparison), so that
    figure over the kernel's time exceeds the HBM peak: it is an algorithmic speed-up, not a fraction -- printed beside `frac`, which
    prices the bytes the implemented search needs, and `traffic_frac`, the counter traffic over the same time."""
    if not roof or not roof.get("avg_launch_ms"):
        return
    sec = roof["avg_launch_ms"] * 1e-3
    if ref_bytes_per_read and reads_per_launch:
        gbs = ref_bytes_per_read * reads_per_launch / sec / 1e9
        roof["survey_8d"] = {"bytes_per_read": ref_bytes_per_read, "bytes_per_launch": ref_bytes_per_read * reads_per_launch, "GBps": gbs, "over_peak": gbs / HBM_PEAK_GBS,
                             "label": ("algorithmic speed-up over the reference's block-per-step search, not a roofline fraction" if gbs > HBM_PEAK_GBS
                                       else "SURVEY 8(d) bytes / launch time / peak")}
        roof["frac_survey_8d"] = gbs / HBM_PEAK_GBS
    if roof.get("traffic"):
        roof["traffic_frac"] = roof["traffic"] / sec / 1e9 / HBM_PEAK_GBS
        roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"] if roof.get("algorithmic_bytes_per_launch") else None


def kernel_traffic(args):
    """per-kernel HBM bytes per step from the committed PMC passes of this very command (profiles/*_pmc_summary.json, `_kernel_traffic`:
    {kernel: bytes per step}, made by tools/profile_to_profiles.py from rocprofv3 --pmc passes with the guide's gfx950 corrections);
    only when the profiled run had this run's shape"""
    best = ({}, None)
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not f.endswith("_pmc_summary.json"):
            continue
        try:
            t = json.load(open(os.path.join(pdir, f))).get("_kernel_traffic")
        except Exception:
            t = None
        if not t or t.get("pairs_per_step") != getattr(args, "pairs", None) or t.get("genome_len", GENOME_LEN) != args.genome_len:
            continue
        if t.get("seed_group") != seed_group_setting()[0] or t.get("stream_lanes") != seed_group_setting()[1] or t.get("stream_reads") != stream_reads_setting():
            continue
        if t.get("sa_mode", "full") != resolved_sa(args):
            continue
        best = (t.get("bytes_per_step", {}), "profiles/" + f)
    return best


def kernel_entries(stats, steps, n_reads, search_ms_per_step, args):
    """achieved / peak / frac for every kernel of the timed region (rank 0): ms = HIP events around each launch of the timed steps."""
    ms = [sum(float(st.kernel_ms[i]) for st in stats) / steps for i in range(len(KERNEL_SLOTS))]
    launches = [sum(int(st.kernel_launches[i]) for st in stats) / steps for i in range(len(KERNEL_SLOTS))]
    cnt = [sum(float(st.aln_counts[i]) for st in stats) / steps for i in range(8)]
    C = sum(float(st.candidates) for st in stats) / steps
    S = sum(float(st.candidate_seeds) for st in stats) / steps
    t_in = sum(float(st.text_in_bytes) for st in stats) / steps
    t_out = sum(float(st.text_out_bytes) for st in stats) / steps
    R = float(n_reads)
    spills, jobs, op_bytes, part_tasks, resc_tasks, trivial_pairs, slow = cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[6], cnt[7]
    # aln_trivial_kernel decides the trivial pairs start to finish; the kernels behind it see the candidates / reads of the other pairs only
    share = 1.0 - (2 * trivial_pairs / R if R else 0.0)
    C_all, S_all, R_all = C, S, R
    per_cand = 32 + 78 + READ_LEN + READ_LEN / 4          # candidate, its report, the read's characters, the text under it
    # bytes every kernel has to move (inputs read once + outputs written once, HBM layout of DESIGN.md section 3)
    alg = {
        "chain": 32 * S + 32 * C + 24 * R,                                           # seeds in (16 B), candidate seeds out (16 B), candidates (32 B), per-read counts / offsets
        "aln_trivial": 24 * R_all + 32 * C_all + 16 * S_all + (1 - share) * (2 * R_all + 4 * C_all) + (2 * trivial_pairs) * (112 + 8 + 8),   # every pair's offsets, candidates and seeds in; the lists of what it leaves; per decided read: the record, the gap characters and text words it compares
        "aln_pair": share * (32 * C + 12 * C + 16 * R),                              # candidates in, score / mate / read per candidate out, per-read flags
        "aln_rescue": resc_tasks * (40 + 1650 / 4 + READ_LEN + 12 * 16),             # task, window text (2 bit/base), the mate's characters, the rescued seeds
        "aln_plan_fast": share * (C * per_cand + 16 * S + 4 * C),                              # every candidate + its seeds; the list of the ones it leaves
        "aln_plan": slow * (per_cand + 536) + 16 * S * (slow / C if C else 0) + 24 * jobs,   # the candidates left to it, their spill slots (536 B), job descriptors
        "aln_partition": part_tasks * (32 + 2 * 255 / 2 + 64),                       # task, both fragments (characters / 2-bit text), plan + pieces
        "nw": None,
        "aln_finish": spills * (536 + 78) + 2 * op_bytes + spills * READ_LEN,        # spill slot in, report out, op strings, the read's characters
        "aln_final": share * (112 * R + 78 * C + 12 * C),                                      # record out, reports + scores / mates in
        "sam_size": (112 + 24 + 4) * R,                                              # record + record table in, length out
        "sam_format": t_in * 0.75 + t_out + (112 + 24 + 8) * R,                      # name + bases + qualities of the FASTQ text (the '+' line and newlines are not read), SAM text out, records
        "fq_parse": 2 * t_in + 16 * R + 24 * R,                                      # the text twice (line count, line index), line ends (4 x 4 B), record table
        "fq_materialise": (READ_LEN + READ_LEN + 8) * R,                             # bases in, characters out, offsets
    }
    traffic, traffic_src = kernel_traffic(args)
    total = sum(ms) + search_ms_per_step
    out = {"what": "rank 0, per step: HIP events around each launch on the lanes' streams (other lanes' kernels share the device, so a launch's time includes what "
                   "they took from it); achieved = algorithmic bytes / that time; peak = %.0f GB/s (HBM); traffic = counter bytes of the same command "
                   "(profiles/, null when no pass of this run's shape is committed)" % HBM_PEAK_GBS,
           "timed_kernel_ms_per_step": total, "counts_per_step": {"candidates": C, "candidate_seeds": S, "parked_candidates": spills, "nw_jobs": jobs, "nw_op_bytes": op_bytes,
                                                                 "partition_tasks": part_tasks, "rescue_windows": resc_tasks, "candidates_left_to_aln_plan": slow,
                                                                 "pairs_decided_by_aln_trivial": trivial_pairs, "share_of_pairs_left_to_the_general_kernels": share},
           "traffic_source": traffic_src}
    for i, name in enumerate(KERNEL_SLOTS):
        if ms[i] <= 0:
            continue
        e = {"ms_per_step": ms[i], "launches_per_step": launches[i], "share_of_timed_kernels": ms[i] / total if total else None}
        if name == "nw":
            e.update({"bound": "valu", "note": "integer max-plus DP: priced in cells per second by the nw_kernels leg of this line (GCUPS per size class against %.0f T int32 "
                                               "VALU operations/s at ~%.0f per cell); in the timed region: %.0f jobs, %.0f op bytes per step" % (VALU_INT32_PEAK_TOPS, NW_OPS_PER_CELL, jobs, op_bytes)})
        elif alg.get(name):
            gbs = alg[name] / (ms[i] * 1e-3) / 1e9
            e.update({"bound": "hbm", "algorithmic_bytes_per_step": alg[name], "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
        t = traffic.get(name)
        e["traffic"] = t
        if t:
            e["traffic_frac"] = t / (ms[i] * 1e-3) / 1e9 / HBM_PEAK_GBS
        out[name] = e
    return out


_HOSTOUT_WORKER = r"""
import mmap, os, sys, time
import numpy as np
path, total, blk, k, n, cpus = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
if cpus:
    try:
        os.sched_setaffinity(0, {int(c) for c in cpus.split(",")})
    except OSError:
        pass
fd = os.open(path, os.O_RDWR)
mm = mmap.mmap(fd, total)
dst = np.frombuffer(mm, dtype=np.uint8)
src = np.full(blk, 65, dtype=np.uint8)
sys.stdout.write("ready\n"); sys.stdout.flush()
sys.stdin.readline()                      # all workers start together
t0 = time.perf_counter()
for at in range(k * blk, total, n * blk):
    np.copyto(dst[at:at + blk], src)
sys.stdout.write("%.6f\n" % (time.perf_counter() - t0)); sys.stdout.flush()
"""


def _one_l3_domain():
    """the CPUs that share a last-level cache with the CPU this process runs on (the pipeline keeps its writers on one such domain)"""
    try:
        cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else 0
        txt = open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % cpu).read().strip()
        cpus = []
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0)
        return [c for c in cpus if c in allowed]
    except Exception:
        return []


def host_output_entry(workdir, sam_bytes, step_seconds, threads):
    """What bounds a step is the host side of the output: fresh page-cache pages of the SAM file.  peak = this box's
    rate for exactly that, measured here: N processes (no GIL between them), kept on the CPUs of ONE last-level cache as the pipeline's
    writers are, copying 16 MB blocks into shared mappings of one fresh file in the same directory -- every 4 KB page a write fault that
    allocates it; N = 8, one more than the pipeline's mapping writers."""
    n_w = 8
    blk = 16 << 20
    # (the probe's file lives where the outputs live -- tmpfs pages are RAM: 8 GB where the host has plenty, else a twentieth of what it can still hold, at least 1 GB)
    usable = host_memory().get("usable")
    total = (8 << 30) if usable is None or usable >= (160 << 30) else max(1 << 30, int(usable * 0.05) // blk * blk)
    path = os.path.join(workdir, "kart_bench_hostout_%d.bin" % os.getpid())
    cpus = _one_l3_domain()
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o600)
    procs = []
    try:
        os.ftruncate(fd, total)
        for k in range(n_w):
            procs.append(subprocess.Popen([sys.executable, "-c", _HOSTOUT_WORKER, path, str(total), str(blk), str(k), str(n_w), ",".join(map(str, cpus))],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL))
        for p_ in procs:
            assert p_.stdout.readline().strip() == b"ready"
        t0 = time.perf_counter()
        for p_ in procs:
            p_.stdin.write(b"go\n"); p_.stdin.flush()
        per = [float(p_.stdout.readline()) for p_ in procs]
        dt = time.perf_counter() - t0
        for p_ in procs:
            p_.wait(timeout=60)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
        os.close(fd)
        try:
            os.remove(path)
        except OSError:
            pass
    peak = total / dt / 1e9
    achieved = sam_bytes / step_seconds / 1e9
    return {"bound": "host page cache", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak, "bytes_per_step": sam_bytes,
            "peak_source": "%d processes on the %d CPUs of one last-level cache, 16 MB copies into mappings of one fresh %d GB file in %s, measured in this run (%.2f s; the "
                           "slowest worker %.2f s)" % (n_w, len(cpus), total >> 30, workdir, dt, max(per)),
            "note": "the step's SAM text over the step's wall time against the rate at which this box hands out fresh pages of one tmpfs file to eight writers on one cache "
                    "domain; the pipeline's seven mapping writers + one pwrite thread share their cores with nothing, but the lanes' input copies and page-locked transfers run beside them"}


def measured_traffic(n_reads, args, tag=None):
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
            if not t or t.get("tag") != tag or t.get("genome_len", GENOME_LEN) != args.genome_len:
                continue
            if tag is None and t.get("reads_per_launch") != n_reads:
                continue
            if t.get("sa_mode", "full") != resolved_sa(args):
                continue
            # the timed region: the same reads per step and the same seeding configuration as the profiled run (a launch of a run with
            # other batch sizes fetches other bytes -- round 3 quoted the 20 M-read run's figure for the 100 M-read run)
            if tag is not None and (t.get("pairs_per_step") != getattr(args, "pairs", None) or t.get("seed_group") != seed_group_setting()[0]
                                    or t.get("stream_lanes") != seed_group_setting()[1] or t.get("stream_reads") != stream_reads_setting()):
                continue
            best = (t["traffic_bytes_per_launch"], "profiles/" + f)
    return best if best else (None, None)


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

        max_len = int(max(m.max(), k.max()))          # (outside the timed calls: a numpy reduction over millions of lengths costs more than the kernels)

        def call():
            rc = ix.lib.kg_nw_batch_device(ix.h, f1.data_ptr(), d1.data_ptr(), f2.data_ptr(), d2.data_ptr(), n, max_len, ops.data_ptr(), ln.data_ptr(), stream)
            assert rc == 0, ix.lib.kg_last_error()
        call(); torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(3):
            call()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t) / 3
        gcups = float((m.astype(np.float64) * k).sum()) / dt / 1e9
        out[name] = {"pairs_per_s": round(n / dt), "GCUPS": round(gcups, 1), "bound": "valu", "peak_GCUPS": round(VALU_INT32_PEAK_TOPS * 1e3 / NW_OPS_PER_CELL, 1),
                     "frac": gcups / (VALU_INT32_PEAK_TOPS * 1e3 / NW_OPS_PER_CELL)}
        del f1, f2, ops, ln
    return out




if __name__ == "__main__":
    sys.exit(main() or 0)
