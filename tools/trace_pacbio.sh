# usage (GPU box): bash tools/trace_pacbio.sh <tag> [reads] -- kernel trace of one kart-amd -pacbio run (7 kb reads, 15 % error) on the hg38-sized index
TAG=${1:-tp}; N=${2:-50000}; R=$GRAFT_REPO_ROOT
cd $R
python3 - <<PY
import os, sys, subprocess, torch
sys.path.insert(0, "$R")
import bench
L = bench.HG38_LEN
wd = bench.pick_workdir(80 << 30)
subprocess.run([sys.executable, "bench.py", "--pairs", "1000000", "--leg", "seeding", "--seed-steps", "1"], stdout=subprocess.DEVNULL)
dev = torch.device("cuda", 0)
codes = bench.make_large_codes(L, 3, dev)
g = torch.Generator(device=dev); g.manual_seed(31)
acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
ar = torch.arange(7000, device=dev)
with open(os.path.join(wd, "pb.fq"), "wb") as fh:
    for s in range(0, $N, 20000):
        m = min(20000, $N - s)
        pos = bench.DECOY_LEN + (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (L - 7001)).long()
        r = codes[pos[:, None] + ar]
        e = torch.rand(r.shape, generator=g, device=dev) < 0.15
        r = torch.where(e, (r + torch.randint(1, 4, r.shape, generator=g, device=dev, dtype=torch.uint8)) & 3, r)
        txt = acgt[r.long()].cpu().numpy()
        q = b"5" * 7000
        for i in range(m):
            fh.write(b"@L%d\n" % (s + i) + txt[i].tobytes() + b"\n+\n" + q + b"\n")
print(wd)
PY
WD=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.pick_workdir(60<<30))")
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- $R/kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/pb.fq -pacbio -o $WD/pb.sam -t 32 > $R/gpurun_out/${TAG}_trace.log 2>&1
f=$(find $R/gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
grep -v -E "build_|expand_sa|qtab|planes2_|fillBuffer" $R/gpurun_out/${TAG}_kernel_stats.csv | head -14 | cut -d, -f1-4 | cut -c1-130
