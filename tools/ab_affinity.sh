# usage (GPU box): bash tools/ab_affinity.sh -- kart-amd FASTQ -> SAM (20 M reads, hg38-sized index) unpinned and pinned to sets of logical CPUs
cd $GRAFT_REPO_ROOT
E2E_NO_REF=1 timeout 600 python tools/e2e_large.py 3100000000 10000000 > /dev/null 2>&1
WD=$(python3 -c "import sys; sys.path.insert(0,'.'); import bench; print(bench.pick_workdir(60<<30))")
lscpu | grep -E "NUMA node|Socket|Thread|Model name" | head -12
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
run() { rm -f $WD/o.sam; KART_AMD_VERBOSE=1 $1 kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/l1.fq -f2 $WD/l2.fq -o $WD/o.sam -t 32 | grep -E "mapping seconds" | tr '\n' ' '; echo " <- $1"; }
for i in 1 2 3 4; do
  run ""
  run "taskset -c 0-63,128-191"
  run "taskset -c 0-31"
  run "taskset -c 0-31,128-159"
  run "taskset -c 64-127,192-255"
done
