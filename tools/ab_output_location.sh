cd $GRAFT_REPO_ROOT
E2E_NO_REF=1 timeout 600 python tools/e2e_large.py 3100000000 10000000 > gpurun_out/ol0.json 2>/dev/null
WD=$(python3 -c "import sys; sys.path.insert(0,'.'); import bench; print(bench.pick_workdir(60<<30))")
df -T /tmp | tail -1
for i in 1 2 3; do
  for out in $WD/o.sam /tmp/o.sam; do
    rm -f $out
    KART_AMD_VERBOSE=1 kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/l1.fq -f2 $WD/l2.fq -o $out -t 32 | grep -E "mapping seconds|cpu seconds" | tr '\n' ' '; echo " -> $out"
  done
done
rm -f /tmp/o.sam
