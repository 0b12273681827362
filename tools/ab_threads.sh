cd $GRAFT_REPO_ROOT
E2E_NO_REF=1 timeout 600 python tools/e2e_large.py 3100000000 10000000 > /dev/null 2>&1
WD=$(python3 -c "import sys; sys.path.insert(0,'.'); import bench; print(bench.pick_workdir(60<<30))")
run() { rm -f $WD/o.sam; KART_AMD_VERBOSE=1 kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/l1.fq -f2 $WD/l2.fq -o $WD/o.sam -t $1 | grep -E "mapping seconds" | tr '\n' ' '; echo " <- -t $1"; }
for i in 1 2 3; do for t in 16 24 32 48; do run $t; done; done
