cd $GRAFT_REPO_ROOT; mkdir -p /tmp/fz && cd /tmp/fz
export KART_FUZZ_BIN=$GRAFT_REPO_ROOT/kart_amd/bin/kart-amd
timeout 900 python $GRAFT_REPO_ROOT/tools/fuzz_vs_reference.py ${FZ_A:-101 140} 2>&1 | tail -4
timeout 600 python $GRAFT_REPO_ROOT/tools/fuzz_speculation_vs_reference.py ${FZ_B:-11 14} 2>&1 | tail -3
timeout 900 python $GRAFT_REPO_ROOT/tools/fuzz_pacbio_vs_reference.py ${FZ_C:-0 20} 2>&1 | tail -3
mkdir -p /tmp/fz/odd && cd /tmp/fz/odd && timeout 900 python $GRAFT_REPO_ROOT/tools/probe_odd_inputs_vs_reference.py 2>&1 | grep -v " same | same$" | tail -20
mkdir -p /tmp/fz/ix && cd /tmp/fz/ix && timeout 900 python $GRAFT_REPO_ROOT/tools/probe_odd_indexes_vs_reference.py 2>&1 | tail -3
