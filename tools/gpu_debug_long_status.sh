cd $GRAFT_REPO_ROOT
N=${1:-100000}
RUN_CONFIGS_NO_REF=1 RUN_CONFIGS_KEEP_INPUTS=1 python3 tools/run_configs.py $N 0 > gpurun_out/dbg_first.json 2> gpurun_out/dbg_first.err
CMD=$(python3 -c "import json; d=json.load(open('gpurun_out/dbg_first.json')); print(' '.join(d['configs[3] -pacbio']['command']))")
KG_LONG_DEBUG_STATUS=1 KART_AMD_VERBOSE=1 $CMD 2>&1 | grep -E "KG_LONG_DEBUG_STATUS|long-read report" | head -40 | cut -c1-400
