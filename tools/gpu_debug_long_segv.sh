# the segmentation fault of kart-amd -pacbio with batches doubling up to 16384 chunks (profiles/r05h_ab_long_2m.log): a backtrace
cd $GRAFT_REPO_ROOT
N=${1:-600000}
RUN_CONFIGS_NO_REF=1 RUN_CONFIGS_KEEP_INPUTS=1 python3 tools/run_configs.py $N 0 > gpurun_out/dbg_first.json 2> gpurun_out/dbg_first.err
CMD=$(python3 -c "import json; d=json.load(open('gpurun_out/dbg_first.json')); print(' '.join(d['configs[3] -pacbio']['command']))")
export KART_AMD_PACBIO_CHUNKS=${C0:-2048} KART_AMD_PACBIO_MAX_CHUNKS=${C1:-16384} KART_AMD_VERBOSE=1
ulimit -c 0
/opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex run -ex "bt 30" -ex "info threads" --args $CMD 2>&1 | grep -v "^\[New Thread\|^\[Thread.*exited\|^warning" | tail -60
