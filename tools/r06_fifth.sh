#!/bin/bash
# round 6, fifth GPU call: batch-size A/B (tail latency of the per-batch kernels), timeline of kernels vs copies
mkdir -p gpurun_out
export TMPDIR=/tmp
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity"
timeout 900 python bench.py $A > gpurun_out/r06e_bench_default.json 2> gpurun_out/r06e_bench_default.err
KART_AMD_STREAM_READS=2000000 timeout 900 python bench.py $A > gpurun_out/r06e_bench_2m.json 2> gpurun_out/r06e_bench_2m.err
KART_AMD_STREAM_READS=2000000 KART_AMD_STREAM_LANES=6 KART_AMD_SEED_GROUP=3 timeout 900 python bench.py $A > gpurun_out/r06e_bench_2m_6lanes.json 2> gpurun_out/r06e_bench_2m_6lanes.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06e_trace -- python3 $GRAFT_REPO_ROOT/bench.py --pairs 50000000 --steps 2 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline > $GRAFT_REPO_ROOT/gpurun_out/r06e_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline_overlap.py gpurun_out/r06e_trace > gpurun_out/r06e_timeline.json 2> gpurun_out/r06e_timeline.err
rm -rf gpurun_out/r06e_trace
for f in default 2m 2m_6lanes; do tail -c 300 gpurun_out/r06e_bench_$f.err; done; cat gpurun_out/r06e_timeline.json | head -40
