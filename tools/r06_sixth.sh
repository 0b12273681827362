#!/bin/bash
# round 6, sixth GPU call: GPU suite on the read-per-lane trivial kernel, wave chaining of heavy reads and the lazy copies back; CHECK_ALIGN; batch-size A/B twice over; timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --maxfail=3 > gpurun_out/r06f_gpu_tests.log 2>&1
E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06f_check_align.json 2> gpurun_out/r06f_check_align.err
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity"
for round in 1 2; do
  for sr in 1120000 2000000 3000000; do
    KART_AMD_STREAM_READS=$sr timeout 900 python bench.py $A > gpurun_out/r06f_bench_${sr}_$round.json 2> gpurun_out/r06f_bench_${sr}_$round.err
  done
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06f_trace -- python3 $GRAFT_REPO_ROOT/bench.py --pairs 50000000 --steps 2 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline > $GRAFT_REPO_ROOT/gpurun_out/r06f_trace.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r06f_trace -type f | head -20 > gpurun_out/r06f_trace_files.txt
for f in $(find gpurun_out/r06f_trace -name "*.csv" | head -8); do echo "== $f"; head -2 $f | cut -c1-600; done >> gpurun_out/r06f_trace_files.txt
python tools/timeline_overlap.py gpurun_out/r06f_trace > gpurun_out/r06f_timeline.json 2> gpurun_out/r06f_timeline.err
rm -rf gpurun_out/r06f_trace
tail -c 600 gpurun_out/r06f_gpu_tests.log; tail -c 400 gpurun_out/r06f_check_align.json; cat gpurun_out/r06f_trace_files.txt | head -30
