#!/bin/bash
# round 6, eighth GPU call: the round's profile -- rocprofv3 kernel trace + separate PMC passes of bench.py (tools/profile_r06.sh), then the kernel / copy timeline on a smaller job
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_r06.sh r06h 50000000 > gpurun_out/r06h_profile.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06h_trace -- python3 $GRAFT_REPO_ROOT/bench.py --pairs 10000000 --steps 2 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline > $GRAFT_REPO_ROOT/gpurun_out/r06h_trace.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r06h_trace -type f | head -20 > gpurun_out/r06h_trace_files.txt
for f in $(find gpurun_out/r06h_trace -name "*.csv" | head -8); do echo "== $f"; head -2 $f | cut -c1-600; done >> gpurun_out/r06h_trace_files.txt
python tools/timeline_overlap.py gpurun_out/r06h_trace > gpurun_out/r06h_timeline.json 2> gpurun_out/r06h_timeline.err
rm -rf gpurun_out/r06h_trace
tail -5 gpurun_out/r06h_profile.log | cut -c1-300; cat gpurun_out/r06h_timeline.json | head -30
