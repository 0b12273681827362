#!/bin/bash
# round 6, tenth GPU call: the driver's command on the round's state
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06j_bench_default_steps20.log 2> gpurun_out/r06j_bench_default_steps20.err ) 2> gpurun_out/r06j_time.txt
tail -c 300 gpurun_out/r06j_bench_default_steps20.err; cat gpurun_out/r06j_time.txt
