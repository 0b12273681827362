#!/bin/bash
# round 6, third GPU call: the whole GPU suite on everything so far (trivial-pair kernel, wide index, checksum leg), then the default bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06c_gpu_tests.log 2>&1
timeout 1500 python bench.py --steps 5 --warmup 1 > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err
tail -c 1200 gpurun_out/r06c_gpu_tests.log; tail -c 600 gpurun_out/r06c_bench.err
