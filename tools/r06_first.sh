#!/bin/bash
# round 6, first GPU call: the tests the round's first changes touch, then a short bench on the wgsim-model reads
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_seed_gpu.py tests/test_sam_gpu.py tests/test_ecoli_gpu.py -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r06a_tests.log
( timeout 1500 python -m pytest tests/test_hg38_gpu.py -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r06a_hg38.log
timeout 1500 python bench.py --steps 5 --warmup 1 > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
tail -c 600 gpurun_out/r06a_tests.log; tail -c 600 gpurun_out/r06a_hg38.log; tail -c 300 gpurun_out/r06a_bench.err
