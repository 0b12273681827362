#!/bin/bash
# round 6, second GPU call: the two new tests by themselves (whole output), the GPU suite on the trivial-pair kernel, bench A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_seed_gpu.py -x -q -m gpu -k "exact_long" > gpurun_out/r06b_exact.log 2>&1
timeout 900 python -X faulthandler -m pytest tests/test_sam_gpu.py -x -q -m gpu -k "full_lists" > gpurun_out/r06b_lists.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06b_gpu_tests.log 2>&1
timeout 1200 python bench.py --steps 5 --warmup 1 --no-other-configs --no-seeding-leg > gpurun_out/r06b_bench_trivial.json 2> gpurun_out/r06b_bench_trivial.err
KG_ALN_NO_TRIVIAL=1 timeout 1200 python bench.py --steps 5 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity > gpurun_out/r06b_bench_notrivial.json 2> gpurun_out/r06b_bench_notrivial.err
head -c 2500 gpurun_out/r06b_exact.log; echo; head -c 2500 gpurun_out/r06b_lists.log; echo; tail -c 1500 gpurun_out/r06b_gpu_tests.log
