#!/bin/bash
# round 6, fourth GPU call: the whole GPU suite, KART_AMD_CHECK_ALIGN on 4 M reads at hg38 size, the candidate histogram, bench with and without the heavy-pair path
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --maxfail=3 > gpurun_out/r06d_gpu_tests.log 2>&1
E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06d_check_align.json 2> gpurun_out/r06d_check_align.err
timeout 900 python tools/cand_histogram.py 1000000 > gpurun_out/r06d_cand_histogram.log 2>&1
timeout 1200 python bench.py --steps 5 --warmup 1 --no-other-configs --no-seeding-leg > gpurun_out/r06d_bench.json 2> gpurun_out/r06d_bench.err
KG_ALN_NO_HEAVY=1 timeout 1200 python bench.py --steps 5 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline > gpurun_out/r06d_bench_noheavy.json 2> gpurun_out/r06d_bench_noheavy.err
tail -c 800 gpurun_out/r06d_gpu_tests.log; tail -c 600 gpurun_out/r06d_check_align.json; tail -c 1500 gpurun_out/r06d_cand_histogram.log
