#!/bin/bash
# round 6, seventh GPU call: where the host faulted in call six (KART_AMD_BACKTRACE), then the suite, CHECK_ALIGN and the batch-size A/B again
mkdir -p gpurun_out
export TMPDIR=/tmp
A="--steps 2 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity"
KART_AMD_BACKTRACE=1 timeout 900 python bench.py $A --no-gpu-pipeline > gpurun_out/r06g_bench_bt.json 2> gpurun_out/r06g_bench_bt.err
KART_AMD_BACKTRACE=1 KART_AMD_FETCH_ALL=1 timeout 900 python bench.py $A --no-gpu-pipeline > gpurun_out/r06g_bench_bt_fetchall.json 2> gpurun_out/r06g_bench_bt_fetchall.err
timeout 2400 python -m pytest tests -q -m gpu --maxfail=3 > gpurun_out/r06g_gpu_tests.log 2>&1
E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06g_check_align.json 2> gpurun_out/r06g_check_align.err
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity"
for sr in 1120000 2000000 3000000; do
  KART_AMD_STREAM_READS=$sr timeout 900 python bench.py $A > gpurun_out/r06g_bench_${sr}.json 2> gpurun_out/r06g_bench_${sr}.err
done
grep -v "^    @" gpurun_out/r06g_bench_bt.err | tail -40 | cut -c1-300; tail -c 600 gpurun_out/r06g_gpu_tests.log; tail -c 300 gpurun_out/r06g_check_align.json
