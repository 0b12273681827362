#!/bin/bash
# round 6, ninth GPU call: the planning lanes' own 8 x 8 alignments (nw8_inline): suite, CHECK_ALIGN, smoke(), bench with and without
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --maxfail=3 > gpurun_out/r06i_gpu_tests.log 2>&1
E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06i_check_align.json 2> gpurun_out/r06i_check_align.err
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06i_smoke.log 2>&1
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline"
timeout 900 python bench.py $A > gpurun_out/r06i_bench_inline.json 2> gpurun_out/r06i_bench_inline.err
KG_ALN_NO_INLINE=1 timeout 900 python bench.py $A > gpurun_out/r06i_bench_noinline.json 2> gpurun_out/r06i_bench_noinline.err
KART_AMD_STREAM_LANES=12 timeout 900 python bench.py $A > gpurun_out/r06i_bench_12lanes.json 2> gpurun_out/r06i_bench_12lanes.err
tail -c 400 gpurun_out/r06i_gpu_tests.log; tail -c 300 gpurun_out/r06i_check_align.json; tail -2 gpurun_out/r06i_smoke.log
