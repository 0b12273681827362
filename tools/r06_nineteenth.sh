#!/bin/bash
# round 6, nineteenth GPU call: pass 1 of the report by groups of eight lanes (aln_plan_group_kernel, KG_ALN_PLAN_GROUP): CHECK_ALIGN, the SAM / stream /
# hg38-sized identity tests with it, and the bench with and without
mkdir -p gpurun_out
export TMPDIR=/tmp
KG_ALN_PLAN_GROUP=1 E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06s_check_align_plan_group.json 2> gpurun_out/r06s_check_align_plan_group.err
grep -o "CHECK_ALIGN[^\"]*" gpurun_out/r06s_check_align_plan_group.json | head -2
KG_ALN_PLAN_GROUP=1 timeout 1500 python -m pytest tests/test_sam_gpu.py tests/test_stream_gpu.py tests/test_hg38_gpu.py tests/test_large_gpu.py tests/test_ecoli_gpu.py -q -m gpu --maxfail=5 > gpurun_out/r06s_tests_plan_group.log 2>&1
tail -c 300 gpurun_out/r06s_tests_plan_group.log
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline"
KG_ALN_PLAN_GROUP=1 timeout 900 python bench.py $A > gpurun_out/r06s_bench_plan_group.json 2> gpurun_out/r06s_bench_plan_group.err
timeout 900 python bench.py $A > gpurun_out/r06s_bench_plan_lanes.json 2> gpurun_out/r06s_bench_plan_lanes.err
python - <<'PY'
import json, re
for n in ("plan_group", "plan_lanes"):
    try:
        t = open("gpurun_out/r06s_bench_%s.json" % n).read()
        st = [m.start() for m in re.finditer(r'\{"metric"', t)][-1]
        d = json.loads(t[st:t.index("\n", st)])
        k = d["kernels"]
        print(n, round(d["value"] / 1e6, 2), "M reads/s; stage", round(d["alignment_stage"]["ms_per_step"], 1), "aln_plan", round(k["aln_plan"]["ms_per_step"], 1), "aln_finish", round(k["aln_finish"]["ms_per_step"], 1))
    except Exception as e:
        print(n, "unreadable", e)
PY
