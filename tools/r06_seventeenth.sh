#!/bin/bash
# round 6, seventeenth GPU call: the finish pass with eight lanes per candidate against sixteen; the gz tests once more
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_hostpath_gpu.py -q -m gpu > gpurun_out/r06q_gz_tests.log 2>&1
tail -2 gpurun_out/r06q_gz_tests.log
KG_ALN_FINISH_G8=1 E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06q_check_align_g8.json 2> gpurun_out/r06q_check_align_g8.err
grep -o "CHECK_ALIGN[^\"]*" gpurun_out/r06q_check_align_g8.json | head -2
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline"
KG_ALN_FINISH_G8=1 timeout 900 python bench.py $A > gpurun_out/r06q_bench_finish_g8.json 2> gpurun_out/r06q_bench_finish_g8.err
timeout 900 python bench.py $A > gpurun_out/r06q_bench_finish_g16.json 2> gpurun_out/r06q_bench_finish_g16.err
python - <<'PY'
import json, re
for n in ("finish_g8", "finish_g16"):
    try:
        t = open("gpurun_out/r06q_bench_%s.json" % n).read()
        st = [m.start() for m in re.finditer(r'\{"metric"', t)][-1]
        d = json.loads(t[st:t.index("\n", st)])
        k = d["kernels"]
        print(n, round(d["value"] / 1e6, 2), "M reads/s; stage", round(d["alignment_stage"]["ms_per_step"], 1), "aln_finish", round(k["aln_finish"]["ms_per_step"], 1))
    except Exception as e:
        print(n, "unreadable", e)
PY
