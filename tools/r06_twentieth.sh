#!/bin/bash
# round 6, twentieth GPU call: aln_pair_kernel held to five waves per SIMD (96 registers, 18 spilled) against four (116)
mkdir -p gpurun_out
export TMPDIR=/tmp
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline"
timeout 900 python bench.py $A > gpurun_out/r06t_bench_pair_5waves.json 2> gpurun_out/r06t_bench_pair_5waves.err
E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06t_check_align.json 2> gpurun_out/r06t_check_align.err
grep -o "CHECK_ALIGN[^\"]*" gpurun_out/r06t_check_align.json | head -2
python - <<'PY'
import json, re
t = open("gpurun_out/r06t_bench_pair_5waves.json").read()
st = [m.start() for m in re.finditer(r'\{"metric"', t)][-1]
d = json.loads(t[st:t.index("\n", st)])
k = d["kernels"]
print(round(d["value"] / 1e6, 2), "M reads/s; stage", round(d["alignment_stage"]["ms_per_step"], 1), {n: round(k[n]["ms_per_step"], 1) for n in ("aln_pair", "aln_rescue", "aln_trivial", "aln_plan", "aln_finish")})
PY
