#!/bin/bash
# round 6, eleventh GPU call: the finish pass with a candidate per WAVE (aln_finish_wave_kernel) against a candidate per lane; the gz leg
# (several-thread reader of ordinary gzip files); CHECK_ALIGN, the suite
mkdir -p gpurun_out
export TMPDIR=/tmp
E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06k_check_align.json 2> gpurun_out/r06k_check_align.err
tail -c 300 gpurun_out/r06k_check_align.json
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline"
timeout 900 python bench.py $A > gpurun_out/r06k_bench_finish_wave.json 2> gpurun_out/r06k_bench_finish_wave.err
KG_ALN_FINISH_LANES=1 timeout 900 python bench.py $A > gpurun_out/r06k_bench_finish_lanes.json 2> gpurun_out/r06k_bench_finish_lanes.err
KART_AMD_PGZ_DEBUG=1 timeout 1500 python bench.py --steps 2 --warmup 1 --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline > gpurun_out/r06k_bench_gz_leg.json 2> gpurun_out/r06k_bench_gz_leg.err
timeout 2400 python -m pytest tests -q -m gpu --maxfail=3 > gpurun_out/r06k_gpu_tests.log 2>&1
tail -c 400 gpurun_out/r06k_gpu_tests.log
python - <<'PY'
import json
for n in ("finish_wave", "finish_lanes", "gz_leg"):
    try:
        d = json.loads([l for l in open("gpurun_out/r06k_bench_%s.json" % n) if l.startswith("{")][-1])
        k = d["kernels"]
        print(n, round(d["value"] / 1e6, 2), "M reads/s; stage", round(d["alignment_stage"]["ms_per_step"], 1), "aln_finish", round(k["aln_finish"]["ms_per_step"], 1))
        if "other_configs" in d and "gz_input" in d["other_configs"]:
            print(json.dumps(d["other_configs"]["gz_input"])[:1500])
    except Exception as e:
        print(n, "unreadable", e)
PY
