#!/bin/bash
# round 6, thirteenth GPU call: gz libraries through the device stream (GzProducer) and the finish pass by groups of sixteen lanes
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_hostpath_gpu.py tests/test_stream_gpu.py -q -m gpu --maxfail=5 > gpurun_out/r06m_gz_stream_tests.log 2>&1
tail -c 1500 gpurun_out/r06m_gz_stream_tests.log
E2E_CHECK_ALIGN=1 E2E_NO_REF=1 timeout 1200 python tools/e2e_large.py 3100000000 2000000 > gpurun_out/r06m_check_align.json 2> gpurun_out/r06m_check_align.err
tail -c 300 gpurun_out/r06m_check_align.json
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline"
timeout 900 python bench.py $A > gpurun_out/r06m_bench_finish_group.json 2> gpurun_out/r06m_bench_finish_group.err
KG_ALN_FINISH_WAVE=1 timeout 900 python bench.py $A > gpurun_out/r06m_bench_finish_wave.json 2> gpurun_out/r06m_bench_finish_wave.err
KART_BENCH_GZ_PAIRS=20000000 KART_BENCH_ONLY_GZ_LEG=1 timeout 1500 python bench.py --steps 1 --warmup 0 --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline > gpurun_out/r06m_bench_gz_leg_40m.json 2> gpurun_out/r06m_bench_gz_leg_40m.err
python - <<'PY'
import json, re
for n in ("finish_group", "finish_wave", "gz_leg_40m"):
    try:
        t = open("gpurun_out/r06m_bench_%s.json" % n).read()
        st = [m.start() for m in re.finditer(r'\{"metric"', t)][-1]
        d = json.loads(t[st:t.index("\n", st)])
        k = d["kernels"]
        print(n, round(d["value"] / 1e6, 2), "M reads/s; stage", round(d["alignment_stage"]["ms_per_step"], 1), "aln_finish", round(k["aln_finish"]["ms_per_step"], 1))
        if "other_configs" in d and "gz_input" in d["other_configs"]:
            print(json.dumps(d["other_configs"]["gz_input"])[:1500])
    except Exception as e:
        print(n, "unreadable", e)
PY
