#!/bin/bash
# round 6, eighteenth GPU call -- the round's final state: the GPU suite, smoke(), the profile passes (trace + counters), the driver's command
# (which prices its kernels' traffic with THIS call's counter summary), 2 M reads of the timed files against kart -t 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --maxfail=5 > gpurun_out/r06r_gpu_tests.log 2>&1
tail -c 300 gpurun_out/r06r_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06r_smoke.log 2>&1
tail -1 gpurun_out/r06r_smoke.log
bash tools/profile_r06.sh r06r > gpurun_out/r06r_profile.log 2>&1
cp gpurun_out/r06r_bench_pmc_summary.json profiles/r06r_bench_pmc_summary.json
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06r_bench_default_steps20.log 2> gpurun_out/r06r_bench_default_steps20.err
tail -3 gpurun_out/r06r_bench_default_steps20.err > gpurun_out/r06r_time.txt
KART_BENCH_IDENT_PAIRS=1000000 timeout 1500 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-seeding-leg --no-other-configs --no-gpu-pipeline > gpurun_out/r06r_identity_2m.log 2> gpurun_out/r06r_identity_2m.err
python - <<'PY'
import json
for n in ("bench_default_steps20", "identity_2m"):
    try:
        d = json.loads([l for l in open("gpurun_out/r06r_%s.log" % n) if l.startswith("{")][-1])
        print(n, round(d["value"] / 1e6, 2), "M reads/s; stage", round(d["alignment_stage"]["ms_per_step"], 1), "parity", json.dumps(d.get("parity"))[:300])
        if "other_configs" in d: print(json.dumps({k: v.get("value") for k, v in d["other_configs"].items() if isinstance(v, dict)}))
    except Exception as e:
        print(n, "unreadable", e)
PY
cat gpurun_out/r06r_time.txt
