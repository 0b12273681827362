#!/bin/bash
# round 6, twenty-first GPU call: the driver's command on the round's last binary (the finish pass's report tail as a shared function, the gz leg at 40 M reads),
# then the larger identity runs by hand: 8 M reads of the timed files against kart -t 1; 100 000 x 7 kb long reads (+ 0.5 M short, 0.2 M -m) in the hg38-sized tests
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06u_bench_default_steps20.log 2> gpurun_out/r06u_bench_default_steps20.err
tail -3 gpurun_out/r06u_bench_default_steps20.err > gpurun_out/r06u_time.txt
KART_BENCH_IDENT_PAIRS=4000000 timeout 2400 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-seeding-leg --no-other-configs --no-gpu-pipeline > gpurun_out/r06u_identity_8m.log 2> gpurun_out/r06u_identity_8m.err
KART_TEST_N_LONG=100000 KART_TEST_LONG_SLICES=16 timeout 1500 python3 -m pytest tests/test_hg38_gpu.py -q -m gpu > gpurun_out/r06u_hg38_long_100k.log 2>&1
tail -2 gpurun_out/r06u_hg38_long_100k.log
python - <<'PY'
import json
for n in ("bench_default_steps20", "identity_8m"):
    try:
        d = json.loads([l for l in open("gpurun_out/r06u_%s.log" % n) if l.startswith("{")][-1])
        print(n, round(d["value"] / 1e6, 2), "M reads/s; stage", round(d["alignment_stage"]["ms_per_step"], 1), "parity", json.dumps(d.get("parity"))[:300])
        if "other_configs" in d: print(json.dumps({k: v.get("value") for k, v in d["other_configs"].items() if isinstance(v, dict)}))
    except Exception as e:
        print(n, "unreadable", e)
PY
cat gpurun_out/r06u_time.txt
