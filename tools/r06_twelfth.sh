#!/bin/bash
# round 6, twelfth GPU call: the gz leg at 40 M reads (the general path's rate behind the several-thread gzip reader once the pipeline has filled)
mkdir -p gpurun_out
export TMPDIR=/tmp
KART_BENCH_GZ_PAIRS=20000000 KART_BENCH_ONLY_GZ_LEG=1 KART_AMD_VERBOSE=1 timeout 1500 python bench.py --steps 1 --warmup 0 --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline > gpurun_out/r06l_bench_gz_leg_40m.json 2> gpurun_out/r06l_bench_gz_leg_40m.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06l_bench_gz_leg_40m.json") if l.startswith("{")][-1])
print(json.dumps(d["other_configs"]["gz_input"])[:1500])
PY
grep -c "pgz round" gpurun_out/r06l_bench_gz_leg_40m.err
