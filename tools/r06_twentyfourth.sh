#!/bin/bash
# round 6, twenty-fourth GPU call: the host library's last form (pages given back by their own thread, huge pages on): the stream / gz tests, smoke(), the gz leg
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_hostpath_gpu.py tests/test_stream_gpu.py tests/test_sam_gpu.py -q -m gpu > gpurun_out/r06x_tests.log 2>&1
tail -2 gpurun_out/r06x_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06x_smoke.log 2>&1
tail -1 gpurun_out/r06x_smoke.log
KART_BENCH_ONLY_GZ_LEG=1 timeout 900 python bench.py --steps 1 --warmup 0 --pairs 2000000 --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline > gpurun_out/r06x_bench_gz.json 2> gpurun_out/r06x_bench_gz.err
python - <<'PY'
import json, re
t = open("gpurun_out/r06x_bench_gz.json").read()
st = [m.start() for m in re.finditer(r'\{"metric"', t)][-1]
g = json.loads(t[st:t.index("\n", st)])["other_configs"]["gz_input"]
print("gz", g["map_seconds"], "plain", g["plain_files"]["map_seconds"], "same", g["same_sam_bytes_all_four"], round(g["value"] / 1e6, 2), "M reads/s")
PY
