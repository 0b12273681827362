#!/bin/bash
# round 6, twenty-third GPU call: the gz leg with the growing block's pages given back by a thread of their own (not by the in-order commit), without and with
# transparent huge pages for the block; then the gz tests
mkdir -p gpurun_out
export TMPDIR=/tmp
A="--steps 1 --warmup 0 --pairs 2000000 --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline"
KART_BENCH_ONLY_GZ_LEG=1 timeout 900 python bench.py $A > gpurun_out/r06w_bench_gz_async_release.json 2> gpurun_out/r06w_bench_gz_async_release.err
KART_AMD_GZ_THP=1 KART_BENCH_ONLY_GZ_LEG=1 timeout 900 python bench.py $A > gpurun_out/r06w_bench_gz_thp.json 2> gpurun_out/r06w_bench_gz_thp.err
timeout 600 python -m pytest tests/test_zz_hostpath_gpu.py -q -m gpu > gpurun_out/r06w_gz_tests.log 2>&1
tail -2 gpurun_out/r06w_gz_tests.log
python - <<'PY'
import json, re
for n in ("gz_async_release", "gz_thp"):
    try:
        t = open("gpurun_out/r06w_bench_%s.json" % n).read()
        st = [m.start() for m in re.finditer(r'\{"metric"', t)][-1]
        d = json.loads(t[st:t.index("\n", st)])
        g = d["other_configs"]["gz_input"]
        print(n, "gz", g["map_seconds"], "plain", g["plain_files"]["map_seconds"], "host reader", g["several_threads_into_the_hosts_gz_reader"]["map_seconds"], "same", g["same_sam_bytes_all_four"], round(g["value"] / 1e6, 2), "M reads/s")
    except Exception as e:
        print(n, "unreadable", e)
PY
