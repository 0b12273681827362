#!/bin/bash
# round 6, fourteenth GPU call: the whole GPU suite on the gz-stream / group-finish code, the odd inputs and the fuzzers against the live reference
# (gz cases through the device stream, with and without the several-thread reader on small files), the heavy-pair threshold
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --maxfail=5 > gpurun_out/r06n_gpu_tests.log 2>&1
tail -c 600 gpurun_out/r06n_gpu_tests.log
( FZ_A="201 230" FZ_B="21 23" FZ_C="50 62" bash tools/fuzz_on_gpu_box.sh ) > gpurun_out/r06n_fuzz_on_box.log 2>&1
( mkdir -p /tmp/fz/odd2 && cd /tmp/fz/odd2 && KART_FUZZ_BIN=$GRAFT_REPO_ROOT/kart_amd/bin/kart-amd KART_AMD_PGZ_MIN_KB=0 KART_AMD_PGZ_CHUNK_KB=8 timeout 900 python $GRAFT_REPO_ROOT/tools/probe_odd_inputs_vs_reference.py 2>&1 | grep -v " same | same$" | tail -20 ) > gpurun_out/r06n_odd_inputs_small_chunks.log 2>&1
tail -30 gpurun_out/r06n_fuzz_on_box.log; cat gpurun_out/r06n_odd_inputs_small_chunks.log
A="--steps 4 --warmup 1 --no-other-configs --no-seeding-leg --no-cpu-baseline --no-parity --no-gpu-pipeline"
for h in 8 16; do
  KG_ALN_PAIR_HEAVY=$h timeout 900 python bench.py $A > gpurun_out/r06n_bench_pair_heavy_$h.json 2> gpurun_out/r06n_bench_pair_heavy_$h.err
done
python - <<'PY'
import json, re
for n in ("pair_heavy_8", "pair_heavy_16"):
    try:
        t = open("gpurun_out/r06n_bench_%s.json" % n).read()
        st = [m.start() for m in re.finditer(r'\{"metric"', t)][-1]
        d = json.loads(t[st:t.index("\n", st)])
        k = d["kernels"]
        print(n, round(d["value"] / 1e6, 2), "M reads/s; stage", round(d["alignment_stage"]["ms_per_step"], 1), "aln_pair", round(k["aln_pair"]["ms_per_step"], 1), "aln_finish", round(k["aln_finish"]["ms_per_step"], 1))
    except Exception as e:
        print(n, "unreadable", e)
PY
