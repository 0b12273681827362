# usage (GPU box): bash tools/ab_sam_buf.sh -- sam_format_kernel's LDS line buffer (KG_SAM_BUF bytes; 0 = every line straight to the output, rounds 3-4):
# the kernel's own time (HIP events around its launches, bench.py `kernels`) and the step at 20 M reads per step, alternating
cd $GRAFT_REPO_ROOT
A="--pairs ${PAIRS:-10000000} --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 bench.py $A > /dev/null 2>&1        # builds + caches the index
for rep in 1 2; do
for b in ${BUFS:-0 2048 4096 8192 16384}; do
  KG_SAM_BUF=$b python3 bench.py $A 2>/dev/null | python3 -c "
import json, sys
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
k = d['kernels']
print('KG_SAM_BUF=%-6s value %.2f M reads/s | sam_format %.1f ms per step (%.0f launches), sam_size %.1f, fq_materialise %.1f, timed kernels %.0f ms' % ('$b', d['value'] / 1e6, k['sam_format']['ms_per_step'], k['sam_format']['launches_per_step'], k['sam_size']['ms_per_step'], k['fq_materialise']['ms_per_step'], k['timed_kernel_ms_per_step']))"
done
done
