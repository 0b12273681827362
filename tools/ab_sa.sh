# usage (GPU box): bash tools/ab_sa.sh -- the suffix array's placement at 100 M reads per step: `full` (168 GB resident) against `compact`
# (66.9 GB, 5-byte entries), alternating on one box: value, the steps, roofline.frac, the seeding kernels' time per step
cd $GRAFT_REPO_ROOT
A="--steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 bench.py $A --steps 1 --warmup 0 > /dev/null 2>&1        # builds + caches the index
for rep in 1 2 3; do
for sa in full compact; do
  python3 bench.py $A --sa $sa 2>/dev/null | python3 -c "
import json, sys
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
k = d['kernels']
ss = d['step_seconds']; r = d['roofline']
print('sa %-8s run $rep: %.2f M mapped reads/s  steps min %.3f median %.3f max %.3f  frac %.4f  search %.1f ms per step (%.2f ms per launch), timed kernels %.0f ms per step' % ('$sa', d['value'] / 1e6, ss['min'], ss['median'], ss['max'], r['frac'], r['search_kernel_ms_per_step'], r['avg_launch_ms'], k['timed_kernel_ms_per_step']))"
done
done
