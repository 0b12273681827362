# usage (GPU box): bash tools/ab_index_modes.sh [modes...] -- the index modes on the hg38-sized synthetic index: resident bytes, the seeding stage alone
# (20 M reads per launch) and the FASTQ -> SAM run (20 M reads per step)
cd $GRAFT_REPO_ROOT
for sa in ${@:-full compact dense4 dense8}; do
  python bench.py --pairs 10000000 --leg seeding --seed-steps 3 --sa $sa 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)['seeding_stage']; print('seeding', d['sa_mode'], '| index GB', round(d['index_bytes']/1e9,1), '| reads/s', round(d['value']), '| kernels ms', {k: round(v, 2) for k, v in d['kernels_ms'].items()})
"
done
for sa in ${@:-full compact dense4 dense8}; do
  python bench.py --pairs 10000000 --steps 3 --warmup 1 --sa $sa --no-other-configs --no-cpu-baseline --no-parity --no-seeding-leg 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('fastq->sam', '$sa', '| reads/s', round(d['value']), '| step s', {k: round(v, 3) for k, v in d.get('step_seconds', {}).items()}, '| seed ms/step', round(d.get('device_ms_per_step', {}).get('seed', 0), 1))
"
done
