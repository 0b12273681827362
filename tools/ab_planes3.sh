run() { timeout 600 python bench.py --leg seeding --seed-steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['seeding_stage']; print('$1', s['kernels_ms']['search'], s['oracle_sample'] if 'oracle_sample' in s else '', {k:round(v,2) for k,v in s['fetched_per_read'].items() if 'step' in k}, s['index_bytes'])"; }
timeout 300 python -m pytest tests/test_seed_gpu.py tests/test_large_gpu.py -x -q -m gpu 2>&1 | tail -2
run k3
KG_NO_PLANES3=1 run k2
python - <<'PY'
p='kart_amd/csrc/kernels/search.inc'
s=open(p).read()
s=s.replace("template <typename idx_t, bool kRaw>\n__global__ __launch_bounds__(256) void search_kernel","template <typename idx_t, bool kRaw>\n__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void search_kernel")
open(p,'w').write(s)
PY
make -C kart_amd/csrc -j8 > /dev/null 2>&1
run k3_w5
KG_NO_PLANES3=1 run k2_w5
