run() { timeout 600 python bench.py --leg seeding --seed-steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['seeding_stage']['kernels_ms'])"; }
run base4
KG_SEARCH_BLOCKS_PER_CU=5 run base5
KG_SEARCH_BLOCKS_PER_CU=8 run base8
python - <<'PY'
p='kart_amd/csrc/kernels/search.inc'
s=open(p).read()
s=s.replace("template <typename idx_t, bool kRaw>\n__global__ __launch_bounds__(256) void search_kernel","template <typename idx_t, bool kRaw>\n__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void search_kernel")
open(p,'w').write(s)
PY
make -C kart_amd/csrc -j8 > /dev/null 2>&1
KG_SEARCH_BLOCKS_PER_CU=5 run w5_5
KG_SEARCH_BLOCKS_PER_CU=4 run w5_4
