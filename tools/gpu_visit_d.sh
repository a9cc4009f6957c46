export TMPDIR=/tmp
TAG=${1:-r05_j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 120 python -m pytest tests/test_roles_order.py tests/test_split_lazy.py -m gpu -q </dev/null 2>&1 | tail -2
{ timeout 100 python tools/kbench.py --nodevs --reps 9 --streams 8192 </dev/null 2>&1 | tail -1
  timeout 150 python tools/slice_pf_bench.py "" 8 0 </dev/null 2>&1 | tail -1; } | grep -v amdgpu.ids | tee $OUT/kbench.txt
timeout 200 python bench.py --quick --steps 30 --warmup 3 > $OUT/bench_quick.json 2>/dev/null </dev/null; timeout 20 python -c "
import json; d=json.load(open('$OUT/bench_quick.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms']['k_wave_alone'], d['breakdown_ms']['host_dispatch'])" </dev/null
