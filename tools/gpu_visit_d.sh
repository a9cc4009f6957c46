export TMPDIR=/tmp
OUT=gpurun_out/r05_m; mkdir -p $OUT
timeout 60 python -m pytest tests/test_roles_order.py -m gpu -q </dev/null 2>&1 | tail -1
{ timeout 60 python tools/kbench.py --nodevs --reps 9 --streams 8192 </dev/null 2>&1 | tail -1
  timeout 90 python tools/slice_pf_bench.py "" 8 0 </dev/null 2>&1 | tail -1; } | grep -v amdgpu.ids | tee $OUT/kbench.txt
