export TMPDIR=/tmp
TAG=${1:-r05_e}; OUT=gpurun_out/$TAG; mkdir -p $OUT
J='import json,sys
d=json.load(open(sys.argv[1])); print({k: d.get(k) for k in ("value","ms_per_step","breakdown_ms","pcie_inclusive","bitbuffers_to_host_per_step","d2h_bytes_per_step_per_gpu")}, d.get("roofline",{}).get("frac"))'
for x in 3 2; do
  echo "== bench --quick --exclusive $x"
  timeout 200 python bench.py --quick --steps 30 --warmup 3 --exclusive $x > $OUT/bench_quick_x$x.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/bench_quick_x$x.json </dev/null
done
echo "== pytest prefilter + dispatch -m gpu"
timeout 400 python -m pytest tests/test_prefilter.py tests/test_dispatch.py -m gpu -q -x </dev/null 2>&1 | tail -3 | tee $OUT/pytest.txt
echo "== dispatch trace"
timeout 200 python tools/dispatch_trace.py 24 1 </dev/null 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-1200 | tee $OUT/dispatch_trace.txt
echo "== slicers"
timeout 200 python tools/slice_pf_bench.py </dev/null 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/slice_pf.txt
