export TMPDIR=/tmp
TAG=r05_d; OUT=gpurun_out/$TAG; mkdir -p $OUT
J='import json,sys
d=json.load(open(sys.argv[1])); print({k: d.get(k) for k in ("value","ms_per_step","breakdown_ms","pcie_inclusive","bitbuffers_to_host_per_step","d2h_bytes_per_step_per_gpu")}, d.get("roofline",{}).get("frac"))'
for x in 2 3 1; do
  echo "== bench --quick --exclusive $x"
  timeout 200 python bench.py --quick --steps 30 --warmup 3 --exclusive $x > $OUT/bench_quick_x$x.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/bench_quick_x$x.json </dev/null
done
echo "== bench --quick --exclusive 2 --threads 16"
timeout 200 python bench.py --quick --steps 30 --warmup 3 --exclusive 2 --threads 16 > $OUT/bench_quick_t16.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/bench_quick_t16.json </dev/null
echo "== dispatch trace"
timeout 200 python tools/dispatch_trace.py 24 1 </dev/null 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-1200 | tee $OUT/dispatch_trace.txt
