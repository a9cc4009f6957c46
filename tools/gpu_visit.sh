#!/bin/bash
# A short GPU visit (round 5: GPU minutes are counted): every step under its own timeout, everything into gpurun_out/<tag>/.
#   tools/gpu_visit.sh <tag>
TAG=${1:-r05_c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
J='import json,sys
d=json.load(open(sys.argv[1])); print({k: d.get(k) for k in ("value","ms_per_step","breakdown_ms","pcie_inclusive","hbm_resident","bitbuffers_to_host_per_step","d2h_bytes_per_step_per_gpu","parity")}, d.get("roofline",{}).get("frac"))'
echo "== bench --quick (value = resident, pcie_inclusive beside it)"
timeout 300 python bench.py --quick --steps 30 --warmup 3 > $OUT/bench_quick.json 2> $OUT/bench_quick.err </dev/null; tail -2 $OUT/bench_quick.err; timeout 20 python -c "$J" $OUT/bench_quick.json </dev/null
for x in 1 3; do
  echo "== bench --quick --exclusive $x"
  timeout 200 python bench.py --quick --steps 20 --warmup 3 --exclusive $x > $OUT/bench_quick_x$x.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/bench_quick_x$x.json </dev/null
done
echo "== bench --quick --engines 4"
timeout 200 python bench.py --quick --steps 20 --warmup 3 --engines 4 > $OUT/bench_quick_e4.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/bench_quick_e4.json </dev/null
echo "== pytest test_prefilter -m gpu"
timeout 400 python -m pytest tests/test_prefilter.py -m gpu -q -x </dev/null 2>&1 | tail -4 | tee $OUT/pytest_prefilter.txt
echo "== slicers with the pre-filter tables"
timeout 200 python tools/slice_pf_bench.py </dev/null 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/slice_pf.txt
echo "== kbench: k_wave forms, variants"
{ for f in 0 32768 4096; do timeout 120 python tools/kbench.py --nodevs --reps 7 --streams 8192 --debug $f </dev/null 2>&1 | tail -1; done
  for v in cw4 pw3; do [ -f rtl_433_amd/lib/librtl433hip_$v.so ] && timeout 120 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip_$v.so 8192 6 0 </dev/null 2>&1 | tail -1; done
  timeout 120 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so 8192 6 0 </dev/null 2>&1 | tail -1; } | grep -v amdgpu.ids | tee $OUT/kbench.txt
ls $OUT
