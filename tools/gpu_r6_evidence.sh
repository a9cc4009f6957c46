#!/bin/bash
# round 6: evidence beside the closing visit -- a long fuzz run of the final build through the product library, the BASELINE configs as lines of their own
TAG=${1:-r06}; OUT=gpurun_out/${TAG}_evidence; mkdir -p $OUT
export TMPDIR=/tmp
echo "== fuzz: random signals x random flow options against the oracle, the final build on the GPU"
timeout 2400 python tools/fuzz_emu.py --gpu ${FUZZ_CASES:-20000} 600000 </dev/null 2>&1 | tail -4 | tee $OUT/fuzz.txt
for c in 3 5 4; do
  echo "== bench.py --config $c"
  timeout 900 python bench.py --config $c > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err </dev/null; tail -1 $OUT/bench_config$c.err | cut -c1-200
  timeout 20 python -c "
import json
d=json.load(open('$OUT/bench_config$c.json')); print({k: d.get(k) for k in ('value','ms_per_step','parity')}, d.get('roofline',{}).get('frac'), d.get('breakdown_ms',{}).get('detect_ms'), d.get('latency_per_burst_ms',{}).get('p50'))" </dev/null
done
echo "== bench (the driver's form) again, drop-in legs at the end"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err </dev/null; tail -1 $OUT/bench.err | cut -c1-200
timeout 20 python -c "
import json
d=json.load(open('$OUT/bench.json')); print({k: d.get(k) for k in ('value','ms_per_step','parity')}, d['roofline']['frac'], d['roofline']['issue_frac']); dd=d['dropin']; print(dd['rtl_433_hip']['wall_ms'], dd['rtl_433_hip']['median_ms'], dd['rtl_433_hip']['max_over_min'], dd['rtl_433_hip']['wall_ms_back_to_back'], dd['pipeline_host_hip']['wall_ms'])" </dev/null
