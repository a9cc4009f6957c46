#!/bin/bash
# round 6, visit c: the default bench line (new: real decoders behind configs[2] / [4], the list, issue_frac) + the tests touched so far
TAG=${1:-r06_c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err </dev/null; echo "bench rc $? in $(( $(date +%s) - t0 )) s"; tail -3 $OUT/bench.err
timeout 60 python - $OUT/bench.json <<'PY' </dev/null
import json,sys
d=json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value","value_r04_definition","ms_per_step","parity")})
print("roofline", {k: d["roofline"].get(k) for k in ("frac","issue_frac","issue_floor","traffic","slicers")})
print("breakdown", d.get("breakdown_ms"))
for k, v in (d.get("other_configs") or {}).items():
    print(k, {x: v.get(x) for x in ("value","ms_per_step","parity","decoded_messages_per_step","error","cpu_baseline")}, (v.get("roofline") or {}).get("frac"))
    if "latency_per_burst_ms" in v: print("   latency", v["latency_per_burst_ms"])
print("dropin", d.get("dropin"))
PY
timeout 900 python -m pytest tests/test_roles_order.py tests/test_prefilter.py tests/test_dropin.py -m gpu -x -q </dev/null 2>&1 | tail -4 | tee $OUT/pytest_subset.txt
