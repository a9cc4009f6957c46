export TMPDIR=/tmp
TAG=${1:-r05_g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
J='import json,sys
d=json.load(open(sys.argv[1])); print({k: d.get(k) for k in ("value","ms_per_step","breakdown_ms")}, d.get("pcie_inclusive",{}).get("value"), d["config"].get("gpu_waits"), d["config"].get("host_dispatch_threads"))'
echo "== default"; timeout 200 python bench.py --quick --steps 30 --warmup 3 > $OUT/b0.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/b0.json </dev/null
echo "== nap wait"; R433_DEBUG_NAP_WAIT=1 timeout 200 python bench.py --quick --steps 30 --warmup 3 > $OUT/b_nap.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/b_nap.json </dev/null
for t in 16 20 32; do echo "== threads $t"; timeout 200 python bench.py --quick --steps 30 --warmup 3 --threads $t > $OUT/b_t$t.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/b_t$t.json </dev/null; done
echo "== nap wait, threads 20"; R433_DEBUG_NAP_WAIT=1 timeout 200 python bench.py --quick --steps 30 --warmup 3 --threads 20 > $OUT/b_nap20.json 2>/dev/null </dev/null; timeout 20 python -c "$J" $OUT/b_nap20.json </dev/null
