#!/bin/bash
OUT=gpurun_out/r4f
mkdir -p $OUT
export TMPDIR=/tmp
for t in 32 64 96; do
  echo "== quick, $t threads" | tee -a $OUT/quick.txt
  timeout 300 python bench.py --quick --steps 12 --warmup 2 --threads $t 2>$OUT/err_$t.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: j.get(k) for k in ['value','ms_per_step','breakdown_ms','hbm_resident','d2h_bytes_per_step_per_gpu','decoded_messages_per_step','bitbuffers_to_host_per_step']})" | tee -a $OUT/quick.txt
  tail -2 $OUT/err_$t.txt
done
echo "== default bench"
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
tail -3 $OUT/bench.err
python -c "
import json
j=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
for k,v in j.items():
    print(k, ':', json.dumps(v)[:600])
"
