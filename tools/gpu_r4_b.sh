#!/bin/bash
OUT=gpurun_out/r4b
mkdir -p $OUT
export TMPDIR=/tmp
for n in 1024 8192; do
  echo "== timing build, $n captures" | tee -a $OUT/timing.txt
  timeout 300 python tools/kbench.py --nodevs --streams $n --debug 1024 --reps 3 2>&1 | tail -32 | tee -a $OUT/timing.txt
done
echo "== quiet statistics (product build)" | tee -a $OUT/timing.txt
timeout 300 python tools/lazy_stats.py 2>&1 | tail -12 | tee -a $OUT/timing.txt
