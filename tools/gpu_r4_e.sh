#!/bin/bash
OUT=gpurun_out/r4e
mkdir -p $OUT
export TMPDIR=/tmp
nproc | tee $OUT/host.txt
for t in 32 64 128; do
  echo "== real decoders, $t threads" | tee -a $OUT/timeline.txt
  timeout 300 python tools/leg_timeline.py 12 3 2 $t 1 2>&1 | head -16 | tee -a $OUT/timeline.txt
done
echo "== real decoders, 64 threads, full timeline" >> $OUT/timeline.txt
timeout 300 python tools/leg_timeline.py 12 3 2 64 1 >> $OUT/timeline.txt 2>&1
echo "== checksum, 32 threads" | tee -a $OUT/timeline.txt
timeout 300 python tools/leg_timeline.py 12 3 2 32 0 2>&1 | head -16 | tee -a $OUT/timeline.txt
echo "== dispatch trace (one batch of 8192, real decoders, 64 threads)" | tee -a $OUT/timeline.txt
timeout 300 python tools/dispatch_trace.py 64 2>&1 | tail -12 | tee -a $OUT/timeline.txt
