#!/bin/bash
OUT=gpurun_out/r4c
mkdir -p $OUT
export TMPDIR=/tmp
for n in 8192 4096 2048; do
for flags in 0 4096; do
  echo "-- $n captures flags $flags" | tee -a $OUT/forms.txt
  timeout 300 python tools/variant_bench.py rtl_433_amd/lib/librtl433hip.so $n 6 0 $flags 2>&1 | tail -1 | tee -a $OUT/forms.txt
done
done
