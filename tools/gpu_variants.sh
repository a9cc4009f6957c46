#!/bin/bash
# development aid: time a list of library builds over the bench grid in one visit
# usage: tools/gpu_variants.sh <tag> <captures> <lib> [<lib> ...]
TAG=$1; N=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
for lib in "$@"; do
  echo "-- $lib ($N captures)" | tee -a $OUT/variants.txt
  timeout 300 python tools/variant_bench.py $lib $N 8 0 2>&1 | tail -1 | tee -a $OUT/variants.txt
done
