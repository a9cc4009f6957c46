#!/bin/bash
# round 6, visit n: variants of the library (tools/build_variant.py) over the slicers' passes of one bench step: tools/gpu_r6_n.sh <tag> <lib> ...
TAG=${1:-r06_n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; shift
export TMPDIR=/tmp
{ for i in 1 2; do
    timeout 300 python tools/slice_pf_bench.py </dev/null 2>&1 | tail -1
    for lib in "$@"; do
      [ -e "$lib" ] && timeout 300 python tools/slice_pf_bench.py "$lib" </dev/null 2>&1 | tail -1
    done
  done
} | grep -v amdgpu.ids | cut -c1-330 | tee -a $OUT/variants.txt
