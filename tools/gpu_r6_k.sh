#!/bin/bash
# round 6, visit k: the narrow chunks of devices through k_slice_multi (a wavefront takes several packages) against a wavefront per
# package for every chunk (R433_SLICE_NO_MULTI): sizing pass of one bench step with the real decoders' pre-filter tables, digests
# tools/gpu_r6_k.sh [tag] [variant libraries ...]
TAG=${1:-r06_k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; shift
export TMPDIR=/tmp
{ for i in 1 2; do
    for lib in "$@"; do
      [ -n "$lib" ] && [ ! -e "$lib" ] && continue
      timeout 300 python tools/slice_pf_bench.py "$lib" </dev/null 2>&1 | tail -1
    done
    R433_SLICE_NO_MULTI=1 timeout 300 python tools/slice_pf_bench.py </dev/null 2>&1 | tail -1
    date +%T
  done
} | grep -v amdgpu.ids | cut -c1-330 | tee -a $OUT/sizing_multi_ab.txt
