#!/bin/bash
# round 6, visit m: line codes with a handful of decoders share chunks of 64 device rows (host_api.cpp) against a chunk per line code
# (R433_SLICE_NO_PACK): sizing / placing pass of one bench step with the real decoders' pre-filter tables, digest of the records
TAG=${1:-r06_m}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
{ for i in 1 2 3; do
    timeout 300 python tools/slice_pf_bench.py </dev/null 2>&1 | tail -1
    R433_SLICE_NO_PACK=1 timeout 300 python tools/slice_pf_bench.py </dev/null 2>&1 | tail -1
  done
} | grep -v amdgpu.ids | cut -c1-330 | tee -a $OUT/sizing_pack_ab.txt
