#!/bin/bash
# round 6, visit j: the sizing pass after the slicers' branches were merged (one add_bit / add_row site per symbol)
TAG=${1:-r06_j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python tools/kbench.py --bench-batch --make-batch-only --streams 8192 </dev/null >/dev/null 2>&1
{ for i in 1 2 3; do
    timeout 300 python tools/slice_pf_bench.py </dev/null 2>&1 | tail -1
    timeout 300 python tools/kbench.py --reps 5 --streams 8192 --bench-batch </dev/null 2>&1 | tail -1
  done
} | grep -v amdgpu.ids | cut -c1-260 | tee $OUT/sizing.txt
