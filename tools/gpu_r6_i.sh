#!/bin/bash
# round 6, visit i: the placing pass without LDS (A/B), staging-slot tests, the default bench line
TAG=${1:-r06_i}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python tools/kbench.py --bench-batch --make-batch-only --streams 8192 </dev/null >/dev/null 2>&1
{ for e in "" "R433_PLACE_FROM_LDS=1" "" "R433_PLACE_FROM_LDS=1"; do
    echo "== placing pass: ${e:-pulses from HBM (new)}"
    env $e X=1 timeout 300 python tools/slice_pf_bench.py </dev/null 2>&1 | tail -1
    env $e X=1 timeout 300 python tools/kbench.py --reps 5 --streams 8192 --bench-batch </dev/null 2>&1 | tail -1
  done
  for cap in 2048 1024; do
    echo "== 2K / 1K slots (R433_STAGE_CAP=$cap): new / from LDS"
    R433_STAGE_CAP=$cap timeout 300 python tools/kbench.py --reps 5 --streams 8192 --bench-batch </dev/null 2>&1 | tail -1
    R433_STAGE_CAP=$cap R433_PLACE_FROM_LDS=1 timeout 300 python tools/kbench.py --reps 5 --streams 8192 --bench-batch </dev/null 2>&1 | tail -1
  done
} | grep -v amdgpu.ids | cut -c1-260 | tee $OUT/placing_ab.txt
timeout 600 python -m pytest tests/test_staging_slots.py tests/test_gpu_parity.py -m gpu -x -q </dev/null 2>&1 | tail -3 | tee $OUT/pytest_subset.txt
