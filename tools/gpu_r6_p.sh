#!/bin/bash
# round 6, visit p: the producers of the detection pass at 4 (the tree) / 5 / 6 wavefronts to a SIMD (R433_PRODUCER_WAVES variants of the library):
# the detection pass alone on the bench's batch (tools/kbench.py --nodevs --bench-batch), digest of the packages
TAG=${1:-r06_p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; shift
export TMPDIR=/tmp
timeout 200 python tools/kbench.py --bench-batch --make-batch-only --streams 8192 </dev/null >/dev/null 2>&1
{ for i in 1 2; do
    timeout 100 python tools/kbench.py --nodevs --reps 7 --streams 8192 --bench-batch </dev/null 2>&1 | tail -1
    for lib in "$@"; do [ -e "$lib" ] && timeout 100 python tools/kbench.py --nodevs --reps 7 --streams 8192 --bench-batch --lib $lib </dev/null 2>&1 | tail -1; done
  done; } | grep -v amdgpu.ids | cut -c1-300 | tee $OUT/producers.txt
