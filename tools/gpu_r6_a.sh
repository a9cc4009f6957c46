#!/bin/bash
# round 6, visit a: the detection pass on the bench's own batch -- time and per-phase shader clocks (timing build)
TAG=${1:-r06_a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
{ for n in 8192 1024; do
    timeout 300 python tools/kbench.py --nodevs --reps 7 --streams $n --bench-batch </dev/null 2>&1 | tail -2
    timeout 300 python tools/kbench.py --nodevs --reps 3 --streams $n --bench-batch --debug 1024 </dev/null 2>&1 | tail -26
  done
  timeout 300 python tools/kbench.py --nodevs --reps 7 --streams 8192 </dev/null 2>&1 | tail -1
} | grep -v amdgpu.ids | tee $OUT/kbench_phases.txt
