#!/bin/bash
# round 6, visit f: the CLI's slow runs -- twelve traced runs in a row, the timeline of each, the cgroup's throttle counters around them
TAG=${1:-r06_f}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && timeout 300 bash tools/cli_trace.sh 8192 > /dev/null 2>&1 </dev/null
cd /tmp/cli_bench && ARGS=$(ls s*_433.92M_250k.cu8 | head -8192 | sed 's/^/-r /' | tr '\n' ' ')
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
for rep in $(seq 12); do
  t0="$(thr)"; s=$(date +%s%N)
  RTL433_HIP_TRACE=1 $GRAFT_REPO_ROOT/dropin/_build/rtl_433_hip $ARGS -F json:/tmp/cli_bench/hip.json -M level -K FILE 2> $OUT/trace.$rep.txt </dev/null
  e=$(date +%s%N); echo "run $rep: $(( (e - s) / 1000000 )) ms | before: $t0 | after: $(thr)"
  grep -E "^hip flow: \[|GPU opened|exit handlers|leaving" $OUT/trace.$rep.txt | sed 's/^hip flow: //' | cut -c1-110 | tr '\n' ';' ; echo
done | tee $OUT/cli_runs.txt
