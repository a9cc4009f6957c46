#!/bin/bash
# round 6, visit g: is the CLI's slow GPU opening the teardown of the process before it?  runs with a pause between them, and with smaller staging slots
TAG=${1:-r06_g}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && timeout 300 bash tools/cli_trace.sh 8192 > /dev/null 2>&1 </dev/null
cd /tmp/cli_bench && ARGS=$(ls s*_433.92M_250k.cu8 | head -8192 | sed 's/^/-r /' | tr '\n' ' ')
series() { # name, pause, extra env...
  name=$1; pause=$2; shift 2
  for rep in $(seq 8); do
    s=$(date +%s%N)
    env "$@" RTL433_HIP_TRACE=1 $GRAFT_REPO_ROOT/dropin/_build/rtl_433_hip $ARGS -F json:/tmp/cli_bench/hip.json -M level -K FILE 2> $OUT/t.txt </dev/null
    e=$(date +%s%N)
    echo "$name run $rep: wall $(( (e - s) / 1000000 )) ms | $(grep -E 'GPU opened' $OUT/t.txt | sed 's/hip flow: //' | cut -c1-12) GPU open | $(grep -E 'exit handlers' $OUT/t.txt | sed 's/hip flow: exit handlers begin //' | cut -c1-9) inside"
    sleep $pause
  done
}
{ series "back-to-back" 0 X=1
  series "pause-1.5s" 1.5 X=1
  series "stage-2K-back-to-back" 0 R433_STAGE_CAP=2048
  series "pause-0.3s" 0.3 X=1
} | tee $OUT/cli_series.txt
