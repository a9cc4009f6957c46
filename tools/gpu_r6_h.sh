#!/bin/bash
# round 6, visit h: the CLI back to back with its 2 KB staging slots, the tests around them
TAG=${1:-r06_h}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && timeout 300 bash tools/cli_trace.sh 8192 > /dev/null 2>&1 </dev/null
cd /tmp/cli_bench && ARGS=$(ls s*_433.92M_250k.cu8 | head -8192 | sed 's/^/-r /' | tr '\n' ' ')
series() { name=$1; pause=$2; shift 2
  for rep in $(seq 10); do
    s=$(date +%s%N)
    env "$@" RTL433_HIP_TRACE=1 $GRAFT_REPO_ROOT/dropin/_build/rtl_433_hip $ARGS -F json:/tmp/cli_bench/hip.json -M level -K FILE 2> $OUT/t.txt </dev/null
    e=$(date +%s%N)
    echo "$name run $rep: wall $(( (e - s) / 1000000 )) ms | $(grep -E 'GPU opened' $OUT/t.txt | sed 's/hip flow: //' | cut -c1-12) GPU open | $(grep -E 'exit handlers' $OUT/t.txt | sed 's/hip flow: exit handlers begin //' | cut -c1-9) inside"
    sleep $pause
  done; }
{ series "2K-slots-back-to-back" 0 X=1
  series "8K-slots-back-to-back" 0 RTL433_HIP_STAGE_SLOT=8192
  series "1K-slots-back-to-back" 0 RTL433_HIP_STAGE_SLOT=1024
} | tee $OUT/cli_series.txt
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_staging_slots.py tests/test_dropin.py tests/test_pipeline_host.py tests/test_corpus.py tests/test_split.py tests/test_long_streams.py -m gpu -x -q </dev/null 2>&1 | tail -4 | tee $OUT/pytest_subset.txt
