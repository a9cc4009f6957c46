#!/bin/bash
# the drop-in CLI over 8192 files against the stock binary (the second half of tools/gpu_r4_cli.sh alone)
OUT=gpurun_out/r04cli
mkdir -p $OUT
D=/tmp/cli_bench
mkdir -p $D
python - <<PY
import sys, os
sys.path.insert(0, "$PWD")
from rtl_433_amd import synth
for s in range(8192):
    f = "$D/s%05d_433.92M_250k.cu8" % s
    if not os.path.exists(f):
        synth.ook_stream(s)[0].tofile(f)
PY
cd $D
ARGS=$(ls s*_433.92M_250k.cu8 | head -8192 | sed 's/^/-r /' | tr '\n' ' ')
HIP=$GRAFT_REPO_ROOT/dropin/_build/rtl_433_hip
REF=$GRAFT_REPO_ROOT/oracle/_ref/rtl_433_ref
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 ))"; }
{
tr=$(t $REF $ARGS -F json:ref.json -M level -K FILE 2>/dev/null)
echo "8192 files: reference ${tr} ms, $(wc -l < ref.json) lines"
for mode in 1 0 1 0 1 1; do
  rm -f hip.json
  th=$(RTL433_HIP_OVERLAP=$([ $mode = 1 ] && echo "" || echo 0) t $HIP $ARGS -F json:hip.json -M level -K FILE 2>/dev/null)
  echo "8192 files: hip ${th} ms (two passes in flight: $([ $mode = 1 ] && echo on || echo off))  $(cmp -s ref.json hip.json && echo IDENTICAL || echo DIFFERENT)"
done
rm -f hip.json
th=$(RTL433_HIP_PREFILTER=1 t $HIP $ARGS -F json:hip.json -M level -K FILE 2>/dev/null)
echo "8192 files: hip ${th} ms (RTL433_HIP_PREFILTER=1: the decoders asked before the first pass)  $(cmp -s ref.json hip.json && echo IDENTICAL || echo DIFFERENT)"
} | tee $GRAFT_REPO_ROOT/$OUT/cli_bench_8192.txt
