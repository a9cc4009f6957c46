#!/bin/bash
# Where the drop-in CLI spends its time on a list of N config-2 captures (RTL433_HIP_TRACE=1): tools/cli_trace.sh [N]
N=${1:-8192}
D=/tmp/cli_bench
mkdir -p $D
python - <<PY
import sys, os
sys.path.insert(0, "$PWD")
from rtl_433_amd import synth
for s in range($N):
    f = "$D/s%05d_433.92M_250k.cu8" % s
    if not os.path.exists(f):
        synth.ook_stream(s)[0].tofile(f)
PY
HIP=$PWD/dropin/_build/rtl_433_hip
cd $D
ARGS=$(ls s*_433.92M_250k.cu8 | head -$N | sed 's/^/-r /' | tr '\n' ' ')
for rep in 1 2; do
  s=$(date +%s%N)
  rm -f hip.json; RTL433_HIP_TRACE=1 $HIP $ARGS -F json:hip.json -M level -K FILE 2> trace.$rep.txt; rm -f hip.json
  e=$(date +%s%N)
  echo "rep $rep: $(( (e - s) / 1000000 )) ms"
  grep "hip flow" trace.$rep.txt
done
