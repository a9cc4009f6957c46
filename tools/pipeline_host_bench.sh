#!/bin/bash
# The C pipeline host (dropin/_build/pipeline_host_hip) over N config-2 capture files: one engine against three, with and
# without the pre-filter.  tools/pipeline_host_bench.sh [N]
N=${1:-4096}
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
PH=$PWD/dropin/_build/pipeline_host_hip
cd $D
FILES=$(ls s*_433.92M_250k.cu8 | head -$N | tr '\n' ' ')
while read -r shape; do
  for rep in 1 2 3; do $PH -q $shape $FILES 2>&1 | tail -1 | sed "s/^/[$shape] /"; done
done <<SHAPES
-e 1 -b 1024
-e 3 -b 1024
-e 3 -b 1024 -p
SHAPES
