#!/bin/bash
# The drop-in CLI several times over the same list of N config-2 captures against one run of the stock binary: every run
# must give the same file (looks for replay races).  tools/cli_repeat.sh [N] [reps] [out dir]; extra environment is passed on.
N=${1:-8192}
REPS=${2:-6}
OUT=$(realpath -m ${3:-gpurun_out/cli_repeat})
D=/tmp/cli_bench
mkdir -p $D $OUT
python - <<PY
import sys, os
sys.path.insert(0, "$PWD")
from rtl_433_amd import synth
for s in range($N):
    f = "$D/s%05d_433.92M_250k.cu8" % s
    if not os.path.exists(f):
        synth.ook_stream(s)[0].tofile(f)
PY
REF=$PWD/oracle/_ref/rtl_433_ref
HIP=$PWD/dropin/_build/rtl_433_hip
cd $D
ARGS=$(ls s*_433.92M_250k.cu8 | head -$N | sed 's/^/-r /' | tr '\n' ' ')
X="-X n=pwm,m=OOK_PWM,s=300,l=600,r=5000,g=2000,t=150 -X n=ppm,m=OOK_PPM,s=300,l=600,r=5000,g=2000,t=150 -X n=mc,m=OOK_MC_ZEROBIT,s=300,l=300,r=5000"
rm -f ref.json; $REF $ARGS $X -F json:ref.json -M level -K FILE 2>/dev/null
for rep in $(seq $REPS); do
  rm -f hip.json; RTL433_HIP_TRACE=1 $HIP $ARGS $X -F json:hip.json -M level -K FILE 2> hip.err
  if cmp -s ref.json hip.json; then echo "rep $rep: IDENTICAL ($(wc -l < hip.json) lines)"; else
    echo "rep $rep: DIFFERENT ($(wc -l < ref.json) / $(wc -l < hip.json) lines)"
    diff ref.json hip.json | head -400 > $OUT/diff.$rep.txt
    cp hip.err $OUT/err.$rep.txt
  fi
done | tee $OUT/summary.txt
