#!/bin/bash
# Times the drop-in CLI (dropin/_build/rtl_433_hip) against the stock reference CLI (oracle/_ref/rtl_433_ref) on a list
# of config-2 captures, all default decoders + three generic ones, JSON to a file; checks the outputs are identical.
#   tools/cli_bench.sh [N captures, default 1024] [out dir]
N=${1:-1024}
OUT=$(realpath -m ${2:-gpurun_out/cli})
D=/tmp/cli_bench
mkdir -p $D $OUT
python - <<PY
import sys
sys.path.insert(0, "$PWD")
from rtl_433_amd import synth
for s in range($N):
    synth.ook_stream(s)[0].tofile("$D/s%05d_433.92M_250k.cu8" % s)
PY
cd $D
ARGS=$(for f in s*_433.92M_250k.cu8; do echo -n "-r $f "; done)
X="-X n=pwm,m=OOK_PWM,s=300,l=600,r=5000,g=2000,t=150 -X n=ppm,m=OOK_PPM,s=300,l=600,r=5000,g=2000,t=150 -X n=mc,m=OOK_MC_ZEROBIT,s=300,l=300,r=5000"
REF=$GRAFT_REPO_ROOT/oracle/_ref/rtl_433_ref
HIP=$GRAFT_REPO_ROOT/dropin/_build/rtl_433_hip
[ -z "$GRAFT_REPO_ROOT" ] && REF=/root/repo/oracle/_ref/rtl_433_ref && HIP=/root/repo/dropin/_build/${CLI_BIN:-rtl_433_hip}
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) m"; }
rm -f ref.json hip.json
for rep in 1 2 3; do
  tr=$(t $REF $ARGS $X -F json:ref.json -M level -K FILE 2>/dev/null); mv ref.json ref.$rep.json
  th=$(t $HIP $ARGS $X -F json:hip.json -M level -K FILE 2>/dev/null); mv hip.json hip.$rep.json
  echo "rep $rep: reference ${tr}s  hip ${th}s  lines $(wc -l < ref.$rep.json) / $(wc -l < hip.$rep.json)  $(cmp -s ref.$rep.json hip.$rep.json && echo IDENTICAL || echo DIFFERENT)"
done | tee $OUT/cli_bench.txt
