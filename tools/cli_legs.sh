D=/tmp/cli_bench
cd $D
ARGS=$(ls s*_433.92M_250k.cu8 | head -8192 | sed 's/^/-r /' | tr '\n' ' ')
s=$(date +%s%N)
R433_TRACE_LEGS=1 RTL433_HIP_TRACE=1 /root/repo/dropin/_build/rtl_433_hip $ARGS -F json:hip.json -M level -K FILE 2> trace.legs.txt
e=$(date +%s%N)
echo "total $(( (e - s) / 1000000 )) ms"
cat trace.legs.txt
