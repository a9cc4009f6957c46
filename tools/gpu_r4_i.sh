#!/bin/bash
# Round 4, visit i: waiting for the GPU asleep instead of spinning -- CPU budget and the bench line, A/B with R433_DEBUG_SPIN_WAIT
OUT=gpurun_out/r04i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python tools/spin_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/spin_probe.txt
echo "== cpu budget (waits asleep)"
timeout 200 python tools/cpu_budget.py 12 2>&1 | grep -v amdgpu.ids | tee $OUT/cpu_budget.txt
echo "== bench"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python tools/jq.py value ms_per_step breakdown_ms hbm_resident parity < $OUT/bench.json
echo "== bench, spinning waits"
timeout 600 python bench.py --quick --debug 524288 > $OUT/bench_spin.json 2> $OUT/bench_spin.err
python tools/jq.py value ms_per_step breakdown_ms hbm_resident < $OUT/bench_spin.json
echo "== kbench (the wait is in the timings of a leg, not of a kernel)"
timeout 300 python tools/kbench.py --reps 4 --streams 8192 2>&1 | tail -1
timeout 300 python tools/kbench.py --reps 4 --streams 8192 --debug 524288 2>&1 | tail -1
timeout 600 python -m pytest tests -m gpu -q -x -k "dispatch or prefilter or pipeline" 2>&1 | tail -2
