#!/bin/bash
# Round 4, visit h: the exhaustive probe of tiny rows (pre-filter) -- parity, then what it does to the host leg.
OUT=gpurun_out/r04h
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu (pre-filter, dispatch, drop-in)"
( time timeout 900 python -m pytest tests/test_prefilter.py tests/test_dispatch.py tests/test_dropin.py tests/test_pipeline_host.py -m gpu -q 2>&1 | tail -6 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/pytest.txt
echo "== bench (default command)"
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
python tools/jq.py value ms_per_step roofline breakdown_ms hbm_resident parity d2h_bytes_per_step_per_gpu host_replay_call < $OUT/bench.json
echo "== timeline"
timeout 300 python tools/leg_timeline.py 12 3 2 24 1 2>&1 | grep -v amdgpu.ids | tee $OUT/leg_timeline.txt | head -12
echo "== dispatch trace"
timeout 200 python tools/dispatch_trace.py 24 1 2>&1 | grep "r.dispatch\|replay" | cut -c1-400 | tee $OUT/dispatch_trace.txt
echo "== fuzz (GPU, 1000 cases)"
timeout 600 python tools/fuzz_emu.py --gpu 300 500000 2>&1 | tail -1 | tee $OUT/fuzz.txt
