#!/bin/bash
# round 4: parity suite with the heavier tests, counter passes of this build, the default bench line
OUT=gpurun_out/r4g
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16 ) 2>&1 | tee $OUT/pytest.txt
echo "== pmc issue"
R433_PMC_TAG=r4g timeout 900 python tools/pmc_issue.py 2>&1 | tail -30 | tee $OUT/pmc_issue_stdout.txt
echo "== pmc traffic (config4 = the bench grid)"
R433_PMC_TAG=r4g timeout 900 python tools/pmc_traffic.py config4 config3 config5 2>&1 | tail -5 | tee $OUT/pmc_traffic_stdout.txt
echo "== default bench"
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
python tools/jq.py value ms_per_step roofline breakdown_ms hbm_resident parity cpu_baseline d2h_bytes_per_step_per_gpu < $OUT/bench.json
ls $OUT
