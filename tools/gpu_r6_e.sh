#!/bin/bash
# round 6, visit e: cut planning after the O(n) planner, counters of HEAD (issue / slicers / traffic) on the bench's batch, the CLI's timeline and run-to-run times
TAG=${1:-r06_e}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/stream_phases.py 3 5 </dev/null 2>/dev/null | grep -E "^==|^-- pass 1|k_wave|k_tile_max" | head -12 | tee $OUT/stream_phases.txt | cut -c1-200
echo "== SQ counters of the detection pass on the bench's batch"
R433_PMC_TAG=r06_pmc timeout 900 python tools/pmc_issue.py </dev/null 2>&1 | tail -70 > $OUT/pmc_issue.txt; grep -E "simd_ipc|waves_per_simd|duration_ms|valu_only" $OUT/pmc_issue.txt | head -12
echo "== ... of the slicers"
R433_PMC_TAG=r06_pmc R433_PMC_WHAT=slice timeout 900 python tools/pmc_issue.py </dev/null 2>&1 | tail -80 > $OUT/pmc_slice.txt; grep -A12 '"summary"' $OUT/pmc_slice.txt | head -16
echo "== HBM traffic of the detection pass"
R433_PMC_TAG=r06_pmc timeout 600 python tools/pmc_traffic.py config4 </dev/null 2>&1 | tail -2 | cut -c1-700
ls gpurun_out/r06_pmc
echo "== the CLI: timeline of a run, then eight runs in a row"
timeout 600 bash tools/cli_trace.sh 8192 > $OUT/cli_trace.txt 2>&1 </dev/null; grep -E "^rep|engine created|captures queued|exit handlers|GPU pass|opened|warm" $OUT/cli_trace.txt | head -60 | cut -c1-180
cd /tmp/cli_bench && ARGS=$(ls s*_433.92M_250k.cu8 | head -8192 | sed 's/^/-r /' | tr '\n' ' ')
for rep in 1 2 3 4 5 6 7 8; do s=$(date +%s%N); $GRAFT_REPO_ROOT/dropin/_build/rtl_433_hip $ARGS -F json:/tmp/cli_bench/hip.json -M level -K FILE 2>/dev/null </dev/null; e=$(date +%s%N); echo "run $rep: $(( (e - s) / 1000000 )) ms"; done | tee $GRAFT_REPO_ROOT/$OUT/cli_runs.txt
