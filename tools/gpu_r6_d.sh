#!/bin/bash
# round 6, visit d: the single-stream workloads kernel by kernel (measure before building), counters of HEAD on the bench's batch
TAG=${1:-r06_d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/stream_phases.py 3 5 </dev/null 2>/dev/null | tee $OUT/stream_phases.txt | cut -c1-200
echo "== SQ counters of the detection pass on the bench's batch"
R433_PMC_TAG=r06_pmc timeout 500 python tools/pmc_issue.py </dev/null 2>&1 | tail -70 > $OUT/pmc_issue.txt; grep -E "simd_ipc|waves_per_simd|duration_ms|valu_only" $OUT/pmc_issue.txt | head -12
echo "== ... of the slicers"
R433_PMC_TAG=r06_pmc R433_PMC_WHAT=slice timeout 500 python tools/pmc_issue.py </dev/null 2>&1 | tail -80 > $OUT/pmc_slice.txt; grep -A12 '"summary"' $OUT/pmc_slice.txt | head -16
echo "== HBM traffic of the detection pass"
R433_PMC_TAG=r06_pmc timeout 400 python tools/pmc_traffic.py config4 </dev/null 2>&1 | tail -2 | cut -c1-700
ls gpurun_out/r06_pmc
