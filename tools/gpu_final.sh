#!/bin/bash
# The closing GPU visit of a round: everything profiles/<round>_* quotes, from one build, every step under its own timeout.
#   tools/gpu_final.sh <tag>        (round 6: ~15 GPU minutes)
TAG=${1:-r06}; OUT=gpurun_out/${TAG}_final; mkdir -p $OUT profiles
export TMPDIR=/tmp
timeout 200 python tools/kbench.py --bench-batch --make-batch-only --streams 8192 </dev/null >/dev/null 2>&1  # (the counter passes must not fork: NOTES.md)
if [ -z "$SKIP_PMC" ]; then
echo "== SQ counters of the detection pass on the bench's own batch (producers / consumers apart)"
R433_PMC_TAG=${TAG}_pmc timeout 500 python tools/pmc_issue.py </dev/null 2>&1 | tail -60 > $OUT/pmc_issue.txt; tail -12 $OUT/pmc_issue.txt
[ -s gpurun_out/${TAG}_pmc/issue.json ] && cp gpurun_out/${TAG}_pmc/issue.json profiles/${TAG}_pmc_issue.json
echo "== ... of the decoder fan-out"
R433_PMC_TAG=${TAG}_pmc R433_PMC_WHAT=slice timeout 500 python tools/pmc_issue.py </dev/null 2>&1 | tail -80 > $OUT/pmc_slice.txt; grep -A12 '"summary"' $OUT/pmc_slice.txt | head -14
[ -s gpurun_out/${TAG}_pmc/slice.json ] && cp gpurun_out/${TAG}_pmc/slice.json profiles/${TAG}_pmc_slice.json
echo "== HBM traffic of the detection pass (FETCH_SIZE / WRITE_SIZE passes)"
R433_PMC_TAG=${TAG}_pmc timeout 400 python tools/pmc_traffic.py config4 </dev/null 2>&1 | tail -3 | cut -c1-600
[ -s gpurun_out/${TAG}_pmc/traffic.json ] && cp gpurun_out/${TAG}_pmc/traffic.json profiles/${TAG}_pmc_traffic.json
fi
echo "== bench (the driver's form)"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err </dev/null; tail -2 $OUT/bench.err
timeout 20 python -c "
import json,sys
d=json.load(open('$OUT/bench.json')); print({k: d.get(k) for k in ('value','ms_per_step','breakdown_ms','parity','bitbuffers_to_host_per_step','d2h_bytes_per_step_per_gpu')}); print(d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), d.get('pcie_inclusive'), d.get('cpu_baseline')); print({k: (v.get('value'), v.get('roofline',{}).get('frac'), v.get('parity')) for k, v in (d.get('other_configs') or {}).items() if isinstance(v, dict)}); print(json.dumps(d.get('dropin'))[:900])" </dev/null
echo "== rocprofv3 kernel trace of the resident pipeline"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --quick --resident --exclusive 3 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err </dev/null )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && timeout 60 python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt </dev/null | head -30
find $OUT/prof -name '*.db' -size +20M -delete
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q </dev/null 2>&1 | tail -6 | tee $OUT/pytest.txt
echo "== the single-stream workloads kernel by kernel"
timeout 600 python tools/stream_phases.py 3 5 </dev/null 2>/dev/null > $OUT/stream_phases.txt; grep -E "^==|^-- pass 1" $OUT/stream_phases.txt | cut -c1-200
echo "== kernel alone"
{ timeout 100 python tools/kbench.py --nodevs --reps 7 --streams 8192 --bench-batch </dev/null 2>&1 | tail -1
  timeout 100 python tools/kbench.py --nodevs --reps 7 --streams 8192 </dev/null 2>&1 | tail -1
  timeout 100 python tools/kbench.py --nodevs --reps 7 --streams 8192 --debug 8388608 </dev/null 2>&1 | tail -1
  timeout 100 python tools/slice_pf_bench.py </dev/null 2>&1 | tail -1; } | grep -v amdgpu.ids | tee $OUT/kbench.txt
ls $OUT
