#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel trace.  Everything lands in gpurun_out/<tag>/.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" 
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest.txt
echo "== bench"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; cat $OUT/bench.json
echo "== rocprof kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --quick --resident --exclusive 3 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt | head -40
find $OUT/prof -name '*.db' -size +20M -delete
ls -la $OUT
