export TMPDIR=/tmp
TAG=r05_b; OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python tools/jq.py $OUT/bench.json value ms_per_step roofline.frac roofline.achieved 2>/dev/null | head
echo "== rocprof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --quick --resident --exclusive 3 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt | head -30
find $OUT/prof -name '*.db' -size +20M -delete
echo "== kbench"
{ for f in 0 32768 4096; do timeout 300 python tools/kbench.py --nodevs --reps 7 --streams 8192 --debug $f 2>&1 | tail -1; done
  timeout 300 python tools/kbench.py --reps 4 --streams 8192 2>&1 | tail -1; } 2>&1 | grep -v amdgpu.ids | tee $OUT/kbench.txt
echo "== pmc issue"
R433_PMC_TAG=$TAG timeout 900 python tools/pmc_issue.py 2>&1 | tail -5
ls $OUT
