#!/bin/bash
# Round 4, visit j: the slicers with a staging slot per device (one stretch for 8192 packages): kernel by kernel, slot sizes
OUT=gpurun_out/r04j
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python $GRAFT_REPO_ROOT/tools/kbench.py --reps 4 --streams 8192 > $GRAFT_REPO_ROOT/$OUT/kbench_prof.txt 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt | head -14 | cut -c1-150
find $OUT/prof -name '*.db' -size +20M -delete
for cap in 8192 4096 2048; do echo "== R433_STAGE_CAP=$cap"; R433_STAGE_CAP=$cap timeout 300 python tools/kbench.py --reps 5 --streams 8192 2>&1 | tail -1; done | tee $OUT/slot_sizes.txt
