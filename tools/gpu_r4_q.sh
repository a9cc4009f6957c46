#!/bin/bash
# Round 4, last visit: the build with the re-scheduled sizing pass -- the whole GPU suite, fuzz, the bench line, rocprofv3 kernel stats
OUT=gpurun_out/r04q
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/pytest.txt
timeout 400 python tools/fuzz_emu.py --gpu 1500 560000 2>&1 | tail -1 | tee $OUT/fuzz.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python tools/jq.py value ms_per_step breakdown_ms hbm_resident parity d2h_bytes_per_step_per_gpu < $OUT/bench.json | cut -c1-900
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --quick --resident --exclusive 3 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt | head -12 | cut -c1-150
find $OUT/prof -name '*.db' -delete
timeout 200 python tools/kbench.py --reps 6 --streams 8192 2>&1 | tail -1 | tee $OUT/kbench.txt
timeout 200 python tools/dispatch_trace.py 24 1 2>&1 | grep "packages," | cut -c1-400 | tee $OUT/dispatch_trace.txt
