export TMPDIR=/tmp
OUT=gpurun_out/r05_last; mkdir -p $OUT
timeout 170 python bench.py > $OUT/bench.json 2> $OUT/bench.err </dev/null
timeout 20 python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms']['k_wave_alone'], d['breakdown_ms']['host_dispatch'], d.get('parity'), {k: (v.get('value'), v.get('parity')) for k, v in (d.get('other_configs') or {}).items() if isinstance(v, dict)})" </dev/null
( cd /tmp && timeout 80 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --quick --exclusive 3 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err </dev/null )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && timeout 40 python tools/rocprof_summary.py $DB $OUT/kernel_stats.txt </dev/null | head -12 | cut -c1-150
find $OUT/prof -name '*.db' -size +20M -delete
