#!/bin/bash
# PMC counter pass over tools/kbench.py (own run: counters only, no tracing domains besides kernel-trace).
# usage: tools/pmc_run.sh <tag> "<counter list>" [kbench args...]
TAG=$1; shift
PMC=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --pmc $PMC -d $OUT/pmc -o r -- python $GRAFT_REPO_ROOT/tools/kbench.py --reps 2 "$@" > $OUT/kbench.txt 2> $OUT/pmc.err
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/pmc -name '*.db' | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name"):
    if 'k_wave' in r[0] or 'k_slice' in r[0]:
        print(f"{r[0][:50]:50s} {r[1]:24s} n={r[2]} mean={r[3]:.0f}")
PY
find $OUT -name '*.db' -size +8M -delete
