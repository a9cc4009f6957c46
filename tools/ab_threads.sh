#!/bin/bash
# A/B: replay threads of the resident pipeline under the container's CPU quota: tools/ab_threads.sh "16 20 24 28" [reps] [steps]
export TMPDIR=/tmp
T=${1:-"16 20 24 28"}; R=${2:-3}; S=${3:-60}
for rep in $(seq $R); do
  for t in $T; do
    timeout 300 python bench.py --quick --resident --threads $t --steps $S --warmup 3 2>/dev/null > /tmp/ab.json
    python - "$t" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("threads", sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "replay", d["breakdown_ms"]["host_replay_call"], "cpu/step", d["host_cpu"]["cpu_ms_per_step_this_rank"], "throttled", d["host_cpu"]["throttled_ms_per_step"])
PY
  done
done
