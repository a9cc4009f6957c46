#!/bin/bash
# A/B: the GPU legs' waits spinning (hipEventSynchronize) against polled with naps (R433_DEBUG_NAP_WAIT) in the resident pipeline,
# now that the host's CPU quota bounds the step: tools/ab_waits.sh [steps] [reps]
export TMPDIR=/tmp
S=${1:-60}; R=${2:-5}
for rep in $(seq $R); do
 for dbg in 0 2097152; do
  timeout 300 python bench.py --quick --resident --debug $dbg --steps $S --warmup 3 2>/dev/null > /tmp/ab.json
  python - "$dbg" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("debug", sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "replay", d["breakdown_ms"]["host_replay_call"], "gpu leg", d["breakdown_ms"]["gpu_leg_overlapped"],
      "cpu/step", d["host_cpu"]["cpu_ms_per_step_this_rank"], "throttled", d["host_cpu"]["throttled_ms_per_step"])
PY
 done
done
