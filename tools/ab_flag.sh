#!/bin/bash
# A/B of a bench.py flag on the resident pipeline: tools/ab_flag.sh "--exclusive 1" "--exclusive 2" [steps]   (alternating, three runs each)
export TMPDIR=/tmp
A=$1; B=$2; S=${3:-40}
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    timeout 300 python bench.py --quick --resident --steps $S --warmup 3 $v 2>/dev/null > /tmp/ab.json
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "replay", d["breakdown_ms"]["host_replay_call"], "gpu leg", d["breakdown_ms"]["gpu_leg_overlapped"], "k_wave", d["breakdown_ms"]["k_wave_timed_region"], "cpu/step", d["host_cpu"]["cpu_ms_per_step_this_rank"])
PY
  done
done
