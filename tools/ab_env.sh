#!/bin/bash
# A/B of an environment variable on the resident pipeline: tools/ab_env.sh NAME VALUE_A VALUE_B [steps]   (alternating, four runs each)
export TMPDIR=/tmp
N=$1; A=$2; B=$3; S=${4:-40}
for rep in 1 2 3 4; do
  for v in "$A" "$B"; do
    env $N=$v timeout 300 python bench.py --quick --resident --steps $S --warmup 3 2>/dev/null > /tmp/ab.json
    python - "$N=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "replay", d["breakdown_ms"]["host_replay_call"], "cpu/step", d["host_cpu"]["cpu_ms_per_step_this_rank"], "throttled", d["host_cpu"]["throttled_ms_per_step"])
PY
  done
done
