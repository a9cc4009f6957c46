#!/bin/bash
# A/B: how the engines of the resident pipeline share the GPU (bench.py --exclusive 0 / 1 / 2), interleaved runs: tools/ab_exclusive.sh [reps] [steps]
export TMPDIR=/tmp
R=${1:-5}; S=${2:-60}
for rep in $(seq $R); do
  for x in 0 1 2; do
    timeout 300 python bench.py --quick --resident --exclusive $x --steps $S --warmup 3 2>/dev/null > /tmp/ab.json
    python - "$x" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("exclusive", sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "replay", d["breakdown_ms"]["host_replay_call"], "gpu leg", d["breakdown_ms"]["gpu_leg_overlapped"], "k_wave", d["breakdown_ms"]["k_wave_timed_region"], "cpu/step", d["host_cpu"]["cpu_ms_per_step_this_rank"])
PY
  done
done
