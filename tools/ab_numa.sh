#!/bin/bash
# A/B: the bench's process on the CPUs of NUMA node 0 / node 1 / anywhere (the resident pipeline): tools/ab_numa.sh [steps]
export TMPDIR=/tmp
S=${1:-40}
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null || echo $N0)
for rep in 1 2 3; do
  for v in "anywhere" "node0:$N0" "node1:$N1"; do
    cpus=${v#*:}; name=${v%%:*}
    if [ "$name" = anywhere ]; then pre=""; else pre="taskset -c $cpus"; fi
    $pre timeout 300 python bench.py --quick --resident --steps $S --warmup 3 2>/dev/null > /tmp/ab.json
    python - "$name" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "replay", d["breakdown_ms"]["host_replay_call"], "gpu leg", d["breakdown_ms"]["gpu_leg_overlapped"], "cpu/step", d["host_cpu"]["cpu_ms_per_step_this_rank"], "throttled", d["host_cpu"]["throttled_ms_per_step"])
PY
  done
done
