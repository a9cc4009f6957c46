#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && timeout 300 bash tools/cli_trace.sh 8192 > /dev/null 2>&1 </dev/null
mkdir -p /dev/shm/clid && cp /tmp/cli_bench/s*_433.92M_250k.cu8 /dev/shm/clid/
HIP=$GRAFT_REPO_ROOT/dropin/_build/rtl_433_hip
one() { # dir, out
  cd $1; ARGS=$(ls s*_433.92M_250k.cu8 | head -8192 | sed 's/^/-r /' | tr '\n' ' ')
  s=$(date +%s%N); RTL433_HIP_TRACE=1 $HIP $ARGS $2 -M level -K FILE 2> /tmp/t.txt > /tmp/o.txt </dev/null; e=$(date +%s%N)
  echo "  wall $(( (e - s) / 1000000 )) ms | $(grep -E 'GPU opened' /tmp/t.txt | sed 's/hip flow: //' | cut -c1-12) open | $(grep -E 'exit handlers' /tmp/t.txt | sed 's/hip flow: exit handlers begin //' | cut -c1-9) inside"
}
for v in "/tmp/cli_bench -F json:/tmp/x.json" "/dev/shm/clid -F json:/tmp/x.json" "/dev/shm/clid -F json"; do
  set -- $v; echo "== shell, files in $1, output $2 $3"; for r in 1 2 3; do sleep 1.5; one $1 "$2 $3"; done
done
echo "== the same beside a python process that holds a HIP context (nothing allocated)"
python - <<'PY'
import subprocess, time, torch, os
x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
d = "/dev/shm/clid"; names = sorted(f for f in os.listdir(d) if f.endswith(".cu8"))[:8192]
args = [a for f in names for a in ("-r", f)] + ["-F", "json", "-M", "level", "-K", "FILE"]
hip = os.path.join(os.environ["GRAFT_REPO_ROOT"], "dropin/_build/rtl_433_hip")
for r in range(4):
    time.sleep(1.5); t0 = time.perf_counter(); p = subprocess.run([hip] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, RTL433_HIP_TRACE="1"))
    ms = (time.perf_counter() - t0) * 1e3
    err = p.stderr.decode(errors="replace")
    op = [l for l in err.splitlines() if "GPU opened" in l]; ex = [l for l in err.splitlines() if "exit handlers" in l]
    print(f"  wall {ms:.0f} ms | {op[0][10:24] if op else '?'} open | {ex[0][-40:] if ex else '?'}")
del x
PY
echo "== ... and from a python parent WITHOUT a HIP context"
python - <<'PY'
import subprocess, time, os
d = "/dev/shm/clid"; names = sorted(f for f in os.listdir(d) if f.endswith(".cu8"))[:8192]
args = [a for f in names for a in ("-r", f)] + ["-F", "json", "-M", "level", "-K", "FILE"]
hip = os.path.join(os.environ["GRAFT_REPO_ROOT"], "dropin/_build/rtl_433_hip")
for r in range(4):
    time.sleep(1.5); t0 = time.perf_counter(); p = subprocess.run([hip] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, RTL433_HIP_TRACE="1"))
    ms = (time.perf_counter() - t0) * 1e3
    err = p.stderr.decode(errors="replace")
    op = [l for l in err.splitlines() if "GPU opened" in l]; ex = [l for l in err.splitlines() if "exit handlers" in l]
    print(f"  wall {ms:.0f} ms | {op[0][10:24] if op else '?'} open | {ex[0][-40:] if ex else '?'}")
PY
rm -rf /dev/shm/clid
