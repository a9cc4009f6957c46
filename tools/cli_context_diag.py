"""Why the drop-in CLI takes 590 ms under bench.py and 380 ms under tools/cli_bench_files.py over the same kind of files:
the same legs behind different parents.  python tools/cli_context_diag.py"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

def throttled():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat")) if k in ("nr_throttled", "throttled_usec", "usage_usec")}
    except Exception as e:
        return {"err": str(e)}

def legs(tag, host, n=4, settle=1.5, pipe=True, where="/dev/shm"):
    d = os.path.join(where, "r433_diag")
    os.makedirs(d, exist_ok=True)
    names = []
    for k in range(host.shape[0]):
        names.append(f"s{k:05d}_433.92M_250k.cu8")
        p = os.path.join(d, names[-1])
        if not os.path.exists(p):
            host[k].tofile(p)
    args = [a for f in names for a in ("-r", f)] + ["-F", "json", "-M", "level", "-K", "FILE"]
    cli = os.path.join(bench.ROOT, "dropin", "_build", "rtl_433_hip")
    walls = []
    for r in range(n):
        time.sleep(settle)
        t0 = throttled(); c0 = os.times()
        s = time.perf_counter()
        if pipe:
            p = subprocess.run([cli] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        else:
            with open("/tmp/diag_out.json", "wb") as f:
                p = subprocess.run([cli] + args, cwd=d, stdout=f, stderr=subprocess.DEVNULL)
        walls.append(round((time.perf_counter() - s) * 1e3))
        t1 = throttled(); c1 = os.times()
    print(f"{tag}: {walls} ms | last run: throttled +{t1.get('nr_throttled', 0) - t0.get('nr_throttled', 0)} periods, "
          f"+{(t1.get('throttled_usec', 0) - t0.get('throttled_usec', 0)) / 1e3:.0f} ms; cgroup cpu +{(t1.get('usage_usec', 0) - t0.get('usage_usec', 0)) / 1e3:.0f} ms; "
          f"parent cpu +{(c1.user + c1.system - c0.user - c0.system) * 1e3:.0f} ms, children +{(c1.children_user + c1.children_system - c0.children_user - c0.children_system) * 1e3:.0f} ms", flush=True)

print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?", flush=True)
a = bench.ook_batches(0, 8192, 8)
legs("A. one batch made by 8 processes", a)
legs("A'. the same, output to a file", a, pipe=False)
procs = max(1, min(32, os.cpu_count() or 1))
b = [bench.ook_batches(k * 8192, 8192, procs) for k in (1, 2)]
legs(f"B. after two more batches made by {procs} processes (3 GiB held)", a)
last = b[-1]
import shutil; shutil.rmtree("/dev/shm/r433_diag")
legs("C. the files of another batch (seeds 16384..)", last)
import torch
torch.cuda.set_device(0)
legs("D. the same with torch.cuda.set_device(0) called in the parent", last)
x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
legs("E. ... and a HIP context made", last)
shutil.rmtree("/dev/shm/r433_diag")
