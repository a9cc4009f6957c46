"""Kernel-only timing of the detection kernel on the bench workload (HIP events inside the library).
    python tools/kbench.py [--streams N] [--samples M] [--reps R]
Honors R433_DEBUG_FLAGS (phase-skip experiments)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtl_433_amd import synth
from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=1024)
ap.add_argument("--samples", type=int, default=65536)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--nodevs", action="store_true")
a = ap.parse_args()
host = synth.ook_batch(a.streams, a.samples, 250000, seed0=0)
d = torch.from_numpy(host).cuda()
devs = None if a.nodevs else load_device_table()[0]
eng = BatchEngine(flow_cfg(2, 250000), devs, profiling=True)
ts = []
for r in range(a.reps):
    n = eng.run(d)
    ts.append(eng.timing())
best = min(ts, key=lambda t: t["detect_ms"])
print(f"flags={os.environ.get('R433_DEBUG_FLAGS','0')} streams={a.streams} samples={a.samples} pkgs={n} " +
      " ".join(f"{k}={v:.3f}" for k, v in best.items()))
