"""Does the order of the captures in a grid matter?  The bench batch (8192 captures) as it comes, heaviest captures first,
lightest first -- weight = pulses the detector found in the capture on a first run.  Kernel time by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import pyoracle as po
from rtl_433_amd.engine import BatchEngine, flow_cfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
host = bench.ook_batches(0, n, 32)
eng = BatchEngine(flow_cfg(2, 250000), None, profiling=True)


def run(arr, reps=5):
    d = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
    best = 1e9
    for _ in range(reps):
        eng.run(d)
        best = min(best, eng.timing()["detect_ms"])
    return best


t_plain = run(host)
pk = po.parse_packages(eng.packages()[0])
w = np.zeros(n)
for p in pk:
    w[p["stream"]] += p["num"]
order = np.argsort(-w, kind="stable")
print(f"{n} captures: as they come {t_plain:.3f} ms; heaviest first {run(host[order]):.3f} ms; lightest first {run(host[order[::-1]]):.3f} ms; "
      f"pulses per capture min {w.min():.0f} mean {w.mean():.0f} max {w.max():.0f}")
# the floor: every capture alone in the grid would take its own time; sum over slots
