"""One long capture (config 3 / 5 style single stream): unsplit vs split across wavefronts.
    python tools/longbench.py [--msamples 64] [--split 131072]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtl_433_amd import synth
from rtl_433_amd.engine import BatchEngine, flow_cfg

ap = argparse.ArgumentParser()
ap.add_argument("--msamples", type=int, default=64)
ap.add_argument("--split", type=int, default=131072)
ap.add_argument("--sigma", type=float, default=2.0)
ap.add_argument("--cs16", action="store_true", help="config 3: 1024 kS/s cs16, FSK Manchester bursts every ~31 ms, minmax detector")
a = ap.parse_args()
n = a.msamples << 20
rng = np.random.default_rng(1)
# sparse OOK bursts (about one per 100 ms) over a noise floor, 250 kS/s
one = []
for s in range(64):
    one.append(synth.ook_stream(1000 + s, 65536)[0])
    one.append(synth.noise_cu8(2000 + s, 65536 * 3, a.sigma))
base = np.concatenate(one)
cfg = flow_cfg(2, 250000)
if a.cs16:
    # 112 Manchester bits x 2 x 51 samples = 11.4 k samples of burst, then ~20 k samples (20 ms) of noise
    base = np.concatenate([synth.fsk_stream_cs16(3000 + s, 32768, n_bursts=1, lead_in=1000) for s in range(32)])
    cfg = flow_cfg(4, 1024000, fpdm=1, center_frequency=868000000)
reps = (2 * n + base.size - 1) // base.size
iq = np.tile(base, reps)[: 2 * n].copy()
d = torch.from_numpy(iq.reshape(1, -1)).cuda()
res = {}
for split in (0, 1, a.split):
    eng = BatchEngine(cfg, None, profiling=True)
    eng.set_split(split)
    best = None
    for r in range(1 if (a.cs16 and split == 0) else 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        npk = eng.run(d)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res[split] = (npk, eng.packages()[0], best, eng.split_stats())
    eng.close()
    print(f"split={split:7d}  {n/1e6:.0f} Msamples in {best*1e3:8.1f} ms = {n/best/1e6:9.1f} MS/s  packages={npk} stats={res[split][3]}", flush=True)
print("identical packages:", res[0][1] == res[a.split][1])
