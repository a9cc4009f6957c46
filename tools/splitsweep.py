"""Detection time of the config-3 / config-5 streams (bench.py's generators) against the segment length of the split (GPU).
    python tools/splitsweep.py [3|5] [n_samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rtl_433_amd.engine import BatchEngine, flow_cfg

which = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else ((64 << 20) if which == 3 else (256 << 20))
host = bench.fsk_stream_config3(n) if which == 3 else bench.mixed_stream_config5(n)
d = torch.from_numpy(host.view(np.uint8)).cuda().reshape(1, -1)
ref = None
for split in (1, 131072, 65536, 32768, 24576, 16384, 8192):  # 1 = R433_SPLIT_AUTO
    cfg = flow_cfg(4, 1024000, fpdm=1, center_frequency=868000000) if which == 3 else flow_cfg(2, 2000000, fpdm=0, auto_level=1.0, fm_low_pass=0.15)
    eng = BatchEngine(cfg, None, profiling=True)
    eng.set_split(split)
    best = None
    for r in range(3):
        npk = eng.run(d)
        t = eng.timing()["detect_ms"]
        best = t if best is None else min(best, t)
    pk = eng.packages()[0]
    if ref is None:
        ref = pk
    print(f"split {split:6d}: detect {best:7.3f} ms  packages {npk}  {eng.split_stats()}  {'same records' if pk == ref else 'RECORDS DIFFER'}", flush=True)
    eng.close()
