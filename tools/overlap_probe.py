"""Does a pass slow down while a pinned-host -> HBM copy runs on another stream, and do the HIP events of the library see it?
One engine, 8192 captures resident; wall clock around the pass (ground truth) against the library's event times, alone and with a
1 GiB H2D copy in flight."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table

devs = load_device_table()[0]
host = bench.ook_batches(0, 8192, 32)
d = torch.from_numpy(host).cuda()
pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
dst = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
eng = BatchEngine(flow_cfg(2, 250000), devs, profiling=True)
st, cp = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3):
    eng.run(d, stream=st.cuda_stream)
for mode in ("alone", "with a 1 GiB H2D copy in flight", "alone", "with a 1 GiB H2D copy in flight"):
    torch.cuda.synchronize()
    if mode != "alone":
        with torch.cuda.stream(cp):
            dst.copy_(pin, non_blocking=True)
            dst.copy_(pin, non_blocking=True)
    t0 = time.perf_counter()
    eng.run(d, stream=st.cuda_stream)
    wall = (time.perf_counter() - t0) * 1e3
    tm = eng.timing()
    torch.cuda.synchronize()
    print(f"{mode}: wall {wall:.2f} ms; events: detect {tm['detect_ms']:.2f} count {tm['count_ms']:.2f} write {tm['write_ms']:.2f} d2h {tm['d2h_ms']:.2f} total {tm['total_ms']:.2f}")
