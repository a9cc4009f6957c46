"""Development aid (GPU box): the ordered replay of one bench step (8192 captures) into the real decoders with the
library's trace (R433_DEBUG_DISPATCH_TRACE): index / per-level / commit times, the slowest decoders.
    python tools/dispatch_trace.py [threads] [stateless flags: 0 | 1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rtl_433_amd import plugins
from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
stateless = len(sys.argv) > 2 and int(sys.argv[2]) != 0
host = bench.ook_batches(0, 8192, 32)
devs = load_device_table()[0]
plug = plugins.Plugins()
eng = BatchEngine(flow_cfg(2, 250000), devs, profiling=True)
eng.probe_prefilter(plug.devices, helper=plug.helper_probe())
if stateless:
    eng.set_stateless(plug.stateless())
d = torch.from_numpy(host).cuda()
for rep in range(3):
    if rep == 2:
        eng.set_debug(32)
    n = eng.run(d)
    import time
    t0 = time.perf_counter()
    ev = eng.dispatch_ordered(plug.devices, plug.hooks() if hasattr(plug, 'hooks') else None, threads)
    dt = time.perf_counter() - t0
    text, n_msg = plug.take()
print(f"{n} packages, {ev} decoded events, {n_msg} messages, replay {dt * 1e3:.2f} ms on {threads} threads; records {len(eng.events()[0])} B + packages {len(eng.packages()[0])} B; timing {eng.timing()}")
