"""Development aid (GPU box): the slicers' passes over one bench step (8192 captures, the 335 real decoders' timing rows) WITH
the pre-filter tables of the real decoders -- what the pipeline runs -- for one build of the library.
    python tools/slice_pf_bench.py [lib.so] [reps] [debug flags]"""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rtl_433_amd import _lib, plugins
from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table
so = (sys.argv[1] or None) if len(sys.argv) > 1 else None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
debug = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
cache = "/tmp/r433_spb_input.npy"
if os.path.exists(cache):
    host = np.load(cache)
else:
    host = bench.ook_batches(0, 8192, 32)
    np.save(cache, host)
devs = load_device_table()[0]
plug = plugins.Plugins()
eng = BatchEngine(flow_cfg(2, 250000), devs, profiling=True, library=_lib.bind(ctypes.CDLL(os.path.abspath(so))) if so else None)
eng.probe_prefilter(plug.devices, helper=plug.helper_probe())
if debug:
    eng.set_debug(debug)
d = torch.from_numpy(host).cuda()
ts = []
for rep in range(reps):
    n = eng.run(d)
    ts.append(eng.timing())
best = {k: min(t[k] for t in ts[2:] or ts) for k in ts[0]}
print(f"lib={os.path.basename(so) if so else 'default'} debug={debug} grid={os.environ.get('R433_SLICE_GRID', '-')} pkgs={n} records={eng.events()[1]} " + " ".join(f"{k}={v:.3f}" for k, v in best.items())
      + f" evt_digest={hashlib.sha1(bytes(eng.events()[0])).hexdigest()[:12]} packed={'off' if os.environ.get('R433_SLICE_NO_PACK') else 'on'}")
