"""Development aid (GPU box): the kernels of ONE build of the library over the bench's grid, with a digest of the records so that
builds (or R433_DEBUG_* switches of one build) can be compared.
    python tools/variant_bench.py <lib.so> [captures] [reps] [all decoders: 0 | 1] [debug flags]
The 1024 distinct bench captures are made once per box (/tmp/r433_vb_input.npy) and tiled to the grid size."""
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtl_433_amd import _lib, synth
from rtl_433_amd.engine import BatchEngine, flow_cfg

so, streams, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8192, int(sys.argv[3]) if len(sys.argv) > 3 else 6
with_devs = len(sys.argv) > 4 and int(sys.argv[4]) != 0
debug = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
cache = "/tmp/r433_vb_input.npy"
if os.path.exists(cache):
    host = np.load(cache)
else:
    host = synth.ook_batch(1024, 65536, 250000, seed0=0)
    np.save(cache, host)
host = np.tile(host, ((streams + 1023) // 1024, 1))[:streams]
d = torch.from_numpy(host).cuda()
from rtl_433_amd.engine import load_device_table
eng = BatchEngine(flow_cfg(2, 250000), load_device_table()[0] if with_devs else None, profiling=True,
                  library=_lib.bind(ctypes.CDLL(os.path.abspath(so))))
if debug:
    eng.set_debug(debug)
ts = []
for r in range(reps):
    n = eng.run(d)
    ts.append(eng.timing())
pk, _ = eng.packages()
ev, n_ev = eng.events()
best = {k: min(t[k] for t in ts[1:] or ts) for k in ts[0]}
print(f"debug={debug} pkgs={n} events={n_ev} " + " ".join(f"{k}={v:.3f}" for k, v in best.items())
      + f" pkg_digest={hashlib.sha1(bytes(pk)).hexdigest()[:12]} evt_digest={hashlib.sha1(bytes(ev)).hexdigest()[:12]}")
