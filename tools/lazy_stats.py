"""Development aid (GPU box): how many tiles of the bench captures go by unfiltered, how many captures run twice."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtl_433_amd import synth
from rtl_433_amd.engine import BatchEngine, flow_cfg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
host = synth.ook_batch(n, 65536, 250000, seed0=0)
eng = BatchEngine(flow_cfg(2, 250000), None, profiling=True)
eng.run(torch.from_numpy(host).cuda())
buf = np.zeros(n * 64, dtype=np.int32)
sz = eng.L.r433_batch_debug_state(eng.h, C.c_void_p(buf.ctypes.data), buf.nbytes)
st = buf[: n * sz // 4].reshape(n, sz // 4)
q, att = st[:, 2], st[:, 3]
print(f"{n} captures x 32 tiles: unfiltered tiles mean {q.mean():.2f}, min {q.min()}, max {q.max()}; captures run twice: {(att > 0).sum()}")
print("histogram of unfiltered tiles per capture:", np.bincount(q, minlength=33).tolist())
