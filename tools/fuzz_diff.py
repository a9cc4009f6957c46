"""Development aid (GPU box): which package record and field of one fuzz case differ between the GPU and the oracle.
    python tools/fuzz_diff.py <seed>"""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import fuzz_emu as F
from oracle import pyoracle as po

seed = int(sys.argv[1])
got = {}
real_oracle = po.oracle_flow
def gpu(*a, **kw):
    g = F.gpu_run(*a, **kw)
    got["g"] = g
    got["args"] = {k: v for k, v in kw.items() if k != "devs"}
    got["shape"] = [(x.dtype.str, x.nbytes) for x in a[0]], a[1], a[2]
    return g
orc = []
def oracle(*a, **kw):
    o = real_oracle(*a, **kw)
    orc.append(o["packages"])
    return o
po.oracle_flow = oracle
F.BIG = True
print("result:", F.one_case(seed, gpu))
print("captures (dtype, bytes), sample size, rate:", got["shape"]); print("flow options:", got["args"])
a, b = bytes(got["g"]["packages"][0]), b"".join(orc)
names = "total_bytes stream type num_pulses frame ret_pos offset_lo offset_hi start_ago end_ago ook_low ook_high fsk_f1 fsk_f2 sample_rate reserved".split()
pa = pb = 0
k = 0
while pa < len(a) and pb < len(b):
    ha, hb = struct.unpack_from("<16I", a, pa), struct.unpack_from("<16I", b, pb)
    ra, rb = a[pa:pa + ha[0]], b[pb:pb + hb[0]]
    if ra != rb:
        print(f"package {k}: header differences (gpu / oracle):", {n: (x, y) for n, x, y in zip(names, ha, hb) if x != y})
        na, nb = ha[3], hb[3]
        qa = np.frombuffer(ra[64:64 + 8 * na], dtype=np.int32).reshape(-1, 2)
        qb = np.frombuffer(rb[64:64 + 8 * nb], dtype=np.int32).reshape(-1, 2)
        m = min(len(qa), len(qb))
        d = np.nonzero((qa[:m] != qb[:m]).any(axis=1))[0]
        print(f"  pulses {na} / {nb}; first differing (pulse, gap) pairs:", [(int(i), qa[i].tolist(), qb[i].tolist()) for i in d[:6]], "of", len(d))
        print("  type", ha[2], "offset", ha[6], "frame", ha[4])
    pa += ha[0]; pb += hb[0]; k += 1
print("packages compared:", k, "bytes", len(a), len(b))
