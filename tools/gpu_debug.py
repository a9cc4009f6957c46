"""Debug aid (GPU box): run one case on the GPU and report the first record that differs from the oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from oracle import pyoracle as po
from rtl_433_amd import _lib
from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table
from tests.cases import make_case, fpdm_for

name = sys.argv[1] if len(sys.argv) > 1 else "kat"
devs, nums, names = load_device_table()
iq, ss, rate, freq = make_case(name)
lens = np.array([iq.nbytes], dtype=np.uint32)
stride = max(16, (iq.nbytes + 15) // 16 * 16)
host = np.zeros((1, stride), dtype=np.uint8); host[0, :iq.nbytes] = iq.view(np.uint8)
eng = BatchEngine(flow_cfg(ss, rate, fpdm=fpdm_for(freq), center_frequency=freq), devs)
npk = eng.run(torch.from_numpy(host).cuda(), lens)
pk, _ = eng.packages()
o = po.oracle_flow(iq, devs, po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq)))
print("packages gpu", npk, "oracle", o["n_packages"], "equal", pk == o["packages"])
if pk != o["packages"]:
    gp, op = po.parse_packages(pk), po.parse_packages(o["packages"])
    for i, (a, b) in enumerate(zip(gp, op)):
        for k in a:
            same = np.array_equal(a[k], b[k]) if isinstance(a[k], np.ndarray) else a[k] == b[k]
            if not same:
                print("pkg", i, k, a[k] if not isinstance(a[k], np.ndarray) else a[k][:20], b[k] if not isinstance(b[k], np.ndarray) else b[k][:20])
# raw events from the pinned buffer, without the validating walk
p, n, c = C.c_void_p(), C.c_size_t(), C.c_uint32()
L = _lib.lib()
rc = L.r433_batch_events(eng.h, C.byref(p), C.byref(n), C.byref(c))
print("events rc", rc, _lib.last_error() if rc < 0 else "", "len", n.value, "oracle len", len(o["events"]))
raw = C.string_at(p, n.value) if n.value else b""
oe = o["events"]
# walk oracle records, compare region by region
at = 0; idx = 0
while at < len(oe):
    total = int.from_bytes(oe[at:at+4], "little")
    if raw[at:at+total] != oe[at:at+total]:
        dev = int.from_bytes(oe[at+8:at+10], "little"); ordn = int.from_bytes(oe[at+10:at+12], "little")
        print("first differing event #", idx, "at byte", at, "dev", dev, names[dev], "mod", devs[dev]["modulation"], "ordinal", ordn, "size", total)
        print(" oracle:", oe[at:at+min(total,96)].hex())
        print(" gpu   :", raw[at:at+min(total,96)].hex())
        break
    at += total; idx += 1
else:
    print("all", idx, "events identical")
bad = {}
at = 0; idx = 0
while at < len(oe):
    total = int.from_bytes(oe[at:at+4], "little")
    dev = int.from_bytes(oe[at+8:at+10], "little")
    if raw[at:at+total] != oe[at:at+total]:
        bad.setdefault((dev, int(devs[dev]["modulation"])), []).append((idx, at, total, raw[at:at+16] == oe[at:at+16]))
    at += total; idx += 1
print("mismatching devices:", len(bad), "of", len(set(int.from_bytes(oe[a+8:a+10], "little") for a in [0])))
for k, v in sorted(bad.items())[:40]:
    print(k, v[:3])
