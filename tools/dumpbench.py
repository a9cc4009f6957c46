"""HBM rate of the -w dump-format kernels (r433_dump_convert): algorithmic bytes (input read once + output written
once) / launch time, against the 8 TB/s HBM peak.
    python tools/dumpbench.py [--mi 256] [--reps 10]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rtl_433_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--mi", type=int, default=256, help="Mi output values per launch")
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
L = _lib.lib()
n = a.mi << 20
src = torch.randint(0, 256, (4 * n + 64,), dtype=torch.uint8, device="cuda")  # enough for 16 IQ-comps of int16
dst = torch.empty(4 * n + 64, dtype=torch.uint8, device="cuda")
OUT_B = {"cu8": 1, "cs8": 1, "cs16": 2, "cf32": 4, "am.f32": 4, "fm.f32": 4, "i.f32": 4, "q.f32": 4}
print(f"{'format':8s} {'input':5s} {'GB in+out':>10s} {'ms':>8s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
for fmt, ss in (("cs16", 2), ("cs8", 2), ("cf32", 2), ("i.f32", 2), ("cu8", 4), ("cs8", 4), ("cf32", 4), ("q.f32", 4), ("am.f32", 2)):
    in16 = ss == 4 or fmt in ("am.f32", "fm.f32")
    in_b = (2 if in16 else 1) * (2 if fmt in ("i.f32", "q.f32") else 1)
    tot = n * (in_b + OUT_B[fmt])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for r in range(a.reps + 2):
        ev[0].record()
        _lib.check(L.r433_dump_convert(_lib.DUMP_FORMATS[fmt], ss, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, None), "dump")
        ev[1].record()
        torch.cuda.synchronize()
        if r >= 2:
            best = min(best, ev[0].elapsed_time(ev[1]))
    print(f"{fmt:8s} {'cs16' if ss == 4 and not fmt.startswith(('am', 'fm')) else 's16' if in16 else 'cu8':5s} {tot / 1e9:10.3f} {best:8.3f} {tot / best / 1e6:8.0f} {tot / best / 1e6 / 8000:9.3f}")
