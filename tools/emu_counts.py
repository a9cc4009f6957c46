"""Development aid: how often each path of phase C runs for one capture, counted by the CPU wave emulator.
Builds a throw-away emulator library with -DR433_EMU_COUNTERS under /tmp (the tested one stays as it is).

    python tools/emu_counts.py [--seed0 885] [--streams 1] [--samples 65536]
"""
import argparse, ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--seed0", type=int, default=885)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--samples", type=int, default=65536)
a = ap.parse_args()

from tests.emu import build_emu
out_dir = "/tmp/emu_counts"
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "librtl433emu.so")
objs, procs = [], []
for src in build_emu.sources():
    obj = os.path.join(out_dir, os.path.basename(src) + ".o")
    cmd = ["g++"] + build_emu.FLAGS + ["-DR433_EMU_COUNTERS", "-x", "c++", "-I", os.path.join(build_emu.HERE, "include"), "-I", build_emu.INC,
                                     "-I", build_emu.CSRC, "-c", src, "-o", obj]
    procs.append(subprocess.Popen(cmd))
    objs.append(obj)
for p in procs:
    assert p.wait() == 0
subprocess.check_call(["g++", "-shared", "-o", so] + objs)

import ctypes
import numpy as np
from rtl_433_amd import _lib, synth
from rtl_433_amd.engine import BatchEngine, flow_cfg

NAMES = {0: "outer iterations", 1: "block loads", 2: "legs idle", 3: "legs pulse", 4: "legs gap-start", 5: "legs gap", 10: "candidates (legs)",
         11: "general steps", 12: "scalar fallback samples", 13: "engine legs", 14: "engine window loads", 15: "resolve re-runs",
         16: "engine average runs", 17: "engine samples in groups of 8", 18: "engine remainder samples", 19: "engine short-run samples",
         20: "engine general-form samples", 21: "engine candidates", 22: "engine rotated runs", 23: "engine pulse legs"}
host = synth.ook_batch(a.streams, a.samples, 250000, seed0=a.seed0)
raw = ctypes.CDLL(so)
L = _lib.bind(raw)
eng = BatchEngine(flow_cfg(2, 250000), None, profiling=False, library=L)
buf = np.ascontiguousarray(host)
lens = np.full(len(buf), buf.shape[1], dtype=np.uint32)
n = eng.run_ptr(buf.ctypes.data, buf.shape[1], len(buf), lens)
cnt = (ctypes.c_ulonglong * 32).in_dll(raw, "r433_dbg_counts")
print("packages", n)
for i in range(32):
    if cnt[i]:
        print(f"{i:2d} {NAMES.get(i, '?'):34s} {cnt[i]}")
