"""TEST INFRASTRUCTURE: drives the emulator build of the library (tests/emu/build_emu.py) with host
buffers through the very same C ABI and Python engine class the product uses."""
from __future__ import annotations

import ctypes as C

import numpy as np

from rtl_433_amd import _lib
from rtl_433_amd.engine import BatchEngine, flow_cfg

from . import build_emu

_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        _emu = _lib.bind(C.CDLL(build_emu.build()))
    return _emu


def emu_run(iq_list, ss, rate, devs, fpdm=0, taps=False, enable_fm=1, center_frequency=433920000, split=0, debug=0, **kw):
    """Same contract as tests/test_gpu_parity._gpu_run, on the emulator."""
    n = len(iq_list)
    lens = np.array([a.nbytes for a in iq_list], dtype=np.uint32)
    stride = max(16, int((lens.max() + 15) // 16 * 16)) if n else 16
    buf = np.zeros(max(n, 1) * stride + 64, dtype=np.uint8)
    base = (-buf.ctypes.data) % 16
    arena = buf[base:base + max(n, 1) * stride].reshape(max(n, 1), stride)
    for i, a in enumerate(iq_list):
        arena[i, :a.nbytes] = a.view(np.uint8)
    cfg = flow_cfg(ss, rate, fpdm=fpdm, enable_fm=enable_fm, center_frequency=center_frequency, **kw)
    eng = BatchEngine(cfg, devs, profiling=False, library=emu_lib())
    if split:
        eng.set_split(split)
    if debug:
        eng.set_debug(debug)
    tap_bufs = None
    if taps:
        ns = max(1, stride // (ss * (2 if kw.get('input_format') == 2 else 1)))
        tap_bufs = [np.zeros((n, ns), dtype=np.uint16), np.zeros((n, ns), dtype=np.int16), np.zeros((n, ns), dtype=np.int16)]
        _lib.check(eng.L.r433_batch_set_taps(eng.h, *[C.c_void_p(t.ctypes.data) for t in tap_bufs], ns), "set_taps")
    npk = eng.run_ptr(arena.ctypes.data, stride, n, lens)
    out = dict(n_packages=npk, packages=eng.packages(), events=eng.events(), sums=eng.frame_sums(n), split=eng.split_stats())
    if taps:
        out["taps"] = tuple(tap_bufs)
    eng.close()
    return out
