"""The stand-alone function-level entry points of include/r433_hip.h -- r433_envelope_detect, r433_magnitude_est_cu8,
r433_magnitude_est_cs16, r433_level_db, r433_convert_cs8_cu8, r433_convert_cf32_cs16 -- against the functions they stand
in for, called in the unmodified reference itself (oracle/_ref/libr433ref.so exports envelope_detect, magnitude_est_cu8,
magnitude_est_cs16 of src/baseband.c:36-110 like any shared object), and against the oracle where that library is absent.
Emulator build in the CPU suite, product library under -m gpu."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from rtl_433_amd import _lib
from tests.emu import build_emu

BACKENDS = [pytest.param("emu", marks=pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")),
            pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    return request.param


class Dev:
    """device buffers for either backend: numpy memory for the emulator, torch tensors on the GPU"""

    def __init__(self, backend):
        self.backend = backend
        if backend == "gpu":
            import torch
            self.torch = torch
            self.L = _lib.lib()
        else:
            from tests.emu import host
            self.L = host.emu_lib()
        self.keep = []

    def put(self, a):
        a = np.ascontiguousarray(a)
        if self.backend == "gpu":
            t = self.torch.from_numpy(a.view(np.uint8).copy()).cuda()
            self.keep.append(t)
            return t.data_ptr(), t
        buf = np.zeros(a.nbytes + 32, dtype=np.uint8)
        off = (-buf.ctypes.data) % 16
        buf[off:off + a.nbytes] = a.view(np.uint8).ravel()
        self.keep.append(buf)
        return buf.ctypes.data + off, buf[off:off + a.nbytes]

    def empty(self, nbytes):
        return self.put(np.full(nbytes, 0xA5, dtype=np.uint8))

    def get(self, handle, dtype):
        if self.backend == "gpu":
            self.torch.cuda.synchronize()
            return handle.cpu().numpy().view(dtype)
        return handle.view(dtype)


def ref_lib():
    if not po.have_ref():
        return None
    L = C.CDLL(po.REF_SO)
    L.baseband_init()
    for f in (L.envelope_detect, L.magnitude_est_cu8, L.magnitude_est_cs16):
        f.restype = C.c_float
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    return L


def inputs(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "cs16":
        a = rng.integers(-32768, 32768, 2 * n, dtype=np.int64).astype(np.int16)
        k = min(8, a.size)
        a[:k] = [-32768, -32768, 32767, 32767, -32768, 32767, 0, 0][:k]
        return a
    a = rng.integers(0, 256, 2 * n, dtype=np.int64).astype(np.uint8)
    k = min(8, a.size)
    a[:k] = [0, 0, 255, 255, 0, 255, 128, 128][:k]
    return a


@pytest.mark.parametrize("kind,n", [("amp", 1), ("amp", 100003), ("amp", 262144), ("mag", 100003), ("mag", 7),
                                    ("cs16", 100003), ("cs16", 65536), ("cs16", 3)])
def test_envelope_functions_vs_reference(kind, n, backend):
    d = Dev(backend)
    iq = inputs(kind, n, 40 + n % 7)
    fn = {"amp": d.L.r433_envelope_detect, "mag": d.L.r433_magnitude_est_cu8, "cs16": d.L.r433_magnitude_est_cs16}[kind]
    p_in, _ = d.put(iq)
    p_out, h_out = d.empty(2 * n + 16)
    p_sum, h_sum = d.empty(16)
    _lib.check(fn(C.c_void_p(p_in), C.c_void_p(p_out), n, C.c_void_p(p_sum), None), "envelope", d.L)
    got = d.get(h_out, np.uint16)[:n].copy()
    got_sum = int(d.get(h_sum, np.uint32)[0])
    # the oracle's restatement ...
    if kind == "cs16":
        a = iq.astype(np.int64)
        i_, q_ = np.abs(a[0::2]), np.abs(a[1::2])
        want = (((122 * np.maximum(i_, q_) + 51 * np.minimum(i_, q_)) >> 8) & 0xffffffff).astype(np.uint16)
    elif kind == "mag":
        a = iq.astype(np.int64)
        i_, q_ = np.abs(a[0::2] - 128), np.abs(a[1::2] - 128)
        want = (122 * np.maximum(i_, q_) + 51 * np.minimum(i_, q_)).astype(np.uint16)
    else:
        a = iq.astype(np.int64)
        want = ((127 - a[0::2]) ** 2 + (127 - a[1::2]) ** 2).astype(np.uint16)
    assert np.array_equal(got, want)
    assert got_sum == int(want.astype(np.uint64).sum() & 0xffffffff)
    # ... and the reference function itself
    R = ref_lib()
    if R is not None:
        out = np.zeros(n, dtype=np.uint16)
        src = np.ascontiguousarray(iq)
        rf = {"amp": R.envelope_detect, "mag": R.magnitude_est_cu8, "cs16": R.magnitude_est_cs16}[kind]
        db = rf(src.ctypes.data, out.ctypes.data, n)
        assert np.array_equal(got, out)
        d.L.r433_level_db.restype = C.c_float
        mine = d.L.r433_level_db(got_sum, n, 0 if kind == "amp" else 1)
        assert np.float32(mine) == np.float32(db), (mine, db)


def test_level_db_corner_cases(backend):
    """AMP_TO_DB / MAG_TO_DB of sum / len, 1 when sum < len (src/baseband.c:43-44,77-78)."""
    d = Dev(backend)
    R = ref_lib()
    if R is None:
        pytest.skip("needs oracle/_ref")
    f = d.L.r433_level_db
    f.restype = C.c_float
    for n, val in ((64, 128), (64, 127), (1000, 0), (5, 255), (4096, 200)):
        iq = np.full(2 * n, val, dtype=np.uint8)
        out = np.zeros(n, dtype=np.uint16)
        for is_mag, rf in ((0, R.envelope_detect), (1, R.magnitude_est_cu8)):
            db = rf(iq.ctypes.data, out.ctypes.data, n)
            s = int(out.astype(np.uint64).sum() & 0xffffffff)
            assert np.float32(f(s, n, is_mag)) == np.float32(db), (n, val, is_mag)


def test_convert_cs8_cu8(backend):
    """src/rtl_433.c:1829-1833: ((int8_t)x) + 128"""
    d = Dev(backend)
    n = 100001 * 2
    src = np.random.default_rng(3).integers(-128, 128, n, dtype=np.int64).astype(np.int8)
    src[:4] = [-128, 127, 0, -1]
    p_in, _ = d.put(src)
    p_out, h_out = d.empty(n + 32)
    _lib.check(d.L.r433_convert_cs8_cu8(C.c_void_p(p_in), C.c_void_p(p_out), n, None), "cs8", d.L)
    want = (src.astype(np.int16) + 128).astype(np.uint8)
    assert np.array_equal(d.get(h_out, np.uint8)[:n], want)


def test_convert_cf32_cs16(backend):
    """src/rtl_433.c:1812-1826: (int)(f * INT16_MAX) clamped to +-INT16_MAX, x86 semantics for NaN / inf / huge."""
    d = Dev(backend)
    n = 50001 * 2
    rng = np.random.default_rng(4)
    src = rng.uniform(-1.3, 1.3, n).astype(np.float32)
    src[:10] = [1.0, -1.0, 0.99999, -0.99999, np.nan, np.inf, -np.inf, 3e9, -3e9, 1e-9]
    p_in, _ = d.put(src)
    p_out, h_out = d.empty(2 * n + 32)
    _lib.check(d.L.r433_convert_cf32_cs16(C.c_void_p(p_in), C.c_void_p(p_out), n, None), "cf32", d.L)
    # C on x86-64: cvttss2si gives INT_MIN for NaN and out-of-range values, which then clamps to -INT16_MAX
    prod = src * np.float32(32767)
    with np.errstate(invalid="ignore"):
        t = np.where(np.isfinite(prod) & (np.abs(prod) < 2147483648.0), np.trunc(prod), -2147483648.0)
    want = np.clip(t, -32767, 32767).astype(np.int16)
    assert np.array_equal(d.get(h_out, np.int16)[:n], want)
