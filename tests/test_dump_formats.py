"""-w dump formats (reference src/r_flow.c:385-489): the oracle's restatement against vectors from the real
reference CLI (tests/golden/dumps.json, made by tests/golden/gen_dump_golden.py), the emulator build of
r433_dump_convert against the oracle (CPU), and the HIP kernels against both (GPU)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from tests.cases import dump_capture, fpdm_for
from oracle import pyoracle as po
from rtl_433_amd import _lib

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dumps.json")))
RATES = {"cu8": (250000, 433920000), "cs16": (1024000, 868000000)}


def _sources(name):
    """fmt -> (input array for the conversion, sample_size)"""
    iq, ss, _ = dump_capture(name)
    rate, freq = RATES[name]
    o = po.oracle_flow(iq, None, po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq)), taps=True)
    src = {}
    for fmt in po.DUMP_FORMATS:
        src[fmt] = (o["am"] if fmt.startswith("am.") else o["fm"] if fmt.startswith("fm.") else iq, ss)
    return src


@pytest.mark.parametrize("name", ["cu8", "cs16"])
def test_oracle_matches_reference_cli(name):
    src = _sources(name)
    for fmt in po.DUMP_FORMATS:
        data = po.dump_convert(fmt, src[fmt][1], src[fmt][0])
        assert len(data) == GOLD[name][fmt]["bytes"], fmt
        assert hashlib.sha256(data).hexdigest() == GOLD[name][fmt]["sha256"], fmt


def _aligned(nbytes):
    buf = np.zeros(nbytes + 64, dtype=np.uint8)
    off = (-buf.ctypes.data) % 16
    return buf[off:off + nbytes]


def _out_bytes(fmt, n_out):
    return n_out * {"cu8": 1, "cs8": 1, "cs16": 2, "am.s16": 2, "fm.s16": 2}.get(fmt, 4)


@pytest.mark.parametrize("name", ["cu8", "cs16"])
def test_emulator_matches_oracle(name):
    from tests.emu.host import emu_lib
    L = emu_lib()
    src = _sources(name)
    for fmt in po.DUMP_FORMATS:
        arr, ss = src[fmt]
        want = po.dump_convert(fmt, ss, arr)
        n_out = len(arr) // 2 if fmt in ("i.f32", "q.f32") else len(arr)
        for cut in (n_out, 8 * 37, 5, 0):  # whole stream, whole groups only, tail only, nothing
            cut = min(cut, n_out)
            din = _aligned(arr.nbytes)
            din[:] = arr.view(np.uint8)
            dout = _aligned(_out_bytes(fmt, cut) + 16)
            dout[:] = 0xA5
            rc = L.r433_dump_convert(_lib.DUMP_FORMATS[fmt], ss, C.c_void_p(din.ctypes.data), C.c_void_p(dout.ctypes.data), cut, None)
            assert rc == 0, (fmt, _lib.last_error(L))
            nb = _out_bytes(fmt, cut)
            assert dout[:nb].tobytes() == want[:nb], (name, fmt, cut)
            assert (dout[nb:] == 0xA5).all(), (name, fmt, cut)  # nothing written past the end


def test_bad_arguments_emulator():
    from tests.emu.host import emu_lib
    L = emu_lib()
    buf = _aligned(64)
    p = C.c_void_p(buf.ctypes.data)
    assert L.r433_dump_convert(0, 2, p, p, 8, None) < 0
    assert L.r433_dump_convert(11, 2, p, p, 8, None) < 0
    assert L.r433_dump_convert(2, 3, p, p, 8, None) < 0
    assert L.r433_dump_convert(2, 2, C.c_void_p(buf.ctypes.data + 1), p, 8, None) < 0
    assert L.r433_dump_convert(2, 2, None, p, 0, None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cu8", "cs16"])
def test_gpu_matches_oracle_and_reference(name):
    import torch
    L = _lib.lib()
    src = _sources(name)
    for fmt in po.DUMP_FORMATS:
        arr, ss = src[fmt]
        want = po.dump_convert(fmt, ss, arr)
        n_out = len(arr) // 2 if fmt in ("i.f32", "q.f32") else len(arr)
        d_in = torch.from_numpy(np.frombuffer(arr.tobytes(), dtype=np.uint8).copy()).cuda()
        d_out = torch.full((_out_bytes(fmt, n_out) + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        _lib.check(L.r433_dump_convert(_lib.DUMP_FORMATS[fmt], ss, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), n_out, None),
                   "dump_convert")
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        nb = len(want)
        assert got[:nb].tobytes() == want, (name, fmt)
        assert (got[nb:] == 0xA5).all(), (name, fmt)
        assert hashlib.sha256(got[:nb].tobytes()).hexdigest() == GOLD[name][fmt]["sha256"], (name, fmt)


@pytest.mark.gpu
def test_gpu_dump_of_engine_taps_matches_reference():
    """am.f32 / fm.f32 from the taps the detection kernel itself leaves behind == the reference's files."""
    import torch
    from tests.test_gpu_parity import _gpu_run
    L = _lib.lib()
    for name in ("cu8", "cs16"):
        iq, ss, _ = dump_capture(name)
        rate, freq = RATES[name]
        g = _gpu_run([iq], ss, rate, freq, None, taps=True)
        env, am, fm = g["taps"]
        n = iq.nbytes // ss
        for fmt, tap in (("am.f32", am), ("fm.f32", fm)):
            d_in = torch.from_numpy(np.ascontiguousarray(np.asarray(tap)[0, :n]).view(np.uint8).copy()).cuda()
            d_out = torch.zeros(4 * n + 16, dtype=torch.uint8, device="cuda")
            _lib.check(L.r433_dump_convert(_lib.DUMP_FORMATS[fmt], ss, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), n, None), "dump")
            torch.cuda.synchronize()
            data = d_out.cpu().numpy()[: 4 * n].tobytes()
            assert hashlib.sha256(data).hexdigest() == GOLD[name][fmt]["sha256"], (name, fmt)


def _host_variant(L, name):
    """r433_dump_convert_host (what the drop-in's -w / -W dumpers call per frame): unaligned host buffers, odd lengths."""
    src = _sources(name)
    for fmt in po.DUMP_FORMATS:
        arr, ss = src[fmt]
        want = po.dump_convert(fmt, ss, arr)
        n_all = len(arr) // 2 if fmt in ("i.f32", "q.f32") else len(arr)
        for n_out in (n_all, 2 * 4097, 6):
            n_out = min(n_out, n_all)
            raw = np.zeros(arr.nbytes + 3, dtype=np.uint8)
            raw[1:1 + arr.nbytes] = arr.view(np.uint8)  # deliberately misaligned
            out = np.full(_out_bytes(fmt, n_out) + 8, 0xA5, dtype=np.uint8)
            rc = L.r433_dump_convert_host(_lib.DUMP_FORMATS[fmt], ss, C.c_void_p(raw.ctypes.data + 1), C.c_void_p(out.ctypes.data), n_out)
            assert rc == 0, (fmt, _lib.last_error(L))
            nb = _out_bytes(fmt, n_out)
            assert out[:nb].tobytes() == want[:nb], (name, fmt, n_out)
            assert (out[nb:] == 0xA5).all(), (name, fmt, n_out)


@pytest.mark.parametrize("name", ["cu8", "cs16"])
def test_emulator_host_variant(name):
    from tests.emu.host import emu_lib
    _host_variant(emu_lib(), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cu8", "cs16"])
def test_gpu_host_variant(name):
    _host_variant(_lib.lib(), name)
