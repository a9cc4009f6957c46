"""Sample grabber plan (`-S`, reference src/r_flow.c:342-362, src/samp_grab.c:100-165) against the files the real
reference CLI saves (tests/golden/grabs.json, made by tests/golden/gen_grab_golden.py)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from rtl_433_amd import _lib
from rtl_433_amd.engine import BatchEngine, flow_cfg
from tests.cases import GOLD, grab_capture

GRABS = json.load(open(os.path.join(GOLD, "grabs.json")))


def _plan(eng, mode, L):
    arr = (_lib.Grab * 16)()
    n = L.r433_batch_grab_plan(eng.h, mode, C.cast(arr, C.c_void_p), 16)
    assert n >= 0, _lib.last_error(L)
    return [arr[k] for k in range(n)]


def _check(make_engine, L):
    names = sorted(GRABS)
    caps = [grab_capture(n) for n in names]
    eng, n_pkgs = make_engine(caps)
    grabs = _plan(eng, 1, L)
    want = [(ci, g) for ci, n in enumerate(names) for g in GRABS[n]]
    assert len(grabs) == len(want)
    for k, (g, (ci, w)) in enumerate(zip(grabs, want)):
        assert g.stream == ci and g.counter == k + 1 and not g.clipped
        assert (g.byte_offset, g.byte_len) == (w["offset_in_input"], w["bytes"]), (k, g.byte_offset, g.byte_len, w)
        data = caps[ci].tobytes()[g.byte_offset:g.byte_offset + g.byte_len]
        assert hashlib.sha256(data).hexdigest() == w["sha256"]
    # known / unknown need the decode results
    assert L.r433_batch_grab_plan(eng.h, 3, None, 0) < 0
    eng.dispatch((C.POINTER(_lib.RDevice) * 0)())  # no decoders: nothing is ever "known"
    assert len(_plan(eng, 3, L)) == 0 and len(_plan(eng, 2, L)) == len(want)
    assert L.r433_batch_grab_plan(eng.h, 4, None, 0) < 0
    eng.close()


def _emu_engine(caps):
    from tests.emu.host import emu_lib
    lens = np.array([c.nbytes for c in caps], dtype=np.uint32)
    stride = int((lens.max() + 15) // 16 * 16)
    buf = np.zeros(len(caps) * stride + 64, dtype=np.uint8)
    off = (-buf.ctypes.data) % 16
    for i, c in enumerate(caps):
        buf[off + i * stride: off + i * stride + c.nbytes] = c.view(np.uint8)
    eng = BatchEngine(flow_cfg(2, 250000), np.zeros(0, dtype=po.DEV_DTYPE), profiling=False, library=emu_lib())
    eng._keep = buf
    eng.set_split(0)
    return eng, eng.run_ptr(buf.ctypes.data + off, stride, len(caps), lens)


def test_grab_plan_matches_reference_files_emulator():
    from tests.emu.host import emu_lib
    _check(_emu_engine, emu_lib())


@pytest.mark.gpu
def test_grab_plan_matches_reference_files_gpu():
    import torch

    def make(caps):
        lens = np.array([c.nbytes for c in caps], dtype=np.uint32)
        stride = int((lens.max() + 15) // 16 * 16)
        host = np.zeros((len(caps), stride), dtype=np.uint8)
        for i, c in enumerate(caps):
            host[i, :c.nbytes] = c.view(np.uint8)
        eng = BatchEngine(flow_cfg(2, 250000), np.zeros(0, dtype=po.DEV_DTYPE), profiling=False)
        eng._keep = torch.from_numpy(host).cuda()
        return eng, eng.run(eng._keep, lens)
    _check(make, _lib.lib())


def test_sigmf_container_matches_reference_files():
    """`-S sigmf:all`: prefix + grabbed bytes + trailer == the .sigmf file the reference CLI writes (host-only code)."""
    L = _lib.lib()
    for name in sorted(GRABS):
        raw = grab_capture(name).tobytes()
        for g in GRABS[name]:
            buf = C.create_string_buffer(4096)
            n = L.r433_sigmf_prefix(2, 250000, 433920000, g["bytes"], buf, len(buf))
            assert n == 1536
            pre = buf.raw[:n]
            n2 = L.r433_sigmf_trailer(g["bytes"], buf, len(buf))
            whole = pre + raw[g["offset_in_input"]:g["offset_in_input"] + g["bytes"]] + buf.raw[:n2]
            assert len(whole) == g["sigmf_bytes"]
            assert hashlib.sha256(whole).hexdigest() == g["sigmf_sha256"]
    assert L.r433_sigmf_prefix(3, 250000, 1, 1, None, 0) < 0
    assert L.r433_sigmf_trailer(100, None, 0) == 412 + 1024


def test_sigmf_probe_reads_the_container_back():
    L = _lib.lib()
    buf = C.create_string_buffer(4096)
    data = bytes(range(256)) * 5 + b"xyz"  # 1283 bytes: needs record padding
    n = L.r433_sigmf_prefix(4, 1024000, 868000000, len(data), buf, len(buf))
    pre = buf.raw[:n]
    n2 = L.r433_sigmf_trailer(len(data), buf, len(buf))
    arc = pre + data + buf.raw[:n2]
    assert len(arc) % 512 == 0
    info = _lib.SigmfInfo()
    assert L.r433_sigmf_probe(arc, len(arc), C.byref(info)) == 0, _lib.last_error(L)
    assert (info.datatype, info.sample_rate, info.frequency, info.sample_start) == (b"ci16_le", 1024000, 868000000, 0)
    assert arc[info.data_offset:info.data_offset + info.data_len] == data
    assert L.r433_sigmf_probe(arc[:1000], 1000, C.byref(info)) < 0       # cut short
    assert L.r433_sigmf_probe(bytes(2048), 2048, C.byref(info)) < 0      # no stream


def test_grab_plan_needs_a_detection_run_emulator():
    from tests.emu.host import emu_lib
    from rtl_433_amd.engine import load_pulse_text
    L = emu_lib()
    eng = BatchEngine(flow_cfg(2, 250000), np.zeros(0, dtype=po.DEV_DTYPE), profiling=False, library=L)
    text = open(os.path.join(GOLD, "kat.ook"), "rb").read()
    assert eng.run_pulses(load_pulse_text(text, 250000, library=L)) == 1
    assert L.r433_batch_grab_plan(eng.h, 1, None, 0) < 0
    assert "detection run" in _lib.last_error(L)
    eng.close()
