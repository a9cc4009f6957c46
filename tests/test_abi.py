"""The drop-in boundary: struct layouts equal the compiled reference's, the shared library exports
every entry point include/r433_hip.h declares, and nothing computes without a HIP device."""
import ctypes as C
import json
import os
import re

import pytest

from rtl_433_amd import _lib
from rtl_433_amd.engine import DEFAULT_DEVICE_TABLE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "r433_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(r433_[a-z0-9_]+)\s*\(", text))
    names -= set(re.findall(r"\(\*\s*(r433_[a-z0-9_]+)\s*\)", text))  # function-pointer typedefs
    return sorted(names)


def test_header_and_binding_table_agree():
    assert _declared() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("librtl433hip.so not built yet (python -m rtl_433_amd.build)")
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(L, n)]
    assert not missing, missing


def test_struct_layouts_equal_the_compiled_reference():
    sizes = json.load(open(DEFAULT_DEVICE_TABLE))["abi_sizes"]  # sizeof/offsetof from the reference build
    bitbuffer, pulse_data, r_device, off_decode_fn, off_priority, off_ctx, off_pulse, off_low = sizes
    assert (bitbuffer, pulse_data, r_device) == (6604, 9672, 152)
    assert C.sizeof(_lib.RDevice) == r_device
    assert _lib.RDevice.decode_fn.offset == off_decode_fn
    assert _lib.RDevice.priority.offset == off_priority
    assert _lib.RDevice.decode_ctx.offset == off_ctx
    hdr = open(os.path.join(ROOT, "include", "r433_abi.h")).read()
    for n in (bitbuffer, pulse_data, r_device, off_pulse, off_low, off_decode_fn, off_priority, off_ctx):
        assert f"== {n}" in hdr, f"static_assert for {n} missing from r433_abi.h"


def test_no_cpu_fallback_without_a_device():
    """On a box without a GPU every compute entry point must refuse (R433_ENODEV), never compute."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("librtl433hip.so not built yet")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _lib.lib()
    assert L.r433_device_count() == -2  # R433_ENODEV
    cfg = _lib.FlowCfg()
    L.r433_flow_cfg_default(C.byref(cfg), 2, 250000)
    assert not L.r433_batch_create(C.byref(cfg), None, 0)
    assert L.r433_envelope_detect(None, None, 0, None, None) == -2
    assert b"HIP" in L.r433_last_error() or b"device" in L.r433_last_error()
    # the stand-alone detector: the object is host state, a call is device work
    import numpy as np
    det = L.r433_detector_create()
    assert det
    am, fm = np.zeros(64, dtype=np.int16), np.zeros(64, dtype=np.int16)
    pa, pb = _lib.PulseData(), _lib.PulseData()
    assert L.r433_detector_package(det, am.ctypes.data, fm.ctypes.data, 64, 250000, 0, C.byref(pa), C.byref(pb), 0) == -2
    assert L.r433_detector_package(det, None, None, 64, 250000, 0, C.byref(pa), C.byref(pb), 0) == -1  # R433_EINVAL before anything else
    L.r433_detector_destroy(det)


def test_missing_library_is_a_hard_error(monkeypatch):
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librtl433hip.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_public_headers_are_plain_c_and_mirrors_match(tmp_path):
    """include/*.h compile as C99 on their own, and the ctypes mirrors of the result structs have the compiler's sizes."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "r433_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(r433_analysis), sizeof(r433_histogram), sizeof(r433_grab),\n'
                   '  sizeof(r433_pulse_data), sizeof(r433_flow_cfg), offsetof(r433_analysis, device), offsetof(r433_grab, byte_len), sizeof(r433_dev_timing)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE).stdout.split()]
    want = [C.sizeof(_lib.Analysis), C.sizeof(_lib.Histogram), C.sizeof(_lib.Grab), C.sizeof(_lib.PulseData), C.sizeof(_lib.FlowCfg),
            _lib.Analysis.device.offset, _lib.Grab.byte_len.offset, C.sizeof(_lib.DevTimingRow)]
    assert got == want


def test_the_stateful_decoder_lists_match_the_reference_sources():
    """dropin/plugins_shim.c and dropin/r_flow_hip.c tell the library which decoders keep nothing between two calls
    (r433_batch_set_stateless: their calls may then run on several replay threads) from a list of the ones that DO.  Held to
    the reference's sources where the tree is present: a decoder whose file has a mutable static (file scope or inside a
    function) must be on both lists under its registered name, and nothing else is."""
    import glob
    import re
    ref = "/root/reference/src/devices"
    if not os.path.isdir(ref):
        pytest.skip("no reference tree here")
    keeps = set()
    for path in glob.glob(os.path.join(ref, "*.c")):
        src = open(path, errors="replace").read()
        mutable = [l for l in src.splitlines() if re.match(r"^\s*static\s+(?!const\b)[^()]*(=|;)\s*(//.*)?$", l) and "const" not in l]
        if mutable:
            keeps |= set(re.findall(r"\.name\s*=\s*\"([^\"]+)\"", src))
    assert len(keeps) >= 4
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for user in ("dropin/plugins_shim.c", "dropin/r_flow_hip.c"):
        src = open(os.path.join(root, user)).read()
        m = re.search(r"static char const \*const stateful\[\] = \{(.*?)\};", src, re.S)
        assert m, user
        assert set(re.findall(r"\"([^\"]+)\"", m.group(1))) == keeps, user
