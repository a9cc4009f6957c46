"""The `u8` logic dump (-w file.u8): the bytes the detection kernel paints against the file the reference CLI itself writes
(oracle/_ref/rtl_433_ref -r capture -w out.u8) -- OOK trains over 2.5 frames, FSK packages with both detectors, the FSK
candidate every OOK package carries through its first pulse, packages spanning frame boundaries (partial paints at the
frame end, clipped repaints after it)."""
import os
import subprocess

import numpy as np
import pytest

from rtl_433_amd import synth
from tests.cases import fpdm_for, make_case
from tests.emu import build_emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "rtl_433_ref")
BACKENDS = [pytest.param("emu", marks=pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")),
            pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    return request.param


CASES = ["kat", "ook3", "ook_long", "fsk_cu8", "fsk_cu8_minmax", "fsk_cs16", "fsk_cs16_classic", "random", "noise", "span"]


def capture(name):
    if name == "span":  # one long package across the 131072-sample frame boundary
        a = np.concatenate([synth.ook_stream(60, 65536)[0], synth.ook_stream(885, 65536)[0][: 2 * 40000], synth.ook_stream(885, 65536)[0]])
        lead = np.full(2 * 70000, 128, dtype=np.uint8)
        return np.concatenate([lead, synth.ook_stream(885, 65536)[0], a]), 2, 250000, 433920000
    return make_case(name)


def reference_u8(tmp_path, iq, ss, rate, freq):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/rtl_433_ref not present")
    ext = "cu8" if ss == 2 else "cs16"
    name = f"c_{freq / 1e6:g}M_{rate // 1000}k.{ext}"
    iq.tofile(tmp_path / name)
    subprocess.run([REF, "-r", name, "-w", "out.u8"], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    return np.fromfile(tmp_path / "out.u8", dtype=np.uint8)


@pytest.mark.parametrize("name", CASES)
def test_logic_dump_equals_reference_file(name, backend, tmp_path, default_devices):
    from rtl_433_amd.engine import BatchEngine, flow_cfg
    iq, ss, rate, freq = capture(name)
    want = reference_u8(tmp_path, iq, ss, rate, freq)
    n = iq.nbytes // ss
    assert want.size == n
    devs = default_devices[0]  # the CLI registers its default decoders (FM demodulation on)
    cfg = flow_cfg(ss, rate, fpdm=fpdm_for(freq), center_frequency=freq)
    if backend == "gpu":
        import torch
        eng = BatchEngine(cfg, devs)
        eng.enable_logic_dump()
        buf = np.zeros((1, (iq.nbytes + 15) // 16 * 16), dtype=np.uint8)
        buf[0, :iq.nbytes] = iq.view(np.uint8)
        eng.run(torch.from_numpy(buf).cuda(), np.array([iq.nbytes], dtype=np.uint32))
    else:
        from tests.emu import host
        eng = BatchEngine(cfg, devs, library=host.emu_lib())
        eng.enable_logic_dump()
        eng.run_host([iq])
    got = eng.logic_dump([n])[0]
    eng.close()
    if name not in ("noise",):
        assert want.any(), "nothing painted: the case does not test anything"
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:10], got[bad[:10]], want[bad[:10]])
