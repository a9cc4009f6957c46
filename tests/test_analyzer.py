"""Pulse analyzer (`-A`, reference src/pulse_analyzer.c): the pure-Python restatement (oracle/analyzer.py) against
the text of the real reference CLI (tests/golden/analyzer.json, made by tests/golden/gen_analyzer_golden.py), and
r433_batch_analyze + r433_analysis_text (device histograms and guess, host text) against both -- on the emulator
(CPU) and on the GPU."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import analyzer as an
from oracle import pyoracle as po
from rtl_433_amd.engine import BatchEngine, flow_cfg
from tests.cases import GOLD, fpdm_for, make_case

GOLDEN = json.load(open(os.path.join(GOLD, "analyzer.json")))
_libm = C.CDLL("libm.so.6")
_libm.log10f.restype = C.c_float
_libm.log10f.argtypes = [C.c_float]
F = np.float32


def _levels(p, ss):
    """calc_rssi_snr, reference src/r_flow.c:35-64, in the reference's float arithmetic."""
    hi = F(p["high"] if p["high"] > 0 else 1)
    lo = F(p["low"] if p["low"] > 0 else 1)
    asnr = F(min(hi, F(16383)) / lo)
    k, ref = (F(10.0), F(42.1442)) if ss == 2 else (F(20.0), F(84.2884))
    return dict(high=p["high"], low=p["low"], f1=p["f1"], f2=p["f2"], rssi=float(F(k * F(_libm.log10f(hi)) - ref)),
                noise=float(F(k * F(_libm.log10f(lo)) - ref)), snr=float(F(k * F(_libm.log10f(asnr)))))


def _oracle_blocks(name):
    iq, ss, rate, freq = make_case(name)
    o = po.oracle_flow(iq, None, po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq)))
    pk = po.parse_packages(o["packages"])
    return pk, [an.analyze(p["pulse"], p["gap"], p["type"], rate, _levels(p, ss)) for p in pk]


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_matches_reference_cli_text(name):
    pk, res = _oracle_blocks(name)
    assert len(pk) == GOLDEN[name]["packages"]
    for k, want in enumerate(GOLDEN[name]["blocks"]):
        assert res[k][0] == want, (name, k)


def _check_engine(make_engine, names):
    for name in names:
        iq, ss, rate, freq = make_case(name)
        pk, res = _oracle_blocks(name)
        eng, n = make_engine(iq, ss, rate, freq)
        assert n == len(pk) == GOLDEN[name]["packages"]
        results = eng.analyze()
        assert len(results) == n
        for k in range(n):
            a = results[k]
            text = eng.analysis_text(k, a).splitlines()
            assert text == res[k][0], (name, k)                       # == oracle, every package
            if k < len(GOLDEN[name]["blocks"]):
                assert text == GOLDEN[name]["blocks"][k], (name, k)   # == reference CLI
            dev = res[k][1]
            assert a.device.modulation == dev["modulation"]
            if dev["modulation"]:
                for f in ("short_width", "long_width", "reset_limit", "gap_limit", "sync_width", "tolerance"):
                    assert np.float32(getattr(a.device, f)) == np.float32(dev[f]), (name, k, f)
            assert a.num_pulses == pk[k]["num"]
        eng.close()


def _emu_engine(iq, ss, rate, freq):
    from tests.emu.host import emu_lib
    eng = BatchEngine(flow_cfg(ss, rate, fpdm=fpdm_for(freq), center_frequency=freq), None, profiling=False, library=emu_lib())
    nb = iq.nbytes
    buf = np.zeros(nb + 96, dtype=np.uint8)
    off = (-buf.ctypes.data) % 16
    buf[off:off + nb] = iq.view(np.uint8)
    eng._keep = buf
    n = eng.run_ptr(buf.ctypes.data + off, max(16, (nb + 15) // 16 * 16), 1, np.array([nb], dtype=np.uint32))
    return eng, n


def test_emulator_matches_oracle_and_reference():
    _check_engine(_emu_engine, ["kat", "ook1", "ook2", "ook_long", "fsk_cs16", "fsk_cu8_minmax"])


def test_analyze_edge_cases_emulator():
    from tests.emu.host import emu_lib
    L = emu_lib()
    eng = BatchEngine(flow_cfg(2, 250000), None, profiling=False, library=L)
    assert len(eng.analyze()) == 0  # nothing has run yet
    eng.close()


def _gpu_engine(iq, ss, rate, freq):
    import torch
    eng = BatchEngine(flow_cfg(ss, rate, fpdm=fpdm_for(freq), center_frequency=freq), None, profiling=False)
    nb = iq.nbytes
    stride = max(16, (nb + 15) // 16 * 16)
    host = np.zeros((1, stride), dtype=np.uint8)
    host[0, :nb] = iq.view(np.uint8)
    eng._keep = torch.from_numpy(host).cuda()
    return eng, eng.run(eng._keep, np.array([nb], dtype=np.uint32))


@pytest.mark.gpu
def test_gpu_matches_oracle_and_reference():
    _check_engine(_gpu_engine, sorted(n for n in GOLDEN if GOLDEN[n]["packages"]))


def test_analyze_after_pulse_side_door_emulator():
    """Packages that came in through r433_batch_run_pulses are analyzed like detected ones."""
    from tests.emu.host import emu_lib
    from rtl_433_amd.engine import load_pulse_text
    L = emu_lib()
    iq, ss, rate, freq = make_case("kat")
    eng, n = _emu_engine(iq, ss, rate, freq)
    a_det = eng.analyze()[0]
    eng.close()
    eng = BatchEngine(flow_cfg(ss, rate, center_frequency=freq), None, profiling=False, library=L)
    text = open(os.path.join(GOLD, "kat.ook"), "rb").read()
    assert eng.run_pulses(load_pulse_text(text, rate, library=L)) == 1
    a_file = eng.analyze()[0]
    eng.close()
    # the file holds microseconds rounded to integers: 4 us per sample here, so the sample counts survive
    assert bytes(a_file) == bytes(a_det)
