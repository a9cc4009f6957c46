"""The pulse-data side door (`-r file.ook`, reference src/rtl_433.c:1755-1794, src/pulse_data.c:122-224):
r433_batch_run_pulses / r433_pulse_text_load / r433_pulse_text_dump against the oracle and against files and
decodes of the real reference CLI (tests/golden/kat.ook, ook_flex.json; tests/golden/gen_ook_golden.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from rtl_433_amd import _lib
from rtl_433_amd.engine import BatchEngine, dump_pulse_text, flow_cfg, load_device_table, load_pulse_text, make_rdevices
from tests.cases import GOLD, fpdm_for, make_case

KAT_OOK = open(os.path.join(GOLD, "kat.ook"), "rb").read()
OOK_FLEX = json.load(open(os.path.join(GOLD, "ook_flex.json")))
PKG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(_lib.PulseData))


def _pulse_array(pkgs):
    arr = (_lib.PulseData * len(pkgs))()
    for a, p in zip(arr, pkgs):
        a.offset, a.sample_rate, a.start_ago, a.end_ago, a.num_pulses = p["offset"], p["rate"], p["start_ago"], p["end_ago"], p["num"]
        a.ook_low_estimate, a.ook_high_estimate, a.fsk_f1_est, a.fsk_f2_est = p["low"], p["high"], p["f1"], p["f2"]
        for i in range(p["num"]):
            a.pulse[i], a.gap[i] = int(p["pulse"][i]), int(p["gap"][i])
    return arr


def _check_fanout(make_engine, names):
    devs = load_device_table()[0]
    for name in names:
        iq, ss, rate, freq = make_case(name)
        o = po.oracle_flow(iq, devs, po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq)))
        pkgs = po.parse_packages(o["packages"])
        assert pkgs, name
        # the reference sends a loaded package to the FSK decoders iff fsk_f2_est != 0; make that agree with the type
        for p in pkgs:
            if (p["type"] == 2) != (p["f2"] != 0):
                p["f2"] = 1 if p["type"] == 2 else 0
        eng = make_engine(flow_cfg(ss, rate, fpdm=fpdm_for(freq), center_frequency=freq), devs)
        n = eng.run_pulses(_pulse_array(pkgs))
        assert n == len(pkgs)
        got = po.parse_packages(eng.packages()[0])
        for k, (g, p) in enumerate(zip(got, pkgs)):
            assert g["stream"] == k and g["type"] == p["type"] and g["num"] == p["num"]
            assert list(g["pulse"]) == list(p["pulse"]) and list(g["gap"]) == list(p["gap"])
        # same packages, same decoders -> the very bitbuffers the full path hands out
        want = o["events"]
        if any((p["type"] == 2) != (q["f2"] != 0) for p, q in zip(po.parse_packages(o["packages"]), pkgs)):
            want = None
        ev = eng.events()[0]
        if want is not None:
            assert po.canonical_events(ev) == po.canonical_events(want), name
        assert len(po.parse_events(ev)) == o["n_events"], name
        eng.close()


def _emu_engine(cfg, devs):
    from tests.emu.host import emu_lib
    return BatchEngine(cfg, devs, profiling=False, library=emu_lib())


def test_fanout_from_pulses_matches_oracle_emulator():
    _check_fanout(_emu_engine, ["kat", "ook1", "fsk_cs16"])


def test_empty_and_bad_input_emulator():
    from tests.emu.host import emu_lib
    L = emu_lib()
    eng = _emu_engine(flow_cfg(2, 250000), load_device_table()[0])
    assert eng.run_pulses((_lib.PulseData * 0)()) == 0
    bad = (_lib.PulseData * 1)()
    bad[0].num_pulses = 1201
    assert L.r433_batch_run_pulses(eng.h, C.cast(bad, C.c_void_p), 1, None) < 0
    assert "1201" in _lib.last_error(L)
    eng.close()


def _text_lib():
    from tests.emu.host import emu_lib
    return emu_lib()  # host-only functions: the same code in the emulator build and in the product library


def test_text_load_matches_reference_semantics():
    L = _text_lib()
    arr = load_pulse_text(KAT_OOK, 250000, library=L)
    assert len(arr) == 1 and arr[0].num_pulses == 53
    lines = [ln for ln in KAT_OOK.decode().splitlines() if ln and not ln.startswith(";")]
    for i, ln in enumerate(lines):  # (int)(to_sample * us), src/pulse_data.c:170-171
        m, s = ln.split()
        assert arr[0].pulse[i] == int(250000 / 1e6 * int(m)) and arr[0].gap[i] == int(250000 / 1e6 * int(s))
    assert abs(arr[0].freq1_hz - 433965312) < 64 and arr[0].fsk_f2_est == 0 and arr[0].sample_rate == 250000
    # quirks of the reader: a blank line is a 0/0 pulse, a negative value skips the line, two packages in one text
    two = b";pulse data\n;ook\n100 200\n\n-5 7\n300 400\n;end\n;received x\n;ook\n8 9\n"
    arr = load_pulse_text(two, 1000000, library=L)
    assert len(arr) == 2
    assert [arr[0].pulse[i] for i in range(arr[0].num_pulses)] == [100, 0, 300]
    assert [arr[0].gap[i] for i in range(arr[0].num_pulses)] == [200, 0, 400]
    assert arr[1].num_pulses == 1 and arr[1].pulse[0] == 8 and arr[1].gap[0] == 9
    assert len(load_pulse_text(b"", 250000, library=L)) == 0


def _kat_package_via(make_engine):
    iq, ss, rate, freq = make_case("kat")
    eng = make_engine(flow_cfg(ss, rate, center_frequency=freq), None)
    got = []
    return eng, iq, got


def test_text_dump_matches_reference_file_emulator():
    """The package the detection path finds in the reference's own fixture, written by r433_pulse_text_dump, is
    the reference's `-w kat.ook` file (minus the file header and the time stamp)."""
    from tests.emu.host import emu_lib, emu_run
    L = emu_lib()
    iq, ss, rate, freq = make_case("kat")
    lens = np.array([iq.nbytes], dtype=np.uint32)
    buf = np.zeros(iq.nbytes + 80, dtype=np.uint8)
    off = (-buf.ctypes.data) % 16
    stride = (iq.nbytes + 15) // 16 * 16
    buf[off:off + iq.nbytes] = iq.view(np.uint8)
    eng = BatchEngine(flow_cfg(ss, rate, center_frequency=freq), np.zeros(0, dtype=po.DEV_DTYPE), profiling=False, library=L)
    assert eng.run_ptr(buf.ctypes.data + off, stride, 1, lens) == 1
    seen = []
    cb = PKG_FN(lambda user, stream, typ, pd: seen.append(dump_pulse_text(pd.contents, b"T", library=L)))
    eng.dispatch((C.POINTER(_lib.RDevice) * 0)(), pkg_cb=cb)
    eng.close()
    assert len(seen) == 1
    want = [ln for ln in KAT_OOK.decode().splitlines()[4:]]  # after ;pulse data ;version ;timescale ;created
    got = seen[0].decode().splitlines()
    assert got[0] == ";received T" and want[0].startswith(";received ")
    assert got[1:] == want[1:]


def test_reference_cli_decode_of_ook_file_emulator():
    """rtl_433_ref -r kat.ook -R 0 -X n=raw,m=OOK_PWM,... decodes {52}e7a760b94372e and {0}0 from the text; the same
    text through r433_pulse_text_load + r433_batch_run_pulses gives those rows."""
    L = _text_lib()
    flex = np.zeros(1, dtype=po.DEV_DTYPE)
    flex[0] = (6, 500.0, 1000.0, 5000.0, 2000.0, 1500.0, 100.0, 0)  # OOK_PWM s l r g y t, as in ook_flex.json
    eng = _emu_engine(flow_cfg(2, 250000), flex)
    assert eng.run_pulses(load_pulse_text(KAT_OOK, 250000, library=L)) == 1
    evs = po.parse_events(eng.events()[0])
    eng.close()
    ref = OOK_FLEX["events"][0]
    assert len(evs) == 1 and evs[0]["num_rows"] == ref["num_rows"]
    for (bits, _s, data), w in zip(evs[0]["rows"], ref["rows"]):
        assert bits == w["len"] and data[: (bits + 7) // 8].hex()[: (bits + 3) // 4] == w["data"]


@pytest.mark.gpu
def test_fanout_from_pulses_matches_oracle_gpu():
    _check_fanout(lambda cfg, devs: BatchEngine(cfg, devs, profiling=False), ["kat", "ook1", "ook3", "fsk_cs16", "fsk_cu8_minmax"])


@pytest.mark.gpu
def test_reference_cli_decode_of_ook_file_gpu():
    flex = np.zeros(1, dtype=po.DEV_DTYPE)
    flex[0] = (6, 500.0, 1000.0, 5000.0, 2000.0, 1500.0, 100.0, 0)
    eng = BatchEngine(flow_cfg(2, 250000), flex, profiling=False)
    assert eng.run_pulses(load_pulse_text(KAT_OOK, 250000)) == 1
    evs = po.parse_events(eng.events()[0])
    eng.close()
    ref = OOK_FLEX["events"][0]
    assert len(evs) == 1 and evs[0]["num_rows"] == ref["num_rows"]
    for (bits, _s, data), w in zip(evs[0]["rows"], ref["rows"]):
        assert bits == w["len"] and data[: (bits + 7) // 8].hex()[: (bits + 3) // 4] == w["data"]


def _rows_match(evs, ref_events):
    assert len(evs) == len(ref_events)
    for ev, ref in zip(evs, ref_events):
        assert ev["num_rows"] == ref["num_rows"]
        for (bits, _s, data), w in zip(ev["rows"], ref["rows"]):
            assert bits == w["len"] and data[: (bits + 7) // 8].hex()[: (bits + 3) // 4] == w["data"]


def _flex_row(spec):
    kv = dict(item.split("=") for item in spec.split(","))
    mod = {"OOK_PWM": 6, "OOK_PCM": 4}[kv["m"]]
    row = np.zeros(1, dtype=po.DEV_DTYPE)
    row[0] = (mod, float(kv.get("s", 0)), float(kv.get("l", 0)), float(kv.get("r", 0)), float(kv.get("g", 0)),
              float(kv.get("y", 0)), float(kv.get("t", 0)), 0)
    return row


@pytest.mark.parametrize("name", sorted(OOK_FLEX["rfraw"]))
def test_rfraw_lines_decode_like_reference_cli_emulator(name):
    """RfRaw lines (src/rfraw.c) inside a pulse file: what the reference CLI decodes from them at 1 MS/s."""
    g = OOK_FLEX["rfraw"][name]
    L = _text_lib()
    pulses = load_pulse_text(g["text"].encode(), 1000000, library=L)
    assert all(p.sample_rate == 1000000 for p in pulses)
    eng = _emu_engine(flow_cfg(2, 1000000), _flex_row(g["flex"]))
    assert eng.run_pulses(pulses) == len(pulses)
    evs = po.parse_events(eng.events()[0])
    eng.close()
    _rows_match(evs, g["events"])


def test_pulses_at_another_rate_are_refused_emulator():
    L = _text_lib()
    eng = _emu_engine(flow_cfg(2, 250000), load_device_table()[0])
    pulses = load_pulse_text(OOK_FLEX["rfraw"]["kat"]["text"].encode(), 250000, library=L)  # rfraw: 1 MS/s whatever the caller says
    assert L.r433_batch_run_pulses(eng.h, C.cast(pulses, C.c_void_p), len(pulses), None) < 0
    assert "samples/s" in _lib.last_error(L)
    eng.close()


def test_vcd_writer_matches_reference_file_emulator():
    """`-w kat.vcd` of the reference CLI == r433_pulse_vcd_header + r433_pulse_vcd of the package the path detects
    (the $date line carries the time of the run)."""
    from tests.emu.host import emu_lib
    L = emu_lib()
    want = open(os.path.join(GOLD, "kat.vcd"), "rb").read().decode().splitlines()
    iq, ss, rate, freq = make_case("kat")
    buf = np.zeros(iq.nbytes + 80, dtype=np.uint8)
    off = (-buf.ctypes.data) % 16
    buf[off:off + iq.nbytes] = iq.view(np.uint8)
    eng = BatchEngine(flow_cfg(ss, rate, center_frequency=freq), np.zeros(0, dtype=po.DEV_DTYPE), profiling=False, library=L)
    assert eng.run_ptr(buf.ctypes.data + off, (iq.nbytes + 15) // 16 * 16, 1, np.array([iq.nbytes], dtype=np.uint32)) == 1
    out = []

    def on_pkg(user, stream, typ, pd):
        b = C.create_string_buffer(1 << 16)
        n = L.r433_pulse_vcd(pd, ord("'") if typ == 1 else ord('"'), b, len(b))
        out.append(b.raw[:n].decode())
    cb = PKG_FN(on_pkg)
    eng.dispatch((C.POINTER(_lib.RDevice) * 0)(), pkg_cb=cb)
    eng.close()
    hb = C.create_string_buffer(4096)
    n = L.r433_pulse_vcd_header(rate, b"D", hb, len(hb))
    got = (hb.raw[:n].decode() + "".join(out)).splitlines()
    assert got[0] == "$date D $end" and want[0].startswith("$date ")
    assert got[1:] == want[1:]


def test_rfraw_full_package_takes_nothing_more_emulator():
    """A line that fills a package (0xB0 repeat count) and then carries further segments: the reference keeps storing
    past pulse[1200] (src/rfraw.c:141-163); this reader takes untrusted text and must stop at a full package."""
    L = _text_lib()
    # AA B0 len bins=2 repeats=FF, two bins (100 us, 200 us), 30 pulse/gap nibbles "8 1" (-> 15 pairs... x repeats), 55
    seg = "AAB0" + "10" + "02" + "FF" + "0064" + "00C8" + "81" * 30 + "55"
    tail = "+AAB1" + "02" + "0064" + "00C8" + "81" * 60 + "55"
    text = (seg + tail * 6 + "\n").encode()
    assert len(text) < 1023
    arr = (_lib.PulseData * 3)()
    n = L.r433_pulse_text_load(text, len(text), 250000, C.cast(arr, C.c_void_p), 3)
    assert n >= 1
    p = arr[0]
    assert p.num_pulses <= 1200
    assert p.num_pulses >= 1170  # the repeats filled it
    # nothing spilled into the estimates behind gap[] or into the next element
    assert p.ook_low_estimate == 0 and p.ook_high_estimate == 0 and p.fsk_f1_est == 0 and p.fsk_f2_est == 0
    if n == 1:  # the slot of the (empty) next package is cleared and carries the rate, nothing else
        nxt = arr[1]
        assert nxt.num_pulses == 0 and nxt.pulse[0] == 0 and nxt.gap[0] == 0 and nxt.offset == 0 and nxt.ook_low_estimate == 0
