"""Every line-code slicer x random timings, pinned to the REAL reference: tests/golden/slicer_matrix.json holds the
event digests the unmodified reference produced (oracle/gen_slicer_golden.py) when its fan-out ran the synthetic
decoder rows of tests/cases.py::slicer_matrix_rows -- all 13 modulations, among them OOK_PULSE_PIWM_RAW and
OOK_PULSE_NRZS (reference src/pulse_slicer.c:597-657,715-759), which no default-enabled protocol uses.

  CPU : the oracle's events against those digests (and against the reference itself where oracle/_ref is built)
  GPU : the HIP fan-out byte for byte against the oracle, and against the reference's digests
"""
import json
import os
import zlib

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.cases import GOLD, SLICER_CASES, fpdm_for, make_slicer_case, slicer_matrix_rows

META = json.load(open(os.path.join(GOLD, "slicer_matrix.json")))
ROWS = slicer_matrix_rows()


def test_rows_are_the_golden_rows():
    assert len(ROWS) == META["n_rows"] and zlib.crc32(ROWS.tobytes()) == META["rows_crc"]
    seen = {int(k) for c in META["cases"].values() for k in c["events_per_modulation"]}
    assert {8, 12} <= seen, "PIWM_RAW / NRZS must be exercised"
    assert seen >= {3, 4, 5, 6, 8, 9, 11, 12, 13, 16, 17, 18}


@pytest.mark.parametrize("name", SLICER_CASES)
def test_oracle_vs_reference(name):
    iq, ss, rate, freq = make_slicer_case(name)
    m = META["cases"][name]
    assert zlib.crc32(iq.tobytes()) == m["iq_crc"]
    o = po.oracle_flow(iq, ROWS, po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq)))
    assert o["n_packages"] == m["n_packages"]
    assert o["n_events"] == m["n_events"]
    assert str(po.events_digest(o["events"])[0]) == m["digest"]
    per_mod = {}
    for e in po.parse_events(o["events"]):
        k = str(int(ROWS["modulation"][e["dev"]]))
        per_mod[k] = per_mod.get(k, 0) + 1
    assert per_mod == m["events_per_modulation"]


@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built")
def test_reference_live_matches_golden():
    """Where the reference library is present, run it again: the committed digests are what it says today."""
    ref = po.Ref(protocols=[])
    ref.add_rows(ROWS)
    for name in ("ook22", "nrz", "fsk_cu8"):
        iq, ss, rate, freq = make_slicer_case(name)
        ref.clear()
        ref.run(iq, ss, rate, freq, fpdm=2)
        dg, nev, npk = ref.digest()
        assert (str(dg), nev, npk) == (META["cases"][name]["digest"], META["cases"][name]["n_events"], META["cases"][name]["n_packages"])
    ref.close()


@pytest.mark.gpu
def test_gpu_slicer_matrix():
    from tests.test_gpu_parity import _gpu_run
    by_cfg = {}
    for name in SLICER_CASES:
        iq, ss, rate, freq = make_slicer_case(name)
        by_cfg.setdefault((ss, rate, freq), []).append((name, iq))
    hit = set()
    for (ss, rate, freq), items in by_cfg.items():
        g = _gpu_run([iq for _, iq in items], ss, rate, freq, ROWS)
        cfg = po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq))
        pk_all, ev_all, base = b"", b"", 0
        per_case = []
        for s, (name, iq) in enumerate(items):
            o = po.oracle_flow(iq, ROWS, cfg, stream_index=s, pkg_base=base)
            pk_all += o["packages"]
            ev_all += o["events"]
            per_case.append((name, base, o["n_packages"]))
            base += o["n_packages"]
        assert g["packages"][0] == pk_all
        assert g["events"][0] == ev_all
        # ... and the GPU's own records against the reference's digests, capture by capture
        evs = po.parse_events(g["events"][0])
        for e in evs:
            hit.add(int(ROWS["modulation"][e["dev"]]))
        blob = g["events"][0]
        at = 0
        chunks = {name: bytearray() for name, _, _ in per_case}
        while at < len(blob):
            total = int.from_bytes(blob[at:at + 4], "little")
            pkg = int.from_bytes(blob[at + 4:at + 8], "little")
            for name, b0, n in per_case:
                if b0 <= pkg < b0 + n:
                    rec = bytearray(blob[at:at + total])
                    rec[4:8] = (pkg - b0).to_bytes(4, "little")  # the reference numbered each capture from 0
                    chunks[name] += rec
            at += total
        for name, _, _ in per_case:
            m = META["cases"][name]
            d, n = po.events_digest(bytes(chunks[name]))
            assert (str(d), n) == (m["digest"], m["n_events"]), name
    assert {8, 12} <= hit


def test_emu_slicer_matrix():
    """The same fan-out kernels on the CPU wave emulator (tests/emu): slicer_device.hpp against the oracle byte for
    byte before a GPU is involved -- a slice of the matrix that runs in seconds."""
    from tests.emu import build_emu, host
    if not build_emu.available():
        pytest.skip("emulator needs x86-64")
    hit = set()
    for names in (("ook22", "nrz"), ("fsk_cu8",)):
        items = [(n, *make_slicer_case(n)) for n in names]
        ss, rate, freq = items[0][2:]
        g = host.emu_run([it[1] for it in items], ss, rate, ROWS, fpdm=fpdm_for(freq), center_frequency=freq)
        cfg = po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq))
        pk_all, ev_all, base = b"", b"", 0
        for s, it in enumerate(items):
            o = po.oracle_flow(it[1], ROWS, cfg, stream_index=s, pkg_base=base)
            pk_all += o["packages"]
            ev_all += o["events"]
            base += o["n_packages"]
        assert g["packages"][0] == pk_all
        assert g["events"][0] == ev_all
        hit |= {int(ROWS["modulation"][e["dev"]]) for e in po.parse_events(ev_all)}
    assert {8, 12} <= hit
