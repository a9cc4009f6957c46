"""Generates tests/golden/slicer_matrix.json from the REAL reference (oracle/_ref): every line-code slicer of
src/pulse_slicer.c -- including pulse_slicer_piwm_raw (:597-657) and pulse_slicer_nrzs (:715-759), which no
default-enabled protocol uses -- run through the reference's own fan-out with the synthetic decoder rows of
tests/cases.py::slicer_matrix_rows.  TEST INFRASTRUCTURE; run in the build container only:

    python -m oracle.gen_slicer_golden
"""
from __future__ import annotations

import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
from tests.cases import GOLD, SLICER_CASES, make_slicer_case, slicer_matrix_rows  # noqa: E402


def main():
    if not po.build_ref():
        raise SystemExit("reference not available")
    rows = slicer_matrix_rows()
    ref = po.Ref(protocols=[])  # no protocol of its own: only the synthetic rows
    ref.add_rows(rows)
    devs, _, _ = ref.devices()
    assert devs.tobytes() == rows.tobytes()
    meta = dict(rows_crc=zlib.crc32(rows.tobytes()), n_rows=len(rows), cases={})
    for name in SLICER_CASES:
        iq, ss, rate, freq = make_slicer_case(name)
        ref.clear()
        ref.run(iq, ss, rate, freq, fpdm=2)
        ev, nev = ref.events()
        dg, dne, npk = ref.digest()
        per_mod = {}
        for e in po.parse_events(ev):
            m = int(rows["modulation"][e["dev"]])
            per_mod[m] = per_mod.get(m, 0) + 1
        meta["cases"][name] = dict(iq_crc=zlib.crc32(iq.tobytes()), n_packages=npk, n_events=nev, digest=str(dg),
                                   events_per_modulation={str(k): v for k, v in sorted(per_mod.items())})
        print(name, npk, nev, per_mod)
    ref.close()
    json.dump(meta, open(os.path.join(GOLD, "slicer_matrix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
