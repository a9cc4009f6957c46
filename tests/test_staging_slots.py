"""The slicers build a (package, device)'s records in a staging slot and a placing pass copies them into the dense event stream;
a record that outgrows its slot is sliced a second time straight into the stream (csrc/slicer_kernels.hip M_COMPACT).  With the
default 8 KB slots that never happens -- a full bitbuffer is 50 rows x 132 bytes --; hosts that live as long as one file list ask
for smaller ones (r433_batch_set_staging_slot: the drop-in CLI and the C pipeline host use 2 KB).  Whatever the slot, the event
records are the oracle's, byte for byte (reference src/pulse_slicer.c:68-918, src/bitbuffer.c:17-133: what a slicer writes does
not depend on where it is written)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from rtl_433_amd import synth
from rtl_433_amd.engine import BatchEngine, flow_cfg
from tests.emu import build_emu

BACKENDS = [pytest.param("emu", marks=pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")),
            pytest.param("gpu", marks=pytest.mark.gpu)]


def _slot_bytes(blob):
    """bytes of records per (package, device): what one staging slot has to hold"""
    import collections
    out, at = collections.Counter(), 0
    while at + 16 <= len(blob):
        total = int.from_bytes(blob[at:at + 4], "little")
        out[(int.from_bytes(blob[at + 4:at + 8], "little"), int.from_bytes(blob[at + 8:at + 10], "little"))] += total
        at += total
    return list(out.values())


@pytest.mark.parametrize("backend", BACKENDS)
def test_every_slot_size_gives_the_oracles_records(backend, default_devices):
    devs = default_devices[0]
    import bench
    n = 3 if backend == "emu" else 24
    # bursts of every family, protocol transmissions (long packages: rows of thousands of bits under the PCM slicers), noise
    iqs = [synth.ook_stream(500 + k)[0] for k in range(n)] + [bench._synth_one(3 * k + 2) for k in range(n)] + [synth.random_cu8(9, 20000)]
    cfg = po.default_flow_cfg(2, 250000, fpdm=0)
    want, base = b"", 0
    for s, a in enumerate(iqs):
        o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
        want += o["events"]
        base += o["n_packages"]
    sizes = _slot_bytes(want)
    assert sum(1 for x in sizes if x > 2048) > 10 and sum(1 for x in sizes if x > 512) > 100, (max(sizes), len(sizes))  # the captures do outgrow the small slots
    for slot in (0, 8192, 2048, 512):
        if backend == "gpu":
            eng = BatchEngine(flow_cfg(2, 250000), devs)
        else:
            from tests.emu import host
            eng = BatchEngine(flow_cfg(2, 250000), devs, library=host.emu_lib())
        eng.set_staging_slot(slot)
        assert eng.run_host(iqs) == base
        got = bytes(eng.events()[0])
        eng.close()
        assert got == want, f"staging slots of {slot or 8192} bytes"
    with pytest.raises(Exception):
        eng = BatchEngine(flow_cfg(2, 250000), devs, library=None if backend == "gpu" else __import__("tests.emu.host", fromlist=["x"]).emu_lib())
        try:
            eng.set_staging_slot(100)
        finally:
            eng.close()
