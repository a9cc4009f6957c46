"""Split captures (r433_batch_set_split): several wavefronts per long capture, speculative cuts verified at
the stitch.  Whatever the cut positions -- chosen where the signal looks idle, or blindly (R433_DEBUG_SPLIT_BLIND,
most cuts then fail and are dropped) -- the result must be byte-identical to the unsplit run and the oracle."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from rtl_433_amd import synth
from tests.emu import build_emu

pytestmark = pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")


def long_capture(seed, n_bursts=8, sigma=2.0, gap=(30000, 60000), rate=250000):
    rng = np.random.default_rng(seed)
    segs = [(20000, False)]
    for k in range(n_bursts):
        bits = rng.integers(0, 2, 32).astype(np.uint8)
        segs += synth.ook_segments(bits, ["pwm", "ppm", "mc"][k % 3], 100, 200, repeats=1) + [(int(rng.integers(*gap)), False)]
    n = sum(s[0] for s in segs)
    mask = synth._segments_to_mask(segs, n)
    return synth.modulate_cu8(mask, rng, rate, 25e3, float(rng.uniform(50, 110)), sigma)


def _oracle(caps, devs, cfg):
    pk, ev, base = b"", b"", 0
    for s, a in enumerate(caps):
        o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
        pk += o["packages"]
        ev += o["events"]
        base += o["n_packages"]
    return pk, ev, base


@pytest.mark.parametrize("blind", [False, True])
@pytest.mark.parametrize("split", [8192, 40000])
def test_split_cu8(split, blind, default_devices):
    from tests.emu import host
    devs = default_devices[0][:40]
    caps = [long_capture(1), long_capture(2, sigma=1.0), long_capture(3, sigma=0.0), synth.noise_cu8(5, 150000, 3.0),
            synth.ook_stream(7, 40000)[0]]
    cfg = po.default_flow_cfg(2, 250000)
    pk, ev, base = _oracle(caps, devs, cfg)
    g = host.emu_run(caps, 2, 250000, devs, split=split, taps=True, debug=1 if blind else 0)
    assert g["split"]["segments"] > len(caps), "nothing was split"
    assert g["packages"][0] == pk and g["events"][0] == ev
    o0 = po.oracle_flow(caps[0], None, cfg, taps=True)
    n0 = caps[0].nbytes // 2
    assert np.array_equal(g["taps"][1][0, :n0], o0["am"]) and np.array_equal(g["taps"][2][0, :n0], o0["fm"])
    for s, a in enumerate(caps):  # frame sums of split captures come from their own pass
        k = (a.nbytes // 2 + 131071) // 131072
        assert list(g["sums"][s][:k]) == list(po.oracle_flow(a, None, cfg)["frame_sums"][:k])
    if not blind:  # cuts at quiet places mostly verify: only the noiseless capture cannot be cut at all
        assert g["split"]["pieces_rerun"] <= 8


def test_split_cs16_fsk_and_autolevel(default_devices):
    from tests.cases import autolevel_capture
    from tests.emu import host
    devs = default_devices[0][:40]
    # cs16 FSK bursts far enough apart to cut between them (1024 kS/s: 25 ms of quiet = 13 tiles)
    rng = np.random.default_rng(9)
    parts = []
    for k in range(4):
        parts.append(synth.fsk_stream_cs16(20 + k, 60000))
        parts.append((rng.normal(0, 60, 2 * 70000)).astype(np.int16))
    cs = np.concatenate(parts)
    cfg = po.default_flow_cfg(4, 1024000, fpdm=1)
    pk, ev, base = _oracle([cs], devs, cfg)
    g = host.emu_run([cs], 4, 1024000, devs, fpdm=1, center_frequency=868000000, split=65536)
    assert base >= 4 and g["split"]["segments"] > 1
    assert g["packages"][0] == pk and g["events"][0] == ev
    # -Y autolevel: the level changes from frame to frame, segments start inside frames
    iq = autolevel_capture()
    cfg = po.default_flow_cfg(2, 250000, auto_level=1.0)
    pk, ev, base = _oracle([iq], devs, cfg)
    g = host.emu_run([iq], 2, 250000, devs, split=50000, auto_level=1.0)
    assert g["split"]["segments"] > 1 and g["packages"][0] == pk and g["events"][0] == ev


def test_every_variant_is_run_in_every_form(default_devices):
    """The odd variant of a piece sits in the slot behind its twin: a launch of one-wavefront workgroups (R433_DEBUG_ONE_WAVE)
    must run it as a workgroup of its own.  (It did not once: the stitch then read whatever the slot's memory held -- on the
    emulator poison, which reads as "failed" and only costs rounds; on the GPU the state of an earlier run.)  Same records and
    the same number of pieces run again in all three forms."""
    from tests.emu import host
    devs = default_devices[0][:40]
    caps = [long_capture(11), long_capture(12, sigma=1.0)]
    cfg = po.default_flow_cfg(2, 250000)
    pk, ev, base = _oracle(caps, devs, cfg)
    reruns = {}
    for name, flag in (("triple", 0), ("one wavefront", 4096), ("pair", 32768)):
        g = host.emu_run(caps, 2, 250000, devs, split=8192, debug=flag)
        assert g["packages"][0] == pk and g["events"][0] == ev, name
        reruns[name] = g["split"]["pieces_rerun"]
    assert reruns["one wavefront"] == reruns["triple"] == reruns["pair"], reruns
