"""CPU parity of the *kernel sources*: rtl_433_amd/csrc compiled by g++ against the lock-step wave
emulator (tests/emu) and driven through the same C ABI as the product, compared byte for byte with the
oracle and the golden vectors taken from the real reference.  No GPU involved; the product library is
not used here (tests/test_gpu_parity.py runs the same comparisons on the MI355X)."""
import json
import os
import zlib

import numpy as np
import pytest

from oracle import pyoracle as po
from rtl_433_amd import synth
from tests.cases import CASES, GOLD, fpdm_for, make_case
from tests.emu import build_emu

pytestmark = pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")

META = json.load(open(os.path.join(GOLD, "cases.json")))
# the big all-random capture makes 88 packages x 335 slicers: minutes on the emulator, covered on the GPU
EMU_CASES = [c for c in CASES if c != "random"]


def _oracle_batch(iqs, devs, cfg):
    pk_all, ev_all, base = b"", b"", 0
    for s, a in enumerate(iqs):
        o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
        pk_all += o["packages"]
        ev_all += o["events"]
        base += o["n_packages"]
    return pk_all, ev_all, base


@pytest.mark.parametrize("name", EMU_CASES)
def test_case_vs_golden_and_oracle(name, default_devices):
    from tests.emu import host
    devs = default_devices[0]
    iq, ss, rate, freq = make_case(name)
    m = META[name]
    g = host.emu_run([iq], ss, rate, devs, fpdm=fpdm_for(freq), taps=True, center_frequency=freq)
    n = iq.nbytes // ss
    env, am, fm = g["taps"]
    assert zlib.crc32(am[0, :n].tobytes()) == m["am_crc"]  # golden: the real reference's -W am.s16 / fm.s16
    assert zlib.crc32(fm[0, :n].tobytes()) == m["fm_crc"]
    pk, npk = g["packages"]
    ev, nev = g["events"]
    gold_pk = open(os.path.join(GOLD, f"case_{name}.pkg.bin"), "rb").read()
    assert npk == m["n_packages"] and po.strip_ret_pos(pk) == gold_pk
    assert nev == m["n_events"] and str(po.events_digest(ev)[0]) == m["digest"]
    k = (n + (262144 // ss) - 1) // (262144 // ss)
    assert list(g["sums"][0][:k]) == m["frame_sums"][:k]
    o = po.oracle_flow(iq, devs, po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq)), taps=True)
    assert pk == o["packages"] and ev == o["events"]
    assert np.array_equal(env[0, :n], o["env"])


@pytest.mark.parametrize("form", ["pair", "one_wave", "stretches", "strides", "strides_stretches", "skewed_shares", "skewed_one_launch", "nap_wait"])
def test_ragged_batch(form, default_devices):
    """Captures of different lengths (including 0 and non-multiples of the tile) in one launch; as producer / consumer
    wavefront pairs (what launches of up to 1280 captures use) and as single wavefronts (what larger launches use)."""
    from tests.emu import host
    devs = default_devices[0][:40]
    rng = np.random.default_rng(5)
    iqs = []
    for s in range(24):
        n = int(rng.integers(0, 30000)) if s else 0
        kind = s % 4
        if kind == 0:
            a = synth.random_cu8(1000 + s, min(n, 6000))
        elif kind == 1:
            a = synth.noise_cu8(1000 + s, n, 4.0)
        else:
            a = synth.ook_stream(1000 + s, max(n, 1))[0][: 2 * n]
        iqs.append(a)
    g = host.emu_run(iqs, 2, 250000, devs, debug={"one_wave": 4096, "pair": 32768, "stretches": 8, "strides": 65536, "strides_stretches": 65536 | 8,
                                                       "skewed_shares": 1048576, "skewed_one_launch": 1048576 | 131072, "nap_wait": 2097152}[form])  # R433_DEBUG_ONE_WAVE / _PAIR / _SMALL_STRETCH / _STATIC_SLICE / _SKEW_SLICE / _ONE_SLICE_LAUNCH / _NAP_WAIT
    pk, ev, base = _oracle_batch(iqs, devs, po.default_flow_cfg(2, 250000, fpdm=0))
    assert g["packages"][1] == base and g["packages"][0] == pk and g["events"][0] == ev


@pytest.mark.parametrize("variant", ["nofm", "magest", "fixed_level", "lowpass", "slow_lowpass", "short_frames"])
def test_flow_options(variant, default_devices):
    from tests.emu import host
    devs = default_devices[0][:60]
    kw, enable_fm = {}, 1
    if variant == "nofm":
        enable_fm = 0
        devs = devs[devs["modulation"] < 16]
    elif variant == "magest":
        kw = dict(use_mag_est=1)
    elif variant == "fixed_level":
        kw = dict(level_limit_db=-10.0)
    elif variant == "lowpass":
        kw = dict(fm_low_pass=0.15)
    elif variant == "slow_lowpass":  # feedback 0.98: the 96-sample warm-up cannot collapse, everything is resolved serially
        kw = dict(fm_low_pass=0.006)
    elif variant == "short_frames":  # frame boundaries inside tiles and chunks' neighbours
        kw = dict(frame_samples=192)
    iqs = [synth.ook_stream(40 + k, 30000)[0] for k in range(2)] + [synth.fsk_stream_cu8(50, 40000), synth.random_cu8(51, 5000)]
    g = host.emu_run(iqs, 2, 250000, devs, enable_fm=enable_fm, taps=True, **kw)
    okw = dict(kw)
    cfg = po.default_flow_cfg(2, 250000, fpdm=0, enable_fm=enable_fm, **okw)
    pk, ev, base = _oracle_batch(iqs, devs, cfg)
    for s, a in enumerate(iqs):
        o = po.oracle_flow(a, devs, cfg, taps=True)
        n = a.nbytes // 2
        assert np.array_equal(g["taps"][1][s, :n], o["am"]), f"am differs, capture {s}"
        assert np.array_equal(g["taps"][2][s, :n], o["fm"]), f"fm differs, capture {s}"
    assert g["packages"][0] == pk and g["events"][0] == ev


def test_stalled_and_saturated_filters():
    """Constant input parks the truncating low-passes on one of several fixed points (which one depends
    on the history); full-scale input makes the AM filter's int16 x[-1] slot wrap at frame starts."""
    from tests.emu import host
    rng = np.random.default_rng(11)
    n = 3 * 2048 + 777
    parts = []
    for v in (128, 130, 255, 0, 127):
        parts.append(np.full(2 * n, v, dtype=np.uint8))
        burst = rng.integers(0, 256, 2 * 300, dtype=np.uint8)  # leaves the filters on a different fixed point each time
        parts.append(burst)
    iq = np.concatenate(parts)
    sat = np.full(2 * 5000, 255, dtype=np.uint8)
    sat[::7] = 0
    for frame_samples in (None, 64, 2048):
        kw = {} if frame_samples is None else dict(frame_samples=frame_samples)
        g = host.emu_run([iq, sat], 2, 250000, None, taps=True, **kw)
        cfg = po.default_flow_cfg(2, 250000, **kw)
        for s, a in enumerate([iq, sat]):
            o = po.oracle_flow(a, None, cfg, taps=True)
            m = a.nbytes // 2
            assert np.array_equal(g["taps"][0][s, :m], o["env"])
            assert np.array_equal(g["taps"][1][s, :m], o["am"])
            assert np.array_equal(g["taps"][2][s, :m], o["fm"])
            k = len(o["frame_sums"]) - 1
            assert list(g["sums"][s][:k]) == list(o["frame_sums"][:k])
        pk, ev, base = _oracle_batch([iq, sat], None, cfg)
        assert g["packages"][0] == pk


def test_cs16_stalls_and_ragged():
    from tests.emu import host
    rng = np.random.default_rng(12)
    a = np.zeros(2 * 9000, dtype=np.int16)
    a[2 * 3000:2 * 3400] = rng.integers(-20000, 20000, 800)
    b = synth.fsk_stream_cs16(8, 20011)
    c = np.full(2 * 4100, -32768, dtype=np.int16)
    for fpdm in (0, 1):
        g = host.emu_run([a, b, c], 4, 1024000, None, fpdm=fpdm, taps=True)
        cfg = po.default_flow_cfg(4, 1024000, fpdm=fpdm)
        for s, x in enumerate([a, b, c]):
            o = po.oracle_flow(x, None, cfg, taps=True)
            m = x.nbytes // 4
            assert np.array_equal(g["taps"][1][s, :m], o["am"])
            assert np.array_equal(g["taps"][2][s, :m], o["fm"])
        pk, ev, base = _oracle_batch([a, b, c], None, cfg)
        assert g["packages"][0] == pk


def test_noise_floor_tracking():
    """The idle noise-floor estimate is evaluated lazily (parity + two-sided walk over the last 128
    samples, full walk when the two sides do not meet, plain walk when |am - low| may reach 1024).
    Bursts riding on noise floors of very different widths exercise all three, and the estimate is
    visible in every package header (ook_low_estimate) and in every threshold decision."""
    from tests.emu import host
    iqs = []
    for k, sigma in enumerate((0.0, 1.0, 3.0, 6.0, 9.0, 12.0, 16.0, 24.0)):
        rng = np.random.default_rng(700 + k)
        segs = [(6000 + 777 * k, False)]
        for rep in range(3):
            bits = rng.integers(0, 2, 24).astype(np.uint8)
            segs += synth.ook_segments(bits, "pwm", 100, 200, repeats=1) + [(9000 + 1111 * rep, False)]
        n = sum(x[0] for x in segs)
        mask = synth._segments_to_mask(segs, n)
        iqs.append(synth.modulate_cu8(mask, rng, 250000, 20e3, 110.0, sigma))
    g = host.emu_run(iqs, 2, 250000, None, taps=False)
    cfg = po.default_flow_cfg(2, 250000)
    pk, ev, base = _oracle_batch(iqs, None, cfg)
    assert base >= 8
    assert g["packages"][0] == pk


def test_autolevel(default_devices):
    """-Y autolevel: per-frame detection level from the running noise estimate (src/r_flow.c:166-186)."""
    from tests.cases import autolevel_capture
    from tests.emu import host
    devs = default_devices[0][:30]
    iq = autolevel_capture()
    cfg_on = po.default_flow_cfg(2, 250000, auto_level=1.0)
    o_on = po.oracle_flow(iq, devs, cfg_on)
    o_off = po.oracle_flow(iq, devs, po.default_flow_cfg(2, 250000))
    assert o_on["n_packages"] > o_off["n_packages"], "the capture must need autolevel to be decoded"
    g = host.emu_run([iq, iq[: 2 * 200000]], 2, 250000, devs, auto_level=1.0)
    o2 = po.oracle_flow(iq[: 2 * 200000], devs, cfg_on, stream_index=1, pkg_base=o_on["n_packages"])
    assert g["packages"][0] == o_on["packages"] + o2["packages"]
    assert g["events"][0] == o_on["events"] + o2["events"]


def test_mixed_2000k_autolevel_filter(default_devices):
    """BASELINE config 5 in miniature: 2 MS/s, OOK + FSK bursts, noise floor step, -Y autolevel, -Y filter."""
    from tests.cases import mixed_2000k_capture
    from tests.emu import host
    devs = default_devices[0][::4]
    iq, rate = mixed_2000k_capture()
    kw = dict(auto_level=1.0, fm_low_pass=0.15)
    cfg = po.default_flow_cfg(2, rate, fpdm=0, **kw)
    o = po.oracle_flow(iq, devs, cfg, taps=True)
    types = [p["type"] for p in po.parse_packages(o["packages"])]
    assert 1 in types and 2 in types, "the capture must produce both OOK and FSK packages"
    g = host.emu_run([iq], 2, rate, devs, taps=True, **kw)
    n = iq.nbytes // 2
    assert np.array_equal(g["taps"][1][0, :n], o["am"]) and np.array_equal(g["taps"][2][0, :n], o["fm"])
    assert g["packages"][0] == o["packages"] and g["events"][0] == o["events"]


def _cf32_to_cs16_like_c(f):
    """(int)(f * INT16_MAX) clamped to +-INT16_MAX as the reference's file loop does it on x86 (src/rtl_433.c:1814-1824)."""
    p = f.astype(np.float32) * np.float32(32767.0)
    bad = ~np.isfinite(p) | (p >= np.float32(2147483648.0)) | (p < np.float32(-2147483648.0))
    s = np.where(bad, np.int64(-2147483648), np.trunc(np.where(bad, 0, p)).astype(np.int64))
    return np.clip(s, -32767, 32767).astype(np.int16)


def test_input_formats_cs8_cf32(default_devices):
    """cs8 and cf32 inputs are converted on the device like the reference converts them on load."""
    from tests.emu import host
    devs = default_devices[0][:30]
    cu8 = [synth.ook_stream(60, 30000)[0], synth.fsk_stream_cu8(61, 20001)]
    cs8 = [(a ^ 0x80) for a in cu8]  # the int8 file a cs8 recorder would have written
    cfg = po.default_flow_cfg(2, 250000)
    pk, ev, _ = _oracle_batch(cu8, devs, cfg)
    g = host.emu_run(cs8, 2, 250000, devs, input_format=1)
    assert g["packages"][0] == pk and g["events"][0] == ev

    rng = np.random.default_rng(62)
    cs16 = synth.fsk_stream_cs16(63, 30000)
    f = (cs16.astype(np.float32) / np.float32(32767.0)) * np.float32(1.3)  # some samples clip
    f[100:110] = [np.nan, np.inf, -np.inf, 1e20, -1e20, 65536.5, -65536.5, 1.0, -1.0, 0.99999]
    f += rng.normal(0, 1e-6, f.size).astype(np.float32)
    want16 = _cf32_to_cs16_like_c(f)
    cfg4 = po.default_flow_cfg(4, 1024000, fpdm=1)
    pk, ev, _ = _oracle_batch([want16], devs, cfg4)
    g = host.emu_run([f.view(np.uint8)], 4, 1024000, devs, fpdm=1, center_frequency=868000000, input_format=2, taps=True)
    o = po.oracle_flow(want16, devs, cfg4, taps=True)
    n = want16.size // 2
    assert np.array_equal(g["taps"][0][0, :n], o["env"])
    assert g["packages"][0] == pk and g["events"][0] == ev


def test_am_s16_fm_s16_input_files():
    """am.s16 / fm.s16 pseudo-IQ input (R433_IN_S16_AM / R433_IN_S16_FM) against the unmodified reference, on the emulator."""
    if not po.have_ref():
        pytest.skip("oracle/_ref/libr433ref.so not present")
    from tests.emu import host
    from tests import s16_input_case
    s16_input_case.check(lambda iq, devs, fmt, fm: host.emu_run(iq, 2, 250000, devs, fpdm=0, taps=True, enable_fm=fm, input_format=fmt))


def test_oracle_am_s16_fm_s16_against_reference():
    """the restatement's load_format (what the fuzzer checks such input against) pinned to the unmodified reference"""
    if not po.have_ref():
        pytest.skip("oracle/_ref/libr433ref.so not present")
    from tests import s16_input_case
    for fmt, words in s16_input_case.captures():
        devs, pk, npk, ev, nev, taps = s16_input_case.reference_records(fmt, words, 1)
        cfg = po.default_flow_cfg(2, 250000, fpdm=0, load_format=fmt)
        opk, oev, base = b"", b"", 0
        for s, w in enumerate(words):
            o = po.oracle_flow(w.view(np.uint8), devs, cfg, stream_index=s, pkg_base=base, taps=True)
            if w.size:
                assert np.array_equal(o["am"], taps[s][0]) and np.array_equal(o["fm"], taps[s][1])
            opk += o["packages"]
            oev += o["events"]
            base += o["n_packages"]
        assert base == npk and po.strip_ret_pos(opk) == pk
        assert po.events_normalize(oev) == po.events_normalize(po.canonical_events(ev))


def test_heaviest_captures_first(default_devices):
    """Large grids hand their captures out heaviest first (k_capture_weight + k_order_by_weight, forced here for a small one):
    every capture still lands in its own slot -- records and taps as in capture order, for both workgroup forms."""
    from tests.emu import host
    devs = default_devices[0][:30]
    caps = [synth.noise_cu8(81, 30000, 2.0), synth.ook_stream(82, 65536)[0], np.zeros(0, dtype=np.uint8), synth.ook_stream(83, 20000)[0],
            synth.fsk_stream_cu8(84, 50001), synth.ook_stream(85, 65536)[0], synth.random_cu8(86, 777)]
    cfg = po.default_flow_cfg(2, 250000)
    pk, ev, _ = _oracle_batch(caps, devs, cfg)
    for flags in (128, 128 | 4096, 128 | 32768):
        g = host.emu_run(caps, 2, 250000, devs, taps=True, debug=flags)
        assert g["packages"][0] == pk and g["events"][0] == ev, flags
    plain = host.emu_run(caps, 2, 250000, devs, taps=True)
    assert all(np.array_equal(a, b) for a, b in zip(g["taps"], plain["taps"]))


@pytest.mark.parametrize("debug", [0, 1048576])  # R433_DEBUG_SKEW_SLICE
def test_more_than_sixteen_chunks_of_devices(debug, default_devices):
    """1125 decoders -- the default 335 three times over and 120 more: 13 line codes, more than the sixteen chunks of 64 the
    sizing pass keeps shares and measurements for; OOK and FSK captures; records == the oracle's."""
    from tests.emu import host
    devs = default_devices[0]
    many = np.concatenate([devs, devs, devs, devs[:120]])
    iqs = [synth.ook_stream(4000 + k, 24000)[0] for k in range(3)] + [synth.fsk_stream_cu8(4010, 20000)]
    g = host.emu_run(iqs, 2, 250000, many, debug=debug)
    pk, ev, base = _oracle_batch(iqs, many, po.default_flow_cfg(2, 250000, fpdm=0))
    assert base >= 4 and g["packages"][1] == base and g["packages"][0] == pk and g["events"][0] == ev
