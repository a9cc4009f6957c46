"""GPU parity: the HIP path (through the C ABI) against the oracle and the committed golden
vectors taken from the real reference.  Bit-exact: package records and event records are compared
byte for byte, the per-sample taps sample for sample."""
import json
import os
import zlib

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.cases import CASES, GOLD, fpdm_for, make_case

pytestmark = pytest.mark.gpu

META = json.load(open(os.path.join(GOLD, "cases.json")))


def _gpu_run(iq_list, ss, rate, freq, devs, taps=False, enable_fm=1, debug=0, **kw):
    import torch
    from rtl_433_amd.engine import BatchEngine, flow_cfg
    n = len(iq_list)
    lens = np.array([a.nbytes for a in iq_list], dtype=np.uint32)
    stride = max(16, int((lens.max() + 15) // 16 * 16)) if n else 16
    host = np.zeros((n, stride), dtype=np.uint8)
    for i, a in enumerate(iq_list):
        host[i, :a.nbytes] = a.view(np.uint8)
    dev = torch.from_numpy(host).cuda()
    cfg = flow_cfg(ss, rate, fpdm=fpdm_for(freq), enable_fm=enable_fm, center_frequency=freq, **kw)
    eng = BatchEngine(cfg, devs, profiling=True)
    tp = None
    if taps:
        eng.enable_taps(n, max(1, stride // ss))
    if debug:
        eng.set_debug(debug)
    npk = eng.run(dev, lens)
    out = dict(n_packages=npk, packages=eng.packages(), events=eng.events(), sums=eng.frame_sums(n), timing=eng.timing())
    if taps:
        out["taps"] = eng.taps()
    eng.close()
    return out


@pytest.mark.parametrize("name", CASES)
def test_case_vs_golden_and_oracle(name, default_devices):
    devs = default_devices[0]
    iq, ss, rate, freq = make_case(name)
    m = META[name]
    assert zlib.crc32(iq.tobytes()) == m["iq_crc"], "case generator drifted from the golden fixtures"
    g = _gpu_run([iq], ss, rate, freq, devs, taps=True)
    n = iq.nbytes // ss
    env, am, fm = g["taps"]
    # golden (real reference) taps
    assert zlib.crc32(am[0, :n].tobytes()) == m["am_crc"]
    assert zlib.crc32(fm[0, :n].tobytes()) == m["fm_crc"]
    # golden packages / events
    gold_pk = open(os.path.join(GOLD, f"case_{name}.pkg.bin"), "rb").read()
    pk, npk = g["packages"]
    ev, nev = g["events"]
    assert npk == m["n_packages"]
    assert po.strip_ret_pos(pk) == gold_pk
    assert nev == m["n_events"]
    assert str(po.events_digest(ev)[0]) == m["digest"]
    nf = len([s for s in m["frame_sums"]])
    got = g["sums"][0]
    k = (n + (262144 // ss) - 1) // (262144 // ss)
    assert list(got[:k]) == m["frame_sums"][:k]
    # oracle, byte for byte (includes ret_pos and the extent-based row sizes)
    o = po.oracle_flow(iq, devs, po.default_flow_cfg(ss, rate, fpdm=fpdm_for(freq)), taps=True)
    assert pk == o["packages"]
    assert ev == o["events"]
    assert np.array_equal(env[0, :n], o["env"])


@pytest.mark.parametrize("form", ["pair", "one_wave", "stretches", "strides", "strides_stretches", "skewed_shares", "skewed_one_launch", "nap_wait"])
def test_ragged_batch_vs_oracle(form, default_devices):
    """Many captures of different lengths in one launch; every capture must match the oracle run alone.  Both forms of the
    detection kernel: producer / consumer wavefront pairs (launches of up to 1280 captures) and single wavefronts (larger ones)."""
    from rtl_433_amd import synth
    devs = default_devices[0]
    rng = np.random.default_rng(5)
    iqs = []
    for s in range(150):
        n = int(rng.integers(0, 90000))
        kind = s % 5
        if kind == 0:
            a = synth.random_cu8(1000 + s, n)
        elif kind == 1:
            a = synth.noise_cu8(1000 + s, n, 4.0)
        else:
            a = synth.ook_stream(1000 + s, max(n, 1))[0][: 2 * n]
        iqs.append(a)
    g = _gpu_run(iqs, 2, 250000, 433920000, devs, debug={"one_wave": 4096, "pair": 32768, "stretches": 8, "strides": 65536, "strides_stretches": 65536 | 8,
                                                       "skewed_shares": 1048576, "skewed_one_launch": 1048576 | 131072, "nap_wait": 2097152}[form])  # R433_DEBUG_ONE_WAVE / _PAIR / _SMALL_STRETCH / _STATIC_SLICE / _SKEW_SLICE / _ONE_SLICE_LAUNCH / _NAP_WAIT
    cfg = po.default_flow_cfg(2, 250000, fpdm=0)
    pk_all, ev_all, base = b"", b"", 0
    for s, a in enumerate(iqs):
        o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
        pk_all += o["packages"]
        ev_all += o["events"]
        base += o["n_packages"]
    assert g["packages"][1] == base
    assert g["packages"][0] == pk_all
    assert g["events"][0] == ev_all


@pytest.mark.parametrize("variant", ["nofm", "magest", "fixed_level", "lowpass"])
def test_flow_options_vs_oracle(variant, default_devices):
    from rtl_433_amd import synth
    devs = default_devices[0]
    kw = {}
    okw = {}
    enable_fm = 1
    if variant == "nofm":
        enable_fm = 0
        devs = devs[devs["modulation"] < 16]
    elif variant == "magest":
        kw = dict(use_mag_est=1)
    elif variant == "fixed_level":
        kw = dict(level_limit_db=-10.0)
    elif variant == "lowpass":
        kw = dict(fm_low_pass=0.15)
    iqs = [synth.ook_stream(40 + k)[0] for k in range(4)] + [synth.fsk_stream_cu8(50, 100000), synth.random_cu8(51, 50000)]
    g = _gpu_run(iqs, 2, 250000, 433920000, devs, enable_fm=enable_fm, **kw)
    cfg = po.default_flow_cfg(2, 250000, fpdm=0, enable_fm=enable_fm, **kw)
    pk_all, ev_all, base = b"", b"", 0
    for s, a in enumerate(iqs):
        o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
        pk_all += o["packages"]
        ev_all += o["events"]
        base += o["n_packages"]
    assert g["packages"][0] == pk_all
    assert g["events"][0] == ev_all


def test_noise_floor_tracking():
    """Same captures as tests/test_emu_parity.py::test_noise_floor_tracking, on the GPU."""
    from rtl_433_amd import synth
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
    g = _gpu_run(iqs, 2, 250000, 433920000, None)
    cfg = po.default_flow_cfg(2, 250000)
    pk_all, base = b"", 0
    for s, a in enumerate(iqs):
        o = po.oracle_flow(a, None, cfg, stream_index=s, pkg_base=base)
        pk_all += o["packages"]
        base += o["n_packages"]
    assert base >= 8 and g["packages"][0] == pk_all


def test_autolevel(default_devices):
    """-Y autolevel on the GPU: frame-sum pre-pass, host level recurrence, per-frame detection level."""
    from tests.cases import autolevel_capture
    devs = default_devices[0]
    iq = autolevel_capture()
    g = _gpu_run([iq, iq[: 2 * 200000]], 2, 250000, 433920000, devs, auto_level=1.0)
    cfg = po.default_flow_cfg(2, 250000, auto_level=1.0)
    o1 = po.oracle_flow(iq, devs, cfg)
    o2 = po.oracle_flow(iq[: 2 * 200000], devs, cfg, stream_index=1, pkg_base=o1["n_packages"])
    assert o1["n_packages"] >= 4
    assert g["packages"][0] == o1["packages"] + o2["packages"]
    assert g["events"][0] == o1["events"] + o2["events"]


def test_mixed_2000k_autolevel_filter(default_devices):
    """BASELINE config 5 in miniature on the GPU (see tests/test_emu_parity.py)."""
    from tests.cases import mixed_2000k_capture
    devs = default_devices[0]
    iq, rate = mixed_2000k_capture()
    kw = dict(auto_level=1.0, fm_low_pass=0.15)
    o = po.oracle_flow(iq, devs, po.default_flow_cfg(2, rate, fpdm=0, **kw))
    g = _gpu_run([iq], 2, rate, 433920000, devs, **kw)
    assert g["packages"][0] == o["packages"] and g["events"][0] == o["events"]


@pytest.mark.parametrize("blind", [False, True])
def test_split_captures(blind, default_devices):
    """Several wavefronts per capture (r433_batch_set_split): byte-identical to the oracle whatever the cuts."""
    import torch
    from rtl_433_amd.engine import BatchEngine, flow_cfg
    from tests.test_split import long_capture
    from rtl_433_amd import synth
    devs = default_devices[0]
    caps = [long_capture(11, n_bursts=20), long_capture(12, sigma=1.0), long_capture(13, sigma=0.0), synth.noise_cu8(15, 400000, 3.0)]
    lens = np.array([a.nbytes for a in caps], dtype=np.uint32)
    stride = int((lens.max() + 15) // 16 * 16)
    host = np.zeros((len(caps), stride), dtype=np.uint8)
    for i, a in enumerate(caps):
        host[i, :a.nbytes] = a
    eng = BatchEngine(flow_cfg(2, 250000), devs)
    eng.set_split(16384)
    eng.set_debug(1 if blind else 0)  # R433_DEBUG_SPLIT_BLIND
    eng.run(torch.from_numpy(host).cuda(), lens)
    st = eng.split_stats()
    pk, ev = eng.packages()[0], eng.events()[0]
    eng.close()
    cfg = po.default_flow_cfg(2, 250000)
    pk_o, ev_o, base = b"", b"", 0
    for s, a in enumerate(caps):
        o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
        pk_o += o["packages"]
        ev_o += o["events"]
        base += o["n_packages"]
    assert st["segments"] > 50
    assert pk == pk_o and ev == ev_o


@pytest.mark.parametrize("seed", [3, 19, 42, 77, 104, 1189, 5003, 5011])
def test_fuzz_cases_on_gpu(seed):
    """The seeded fuzz slice of tests/test_fuzz_emu.py through the product library (tools/fuzz_emu.py --gpu)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_emu
    assert fuzz_emu.one_case(seed, fuzz_emu.gpu_run) is None


@pytest.mark.parametrize("first", [200000 + 150 * k for k in range(24)])
def test_fuzz_block_on_gpu(first):
    """3600 more random cases (tools/fuzz_emu.py --gpu, 150 a block): random signals x flow options x the forms of the
    detection kernel (lone wavefront, pair, producers and consumers as two launches with their run-again launch) x lazy tiles
    on / off x the split path x the slicer fan-out in stretches and at fixed strides; records, frame sums and (every third
    case) the sample taps against the oracle.  ~12 cases a second on the GPU: the inline-assembly and readfirstlane forms of
    the kernels, which the CPU emulator replaces by plain twins, are only ever checked here."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_emu
    fuzz_emu.BIG = True
    try:
        bad = [seed for seed in range(first, first + 150) if fuzz_emu.one_case(seed, fuzz_emu.gpu_run) is not None]
    finally:
        fuzz_emu.BIG = False
    assert bad == []


def _ref_shard(bounds):
    """(worker process) the unmodified reference over captures [a, b) of the bench's capture list: package records and the
    order-independent checksum of every bitbuffer handed to a decoder"""
    import bench
    a, b = bounds
    host = bench.ook_batches(a, b - a, 1)
    ref = po.Ref(record=True)
    ref.set_digest_mode(2)
    for s in range(b - a):
        ref.run(host[s], 2, 250000, 433920000, fpdm=2, stream_index=a + s)
    pk, n = ref.packages()
    out = (a, bytes(pk), n, ref.digest2(), ref.digest()[1])
    ref.close()
    return out


def test_one_rank_shard_of_config4_vs_reference(default_devices):
    """What ONE rank of BASELINE configs[3] at N = 8 processes -- a shard of 8192 captures of the list (every third a
    protocol-valid transmission), one detection grid -- against the unmodified reference run over the same 8192 captures
    (sixteen worker processes): every package record byte for byte, every bitbuffer by checksum."""
    if not po.have_ref():
        pytest.skip("oracle/_ref/libr433ref.so not present")
    import ctypes as C
    import multiprocessing as mp
    import torch
    import bench
    from rtl_433_amd import _lib
    from rtl_433_amd.engine import BatchEngine, digest_plugin_addr, flow_cfg, make_rdevices
    devs, protocols, names = default_devices
    n = 8192
    host = bench.ook_batches(0, n, 16)
    ctx = _lib.DigestCtx(0, 0)
    rdev_arr, _objs = make_rdevices(devs, digest_plugin_addr(), C.addressof(ctx), names, protocols)
    eng = BatchEngine(flow_cfg(2, 250000), devs)
    npk = eng.run(torch.from_numpy(host).cuda())
    eng.dispatch(rdev_arr, n_threads=16)
    pk = eng.packages()[0]
    eng.close()
    jobs = [(a, min(n, a + 256)) for a in range(0, n, 256)]
    with mp.get_context("fork").Pool(16) as pool:
        parts = sorted(pool.map(_ref_shard, jobs, chunksize=1))
    assert sum(p[2] for p in parts) == npk
    pk_ref = b"".join(p[1] for p in parts)
    assert po.strip_ret_pos(pk) == pk_ref
    assert int(ctx.events) == sum(p[4] for p in parts) and int(ctx.events) > 8000000
    # the reference numbers its packages per worker; the checksum keys every bitbuffer by its package number: re-key by
    # comparing per worker instead -- run the GPU side again shard by shard
    eng = BatchEngine(flow_cfg(2, 250000), devs)
    d = torch.from_numpy(host).cuda()
    for a, _pk, _n, dig, nev in parts[::8]:  # every eighth shard of 256: 1024 captures' bitbuffers by checksum
        ctx.sum = 0
        ctx.events = 0
        eng.run(d[a:a + 256])
        eng.dispatch(rdev_arr, n_threads=16)
        assert (int(ctx.sum), int(ctx.events)) == (dig, nev), a
    eng.close()


def test_two_shards_in_one_process_gather(default_devices):
    """N = 2 without a second GPU: two engines on two HIP streams take the two halves of a capture list side by side, their
    per-shard records go through the same pack / gather / merge path as the ranks' (shard.gather_rank_records), and the
    merged package stream is the one a single engine makes of the whole list."""
    import threading
    import torch
    from rtl_433_amd import shard, synth
    from rtl_433_amd.engine import BatchEngine, flow_cfg
    devs = default_devices[0]
    n = 600
    host = synth.ook_batch(n, 40000, 250000, seed0=70000)
    d = torch.from_numpy(host).cuda()
    whole = BatchEngine(flow_cfg(2, 250000), devs)
    npk = whole.run(d)
    pk_whole, ev_whole = whole.packages()[0], whole.events()[0]
    whole.close()
    bounds = shard.partition(n, 2)
    engines = [BatchEngine(flow_cfg(2, 250000), devs) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    got = [None, None]

    def leg(r):
        torch.cuda.set_device(0)
        got[r] = engines[r].run(d[bounds[r]:bounds[r + 1]], stream=streams[r].cuda_stream)
    ts = [threading.Thread(target=leg, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    payloads, per_rank = [], []
    for r in range(2):
        pk, k = engines[r].packages()
        ev, nev = engines[r].events()
        payloads.append(shard.pack_rank_record(bounds[r], k, nev, 0, len(ev), pk))
        per_rank.append((bounds[r], k, pk, ev))
    merged = shard.gather_rank_records(payloads, False, tail="packages")
    assert sum(p["packages"] for p in merged["per_rank"]) == npk == got[0] + got[1]
    assert merged["merged"] == pk_whole
    assert shard.merge_rank_records(per_rank) == (pk_whole, ev_whole)
    for e in engines:
        e.close()


def test_input_formats_cs8_cf32(default_devices):
    """cs8 / cf32 inputs converted on the device (reference src/rtl_433.c:1811-1834), incl. NaN / out-of-range floats."""
    import torch
    from rtl_433_amd import synth
    from rtl_433_amd.engine import BatchEngine, flow_cfg
    from tests.test_emu_parity import _cf32_to_cs16_like_c
    devs = default_devices[0]

    def run(arr_list, ss, rate, **kw):
        lens = np.array([a.nbytes for a in arr_list], dtype=np.uint32)
        stride = int((lens.max() + 15) // 16 * 16)
        hostbuf = np.zeros((len(arr_list), stride), dtype=np.uint8)
        for i, a in enumerate(arr_list):
            hostbuf[i, :a.nbytes] = a.view(np.uint8)
        eng = BatchEngine(flow_cfg(ss, rate, **kw), devs)
        eng.run(torch.from_numpy(hostbuf).cuda(), lens)
        out = eng.packages()[0], eng.events()[0]
        eng.close()
        return out

    cu8 = [synth.ook_stream(60, 300000)[0], synth.fsk_stream_cu8(61, 200001)]
    cfg = po.default_flow_cfg(2, 250000)
    pk_o, ev_o, base = b"", b"", 0
    for s, a in enumerate(cu8):
        o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base)
        pk_o, ev_o, base = pk_o + o["packages"], ev_o + o["events"], base + o["n_packages"]
    assert run([a ^ 0x80 for a in cu8], 2, 250000, input_format=1) == (pk_o, ev_o)

    rng = np.random.default_rng(62)
    cs16 = synth.fsk_stream_cs16(63, 300000)
    f = (cs16.astype(np.float32) / np.float32(32767.0)) * np.float32(1.3)
    f[100:110] = [np.nan, np.inf, -np.inf, 1e20, -1e20, 65536.5, -65536.5, 1.0, -1.0, 0.99999]
    f += rng.normal(0, 1e-6, f.size).astype(np.float32)
    want16 = _cf32_to_cs16_like_c(f)
    o = po.oracle_flow(want16, devs, po.default_flow_cfg(4, 1024000, fpdm=1))
    assert run([f.view(np.uint8)], 4, 1024000, fpdm=1, center_frequency=868000000, input_format=2) == (o["packages"], o["events"])


def test_full_size_config2_vs_reference(default_devices):
    """BASELINE configs[1] at full size -- 1024 captures x 65536 samples, the 335 default decoders -- against the
    UNMODIFIED reference (oracle/_ref/libr433ref.so, which travels with the repo): every package record (levels,
    start_ago / end_ago, offsets, all pulse / gap widths) and every bitbuffer handed to a decoder, byte for byte."""
    if not po.have_ref():
        pytest.skip("oracle/_ref/libr433ref.so not present")
    from rtl_433_amd import synth
    devs = default_devices[0]
    batch = synth.ook_batch(1024)
    g = _gpu_run([batch[s] for s in range(1024)], 2, 250000, 433920000, devs)
    ref = po.Ref(record=True)
    rdevs, _, _ = ref.devices()
    assert rdevs.tobytes() == np.ascontiguousarray(devs).tobytes()
    for s in range(1024):
        ref.run(batch[s], 2, 250000, 433920000, fpdm=2, stream_index=s)
    pk_ref, n_ref = ref.packages()
    ev_ref, nev_ref = ref.events()
    ref.close()
    pk, npk = g["packages"]
    ev, nev = g["events"]
    assert npk == n_ref and npk >= 1024
    assert po.strip_ret_pos(pk) == pk_ref
    assert nev == nev_ref and nev > 1000000
    # the reference calls its decoders priority level by priority level; the records are (package, device, ordinal)
    assert po.events_normalize(ev) == po.events_normalize(po.canonical_events(ev_ref))


def test_am_s16_fm_s16_input_files():
    """am.s16 / fm.s16 pseudo-IQ input (R433_IN_S16_AM / R433_IN_S16_FM) against the unmodified reference."""
    if not po.have_ref():
        pytest.skip("oracle/_ref/libr433ref.so not present")
    from tests import s16_input_case
    s16_input_case.check(lambda iq, devs, fmt, fm: _gpu_run(iq, 2, 250000, 433920000, devs, taps=True, enable_fm=fm, input_format=fmt))
