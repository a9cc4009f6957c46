"""Decoder dispatch (r433_batch_dispatch / _mt) against the reference's rules (src/r_api.c:438-550,
src/pulse_slicer.c:26-66): packages in detection order, priority levels ascending and only while no
earlier level produced an event, devices in registration order inside a level, bitbuffers in pulse
order, reference-layout bitbuffer_t contents, statistics, invalid return codes are fatal.  Every test runs twice:
on the emulator build of the library (CPU suite) and, under -m gpu, on the product library
(rtl_433_amd/lib/librtl433hip.so: another compiler, a version script -- and r433_batch_run_host in front)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from rtl_433_amd import _lib, synth
from rtl_433_amd.engine import BatchEngine, flow_cfg, make_rdevices
from tests.emu import build_emu

BACKENDS = [pytest.param("emu", marks=pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")),
            pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    return request.param


class BitBuffer(C.Structure):  # bitbuffer_t, reference include/bitbuffer.h:34-40
    _fields_ = [("num_rows", C.c_uint16), ("free_row", C.c_uint16), ("bits_per_row", C.c_uint16 * 50),
                ("syncs_before_row", C.c_uint16 * 50), ("bb", (C.c_uint8 * 128) * 50)]


def _setup(devs, iqs, backend="emu", **cfg_kw):
    if backend == "gpu":  # the product library, fed from host memory like a C host would
        eng = BatchEngine(flow_cfg(2, 250000, **cfg_kw), devs)
        eng.run_host(iqs)
        return eng, iqs
    from tests.emu import host
    lib = host.emu_lib()
    n = len(iqs)
    lens = np.array([a.nbytes for a in iqs], dtype=np.uint32)
    stride = int((lens.max() + 15) // 16 * 16)
    buf = np.zeros(n * stride + 64, dtype=np.uint8)
    off = (-buf.ctypes.data) % 16
    arena = buf[off:off + n * stride].reshape(n, stride)
    for i, a in enumerate(iqs):
        arena[i, :a.nbytes] = a
    eng = BatchEngine(flow_cfg(2, 250000, **cfg_kw), devs, library=lib)
    eng.run_ptr(arena.ctypes.data, stride, n, lens)
    return eng, arena


def _devices():
    devs = np.zeros(6, dtype=po.DEV_DTYPE)
    #          mod  short  long  reset  gap   sync  tol  prio
    devs[0] = (6, 400.0, 800.0, 6000.0, 2000.0, 0.0, 150.0, 0)    # OOK_PWM
    devs[1] = (5, 400.0, 800.0, 6000.0, 0.0, 0.0, 150.0, 10)      # OOK_PPM, low priority (runs late)
    devs[2] = (3, 400.0, 0.0, 6000.0, 0.0, 0.0, 0.0, 0)           # OOK_MC_ZEROBIT
    devs[3] = (4, 400.0, 400.0, 6000.0, 0.0, 0.0, 0.0, 5)         # OOK_PCM, middle priority
    devs[4] = (6, 200.0, 400.0, 3000.0, 1000.0, 0.0, 80.0, 0)     # OOK_PWM, other timing
    devs[5] = (16, 100.0, 100.0, 3000.0, 0.0, 0.0, 0.0, 0)        # FSK_PCM: never sees OOK packages
    return devs


def _expected_order(ev_blob, n_pkgs, prios, hit_dev):
    """Reference order of decode_fn calls given that device hit_dev returns 1 and everything else -1."""
    evs = po.parse_events(ev_blob)
    calls = []
    for pkg in range(n_pkgs):
        mine = [e for e in evs if e["pkg"] == pkg]
        got = False
        for level in sorted(set(prios)):
            if got:
                break
            for dev in range(len(prios)):
                if prios[dev] != level:
                    continue
                for e in (x for x in mine if x["dev"] == dev):
                    calls.append((pkg, dev, e["ordinal"]))
                    got = got or dev == hit_dev
    return calls


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("hit_dev", [0, 3, None])
def test_dispatch_order_priority_and_contents(threads, hit_dev, backend):
    devs = _devices()
    iqs = [synth.ook_stream(900 + k, 30000)[0] for k in range(5)]
    eng, keep = _setup(devs, iqs, backend)
    ev_blob, _ = eng.events()
    pk_blob, n_pkgs = eng.packages()
    by_key = {(e["pkg"], e["dev"], e["ordinal"]): e for e in po.parse_events(ev_blob)}
    calls, bad = [], []
    L = eng.L

    @_lib.DECODE_FN
    def decode(rdev, bits_p):
        info = _lib.DispatchInfo()  # per call: callbacks of several dispatch threads interleave
        L.r433_dispatch_current(C.byref(info))
        key = (info.package, info.device, info.ordinal)
        calls.append(key)
        bb = C.cast(bits_p, C.POINTER(BitBuffer)).contents
        e = by_key[key]
        if bb.num_rows != e["num_rows"] or bb.free_row != e["free_row"]:
            bad.append(("rows", key))
        for r, (bits, syncs, data) in enumerate(e["rows"][:50]):
            if bb.bits_per_row[r] != bits or bb.syncs_before_row[r] != syncs:
                bad.append(("hdr", key, r))
            flat = bytes(C.cast(bb.bb[r], C.POINTER(C.c_uint8 * len(data))).contents) if data else b""
            if flat != data:
                bad.append(("data", key, r))
        return 1 if info.device == hit_dev else -1

    rdevs, objs = make_rdevices(devs, C.cast(decode, C.c_void_p).value, None)
    n_ok = eng.dispatch(rdevs, n_threads=threads)
    assert not bad, bad[:5]
    prios = [int(d["priority"]) for d in devs]
    want = _expected_order(ev_blob, n_pkgs, prios, hit_dev)
    if threads == 1:
        assert calls == want
    else:  # packages are spread over threads: order holds inside each package
        assert sorted(calls) == sorted(want)
        for pkg in range(n_pkgs):
            assert [c for c in calls if c[0] == pkg] == [c for c in want if c[0] == pkg]
    hits = sum(1 for c in want if c[1] == hit_dev)
    assert n_ok == hits
    for d, o in enumerate(objs):  # statistics as account_event keeps them
        mine = [c for c in want if c[1] == d]
        assert o.decode_events == len(mine)
        assert o.decode_ok == (len(mine) if d == hit_dev else 0)
        assert o.decode_fails[1] == (0 if d == hit_dev else len(mine))  # DECODE_ABORT_LENGTH == -1
    eng.close()


def test_invalid_decoder_return_is_fatal(backend):
    devs = _devices()[:1]
    eng, keep = _setup(devs, [synth.ook_stream(901, 30000)[0]], backend)

    @_lib.DECODE_FN
    def decode(rdev, bits_p):
        return -7  # below DECODE_FAIL_SANITY: the reference exits (src/pulse_slicer.c:44-47)

    rdevs, objs = make_rdevices(devs, C.cast(decode, C.c_void_p).value, None)
    with pytest.raises(RuntimeError, match="invalid return value"):
        eng.dispatch(rdevs)
    eng.close()


def test_package_callback_levels(backend):
    """pkg_cb receives a reference-layout pulse_data_t with calc_rssi_snr applied (src/r_flow.c:35-64)."""
    devs = _devices()[:1]
    import os
    iq = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nice_250k.cu8"), dtype=np.uint8)
    # `rtl_433 -R 169` registers no FSK decoder, so FM demodulation is off and the detector reads the raw
    # envelope where it expects FM samples (include/r_private.h:32-36): the CLI's freq 433.955 comes from that
    eng, keep = _setup(devs, [iq], backend, enable_fm=0)
    seen = []

    class PulseData(C.Structure):  # pulse_data_t, include/pulse_data.h:30-50
        _fields_ = [("offset", C.c_uint64), ("sample_rate", C.c_uint32), ("depth_bits", C.c_uint), ("start_ago", C.c_uint),
                    ("end_ago", C.c_uint), ("num_pulses", C.c_uint), ("pulse", C.c_int * 1200), ("gap", C.c_int * 1200),
                    ("ook_low_estimate", C.c_int), ("ook_high_estimate", C.c_int), ("fsk_f1_est", C.c_int),
                    ("fsk_f2_est", C.c_int), ("freq1_hz", C.c_float), ("freq2_hz", C.c_float), ("centerfreq_hz", C.c_float),
                    ("range_db", C.c_float), ("rssi_db", C.c_float), ("snr_db", C.c_float), ("noise_db", C.c_float)]
    assert C.sizeof(PulseData) == 9672

    @_lib.PACKAGE_FN
    def on_pkg(user, stream, typ, pd_p):
        pd = C.cast(pd_p, C.POINTER(PulseData)).contents
        seen.append((stream, typ, pd.num_pulses, pd.pulse[0], pd.gap[0], round(pd.rssi_db, 3), round(pd.snr_db, 3), round(pd.noise_db, 3),
                     round(pd.freq1_hz / 1e6, 3)))

    rdevs, objs = make_rdevices(devs, None, None)
    eng.dispatch(rdevs, pkg_cb=on_pkg)
    # the reference CLI on this capture: rssi -2.312 snr 39.833 noise -42.144 freq 433.955 (tests/golden/kat.json)
    assert seen == [(0, 1, 53, 131, 123, -2.312, 39.833, -42.144, 433.955)]
    eng.close()


def test_dispatch_hooks(backend):
    """r433_batch_dispatch_hooks: package_begin / event_done / package_end around the decoders, in reference order,
    with the r_device counters already moved when event_done fires (account_event, src/pulse_slicer.c:35-47)."""
    devs = _devices()
    iqs = [synth.ook_stream(900 + k, 30000)[0] for k in range(4)]
    eng, keep = _setup(devs, iqs, backend)
    ev_blob, _ = eng.events()
    pk_blob, n_pkgs = eng.packages()
    pkgs = po.parse_packages(pk_blob)
    L = eng.L
    trace = []
    hit_dev = 3

    @_lib.DECODE_FN
    def decode(rdev, bits_p):
        info = _lib.DispatchInfo()
        L.r433_dispatch_current(C.byref(info))
        trace.append(("call", info.package, info.device, info.ordinal))
        return 2 if info.device == hit_dev else -3

    @_lib.HOOK_BEGIN_FN
    def begin(user, rec, pd):
        r, p = rec.contents, pd.contents
        trace.append(("begin", r.stream, r.type, r.num_pulses, r.frame, p.num_pulses, p.pulse[0], p.gap[0], p.start_ago))

    @_lib.HOOK_EVENT_FN
    def event(user, rdev, ret, bits_p):
        trace.append(("event", ret, rdev.contents.decode_events))

    @_lib.HOOK_END_FN
    def end(user, rec, p_events):
        trace.append(("end", rec.contents.stream, p_events))

    rdevs, objs = make_rdevices(devs, C.cast(decode, C.c_void_p).value, None)
    hooks = _lib.DispatchHooks(None, begin, event, end)
    n_ok = eng.dispatch_hooks(rdevs, hooks)
    prios = [int(d["priority"]) for d in devs]
    want_calls = _expected_order(ev_blob, n_pkgs, prios, hit_dev)
    assert [t[1:] for t in trace if t[0] == "call"] == want_calls
    assert n_ok == 2 * sum(1 for c in want_calls if c[1] == hit_dev)
    # shape of the trace: begin, (call, event)*, end per package
    at = 0
    seen_events = [0] * len(devs)
    for k, p in enumerate(pkgs):
        assert trace[at] == ("begin", p["stream"], p["type"], p["num"], p["frame"], p["num"], int(p["pulse"][0]),
                             int(p["gap"][0]), p["start_ago"])
        at += 1
        p_events = 0
        for c in (c for c in want_calls if c[0] == k):
            assert trace[at] == ("call",) + c
            seen_events[c[1]] += 1
            ret = 2 if c[1] == hit_dev else 0  # failures come back as 0 to the hook, like account_event returns them
            assert trace[at + 1] == ("event", ret, seen_events[c[1]])
            p_events += ret
            at += 2
        assert trace[at] == ("end", p["stream"], p_events)
        at += 1
    assert at == len(trace)
    assert list(eng.decoded()) == [t[2] for t in trace if t[0] == "end"]
    assert objs[hit_dev].decode_messages == n_ok and objs[0].decode_fails[3] == seen_events[0]
    eng.close()


def test_run_host_equals_run(backend):
    """r433_batch_run_host (captures in host memory, ragged, not contiguous) == r433_batch_run on the staged layout."""
    devs = _devices()
    iqs = [synth.ook_stream(910 + k, 20000 + 3001 * k)[0] for k in range(5)] + [np.zeros(0, dtype=np.uint8)]
    if backend == "gpu":
        import torch
        eng = BatchEngine(flow_cfg(2, 250000), devs)
        n0 = eng.run_host(iqs)
        got = (eng.packages(), eng.events())
        lens = np.array([a.nbytes for a in iqs], dtype=np.uint32)
        stride = int((lens.max() + 15) // 16 * 16)
        host = np.zeros((len(iqs), stride), dtype=np.uint8)
        for i, a in enumerate(iqs):
            host[i, :a.nbytes] = a
        n1 = eng.run(torch.from_numpy(host).cuda(), lens)
        assert n0 == n1 and got == (eng.packages(), eng.events())
        # contiguous and equally long: the single-copy path
        same = np.ascontiguousarray(np.stack([synth.ook_stream(920 + k, 8192)[0] for k in range(4)]))
        n2 = eng.run_host([same[k] for k in range(4)])
        got2 = (eng.packages(), eng.events())
        assert n2 == eng.run(torch.from_numpy(same).cuda()) and got2 == (eng.packages(), eng.events())
    else:
        from tests.emu import host as emu_host
        eng = BatchEngine(flow_cfg(2, 250000), devs, library=emu_host.emu_lib())
        n0 = eng.run_host(iqs)
        got = (eng.packages(), eng.events())
        eng2, keep = _setup(devs, iqs[:5] + [np.zeros(0, dtype=np.uint8)], "emu")
        assert got == (eng2.packages(), eng2.events())
        eng2.close()
    eng.close()


def test_run_host_all_empty(backend):
    """Every capture empty (pointers may be null): no copy, no package, no crash (the packed path read
    n * 16 bytes from a null pointer)."""
    devs = _devices()
    if backend == "gpu":
        eng = BatchEngine(flow_cfg(2, 250000), devs)
    else:
        from tests.emu import host as emu_host
        eng = BatchEngine(flow_cfg(2, 250000), devs, library=emu_host.emu_lib())
    ptrs = (C.c_void_p * 2)(None, None)
    lens = (C.c_uint32 * 2)(0, 0)
    rc = eng.L.r433_batch_run_host(eng.h, C.cast(ptrs, C.c_void_p), C.cast(lens, C.c_void_p), 2)
    assert rc == 0
    assert eng.packages()[1] == 0 and eng.events()[1] == 0
    eng.close()


@pytest.mark.parametrize("threads,stateless", [(1, False), (4, False), (4, True)])
def test_dispatch_ordered_equals_single_thread(threads, stateless, backend):
    """r433_batch_dispatch_ordered: decoders spread over threads, each decoder's calls in reference order, the priority
    rule kept, and what decoders hand to output_fn / log_fn committed in the order of the single-threaded replay."""
    devs = _devices()
    iqs = [synth.ook_stream(930 + k, 30000)[0] for k in range(6)]
    eng, keep = _setup(devs, iqs, backend)
    L = eng.L
    OUT_FN = C.CFUNCTYPE(None, C.POINTER(_lib.RDevice), C.c_void_p)
    LOG_FN = C.CFUNCTYPE(None, C.POINTER(_lib.RDevice), C.c_int, C.c_void_p)
    hit_dev = 3

    def run(ordered):
        trace, per_dev_calls = [], {}

        @_lib.DECODE_FN
        def decode(rdev, bits_p):
            info = _lib.DispatchInfo()
            L.r433_dispatch_current(C.byref(info))
            per_dev_calls.setdefault(info.device, []).append((info.package, info.ordinal))
            d = rdev.contents
            token = (info.package << 20) | (info.device << 10) | info.ordinal
            # a decoder that talks: one log line for every call of device 0, two outputs for every hit of hit_dev
            if info.device == 0:
                C.cast(d.log_fn, LOG_FN)(rdev, 2, C.c_void_p(token | (1 << 40)))
            if info.device == hit_dev:
                C.cast(d.output_fn, OUT_FN)(rdev, C.c_void_p(token))
                C.cast(d.output_fn, OUT_FN)(rdev, C.c_void_p(token | (1 << 41)))
                return 2
            return -1

        @OUT_FN
        def real_out(rdev, payload):
            trace.append(("out", rdev.contents.protocol_num, payload))

        @LOG_FN
        def real_log(rdev, level, payload):
            trace.append(("log", rdev.contents.protocol_num, level, payload))

        @_lib.HOOK_BEGIN_FN
        def begin(user, rec, pd):
            trace.append(("begin", rec.contents.stream, rec.contents.num_pulses, round(pd.contents.rssi_db, 3)))

        @_lib.HOOK_END_FN
        def end(user, rec, p_events):
            trace.append(("end", rec.contents.stream, p_events))

        rdevs, objs = make_rdevices(devs, C.cast(decode, C.c_void_p).value, None, protocols=list(range(100, 100 + len(devs))))
        for o in objs:
            o.output_fn = C.cast(real_out, C.c_void_p)
            o.log_fn = C.cast(real_log, C.c_void_p)
        null_event = C.cast(None, _lib.HOOK_EVENT_FN)
        hooks = _lib.DispatchHooks(None, begin, null_event, end)
        if ordered and stateless:
            # every decoder declared stateless (r433_batch_set_stateless): its calls go out in stretches of five records to
            # whichever thread is free -- hooks, outputs, statistics and events stay those of the single-threaded replay
            eng.set_stateless((C.c_uint8 * len(devs))(*([1] * len(devs))))
            eng.set_debug(8)
        n = eng.dispatch_ordered(rdevs, hooks, threads) if ordered else eng.dispatch_hooks(rdevs, hooks)
        if ordered and stateless:
            per_dev_calls = {d: sorted(v) for d, v in per_dev_calls.items()}  # (a decoder's calls: the same set, any thread)
            eng.set_stateless(None)
            eng.set_debug(0)
        stats = [(o.decode_events, o.decode_ok, o.decode_messages, list(o.decode_fails)) for o in objs]
        fns = all(o.output_fn == C.cast(real_out, C.c_void_p).value and o.log_fn == C.cast(real_log, C.c_void_p).value for o in objs)
        return n, trace, per_dev_calls, stats, list(eng.decoded()), fns

    a = run(False)
    b = run(True)
    assert a[0] == b[0] and a[0] > 0
    assert a[1] == b[1]            # hooks and committed outputs: same sequence
    if stateless:
        a = (a[0], a[1], {d: sorted(v) for d, v in a[2].items()}) + a[3:]
    assert a[2] == b[2]            # every decoder saw the same calls in the same order (priority gating included)
    assert a[3] == b[3] and a[4] == b[4]
    assert a[5] and b[5]           # output_fn / log_fn restored
    assert any(t[0] == "out" for t in a[1]) and any(t[0] == "log" for t in a[1])
    eng.close()


@pytest.mark.parametrize("ordered", [False, True])
def test_package_filter_leaves_packages_out(ordered, backend):
    """hooks.package_filter: a package it turns down reaches no decoder and no other hook, in the single-threaded and in the
    ordered replay alike; the others are replayed as if it were not there."""
    devs = _devices()
    iqs = [synth.ook_stream(950 + k, 30000)[0] for k in range(5)]
    eng, keep = _setup(devs, iqs, backend)
    L = eng.L
    n_pkgs = eng.packages()[1]
    assert n_pkgs >= 4
    calls, begun, ended, asked = [], [], [], []

    @_lib.DECODE_FN
    def decode(rdev, bits_p):
        info = _lib.DispatchInfo()
        L.r433_dispatch_current(C.byref(info))
        calls.append(info.package)
        return 1 if info.device == 2 else 0

    @_lib.HOOK_BEGIN_FN
    def begin(user, rec, pd):
        begun.append(rec.contents.stream)

    @_lib.HOOK_END_FN
    def end(user, rec, p_events):
        ended.append((rec.contents.stream, p_events))

    @_lib.HOOK_FILTER_FN
    def keep_even_captures(user, rec):
        asked.append(rec.contents.stream)
        return 1 if rec.contents.stream % 2 == 0 else 0

    rdevs, objs = make_rdevices(devs, C.cast(decode, C.c_void_p).value, None)
    hooks = _lib.DispatchHooks(None, begin, C.cast(None, _lib.HOOK_EVENT_FN), end, keep_even_captures)
    n = eng.dispatch_ordered(rdevs, hooks, 3) if ordered else eng.dispatch_hooks(rdevs, hooks)
    pkgs = po.parse_packages(eng.packages()[0])
    kept = [k for k, p in enumerate(pkgs) if p["stream"] % 2 == 0]
    assert sorted(set(asked)) == sorted(set(p["stream"] for p in pkgs))      # every package was offered once
    assert len(asked) == len(pkgs)
    assert sorted(set(calls)) == kept                                          # decoders saw the kept packages only
    assert begun == [pkgs[k]["stream"] for k in kept]
    assert [s for s, _ in ended] == begun and sum(e for _, e in ended) == n
    dec = list(eng.decoded())
    assert all(dec[k] == 0 for k in range(len(pkgs)) if k not in kept) and n == sum(dec) > 0
    eng.close()


def test_engine_on_a_chosen_device(backend):
    """r433_batch_create_on: an engine belongs to one GPU and says which; a GPU that is not there is R433_ENODEV; an engine on
    the last visible device gives the records of one made the old way."""
    if backend == "gpu":
        L = _lib.lib()
    else:
        from tests.emu import host
        L = host.emu_lib()
    n = L.r433_device_count()
    assert n >= 1
    cfg = flow_cfg(2, 250000)
    h = L.r433_batch_create_on(n - 1, C.byref(cfg), None, 0)
    assert h and L.r433_batch_device(h) == n - 1
    L.r433_batch_destroy(h)
    assert not L.r433_batch_create_on(n, C.byref(cfg), None, 0)
    assert "no GPU" in _lib.last_error(L)
    iqs = [synth.ook_stream(3, 20000)[0], synth.ook_stream(4, 24000)[0]]
    e0 = BatchEngine(cfg, None, library=L)
    e1 = BatchEngine(cfg, None, library=L, device=n - 1)
    assert e0.run_host(iqs) == e1.run_host(iqs) >= 2  # host memory: the portable form, whichever GPU the engine is on
    assert e0.packages() == e1.packages()
    e0.close()
    e1.close()


def test_output_render_hook_of_the_ordered_replay(backend):
    """r433_dispatch_hooks.output_render: what the reference's real decoders report is rendered to its JSON line on the replay
    thread that ran the decoder (dropin/plugins_shim.c r433p_render), the commit appends the lines in order -- the same text,
    byte for byte, as when the output handler renders on the committing thread; a plain test plugin sees the hook's return
    value in its output_fn, in reference order."""
    from rtl_433_amd import plugins, protocols
    from rtl_433_amd.engine import load_device_table
    if not plugins.available():
        pytest.skip("dropin/_build/libr433plugins.so not built")
    devs = load_device_table()[0]
    iqs = [protocols.bench_capture(3 * i + 2)[0] if i % 2 else synth.ook_stream(900 + i)[0] for i in range(10)]
    eng, _ = _setup(devs, iqs, backend)
    plug = plugins.Plugins()
    texts = []
    for hooks, threads in ((None, 1), (None, 5), (plug.hooks(), 1), (plug.hooks(), 5)):
        n = eng.dispatch_ordered(plug.devices, hooks, threads)
        text, n_msg = plug.take()
        assert n == n_msg == text.count(b"\n") >= 4
        texts.append(text)
    assert texts[0] == texts[1] == texts[2] == texts[3]
    eng.set_stateless(plug.stateless())  # a decoder's calls on several threads: the rendering, too
    eng.dispatch_ordered(plug.devices, plug.hooks(), 5)
    assert plug.take()[0] == texts[0]
    plug.close()
    eng.close()
