"""The decoder pre-filter (r433_batch_probe_prefilter): records their decoder provably refuses on
num_rows / bits_per_row[0] never leave the device, and nothing a caller can observe changes -- per-decoder statistics
(src/pulse_slicer.c:26-66), events per package, what the decoders that do run are handed.  Plugins with first-line tests
like the reference's (tests/plugins/pf_decoders.c) on the emulator and, under -m gpu, on the product library; there also
the reference's REAL decoders (taken from oracle/_ref/libr433ref.so -- the checker's copy of the unmodified plugins)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from rtl_433_amd import _lib, synth
from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table, make_rdevices
from tests.emu import build_emu

HERE = os.path.dirname(os.path.abspath(__file__))
BACKENDS = [pytest.param("emu", marks=pytest.mark.skipif(not build_emu.available(), reason="wave emulator needs x86-64")),
            pytest.param("gpu", marks=pytest.mark.gpu)]
FNS = ["pf_dec_exact", "pf_dec_onerow", "pf_dec_rows", "pf_dec_data", "pf_dec_moody", "pf_dec_zero"]


@pytest.fixture(params=BACKENDS)
def backend(request):
    return request.param


@pytest.fixture(scope="module")
def plugins():
    out = os.path.join(HERE, "emu", "_build", "libpfdecoders.so")
    src = os.path.join(HERE, "plugins", "pf_decoders.c")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", out, src])
    return C.CDLL(out)


def _engine(devs, backend):
    if backend == "gpu":
        return BatchEngine(flow_cfg(2, 250000), devs)
    from tests.emu import host
    return BatchEngine(flow_cfg(2, 250000), devs, library=host.emu_lib())


def _stats(objs):
    return [(o.decode_events, o.decode_ok, o.decode_messages, tuple(o.decode_fails)) for o in objs]


def _records(blob):
    out, at = [], 0
    while at + 16 <= len(blob):
        total = int.from_bytes(blob[at:at + 4], "little")
        out.append(blob[at:at + total])
        at += total
    return out


def _devices():
    devs = np.zeros(8, dtype=po.DEV_DTYPE)
    #          mod  short  long  reset  gap   sync  tol  prio
    devs[0] = (6, 400.0, 800.0, 6000.0, 2000.0, 0.0, 150.0, 0)   # OOK_PWM -> pf_dec_exact
    devs[1] = (4, 100.0, 100.0, 3000.0, 0.0, 0.0, 0.0, 0)        # OOK_PCM -> pf_dec_onerow
    devs[2] = (5, 400.0, 800.0, 6000.0, 0.0, 0.0, 150.0, 0)      # OOK_PPM -> pf_dec_rows
    devs[3] = (6, 250.0, 500.0, 1200.0, 800.0, 900.0, 120.0, 0)  # OOK_PWM with sync pulses, short reset -> pf_dec_data
    devs[4] = (6, 200.0, 400.0, 3000.0, 1000.0, 0.0, 80.0, 0)    # OOK_PWM -> pf_dec_moody
    devs[5] = (6, 300.0, 600.0, 900.0, 700.0, 0.0, 100.0, 0)     # OOK_PWM, short reset: many bitbuffers -> pf_dec_zero
    devs[6] = (6, 400.0, 800.0, 6000.0, 2000.0, 0.0, 150.0, 5)   # a later priority level -> pf_dec_exact, never filtered
    devs[7] = (5, 250.0, 500.0, 4000.0, 0.0, 0.0, 100.0, 0)      # verbose decoder -> pf_dec_onerow, never filtered
    fns = [0, 1, 2, 3, 4, 5, 0, 1]
    return devs, fns


def test_prefilter_plugins(backend, plugins):
    devs, fns = _devices()
    iqs = [synth.ook_stream(3000 + k, 40000)[0] for k in range(12)] + [synth.random_cu8(77, 5000)]
    runs = {}
    for mode in ("plain", "filtered"):
        eng = _engine(devs, backend)
        arr, objs = make_rdevices(devs)
        for o, f in zip(objs, fns):
            o.decode_fn = C.cast(getattr(plugins, FNS[f]), C.c_void_p).value
        objs[7].verbose = 1
        calls = (C.c_ulong * 8).in_dll(plugins, "pf_calls")
        n_tables = eng.probe_prefilter(arr)  # in both runs: the moody decoder counts its calls
        if mode == "plain":
            eng.set_prefilter(0)
        for k in range(8):
            calls[k] = 0
        C.c_uint.in_dll(plugins, "pf_moody_n").value = 0
        eng.run_host(iqs)
        ev, nev = eng.events()
        dec = eng.dispatch(arr, n_threads=1)
        runs[mode] = dict(stats=_stats(objs), per_pkg=list(eng.decoded()), decoded=dec, records=_records(ev), nev=nev,
                          calls=list(calls), tables=n_tables, dropped=eng.prefilter_counts())
        eng.close()
    a, b = runs["plain"], runs["filtered"]
    # exact, onerow and zero decide on the head; rows does when there is one short row (or none), else it walks on; data
    # reads the payload first, but refuses every one-row bitbuffer of fewer than 8 bits that no sync pulse came before (the
    # exhaustive probe of tiny rows); moody is unsteady; 7 is verbose; 6 is a later priority level: whether a bitbuffer it
    # refuses counts is the replay's to know (only packages the first level decoded nothing of get that far), so the device
    # leaves a 16-byte stub (num_rows 0xffff, the code in free_row) in the record's place and drops nothing
    assert b["tables"] == 6
    assert a["stats"] == b["stats"] and a["per_pkg"] == b["per_pkg"] and a["decoded"] == b["decoded"]
    assert b["nev"] < a["nev"] and b["dropped"].sum() == a["nev"] - b["nev"]
    assert b["dropped"][[4, 6, 7]].sum() == 0 and all(b["dropped"][d].sum() > 0 for d in (0, 1, 3, 5))
    assert b["dropped"][3][[0, 2, 3, 4]].sum() == 0  # (only -1: the sync'd tiny rows' -3 is the decoder's to give)
    stubs = [r for r in b["records"] if len(r) == 16 and int.from_bytes(r[12:14], "little") == 0xffff]
    assert stubs and all(int.from_bytes(r[8:10], "little") == 6 and int.from_bytes(r[14:16], "little") == 1 for r in stubs)  # (-1: not 24 / 25 bits)
    # what still reaches the host is what was there before, in the same order, minus the dropped records and the stubs' originals
    it = iter(a["records"])
    assert all(any(r == x for x in it) for r in b["records"] if r not in stubs)
    assert len(a["records"]) == len(b["records"]) + int(b["dropped"].sum())
    # the decoders were called less: exactly by what was dropped (and, for the later level, by the stubs of the packages that got there)
    for f, devs_of in ((1, (1, 7)), (2, (2,)), (3, (3,)), (5, (5,))):
        assert a["calls"][f] - b["calls"][f] == sum(int(b["dropped"][d].sum()) for d in devs_of)
    assert int(b["dropped"][0].sum()) <= a["calls"][0] - b["calls"][0] <= int(b["dropped"][0].sum()) + len(stubs)
    assert a["calls"][4] == b["calls"][4]


def test_prefilter_leaves_stateful_decoders_alone(plugins):
    """A decoder the host declares as keeping state between calls (r433_batch_set_stateless with a 0 for it; the reference has
    four with file-scope statics, src/devices/secplus_v1.c:142-143, and every create_fn's context) is never asked -- the
    questions would leave made-up half-messages in its state, and its refusals need not be functions of the head -- and none
    of its records is dropped; the others are filtered as ever."""
    if not build_emu.available():
        pytest.skip("wave emulator needs x86-64")
    devs, fns = _devices()
    iqs = [synth.ook_stream(3000 + k, 40000)[0] for k in range(6)]
    calls = (C.c_ulong * 8).in_dll(plugins, "pf_calls")
    got = {}
    for mode in ("all stateless", "decoders 0 and 5 keep state"):
        eng = _engine(devs, "emu")
        eng.L.r433_prefilter_forget()
        arr, objs = make_rdevices(devs)
        for o, f in zip(objs, fns):
            o.decode_fn = C.cast(getattr(plugins, FNS[f]), C.c_void_p).value
        flags = (C.c_uint8 * len(devs))(*([1] * len(devs)))
        if mode != "all stateless":
            flags[0] = flags[5] = 0
        eng.set_stateless(flags)
        for k in range(8):
            calls[k] = 0
        tables = eng.probe_prefilter(arr)
        asked = list(calls)
        eng.run_host(iqs)
        eng.dispatch_ordered(arr, None, 2)
        got[mode] = dict(tables=tables, asked=asked, dropped=eng.prefilter_counts(), stats=_stats(objs))
        eng.L.r433_prefilter_forget()
        eng.close()
    a, b = got["all stateless"], got["decoders 0 and 5 keep state"]
    assert a["stats"] == b["stats"]
    assert b["tables"] == a["tables"] - 2  # (decoder 6, a later priority level, shares decoder 0's decode_fn but is its own object)
    assert a["asked"][5] > 0 and b["asked"][5] == 0  # pf_dec_zero is decoder 5's alone: never called by the probe
    assert a["dropped"][0].sum() > 0 and a["dropped"][5].sum() > 0
    assert b["dropped"][0].sum() == 0 and b["dropped"][5].sum() == 0
    assert (b["dropped"][[1, 3]] == a["dropped"][[1, 3]]).all()


def test_prefilter_asks_a_context_decoder_with_its_context_out_of_reach(plugins):
    """R433_KEEPS_CONTEXT (flag 2 of r433_batch_set_stateless): a decoder whose state is all behind r_device.decode_ctx is asked
    with that pointer on an unreadable page -- a refusal that comes back without a fault has neither read nor written the state.
    The test decoder refuses many-row bitbuffers and long rows without its state (filtered), counts short rows in its state
    before it refuses them (never filtered: the question faults), and its counter is what the unfiltered run leaves."""
    if not build_emu.available():
        pytest.skip("wave emulator needs x86-64")
    devs = np.zeros(2, dtype=po.DEV_DTYPE)
    devs[0] = (4, 100.0, 100.0, 900.0, 0.0, 0.0, 0.0, 0)        # OOK_PCM, short reset: many short one-row bitbuffers
    devs[1] = (4, 100.0, 100.0, 3000.0, 0.0, 0.0, 0.0, 0)       # OOK_PCM: long rows
    iqs = [synth.ook_stream(7000 + k, 50000)[0] for k in range(12)]
    C.c_uint.in_dll(plugins, "pf_ctx_offset").value = _lib.RDevice.decode_ctx.offset
    calls = (C.c_ulong * 8).in_dll(plugins, "pf_calls")
    got = {}
    for mode in ("plain", "context fenced"):
        eng = _engine(devs, "emu")
        eng.L.r433_prefilter_forget()
        arr, objs = make_rdevices(devs)
        state = [(C.c_uint * 2)(0, 0) for _ in objs]
        for o, st in zip(objs, state):
            o.decode_fn = C.cast(plugins.pf_dec_context, C.c_void_p).value
            o.decode_ctx = C.addressof(st)
        eng.set_stateless((C.c_uint8 * 2)(2, 2))
        tables = eng.probe_prefilter(arr) if mode != "plain" else 0
        assert all(st[0] == 0 and st[1] == 0 for st in state), "a question reached the decoder's state"
        assert all(o.decode_ctx == C.addressof(st) for o, st in zip(objs, state))  # (the pointer is back)
        calls[6] = 0
        eng.run_host(iqs)
        eng.dispatch_ordered(arr, None, 2)
        got[mode] = dict(tables=tables, calls=int(calls[6]), state=[(st[0], st[1]) for st in state], stats=_stats(objs),
                         dropped=eng.prefilter_counts() if mode != "plain" else None)
        eng.L.r433_prefilter_forget()
        eng.close()
    a, b = got["plain"], got["context fenced"]
    assert b["tables"] == 2 and a["stats"] == b["stats"]
    assert a["state"] == b["state"] and sum(s[0] for s in a["state"]) > 10  # every short row still reached the decoder, in both runs
    dropped = int(b["dropped"].sum())
    assert dropped > 20 and a["calls"] - b["calls"] == dropped  # the long rows and the many-row bitbuffers stayed on the device


def test_prefilter_ordered_replay_and_switch(backend, plugins):
    """The ordered multi-threaded replay accounts the dropped records too; set_prefilter(0) brings every record back; an
    event_done hook or a package_filter refuses to run over a filtered pass."""
    devs, fns = _devices()
    iqs = [synth.ook_stream(3100 + k, 30000)[0] for k in range(6)]
    eng = _engine(devs, backend)
    arr, objs = make_rdevices(devs)
    for o, f in zip(objs, fns):
        o.decode_fn = C.cast(getattr(plugins, FNS[f]), C.c_void_p).value
    eng.run_host(iqs)
    n_plain = eng.events()[1]
    eng.dispatch_ordered(arr, None, 4)
    want = _stats(objs)
    for o in objs:
        o.decode_events = o.decode_ok = o.decode_messages = 0
        for k in range(5):
            o.decode_fails[k] = 0
    assert eng.probe_prefilter(arr) >= 3
    eng.run_host(iqs)
    assert eng.events()[1] < n_plain
    eng.dispatch_ordered(arr, None, 4)
    assert _stats(objs) == want
    hooks = _lib.DispatchHooks()
    hooks.package_filter = _lib.HOOK_FILTER_FN(lambda user, rec: 1)
    assert eng.L.r433_batch_dispatch_ordered(eng.h, C.cast(arr, C.c_void_p), len(arr), C.byref(hooks), 2) == -1
    eng.set_prefilter(0)
    eng.run_host(iqs)
    assert eng.events()[1] == n_plain and eng.prefilter_counts().size == 0
    eng.close()


def test_prefilter_real_decoders(backend):
    """The reference's real decoders behind the replay: statistics of all 335, decoded events and what every package
    produced are the same with the filter as without, and the filter takes most of the records (head tables: a good third;
    with the exhaustive probe of tiny rows: more than two thirds)."""
    if not po.have_ref():
        pytest.skip("oracle/_ref not built")
    devs, protocols, names = load_device_table()
    iqs = [synth.ook_stream(s)[0] for s in range(96 if backend == "gpu" else 20)]
    res = {}
    for mode in ("plain", "filtered"):
        ref = po.Ref(call_real=True, record=False)  # a fresh set of decoders (some keep state between calls)
        plain = ref.plain_devices()
        objs = [C.cast(p, C.POINTER(_lib.RDevice)).contents for p in plain]
        eng = _engine(devs, backend)
        tables = eng.probe_prefilter(plain) if mode == "filtered" else 0
        eng.run_host(iqs)
        nev = eng.events()[1]
        dec = eng.dispatch_ordered(plain, None, 8)
        res[mode] = dict(stats=_stats(objs), per_pkg=list(eng.decoded()), decoded=dec, nev=nev, tables=tables)
        eng.close()
        ref.close()
    a, b = res["plain"], res["filtered"]
    assert b["tables"] > 200
    assert a["stats"] == b["stats"] and a["per_pkg"] == b["per_pkg"] and a["decoded"] == b["decoded"]
    assert b["nev"] < 0.3 * a["nev"]


HFNS = ["pfh_dec_search_then_length", "pfh_dec_search_two_codes", "pfh_dec_invert_then_length", "pfh_dec_invert_then_payload",
        "pfh_dec_foreign_search", "pfh_dec_two_searches", "pfh_dec_repeated_row", "pfh_dec_repeated_once", "pfh_dec_search_position",
        "pfh_dec_repeated_then_lengths", "pfh_dec_invert_then_search"]
WRAP = "-Wl,--wrap=bitbuffer_invert,--wrap=bitbuffer_search,--wrap=bitbuffer_find_repeated_row,--wrap=bitbuffer_find_repeated_prefix"


@pytest.fixture(scope="module")
def helper_plugins():
    """decoders that call bitbuffer helpers before their length test, with dropin/helper_wrap.c linked between the two
    (ld --wrap: the way dropin/Makefile links the reference's decoders)"""
    out = os.path.join(HERE, "emu", "_build", "libpfhelper.so")
    srcs = [os.path.join(HERE, "plugins", "pf_helper_decoders.c"), os.path.join(HERE, "plugins", "pf_helper_bitbuffer.c"),
            os.path.join(HERE, "..", "dropin", "helper_wrap.c")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-DR433_WRAP_STANDALONE", "-I", os.path.join(HERE, "..", "include"),
                               "-o", out] + srcs + [WRAP])
    return C.CDLL(out)


def _helper_devices():
    devs = np.zeros(len(HFNS), dtype=po.DEV_DTYPE)
    #          mod  short  long  reset  gap   sync  tol  prio
    devs[0] = (4, 100.0, 100.0, 3000.0, 0.0, 0.0, 0.0, 0)        # OOK_PCM: one long row per burst
    devs[1] = (4, 100.0, 100.0, 900.0, 0.0, 0.0, 0.0, 0)         # OOK_PCM, short reset: many short rows
    devs[2] = (6, 400.0, 800.0, 6000.0, 2000.0, 0.0, 150.0, 0)   # OOK_PWM
    devs[3] = (6, 300.0, 600.0, 900.0, 700.0, 0.0, 100.0, 0)     # OOK_PWM, short reset
    devs[4] = (5, 400.0, 800.0, 6000.0, 0.0, 0.0, 150.0, 0)      # OOK_PPM
    devs[5] = (4, 200.0, 200.0, 2000.0, 0.0, 0.0, 0.0, 0)        # OOK_PCM
    devs[6] = (6, 250.0, 500.0, 1200.0, 800.0, 0.0, 120.0, 0)    # OOK_PWM: one- and several-row bitbuffers
    devs[7] = (5, 250.0, 500.0, 4000.0, 0.0, 0.0, 100.0, 0)      # OOK_PPM
    devs[8] = (4, 100.0, 100.0, 900.0, 0.0, 0.0, 0.0, 0)         # OOK_PCM, short reset
    devs[9] = (6, 250.0, 500.0, 1200.0, 800.0, 0.0, 120.0, 0)    # OOK_PWM, as 6
    devs[10] = (4, 100.0, 100.0, 900.0, 0.0, 0.0, 0.0, 0)        # OOK_PCM, short reset, as 1: the same rows, inverted before the search
    return devs


def test_prefilter_helper_probe_plugins(backend, helper_plugins):
    """Decoders that invert, search or look for a repeated row before their length test: with the host's wrappers answering
    the probe (r433_prefilter_set_helper_probe) a head is dropped where EVERY answer the helper could have given leads to the
    same refusal, and only there; statistics, events per package and the calls that still arrive are those of the unfiltered
    run minus exactly what the device says it dropped.  Without the wrappers' block nothing is learned from these decoders."""
    devs = _helper_devices()
    iqs = [synth.ook_stream(7000 + k, 50000)[0] for k in range(30)] + [synth.random_cu8(99, 6000)]
    calls = (C.c_ulong * 12).in_dll(helper_plugins, "pfh_calls")
    helper = C.cast(helper_plugins.r433_host_helper_probe, C.c_void_p).value
    runs = {}
    for mode in ("plain", "fence_only", "filtered"):
        eng = _engine(devs, backend)
        eng.L.r433_prefilter_forget()
        arr, objs = make_rdevices(devs)
        for o, f in zip(objs, HFNS):
            o.decode_fn = C.cast(getattr(helper_plugins, f), C.c_void_p).value
        eng.L.r433_prefilter_set_helper_probe(None)
        tables = 0 if mode == "plain" else eng.probe_prefilter(arr, helper=helper if mode == "filtered" else None)
        for k in range(12):
            calls[k] = 0
        eng.run_host(iqs)
        ev, nev = eng.events()
        dec = eng.dispatch(arr, n_threads=1)
        runs[mode] = dict(stats=_stats(objs), per_pkg=list(eng.decoded()), decoded=dec, records=_records(ev), nev=nev,
                          calls=list(calls), tables=tables, dropped=eng.prefilter_counts() if mode != "plain" else None)
        eng.L.r433_prefilter_set_helper_probe(None)
        eng.L.r433_prefilter_forget()
        eng.close()
    a, f, b = runs["plain"], runs["fence_only"], runs["filtered"]
    for r in (f, b):
        assert a["stats"] == r["stats"] and a["per_pkg"] == r["per_pkg"] and a["decoded"] == r["decoded"]
        assert int(r["dropped"].sum()) == a["nev"] - r["nev"]
        for i in range(len(HFNS)):
            assert a["calls"][i] - r["calls"][i] == int(r["dropped"][i].sum())
        it = iter(a["records"])
        assert all(any(x == y for y in it) for x in r["records"])
    # (under the bare fence: other row counts on the head, a short row 0 where that comes first, tiny rows asked one by one)
    print("dropped under the fence alone / with the wrappers:", [int(x.sum()) for x in f["dropped"]], [int(x.sum()) for x in b["dropped"]])
    # with the wrappers: the searches (0, 1, 8), the inversion in front of a length test (2), the repeated-row tests (6, 7)
    for d in (0, 1, 2, 6, 7, 8):
        assert int(b["dropped"][d].sum()) > int(f["dropped"][d].sum()), d
    # 6: bitbuffers of several rows, none long enough for its repeated-row test, go by the short-rows verdicts (the slicer
    # kernel knows the longest row); 9 makes the same test and then walks the row lengths itself: only what one-row heads give
    many = lambda recs, d: sum(1 for r in recs if int.from_bytes(r[8:10], "little") == d and 2 <= int.from_bytes(r[12:14], "little") < 0xffff)
    assert many(b["records"], 6) < many(f["records"], 6) and many(b["records"], 9) == many(f["records"], 9) > 0
    # ... 3 looks at the payload whatever the length; 4 and 5 gain nothing (a foreign bitbuffer; a second search)
    assert int(b["dropped"][3].sum()) == 0
    assert int(b["dropped"][4].sum()) == int(f["dropped"][4].sum()) and int(b["dropped"][5].sum()) == int(f["dropped"][5].sum())
    # 1: only rows too short for the preamble go ("not found" and "found, too short" are different codes): all under ABORT_EARLY
    assert int(b["dropped"][1][1]) == 0 and int(b["dropped"][1][2]) > 0
    # 10: the same decoder behind a bitbuffer_invert.  Its search runs over the inverted row, so the device must not run it over
    # the row as sliced (no search rule): what goes is what the length alone decides, never more than decoder 1 loses -- and the
    # statistics above are equal, which they are not when the raw row is searched for the decoder's pattern
    assert 0 < int(b["dropped"][10].sum()) <= int(b["dropped"][1].sum())
    assert b["nev"] < f["nev"] < a["nev"]


def test_wrapped_bitbuffer_search_is_the_reference_one():
    """dropin/helper_wrap.c answers the decoders' bitbuffer_search calls with a word-parallel search of its own outside the
    probe: it must return what src/bitbuffer.c:228-253 returns, for every row, start and pattern (the library holds both and
    compares them on random cases: rows around every length edge, patterns cut from the row itself half of the time)."""
    from rtl_433_amd import plugins
    if not plugins.available():
        pytest.skip("dropin/_build/libr433plugins.so not built")
    L = C.CDLL(os.path.abspath(plugins.LIB_PATH))
    L.r433_host_wrap_selftest.restype = C.c_uint
    for seed in range(3):
        assert L.r433_host_wrap_selftest(seed, 1000000) == 0


def test_prefilter_helper_probe_real_decoders(backend):
    """The plugin library's real decoders (dropin/_build/libr433plugins.so, linked with dropin/helper_wrap.c) asked with their
    helpers wrapped: statistics of all 335, decoded events and what every package produced are the same as without the filter,
    and fewer records cross than under the fence alone (Neptune R900's search-then-length alone was a quarter of them)."""
    from rtl_433_amd import plugins
    if not plugins.available():
        pytest.skip("dropin/_build/libr433plugins.so not built")
    devs, protocols, names = load_device_table()
    iqs = [synth.ook_stream(s)[0] for s in range(96 if backend == "gpu" else 20)]
    res = {}
    for mode in ("plain", "fence_only", "filtered"):
        plug = plugins.Plugins()  # a fresh set of decoders (some keep state between calls)
        assert plug.helper_probe() is not None, "libr433plugins.so was built without dropin/helper_wrap.c: make -C dropin plugins"
        objs = [C.cast(p, C.POINTER(_lib.RDevice)).contents for p in plug.devices]
        eng = _engine(devs, backend)
        eng.L.r433_prefilter_forget()
        eng.L.r433_prefilter_set_helper_probe(None)
        tables = 0 if mode == "plain" else eng.probe_prefilter(plug.devices, helper=plug.helper_probe() if mode == "filtered" else None)
        eng.run_host(iqs)
        nev = eng.events()[1]
        dec = eng.dispatch_ordered(plug.devices, None, 8)
        res[mode] = dict(stats=_stats(objs), per_pkg=list(eng.decoded()), decoded=dec, nev=nev, tables=tables,
                         neptune=int(eng.prefilter_counts()[names.index("Neptune R900 flow meters")].sum()) if mode != "plain" else 0)
        eng.L.r433_prefilter_set_helper_probe(None)
        eng.L.r433_prefilter_forget()
        eng.close()
        plug.take()
        plug.close()
    a, f, b = res["plain"], res["fence_only"], res["filtered"]
    assert b["tables"] >= f["tables"] > 200
    for r in (f, b):
        assert a["stats"] == r["stats"] and a["per_pkg"] == r["per_pkg"] and a["decoded"] == r["decoded"]
    assert b["nev"] < 0.8 * f["nev"] and f["nev"] < 0.3 * a["nev"]  # (the wrapped search is lazier than the reference's: the fence alone learns the rows a pattern cannot fit in)
    assert b["neptune"] > f["neptune"]


class _Step(C.Structure):
    _fields_ = [("op", C.c_int), ("a", C.c_int), ("b", C.c_int), ("code", C.c_int)]


class _Prog(C.Structure):  # pf_prog, tests/plugins/pf_decoders.c
    _fields_ = [("n", C.c_int), ("s", _Step * 10), ("calls", C.c_ulong)]


def _random_case(seed, n_dev):
    """timing rows of the common OOK line codes and a program of first-line tests per decoder"""
    rng = np.random.default_rng(seed)
    devs = np.zeros(n_dev, dtype=po.DEV_DTYPE)
    progs = (_Prog * n_dev)()
    for i in range(n_dev):
        mod = int(rng.choice([3, 4, 5, 6, 6, 5, 4]))
        short = float(rng.choice([80, 100, 200, 250, 400, 500]))
        long_ = short if mod in (3, 4) else short * float(rng.choice([2, 3]))
        reset = float(rng.choice([600, 900, 1500, 3000, 6000, 10000]))
        gap = 0.0 if mod in (3, 4) else float(rng.choice([0, 700, 1000, 2000])) if reset > 2500 else float(rng.choice([0, 500]))
        sync = float(rng.choice([0, 0, 900, 1200])) if mod == 6 else 0.0
        tol = 0.0 if mod in (3, 4) else float(rng.choice([0, 80, 150]))
        devs[i] = (mod, short, 0.0 if mod == 3 else long_, reset, gap, sync, tol, 0)
        p = progs[i]
        p.n = int(rng.integers(1, 7))
        for k in range(p.n):
            op = int(rng.choice([0, 1, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10]))
            a, b = int(rng.integers(0, 40)), int(rng.integers(0, 64))
            if op == 0:
                a = int(rng.choice([1, 1, 2, 3]))
            elif op == 1:
                a, b = int(rng.integers(1, 4)), int(rng.integers(3, 30))
            elif op in (2, 8):
                a = int(rng.choice([4, 8, 12, 16, 24, 40, 64]))
            elif op == 3:
                b = int(rng.choice([16, 40, 80, 200, 600]))
            elif op in (5, 7):
                a = int(rng.choice([0x00, 0xff, 0xaa, 0x55, int(rng.integers(0, 256))]))
            elif op == 9:
                a, b = int(rng.integers(2, 9)), int(rng.integers(0, 2))
            p.s[k] = _Step(op, a, b, -int(rng.integers(0, 5)))
    return devs, progs


EMU_SEEDS = (0, 1, 7, 9)  # (7: the case that fails when BitSink::fire forgets the sync count of a tiny row)


# (every seed on the GPU, four of them on the emulator as well: spelled out as pairs so that no combination is collected only to be skipped)
@pytest.mark.parametrize("backend, seed", [pytest.param(b, s, marks=b.marks, id=f"{b.values[0]}-{s}") for b in BACKENDS for s in range(12)
                                          if b.values[0] != "emu" or s in EMU_SEEDS])
def test_prefilter_random_decoders(backend, plugins, seed):
    """Decoders drawn at random: 16 timing rows of the OOK line codes, each with a program of one to six first-line tests of
    the kinds the reference's decoders open with (head, other rows, sync count, content after an in-place inversion, a search, a
    checksum).  Whatever the probe makes of them -- head tables, tiny-row tables, nothing -- statistics, events per package and
    the calls that still arrive are those of the unfiltered run minus exactly what the device says it dropped."""
    n_dev = 16
    devs, progs = _random_case(1000 + seed, n_dev)
    iqs = [synth.ook_stream(5000 + 17 * seed + k, 50000)[0] for k in range(14)] + [synth.random_cu8(300 + seed, 6000)]
    runs = {}
    for mode in ("plain", "filtered"):
        eng = _engine(devs, backend)
        arr, objs = make_rdevices(devs)
        for i, o in enumerate(objs):
            o.decode_fn = C.cast(plugins.pf_dec_random, C.c_void_p).value
            o.decode_ctx = C.addressof(progs[i])
        tables = eng.probe_prefilter(arr) if mode == "filtered" else 0
        for p in progs:
            p.calls = 0
        eng.run_host(iqs)
        ev, nev = eng.events()
        dec = eng.dispatch(arr, n_threads=1)
        runs[mode] = dict(stats=_stats(objs), per_pkg=list(eng.decoded()), decoded=dec, nev=nev, records=_records(ev),
                          calls=[int(p.calls) for p in progs], tables=tables,
                          dropped=eng.prefilter_counts() if mode == "filtered" else None)
        eng.close()
    a, b = runs["plain"], runs["filtered"]
    assert a["stats"] == b["stats"] and a["per_pkg"] == b["per_pkg"] and a["decoded"] == b["decoded"]
    assert a["nev"] > 200  # (the case has something to filter)
    dropped = b["dropped"]
    assert int(dropped.sum()) == a["nev"] - b["nev"]
    for i in range(n_dev):
        assert a["calls"][i] - b["calls"][i] == int(dropped[i].sum())
    it = iter(a["records"])
    assert all(any(r == x for x in it) for r in b["records"])
    if b["tables"]:
        assert b["nev"] < a["nev"]
