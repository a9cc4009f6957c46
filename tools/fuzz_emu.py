"""Fuzzing against the oracle: random signals x random flow options, on the wave emulator (CPU, default) or on
the MI355X through the product library (--gpu).
    python tools/fuzz_emu.py [--gpu] [n_cases] [first_seed]
Every case: 1-3 captures in one launch, byte-for-byte comparison of package and event records and of the
am/fm taps.  Prints the seed of every failing case."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle as po
from rtl_433_amd import synth
from rtl_433_amd.engine import load_device_table
from tests.emu import host

DEVS = load_device_table()[0]


def rand_ook(rng, rate, n_max):
    """OOK bursts with adversarial timing: very short (spurious) pulses, very long packages, gaps around the
    end-of-package rules, amplitude steps."""
    segs = [(int(rng.integers(0, 3000)), False)]
    for _ in range(int(rng.integers(1, 6))):
        style = rng.integers(0, 6)
        short = int(rng.integers(3, 200)) if style != 1 else int(rng.integers(3, 12))
        nb = int(rng.integers(4, 80)) if style != 2 else int(rng.integers(1300, 1500))  # > PD_MAX_PULSES
        for _b in range(nb):
            w = short * int(rng.integers(1, 4))
            g = short * int(rng.integers(1, 4))
            if style == 3 and rng.random() < 0.1:
                g = int(rng.integers(2000, 30000))  # around 10 ms / 100 ms at 250k
            segs += [(w, True), (g, False)]
        segs.append((int(rng.integers(100, 40000)), False))
    n = min(n_max, sum(s[0] for s in segs))
    return synth._segments_to_mask(segs, n)


def make_capture(rng, ss, rate):
    kind = rng.integers(0, 8)
    n_max = int(rng.integers(3000, 140000))
    if ss == 2:
        if kind == 0:
            return synth.random_cu8(int(rng.integers(1 << 30)), int(rng.integers(1, 20000)))
        if kind == 1:
            return synth.noise_cu8(int(rng.integers(1 << 30)), n_max, float(rng.choice([0.0, 0.5, 1, 3, 8, 20, 40])))
        if kind == 2:
            if rng.integers(0, 2):  # FSK of every shape: narrow / wide shift, short / long bits, weak and noisy
                return synth.fsk_stream_cu8(int(rng.integers(1 << 30)), n_max, rate=rate, n_bursts=int(rng.integers(1, 4)),
                                            dev_hz=float(rng.choice([4e3, 12e3, 30e3, 70e3])), bit_us=float(rng.choice([24, 60, 100, 400])),
                                            coding=str(rng.choice(["pcm", "mc"])), nbits=int(rng.integers(8, 300)),
                                            amp=float(rng.choice([8, 20, 60, 120])), sigma=float(rng.choice([0, 1, 3, 8])))
            return synth.fsk_stream_cu8(int(rng.integers(1 << 30)), n_max, rate=rate, n_bursts=int(rng.integers(1, 4)))
        if kind == 3:  # FSK with very many transitions (> 1200 FSK pulses: ring overflow)
            n = n_max
            t = np.arange(n)
            lv = np.sign(np.sin(2 * np.pi * t / float(rng.integers(24, 60))))
            ph = 2 * np.pi * np.cumsum(lv * 40e3) / rate
            a = 90.0 * (t > 2000) * (t < n - 3000)
            sg = float(rng.choice([0, 1, 2]))
            i = 128 + a * np.cos(ph) + rng.normal(0, 1, n) * sg
            q = 128 + a * np.sin(ph) + rng.normal(0, 1, n) * sg
            out = np.empty(2 * n, dtype=np.uint8)
            out[0::2] = np.clip(np.rint(i), 0, 255)
            out[1::2] = np.clip(np.rint(q), 0, 255)
            return out
        mask = rand_ook(rng, rate, n_max)
        amp = float(rng.choice([6, 12, 25, 60, 110, 127]))
        sig = float(rng.choice([0, 0, 1, 2, 5, 12]))
        iq = synth.modulate_cu8(mask, rng, rate, float(rng.uniform(-80e3, 80e3)), amp, sig)
        if kind == 7 and len(iq) > 4000:  # amplitude step / saturated stretch
            k = int(rng.integers(0, len(iq) // 2 - 1000)) * 2
            iq[k:k + 2000] = rng.choice([0, 255])
        return iq
    # cs16
    if kind < 3:
        return synth.fsk_stream_cs16(int(rng.integers(1 << 30)), n_max, rate=rate, n_bursts=int(rng.integers(1, 4)),
                                     coding=str(rng.choice(["mc", "pcm"])), sigma=float(rng.choice([0.0, 0.01, 0.05])))
    if kind == 3:
        return rng.integers(-32768, 32768, 2 * int(rng.integers(1, 20000)), dtype=np.int64).astype(np.int16)
    mask = rand_ook(rng, rate, n_max)
    n = len(mask)
    ph = 2 * np.pi * float(rng.uniform(-0.2, 0.2)) * np.arange(n)
    a = float(rng.choice([300, 3000, 20000, 32000])) * mask
    sg = float(rng.choice([0, 30, 300]))
    out = np.empty(2 * n, dtype=np.int16)
    out[0::2] = np.clip(np.rint(a * np.cos(ph) + rng.normal(0, 1, n) * sg), -32768, 32767)
    out[1::2] = np.clip(np.rint(a * np.sin(ph) + rng.normal(0, 1, n) * sg), -32768, 32767)
    return out


STATS = {"cases": 0, "roles": 0, "run_again": 0}  # cases run, of them as two launches, captures their run-again launch took
BIG = False  # --gpu: some cases are captures long enough for the automatic split (>= 2^20 samples)


def long_ook(rng, rate):
    n = int(rng.integers(1100000, 2600000))
    segs = [(int(rng.integers(500, 30000)), False)]
    while sum(s[0] for s in segs) < n:
        short = int(rng.integers(20, 300))
        for _b in range(int(rng.integers(8, 120))):
            segs += [(short * int(rng.integers(1, 4)), True), (short * int(rng.integers(1, 4)), False)]
        segs.append((int(rng.integers(2000, 300000)), False))
    mask = synth._segments_to_mask(segs, n)
    return synth.modulate_cu8(mask, rng, rate, float(rng.uniform(-80e3, 80e3)), float(rng.choice([15, 40, 100])), float(rng.choice([0, 1, 2, 4, 9])))


def one_case(seed, run=None):
    """run(caps, ss, rate, devs, fpdm=, taps=, enable_fm=, split=, **flow options) -> dict like tests.emu.host.emu_run"""
    run = run or host.emu_run
    rng = np.random.default_rng(seed)
    ss = int(rng.choice([2, 2, 2, 4]))
    rate = int(rng.choice([250000, 250000, 1000000, 1024000, 2000000, 48000]))
    kw = {}
    if rng.random() < 0.2 and ss == 2:
        kw["use_mag_est"] = 1
    if rng.random() < 0.2:
        kw["level_limit_db"] = float(rng.choice([-5.0, -10.0, -20.0]))
    if rng.random() < 0.3:
        kw["min_level_db"] = float(rng.choice([-6.0, -20.0, -30.0]))
    if rng.random() < 0.2:
        kw["min_snr_db"] = float(rng.choice([3.0, 6.0, 12.0]))
    if rng.random() < 0.3:
        kw["fm_low_pass"] = float(rng.choice([0.02, 0.05, 0.15, 0.3, 0.45, 50.0, 30000.0]))
    if rng.random() < 0.3:
        kw["frame_samples"] = int(rng.choice([64, 192, 2048, 4096, 10048, 65536]))
    if rng.random() < 0.2:
        kw["auto_level"] = 1.0
    fpdm = int(rng.integers(0, 2))
    enable_fm = int(rng.random() < 0.85)
    devs = DEVS[rng.choice(len(DEVS), int(rng.integers(0, 30)), replace=False)] if rng.random() < 0.8 else None
    if devs is not None:
        devs = devs[np.argsort(rng.random(len(devs)))]
        if not enable_fm:
            devs = devs[devs["modulation"] < 16]
        if len(devs) == 0:
            devs = None
    caps = [make_capture(rng, ss, rate) for _ in range(int(rng.integers(1, 4)))]
    split = int(rng.choice([0, 0, 4096, 8192, 20000]))
    load_format = 0
    if ss == 2 and rng.random() < 0.12:  # the same bytes taken for an am.s16 / fm.s16 file (any int16 word, negative AM included)
        load_format = int(rng.integers(1, 3))
    if BIG and ss == 2 and rng.random() < 0.25:
        caps = [long_ook(rng, rate)] + caps[:1]
        split = int(rng.choice([1, 1, 65536, 200000]))  # 1 = R433_SPLIT_AUTO
        kw.pop("frame_samples", None) if kw.get("frame_samples", 65536) < 2048 else None
    blind = 1 if split and rng.random() < 0.5 else 0  # R433_DEBUG_SPLIT_BLIND
    form = (0, 4096, 32768)[seed % 3]  # the launch's own choice / R433_DEBUG_ONE_WAVE / R433_DEBUG_PAIR: both forms of the detection kernel
    if seed % 4 == 1:
        form |= 128  # R433_DEBUG_FORCE_ORDER: the workgroups take the captures heaviest first (a list made on the device)
    if seed % 5 == 2:
        form |= 8  # R433_DEBUG_SMALL_STRETCH: the slicer fan-out three packages at a time (cursors rewound per stretch)
    if seed % 7 == 3:
        form |= 65536  # R433_DEBUG_STATIC_SLICE: slicer workgroups at fixed strides instead of drawing from the cursors
    if seed % 11 == 5:
        form |= 262144  # R433_DEBUG_NO_LAZY: every tile filtered
    if seed % 9 == 4:
        form |= 1048576  # R433_DEBUG_SKEW_SLICE: the chunks of devices get unequal shares of the sizing pass's workgroups
    if seed % 13 == 6:
        form |= 131072  # R433_DEBUG_ONE_SLICE_LAUNCH: the slicers' sizing pass as one launch instead of large / small packages apart
    if seed % 4 == 2 and not (form & 4096):
        form |= 4194304  # R433_DEBUG_SPLIT_ROLES: producers and consumers as two launches over tile records in HBM (+ the run-again launch)
    form |= int(os.environ.get("R433_FUZZ_DEBUG", "0"), 0)  # (exploring: a switch for every case of a sweep)
    # One case in three with the sample taps; without them the detection kernel leaves tiles that cannot move the detector
    # unfiltered (lazy tiles), and the per-frame envelope sums are compared instead.
    taps = seed % 3 == 0
    g = run(caps, ss, rate, devs, fpdm=fpdm, taps=taps, enable_fm=enable_fm, split=split, debug=blind | form,
            **(dict(kw, input_format=2 + load_format) if load_format else kw))
    cfg = po.default_flow_cfg(ss, rate, fpdm=fpdm, enable_fm=enable_fm, load_format=load_format, **kw)
    pk, ev, base = b"", b"", 0
    for s, a in enumerate(caps):
        o = po.oracle_flow(a, devs, cfg, stream_index=s, pkg_base=base, taps=taps)
        n = a.nbytes // ss
        if taps and n and not (np.array_equal(g["taps"][1][s, :n], o["am"]) and np.array_equal(g["taps"][2][s, :n], o["fm"])
                               and np.array_equal(g["taps"][0][s, :n], o["env"])):
            return f"taps differ (capture {s})"
        if "sums" in g and n:
            nf = (n + cfg.frame_samples - 1) // cfg.frame_samples
            if not np.array_equal(g["sums"][s, :nf], o["frame_sums"][:nf]):
                return f"frame sums differ (capture {s})"
        pk += o["packages"]
        ev += o["events"]
        base += o["n_packages"]
    STATS["cases"] += 1
    STATS["roles"] += int(g.get("split", {}).get("detect_form", 0) == 45)
    STATS["run_again"] += int(g.get("split", {}).get("pieces_rerun", 0)) if g.get("split", {}).get("detect_form", 0) == 45 else 0
    if g["packages"][0] != pk:
        return f"packages differ ({g['n_packages']} vs {base})"
    if g["events"][0] != ev:
        return "events differ"
    return None


def gpu_run(caps, ss, rate, devs, fpdm=0, taps=False, enable_fm=1, split=0, debug=0, **kw):
    """The same contract on the MI355X through the product library."""
    import torch
    from rtl_433_amd.engine import BatchEngine, flow_cfg
    n = len(caps)
    lens = np.array([a.nbytes for a in caps], dtype=np.uint32)
    stride = max(16, int((lens.max() + 15) // 16 * 16))
    hostbuf = np.zeros((n, stride), dtype=np.uint8)
    for i, a in enumerate(caps):
        hostbuf[i, :a.nbytes] = a.view(np.uint8)
    eng = BatchEngine(flow_cfg(ss, rate, fpdm=fpdm, enable_fm=enable_fm, **kw), devs)
    eng.set_split(split)
    eng.set_debug(debug)
    if taps:
        eng.enable_taps(n, max(1, stride // ss))
    npk = eng.run(torch.from_numpy(hostbuf).cuda(), lens)
    out = dict(n_packages=npk, packages=eng.packages(), events=eng.events(), split=eng.split_stats(), sums=eng.frame_sums(n))
    if taps:
        out["taps"] = eng.taps()
    eng.close()
    return out


if __name__ == "__main__":
    runner = None
    if "--gpu" in sys.argv:
        sys.argv.remove("--gpu")
        runner = gpu_run
        BIG = True
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    bad = []
    for seed in range(first, first + n_cases):
        try:
            r = one_case(seed, runner)
        except Exception as e:  # noqa
            r = "exception: " + repr(e)
            traceback.print_exc()
        if r:
            bad.append(seed)
            print(f"seed {seed}: {r}", flush=True)
        if (seed - first + 1) % 50 == 0:
            print(f"... {seed - first + 1} cases, {len(bad)} failures so far, {time.time() - t0:.0f} s", flush=True)
    print(f"{n_cases} cases, {len(bad)} failures {bad} in {time.time() - t0:.0f} s; {STATS}")
