"""Seeded parity cases shared by the golden generator (oracle/gen_golden.py) and the tests."""
from __future__ import annotations

import os

import numpy as np

from rtl_433_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = [
    "kat", "ook0", "ook1", "ook2", "ook3", "ook4", "ook5", "ook6", "ook7", "ook_long", "noise", "random",
    "silence", "fsk_cu8", "fsk_cu8_minmax", "fsk_cs16", "fsk_cs16_classic", "fsk_cs16_pcm", "empty", "tiny", "ragged",
]


def make_case(name):
    """-> (iq array, sample_size, sample rate, centre frequency)"""
    if name == "kat":
        return np.fromfile(os.path.join(GOLD, "nice_250k.cu8"), dtype=np.uint8), 2, 250000, 433920000
    if name.startswith("ook") and name[3:].isdigit():
        iq, _ = synth.ook_stream(int(name[3:]))
        return iq, 2, 250000, 433920000
    if name == "ook_long":  # five bursts over 2.5 reference frames (131072 samples each)
        iq = np.concatenate([synth.ook_stream(100 + k)[0] for k in range(5)])
        return iq, 2, 250000, 433920000
    if name == "noise":
        return synth.noise_cu8(1, 300000, 6.0), 2, 250000, 433920000
    if name == "random":
        return synth.random_cu8(2, 200000), 2, 250000, 433920000
    if name == "silence":
        return np.full(2 * 140000, 128, dtype=np.uint8), 2, 250000, 433920000
    if name == "fsk_cu8":
        return synth.fsk_stream_cu8(3, 150000), 2, 250000, 433920000
    if name == "fsk_cu8_minmax":
        return synth.fsk_stream_cu8(4, 150000, n_bursts=3), 2, 250000, 868300000
    if name == "fsk_cs16":
        return synth.fsk_stream_cs16(3, 200000), 4, 1024000, 868000000
    if name == "fsk_cs16_classic":
        return synth.fsk_stream_cs16(5, 150000), 4, 1024000, 433920000
    if name == "fsk_cs16_pcm":
        return synth.fsk_stream_cs16(6, 150000, coding="pcm", halfbit_us=52.0, nbits=64), 4, 1024000, 868000000
    if name == "empty":
        return np.zeros(0, dtype=np.uint8), 2, 250000, 433920000
    if name == "tiny":
        return synth.random_cu8(7, 10), 2, 250000, 433920000
    if name == "ragged":
        return synth.ook_stream(9, 70001)[0], 2, 250000, 433920000
    raise KeyError(name)


def fpdm_for(freq):
    """FSK detector resolution for AUTO mode, reference src/rtl_433.c:1094-1102."""
    return 1 if freq > 800000000 else 0


def autolevel_capture(seed=31, n_frames=5, amp=15.0, sigma=1.0):
    """Several reference frames of quiet noise with OOK bursts too weak for the default -12 dB level:
    only -Y autolevel (which lowers the level frame by frame as the noise estimate settles) sees them."""
    rng = np.random.default_rng(seed)
    n = n_frames * 131072 + 4321
    segs = [(30000, False)]
    while sum(s[0] for s in segs) < n - 40000:
        bits = rng.integers(0, 2, 32).astype(np.uint8)
        segs += synth.ook_segments(bits, "pwm", 120, 240, repeats=1) + [(int(rng.integers(50000, 90000)), False)]
    mask = synth._segments_to_mask(segs, n)
    return synth.modulate_cu8(mask, rng, 250000, 15e3, amp, sigma)


def mixed_2000k_capture(seed=77, n=900000):
    """Config-5 style: one 2 MS/s cu8 stream with OOK bursts and FSK bursts over a noise floor that
    steps up half way (exercises -Y autolevel, the FM low-pass option and both detectors in one capture)."""
    rng = np.random.default_rng(seed)
    rate = 2000000
    t = np.arange(n) / rate
    amp = np.zeros(n)
    phase = np.zeros(n)
    pos = 40000
    k = 0
    while pos < n - 120000:
        if k % 2 == 0:  # OOK PWM burst, 200/400 us pulses
            bits = rng.integers(0, 2, 24).astype(np.uint8)
            segs = synth.ook_segments(bits, "pwm", 400, 800, repeats=1)
            m = synth._segments_to_mask(segs, sum(s[0] for s in segs))
            a = float(rng.uniform(20, 90))
            amp[pos:pos + len(m)] = a * m[: n - pos]
            phase[pos:pos + len(m)] = 2 * np.pi * 30e3 * t[: len(m)]
            pos += len(m)
        else:  # FSK PCM burst, 100 us bits, +-50 kHz
            nb = 64
            bits = np.concatenate([np.tile([1, 0], 12), rng.integers(0, 2, nb - 24)])
            lv = np.repeat(2.0 * bits - 1.0, 200)
            ph = 2 * np.pi * np.cumsum(lv * 50e3) / rate
            amp[pos:pos + len(ph)] = 80.0
            phase[pos:pos + len(ph)] = ph
            pos += len(ph)
        pos += int(rng.integers(60000, 110000))
        k += 1
    sigma = np.where(np.arange(n) < n // 2, 1.0, 3.0)
    i = 128 + amp * np.cos(phase) + rng.normal(0, 1, n) * sigma
    q = 128 + amp * np.sin(phase) + rng.normal(0, 1, n) * sigma
    out = np.empty(2 * n, dtype=np.uint8)
    out[0::2] = np.clip(np.rint(i), 0, 255)
    out[1::2] = np.clip(np.rint(q), 0, 255)
    return out, rate


def dump_capture(name):
    """Inputs of the -w dump-format vectors (tests/golden/gen_dump_golden.py): (iq, sample_size, file name).
    Lengths that are no multiple of the kernels' group size, so the ragged tail is covered."""
    if name == "cu8":
        return synth.ook_batch(1, 32768, 250000, seed0=5)[0][: 2 * 32765].copy(), 2, "g_433.92M_250k.cu8"
    iq = np.asarray(synth.fsk_stream_cs16(3, 40003))
    return iq[: 2 * 40003].copy(), 4, "g_868M_1024k.cs16"


def grab_capture(name):
    """Inputs of the sample-grabber vectors (tests/golden/gen_grab_golden.py): long quiet stretches around bursts so that
    every window the reference writes lies inside the file."""
    quiet = lambda n: np.tile(np.array([128, 127], dtype=np.uint8), n)
    if name == "one_frame":  # five close bursts: one tracked frame
        return np.concatenate([quiet(200000), make_case("ook_long")[0]])
    return np.concatenate([quiet(300000), synth.ook_stream(1)[0], quiet(400000), synth.ook_stream(2)[0], quiet(300000)])


# ---- slicer matrix: every line-code slicer x random timings (oracle/gen_slicer_golden.py, tests/test_slicer_matrix.py) ----

# enum modulation_types, reference include/r_device.h:24-40
ALL_MODULATIONS = (3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 16, 17, 18)


def slicer_matrix_rows(seed=77, per_modulation=12):
    """Synthetic decoder rows: each of the 13 modulations with `per_modulation` random timing sets (us), including the
    two no default-enabled protocol uses (OOK_PULSE_PIWM_RAW = 8, OOK_PULSE_NRZS = 12), Klimalogg's own row
    (reference src/devices/klimalogg.c:112-121) and a few degenerate ones (zero widths)."""
    from oracle.pyoracle import DEV_DTYPE
    rng = np.random.default_rng(seed)
    rows = []
    for mod in ALL_MODULATIONS:
        for k in range(per_modulation):
            short = float(rng.choice([26, 40, 52, 100, 200, 300, 400, 500, 800])) * float(rng.uniform(0.9, 1.1))
            long_ = short * float(rng.choice([1.0, 1.5, 2.0, 3.0]))
            reset = float(rng.choice([600, 1000, 2500, 6000, 12000]))
            gap = float(rng.choice([0, 0, 700, 1500, 3000]))
            sync = float(rng.choice([0, 0, 0, 700, 1500]))
            tol = float(rng.choice([0, 0, 20, 60, 150]))
            if k == per_modulation - 1:  # degenerate: a width that rounds to zero samples / no long width
                # (a zero long width is a SIGFPE in the reference's PCM / PPM / RZI slicers, e.g. src/pulse_slicer.c:97,
                # not a case; OSv1 (10), NRZS (12) and FSK Manchester (18) never divide by it)
                short, long_ = (short, 0.0) if mod in (10, 12, 18) else (2.0, 3.0)
            rows.append((mod, np.float32(short), np.float32(long_), np.float32(reset), np.float32(gap), np.float32(sync),
                         np.float32(tol), 0))
    rows.append((12, 26.0, 0.0, 1000.0, 0.0, 0.0, 0.0, 0))  # Klimalogg
    return np.array(rows, dtype=DEV_DTYPE)


SLICER_CASES = ["ook20", "ook21", "ook22", "ook23", "ook24", "ook25", "nrz", "fsk_cu8", "fsk_cu8_mc", "random"]


def make_slicer_case(name):
    """-> (iq, sample_size, rate, centre frequency): what the slicer matrix is run over."""
    if name.startswith("ook"):
        return synth.ook_stream(int(name[3:]))[0], 2, 250000, 433920000
    if name == "nrz":  # level shifts at multiples of 100 us: what PIWM_RAW / NRZS / PCM / RZI expect
        rng = np.random.default_rng(5)
        segs = [(3000, False)]
        for rep in range(3):
            level = True
            for _ in range(60):
                segs.append((25 * int(rng.integers(1, 5)), level))
                level = not level
            segs.append((4000 + 1500 * rep, False))
        mask = synth._segments_to_mask(segs, 65536)
        return synth.modulate_cu8(mask, rng, 250000, 20e3, 90.0, 1.0), 2, 250000, 433920000
    if name == "fsk_cu8":
        return synth.fsk_stream_cu8(13, 120000), 2, 250000, 433920000
    if name == "fsk_cu8_mc":
        return synth.fsk_stream_cu8(14, 120000, coding="mc", n_bursts=3), 2, 250000, 868300000
    if name == "random":
        return synth.random_cu8(15, 120000), 2, 250000, 433920000
    raise KeyError(name)
