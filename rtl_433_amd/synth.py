"""Deterministic synthetic captures for parity tests and bench.py.

The recipes follow SURVEY.md section 8(d): OOK bursts modelled on the reference's own
end-to-end fixture (reference tests/rtl_tcp_serve.py:46-71: a tone keyed on/off around the
128,128 bias), FSK bursts as constant-envelope 2-FSK.  Everything is seeded numpy so the
oracle, the reference harness and the GPU path can be fed byte-identical input.
"""
from __future__ import annotations

import numpy as np

CU8_FRAME_SAMPLES = 131072   # reference include/rtl_433.h:17 (262144-byte frames)
CS16_FRAME_SAMPLES = 65536


def _segments_to_mask(segs, n_total):
    """segs: list of (n_samples, on) -> bool mask of length n_total (zero padded / clipped)."""
    if not segs:
        return np.zeros(n_total, dtype=bool)
    lens = np.array([s[0] for s in segs], dtype=np.int64)
    vals = np.array([s[1] for s in segs], dtype=bool)
    m = np.repeat(vals, lens)
    if m.size >= n_total:
        return m[:n_total]
    return np.concatenate([m, np.zeros(n_total - m.size, dtype=bool)])


def ook_segments(bits, family, short, long_, gap=None, sync=0, repeats=1, repeat_gap=0):
    """Key-on/key-off schedule (in samples) of one OOK message.

    family 'pwm': 1 = short pulse, 0 = long pulse, fixed gap (reference OOK_PULSE_PWM convention)
    family 'ppm': fixed short pulse, 0 = short gap, 1 = long gap (OOK_PULSE_PPM)
    family 'mc' : Manchester, half-bit = short: 1 = on,off  0 = off,on (OOK_PULSE_MANCHESTER_ZEROBIT)
    """
    gap = short if gap is None else gap
    one = []
    if family == "pwm":
        for b in bits:
            one.append((short if b else long_, True))
            one.append((gap, False))
        if sync:
            one.append((sync, True))
            one.append((gap, False))
    elif family == "ppm":
        for b in bits:
            one.append((short, True))
            one.append((long_ if b else short, False))
        one.append((short, True))
        one.append((short, False))
    elif family == "mc":
        for b in bits:
            one.append((short, bool(b)))
            one.append((short, not b))
    else:
        raise ValueError(family)
    segs = []
    for r in range(repeats):
        segs.extend(one)
        if r + 1 < repeats:
            segs.append((repeat_gap, False))
    return segs


def modulate_cu8(mask, rng, rate, tone_hz, amplitude, sigma):
    """cu8 IQ: 128 + A*(cos, sin)(2 pi f t) while keyed, plus AWGN; rounded and clipped."""
    n = mask.size
    t = np.arange(n, dtype=np.float64)
    ph = 2.0 * np.pi * tone_hz / rate * t
    a = amplitude * mask
    i = 128.0 + a * np.cos(ph)
    q = 128.0 + a * np.sin(ph)
    if sigma > 0:
        i = i + rng.normal(0.0, sigma, n)
        q = q + rng.normal(0.0, sigma, n)
    out = np.empty(2 * n, dtype=np.uint8)
    out[0::2] = np.clip(np.rint(i), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(q), 0, 255).astype(np.uint8)
    return out


def ook_stream(seed, n_samples=65536, rate=250000, families=("pwm", "ppm", "mc")):
    """One config-2 style capture: a single OOK burst with seeded parameters.

    Returns (cu8 bytes as uint8 array of 2*n_samples, dict describing the burst)."""
    rng = np.random.default_rng(seed)
    fam = families[int(rng.integers(0, len(families)))]
    nbits = int(rng.integers(24, 65))
    bits = rng.integers(0, 2, nbits).astype(np.uint8)
    short_us = float(rng.integers(200, 501))
    us = rate / 1e6
    short = max(1, int(round(short_us * us)))
    long_ = 2 * short
    tone = float(rng.uniform(-60e3, 60e3))
    amp = float(rng.uniform(40.0, 120.0))
    sigma = float(rng.integers(0, 3))
    lead_in = int(rng.integers(1400, 4000))
    repeats = int(rng.integers(1, 4)) if fam != "mc" else 1
    segs = [(lead_in, False)] + ook_segments(bits, fam, short, long_, repeats=repeats, repeat_gap=6 * short)
    used = sum(s[0] for s in segs)
    mask = _segments_to_mask(segs, n_samples)
    iq = modulate_cu8(mask, rng, rate, tone, amp, sigma)
    meta = dict(family=fam, nbits=nbits, short=short, long=long_, tone=tone, amp=amp, sigma=sigma,
                lead_in=lead_in, repeats=repeats, used=used, bits=bits)
    return iq, meta


def ook_batch(n_streams, n_samples=65536, rate=250000, seed0=0):
    """Batch of independent cu8 captures, row s generated from seed0+s.  Shape (n_streams, 2*n_samples)."""
    out = np.empty((n_streams, 2 * n_samples), dtype=np.uint8)
    for s in range(n_streams):
        out[s], _ = ook_stream(seed0 + s, n_samples, rate)
    return out


def fsk_phase(bits_levels, rate, dev_hz):
    """Phase track of a 2-FSK signal; bits_levels is +1/-1 per sample."""
    return np.cumsum(2.0 * np.pi * dev_hz * bits_levels / rate)


def fsk_stream_cs16(seed, n_samples, rate=1024000, dev_hz=40e3, halfbit_us=50.0, coding="mc",
                    n_bursts=2, nbits=96, amp=0.8, sigma=0.01, lead_in=6000, gap=20000):
    """cs16 capture with constant-envelope FSK bursts (SURVEY 8c config-3 recipe).

    coding 'mc': Manchester, 1 -> (+dev,-dev) 0 -> (-dev,+dev), each half `halfbit_us`;
    coding 'pcm': NRZ, one level per bit of `halfbit_us`.
    Returns int16 array of 2*n_samples (interleaved I,Q)."""
    rng = np.random.default_rng(seed)
    hb = max(1, int(round(halfbit_us * rate / 1e6)))
    level = np.zeros(n_samples, dtype=np.float64)
    keyed = np.zeros(n_samples, dtype=bool)
    pos = lead_in
    for _ in range(n_bursts):
        bits = np.concatenate([np.zeros(16, dtype=np.uint8), rng.integers(0, 2, nbits).astype(np.uint8)])
        if coding == "mc":
            halves = np.empty(2 * bits.size, dtype=np.float64)
            halves[0::2] = np.where(bits == 1, 1.0, -1.0)
            halves[1::2] = -halves[0::2]
        else:
            pre = np.tile(np.array([1.0, -1.0]), 16)
            halves = np.concatenate([pre, np.where(rng.integers(0, 2, nbits) == 1, 1.0, -1.0)])
        track = np.repeat(halves, hb)
        end = min(n_samples, pos + track.size)
        if end <= pos:
            break
        level[pos:end] = track[: end - pos]
        keyed[pos:end] = True
        pos = end + gap
    ph = fsk_phase(level, rate, dev_hz)
    a = amp * 32767.0 * keyed
    i = a * np.cos(ph) + rng.normal(0.0, sigma * 32767.0, n_samples)
    q = a * np.sin(ph) + rng.normal(0.0, sigma * 32767.0, n_samples)
    out = np.empty(2 * n_samples, dtype=np.int16)
    out[0::2] = np.clip(np.rint(i), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.rint(q), -32768, 32767).astype(np.int16)
    return out


def fsk_stream_cu8(seed, n_samples, rate=250000, dev_hz=30e3, bit_us=100.0, coding="pcm",
                   n_bursts=2, nbits=64, amp=100.0, sigma=1.0, lead_in=3000, gap=8000):
    """cu8 capture with FSK bursts (PCM/NRZ by default), for the cu8 FM path."""
    rng = np.random.default_rng(seed)
    hb = max(1, int(round(bit_us * rate / 1e6)))
    level = np.zeros(n_samples, dtype=np.float64)
    keyed = np.zeros(n_samples, dtype=bool)
    pos = lead_in
    for _ in range(n_bursts):
        if coding == "mc":
            bits = np.concatenate([np.zeros(8, dtype=np.uint8), rng.integers(0, 2, nbits).astype(np.uint8)])
            halves = np.empty(2 * bits.size, dtype=np.float64)
            halves[0::2] = np.where(bits == 1, 1.0, -1.0)
            halves[1::2] = -halves[0::2]
        else:
            halves = np.concatenate([np.tile(np.array([1.0, -1.0]), 12),
                                     np.where(rng.integers(0, 2, nbits) == 1, 1.0, -1.0)])
        track = np.repeat(halves, hb)
        end = min(n_samples, pos + track.size)
        if end <= pos:
            break
        level[pos:end] = track[: end - pos]
        keyed[pos:end] = True
        pos = end + gap
    ph = fsk_phase(level, rate, dev_hz)
    a = amp * keyed
    i = 128.0 + a * np.cos(ph) + rng.normal(0.0, sigma, n_samples)
    q = 128.0 + a * np.sin(ph) + rng.normal(0.0, sigma, n_samples)
    out = np.empty(2 * n_samples, dtype=np.uint8)
    out[0::2] = np.clip(np.rint(i), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(q), 0, 255).astype(np.uint8)
    return out


def noise_cu8(seed, n_samples, sigma=3.0):
    """Pure noise capture (no bursts): exercises the idle estimator and spurious-pulse rules."""
    rng = np.random.default_rng(seed)
    v = 128.0 + rng.normal(0.0, sigma, 2 * n_samples)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def random_cu8(seed, n_samples):
    """Uniform random bytes: worst case for every integer corner (saturation, 32768 envelope)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, 2 * n_samples, dtype=np.uint8)
