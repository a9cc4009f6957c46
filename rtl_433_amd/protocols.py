"""Protocol-valid synthetic transmissions: signals the reference's REAL decoders (src/devices/*.c) accept.

The config-2 bursts of synth.py carry random payloads: they exercise detection and the slicers, but of 335 decoders
hardly one ever says yes.  Here every transmission is a frame its decoder checks out -- length, fixed fields, checksum --
keyed with the line code and the timing of that decoder's r_device entry, so the decoded JSON (model, id, values,
mod / freq / rssi) can be compared with the reference CLI's output, priority gating and stateful decoders included.

Frames come from rtl_433_amd/data/protocol_frames.json (made by tools/gen_protocol_frames.py: payload fields drawn at
random, integrity fields found by asking the reference's own decode_fn, so nothing here restates a decoder); this
module only turns bits into a key-on / key-off (or frequency) schedule and that into IQ samples.  The line codes are the
inverses of the reference's slicers (src/pulse_slicer.c:68-918), written from their bit conventions:

    pwm      short pulse = 1, long pulse = 0, fixed gap; optional sync pulse in front of a row
    ppm      fixed pulse, short gap = 0, long gap = 1, a closing pulse
    mc       Manchester, half bit = short_width; the slicer starts every row with a 0 of its own
    dmc      differential Manchester: a level change in mid-bit = 1 (two short symbols), none = 0 (one long symbol)
    piwm     pulse-interval and -width: every symbol (pulse or gap alike) short = 1, long = 0
    osv1     Oregon Scientific v1 Manchester with its sync pulses
    pcm      NRZ or RZ bits of fixed width (OOK or FSK)
"""
from __future__ import annotations

import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FRAMES_PATH = os.path.join(HERE, "data", "protocol_frames.json")
_frames = None


def frames():
    global _frames
    if _frames is None:
        with open(FRAMES_PATH) as f:
            _frames = json.load(f)
    return _frames


# ---------------------------------------------------------------- line codes: bits -> [(microseconds, level)]
# level: 1 = carrier on (OOK) or the upper frequency (FSK), 0 = carrier off, -1 = the lower frequency (FSK)

def _pairs(pulses):
    out = []
    for on, off in pulses:
        out.append((on, 1))
        if off:
            out.append((off, 0))
    return out


def code_pwm(bits, short, long_, gap, sync=0, sync_gap=None):
    p = []
    if sync:
        p.append((sync, gap if sync_gap is None else sync_gap))
    for b in bits:
        p.append((short if b else long_, gap))
    return _pairs(p)


def code_ppm(bits, pulse, gap0, gap1, closing=True):
    p = [(pulse, gap1 if b else gap0) for b in bits]
    out = _pairs(p)
    if closing:
        out.append((pulse, 1))
    return out


def code_mc(bits, half, lead_one=False):
    """Manchester as pulse_slicer_manchester_zerobit reads it: a 1 is on-then-off, a 0 off-then-on.  The slicer puts a 0
    in front of what it sees on its own account; `bits` is what should FOLLOW that 0."""
    out = []
    for b in bits:
        out += [(half, 1), (half, 0)] if b else [(half, 0), (half, 1)]
    return out


def code_levels(levels, unit):
    """one level per unit of time (NRZ), merged into runs"""
    out = []
    for lv in levels:
        if out and out[-1][1] == lv:
            out[-1] = (out[-1][0] + unit, lv)
        else:
            out.append((unit, lv))
    return out


def code_dmc(bits, short, long_, start_level=1):
    """Differential Manchester, pulse_slicer_dmc: a short symbol followed by a short symbol = 1, one long symbol = 0;
    symbols alternate pulse / gap."""
    out, lv = [], start_level
    for b in bits:
        if b:
            out += [(short, lv), (short, 1 - lv)]
        else:
            out.append((long_, lv))
            lv = 1 - lv
    return out


def code_piwm(bits, short, long_, start_level=1):
    """pulse_slicer_piwm_dc: every symbol, pulse or gap, short = 1 / long = 0"""
    out, lv = [], start_level
    for b in bits:
        out.append((short if b else long_, lv))
        lv = 1 - lv
    return out


def merge(schedule):
    out = []
    for us, lv in schedule:
        if us <= 0:
            continue
        if out and out[-1][1] == lv:
            out[-1] = (out[-1][0] + us, lv)
        else:
            out.append((us, lv))
    return out


# ---------------------------------------------------------------- schedule -> IQ

def render_cu8(schedule, rate, rng, fsk=False, lead_us=8000.0, tail_us=None, tone_hz=None, amp=None, sigma=None, dev_hz=40e3,
               n_samples=None):
    """cu8 IQ of one transmission.  OOK: a tone keyed on and off; FSK: a constant envelope whose frequency steps between
    +-dev_hz.  Random carrier offset, amplitude and noise like synth.ook_stream unless given."""
    us = rate / 1e6
    sched = merge(schedule)
    lens = [max(1, int(round(d * us))) for d, _ in sched]
    lead = int(round(lead_us * us))
    tail = int(round((tail_us if tail_us is not None else 12000.0) * us))
    n = lead + sum(lens) + tail
    if n_samples is not None:
        n = n_samples
    level = np.zeros(n, dtype=np.float64)
    keyed = np.zeros(n, dtype=bool)
    pos = lead
    for ln, (_, lv) in zip(lens, sched):
        end = min(n, pos + ln)
        if end > pos:
            if fsk:
                keyed[pos:end] = lv != 0
                level[pos:end] = float(lv)
            else:
                keyed[pos:end] = lv == 1
        pos += ln
    tone = float(rng.uniform(-40e3, 40e3)) if tone_hz is None else tone_hz
    a = float(rng.uniform(50.0, 110.0)) if amp is None else amp
    sg = float(rng.integers(0, 3)) if sigma is None else sigma
    t = np.arange(n, dtype=np.float64)
    if fsk:
        ph = np.cumsum(2.0 * np.pi * (dev_hz * level) / rate)
        if tone_hz is not None:
            ph = ph + 2.0 * np.pi * tone_hz / rate * t
    else:
        ph = 2.0 * np.pi * tone / rate * t
    i = 128.0 + a * keyed * np.cos(ph)
    q = 128.0 + a * keyed * np.sin(ph)
    if sg > 0:
        i = i + rng.normal(0.0, sg, n)
        q = q + rng.normal(0.0, sg, n)
    out = np.empty(2 * n, dtype=np.uint8)
    out[0::2] = np.clip(np.rint(i), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.rint(q), 0, 255).astype(np.uint8)
    return out


# ---------------------------------------------------------------- the protocols

def _bits(s):
    return [int(c) for c in s if c in "01"]


def _rows(frame):
    """a frame of the JSON file: one bit string, or a list of them (the rows of one transmission)"""
    return [_bits(r) for r in (frame if isinstance(frame, list) else [frame])]


def _repeat(one_row_fn, rows, repeats, row_gap):
    """rows keyed one after the other, `row_gap` microseconds of silence between them, the lot `repeats` times"""
    out = []
    k = 0
    for _ in range(repeats):
        for r in rows:
            if k:
                out.append((row_gap, 0))
            out += one_row_fn(r)
            k += 1
    return out


# name -> dict(model, protocol, fsk, rate, schedule(frame) -> [(us, level)], and what the reference CLI needs to be told)
PROTOCOLS = {}


def protocol(name, **kw):
    def deco(fn):
        PROTOCOLS[name] = {**dict(name=name, schedule=fn, fsk=False, rate=250000, freq=433920000), **kw}
        return fn
    return deco


def transmission(name, seed, rate=None, n_samples=None, **render_kw):
    """-> (cu8 IQ as uint8 array, dict(name, model, frame, rate, freq)) : frame number `seed` of the protocol's list."""
    p = PROTOCOLS[name]
    fr = frames()[name]
    frame = fr[seed % len(fr)]
    rng = np.random.default_rng(1000003 * (seed + 1) + sum(map(ord, name)))
    rate = rate or p["rate"]
    iq = render_cu8(p["schedule"](frame), rate, rng, fsk=p["fsk"], n_samples=n_samples, **{**p.get("render", {}), **render_kw})
    return iq, dict(name=name, model=p["model"], frame=frame, rate=rate, freq=p["freq"])


def file_name(name, seed, rate, freq):
    """a file name the reference CLI reads rate and frequency from (include/fileformat.h:100-128)"""
    f = f"{freq / 1e6:.2f}M" if freq % 1000000 else f"{freq // 1000000}M"
    return f"p_{name}_{seed:04d}_{f}_{rate // 1000}k.cu8"


# ---- OOK_PULSE_PPM ----

def _ppm_rows(frame, pulse, gap0, gap1, row_gap, repeats):
    return _repeat(lambda r: code_ppm(r, pulse, gap0, gap1), _rows(frame), repeats, row_gap)


@protocol("rubicson", model="Rubicson-Temperature", protocol=2)
def _rubicson(frame):  # src/devices/rubicson.c: 36 bits, at least three equal rows
    return _ppm_rows(frame, 500, 1000, 2000, 4000, 4)


@protocol("nexus", model="Nexus-TH", protocol=19)
def _nexus(frame):  # src/devices/nexus.c: priority 10 -- runs only where no priority-0 decoder (Rubicson) had an event
    return _ppm_rows(frame, 500, 1000, 2000, 4000, 4)


@protocol("prologue", model="Prologue-TH", protocol=3)
def _prologue(frame):  # src/devices/prologue.c: 36 bits, four equal rows; priority 10
    return _ppm_rows(frame, 500, 2000, 4000, 8500, 5)


# ---- OOK_PULSE_PWM ----

@protocol("generic_remote", model="Generic-Remote", protocol=30)
def _generic_remote(frame):  # src/devices/generic_remote.c: one row of 25 bits, fixed period
    return code_pwm(_rows(frame)[0], 464, 1404, 464) + [(20000, 0)] + code_pwm(_rows(frame)[0], 464, 1404, 464)


@protocol("s3318p", model="Conrad-S3318P", protocol=47)
def _s3318p(frame):  # src/devices/s3318p.c: 42 bits, four equal rows
    return _ppm_rows(frame, 500, 1900, 3800, 5200, 5)


@protocol("tfa_pool", model="TFA-Pool", protocol=56)
def _tfa_pool(frame):  # src/devices/tfa_pool_thermometer.c: 28 bits, seven equal rows
    return _ppm_rows(frame, 500, 2000, 4600, 8800, 8)


@protocol("thermopro_tp11", model="Thermopro-TP11", protocol=84)
def _tp11(frame):  # src/devices/thermopro_tp11.c: 32 bits, two equal rows
    return _ppm_rows(frame, 500, 500, 1500, 3000, 3)


@protocol("kerui", model="Kerui-Security", protocol=68)
def _kerui(frame):  # src/devices/kerui.c: 25 bits, nine equal rows
    return _repeat(lambda r: code_pwm(r, 420, 960, 960), _rows(frame), 10, 7000)  # (silence between rows = this + the last bit's gap, below the reset limit)


@protocol("quhwa", model="Quhwa-Doorbell", protocol=49)
def _quhwa(frame):  # src/devices/quhwa.c: 18 bits, five equal rows
    return _repeat(lambda r: code_pwm(r, 360, 1070, 700), _rows(frame), 6, 4500)


@protocol("waveman", model="Waveman-Switch", protocol=4)
def _waveman(frame):  # src/devices/waveman.c: one row of 25 bits
    return code_pwm(_rows(frame)[0], 357, 1064, 700)


# ---- OOK_PULSE_PCM ----

@protocol("secplus_v1", model="Secplus-v1", protocol=178, freq=315000000)
def _secplus(frame):  # src/devices/secplus_v1.c: two halves of 21 ternary symbols, a package each; the decoder keeps the first
    out = []
    for k, half in enumerate(_rows(frame)):
        if k:
            out.append((60000, 0))
        out += code_levels(half, 500)
    return out


# ---- OOK_PULSE_MANCHESTER_ZEROBIT ----

@protocol("ambient_f007th", model="Ambientweather-F007TH", protocol=20)
def _ambient(frame):  # src/devices/ambient_weather.c: preamble 0x01 0x45 then 5 bytes and an LFSR digest; three copies in a row
    row = _rows(frame)[0]
    return code_mc(row + row + row, 500)


# ---- OOK_PULSE_DMC ----

@protocol("wt450", model="WT450-TH", protocol=33)
def _wt450(frame):  # src/devices/wt450.c: 36 bits behind four preamble ones
    return _repeat(lambda r: code_dmc(r, 976, 1952), _rows(frame), 2, 30000)


# ---- OOK_PULSE_PWM_OSV1 ----

@protocol("oregon_v1", model="Oregon-v1", protocol=50)
def _osv1(frame):  # src/devices/oregon_scientific_v1.c behind pulse_slicer_osv1: twelve preamble pulses, the sync, 32 Manchester bits
    h = 1465
    out = []
    for k in range(12):
        out += [(h, 1), (4200 if k == 11 else h, 0)]
    out += [(5780, 1), (5200, 0)]
    for b in _rows(frame)[0]:
        out += [(h, 1), (h, 0)] if b else [(h, 0), (h, 1)]
    return out


# ---- FSK ----

def _fsk(levels_schedule):
    """0/1 levels -> lower / upper frequency"""
    return [(us, 1 if lv else -1) for us, lv in levels_schedule]


@protocol("lacrosse_tx29", model="LaCrosse-TX29IT", protocol=76, fsk=True, freq=868300000, rate=1000000, render=dict(dev_hz=60e3, lead_us=3000.0, tail_us=3000.0, amp=100.0, sigma=1.0))
def _tx29(frame):  # src/devices/lacrosse_tx35.c: preamble aa.., sync 2d d4, 5 bytes with CRC-8, NRZ at 55 us
    return _fsk(code_levels(_rows(frame)[0], 55))


@protocol("steelmate", model="Steelmate", protocol=59, fsk=True, rate=1000000, render=dict(dev_hz=60e3, lead_us=3000.0, tail_us=3000.0, amp=100.0, sigma=1.0))
def _steelmate(frame):  # src/devices/steelmate.c: Manchester at 50 us half bits
    return [(800, -1)] + _fsk(code_mc(_rows(frame)[0], 50))  # (a stretch of plain carrier first: the FSK detector settles on it)


@protocol("efergy_e2", model="Efergy-e2CT", protocol=36, fsk=True)
def _efergy(frame):  # src/devices/efergy_e2_classic.c: FSK PWM, a 500 us sync, 64 bits
    return _fsk(_pairs([(500, 136)] + [(64 if b else 136, 136 if b else 64) for b in _rows(frame)[0]]))


@protocol("acurite_606", model="Acurite-606TX", protocol=55)
def _acurite606(frame):  # src/devices/acurite.c acurite_606_decode: 32 bits, three equal rows
    return _ppm_rows(frame, 500, 2000, 4000, 8500, 4)


@protocol("thermopro_tp12", model="Thermopro-TP12", protocol=97)
def _tp12(frame):  # src/devices/thermopro_tp12.c: 41 bits, rows repeated
    return _ppm_rows(frame, 500, 500, 1500, 3000, 4)


@protocol("gt_wt_02", model="GT-WT02", protocol=25)
def _gtwt02(frame):  # src/devices/gt_wt_02.c: 37 bits a row, at least two rows
    return _ppm_rows(frame, 500, 2500, 5000, 9500, 3)


@protocol("bresser_3ch", model="Bresser-3CH", protocol=52)
def _bresser3ch(frame):  # src/devices/bresser_3ch.c: sync pulses, then 40 bits in a 750 us period; three equal rows
    def row(r):
        return _pairs([(750, 750)] * 4 + [(250, 500) if b else (500, 250) for b in r])
    out = []
    for _ in range(4):
        out += row(_rows(frame)[0])
    return out + _pairs([(750, 0)])


@protocol("ht680", model="HT680-Remote", protocol=46)
def _ht680(frame):  # src/devices/ht680.c: 41 bits, sync 10101 in front
    return _repeat(lambda r: code_pwm(r, 200, 600, 400), _rows(frame), 3, 9000)


_fits = {}


def bench_capture(seed, n_samples=65536, rate=250000):
    """A config-2 sized capture (one burst in n_samples at `rate`) that carries a protocol-valid transmission: the protocol
    is drawn from those whose transmission fits the capture at this rate, the frame and the channel (carrier offset,
    amplitude, noise) from the seed.  -> (cu8 IQ, meta)"""
    tail = int(0.016 * rate)  # the end-of-package count has to fire inside the capture (src/pulse_detect.c:446-450)
    key = (n_samples, rate)
    if key not in _fits:
        ok = []
        for name in sorted(PROTOCOLS):
            p = PROTOCOLS[name]
            if p["rate"] != rate:
                continue
            busy = max(sum(us for us, _ in merge(p["schedule"](f))) for f in frames()[name]) * rate / 1e6
            if 4000 + busy + tail <= n_samples:
                ok.append(name)
        _fits[key] = ok
    names = _fits[key]
    rng = np.random.default_rng(7919 * (seed + 1))
    name = names[int(rng.integers(0, len(names)))]
    p = PROTOCOLS[name]
    fr = frames()[name]
    frame = fr[int(rng.integers(0, len(fr)))]
    lead = int(rng.integers(1400, 4000))
    iq = render_cu8(p["schedule"](frame), rate, rng, fsk=p["fsk"], n_samples=n_samples, **{**p.get("render", {}), "lead_us": lead * 1e6 / rate})
    return iq, dict(name=name, model=p["model"], frame=frame, rate=rate, freq=p["freq"])
