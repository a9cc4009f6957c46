"""Pure-Python restatement of the reference's pulse analyzer (`-A`, src/pulse_analyzer.c) up to and including the
flex-decoder suggestion line.  TEST INFRASTRUCTURE ONLY (small inputs: <= 1200 pulses per package).

Every function cites the reference lines it follows.  Floats are kept in the C types of the reference
(float32 for `tolerance` and the r_device timing fields, double for `to_us`/`to_ms`)."""
from __future__ import annotations

import numpy as np

MAX_HIST_BINS = 16          # src/pulse_analyzer.c:20
TOLERANCE = np.float32(0.2)  # :211
F32 = np.float32

# enum modulation_types, include/r_device.h:24-40
MOD = {"OOK_PULSE_MANCHESTER_ZEROBIT": 3, "OOK_PULSE_PCM": 4, "OOK_PULSE_PPM": 5, "OOK_PULSE_PWM": 6,
       "FSK_PULSE_PCM": 16, "FSK_PULSE_PWM": 17, "FSK_PULSE_MANCHESTER_ZEROBIT": 18}


class Hist:
    def __init__(self):
        self.bins = []  # dicts count, sum, mean, min, max


def _cdiv(a, b):  # C integer division (truncation toward zero)
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _within(bn, bm):  # abs(bn - bm) < (tolerance * MAX(bn, bm)), float32 product, :46 and :142
    return F32(abs(bn - bm)) < TOLERANCE * F32(max(bn, bm))


def histogram_sum(h, data):  # :38-66
    for v in data:
        v = int(v)
        for b in h.bins:
            if _within(v, b["mean"]):
                b["count"] += 1
                b["sum"] += v
                b["mean"] = _cdiv(b["sum"], b["count"])
                b["min"] = min(v, b["min"])
                b["max"] = max(v, b["max"])
                break
        else:
            if len(h.bins) < MAX_HIST_BINS:
                h.bins.append(dict(count=1, sum=v, mean=v, min=v, max=v))


def histogram_fuse_bins(h):  # :130-154
    if len(h.bins) < 2:
        return
    n = 0
    while n < len(h.bins) - 1:
        m = n + 1
        while m < len(h.bins):
            a, b = h.bins[n], h.bins[m]
            if _within(a["mean"], b["mean"]):
                a["count"] += b["count"]
                a["sum"] += b["sum"]
                a["mean"] = _cdiv(a["sum"], a["count"])
                a["min"] = min(a["min"], b["min"])
                a["max"] = max(a["max"], b["max"])
                del h.bins[m]
            else:
                m += 1
        n += 1


def _bubble(h, key):  # histogram_sort_mean :96-110 / histogram_sort_count :113-127 (the exact exchange order matters for ties)
    n_bins = len(h.bins)
    for n in range(n_bins - 1):
        for m in range(n + 1, n_bins):
            if h.bins[m][key] < h.bins[n][key]:
                h.bins[m], h.bins[n] = h.bins[n], h.bins[m]


def histogram_find_bin_index(h, width):  # :157-165
    for n, b in enumerate(h.bins):
        if b["min"] <= width <= b["max"]:
            return n
    return -1


def histogram_print(h, rate, out):  # :168-178
    for n, b in enumerate(h.bins):
        out.append(" [%2u] count: %4u,  width: %4.0f us [%.0f;%.0f]\t(%4i S)" % (
            n, b["count"], b["mean"] * 1e6 / rate, b["min"] * 1e6 / rate, b["max"] * 1e6 / rate, b["mean"]))


def analyze(pulse, gap, package_type, rate, levels):
    """pulse_analyzer(), :279-560.  levels: dict high, low, f1, f2, rssi, snr, noise.  Returns (lines, device) where
    device is a dict of the guessed r_device fields (modulation 0: no guess)."""
    num = len(pulse)
    out = []
    if num == 0:
        return ["No pulses detected."], dict(modulation=0)
    pulse = [int(x) for x in pulse]
    gap = [int(x) for x in gap]
    to_ms, to_us = 1e3 / rate, 1e6 / rate
    pg = [pulse[n] + gap[n] for n in range(num)]                       # :291-297
    total = sum(pg) - gap[num - 1]
    gp = [pulse[0]] + [pulse[n] + gap[n - 1] for n in range(1, num)]    # :299-304
    hp, hg, hpg, hgp, ht = Hist(), Hist(), Hist(), Hist(), Hist()
    histogram_sum(hp, pulse)                                            # :313-318
    histogram_sum(hg, gap[: num - 1])
    histogram_sum(hpg, pg[: num - 1])
    histogram_sum(hgp, gp)
    histogram_sum(ht, pulse)
    histogram_sum(ht, gap)
    for h in (hp, hg, hpg, ht):                                         # :321-324
        histogram_fuse_bins(h)
    out.append("Analyzing pulses...")
    out.append("Total count: %4u,  width: %4.2f ms\t\t(%5i S)" % (num, total * to_ms, total))
    for title, h in (("Pulse width distribution:", hp), ("Gap width distribution:", hg), ("Pulse+gap period distribution:", hpg),
                     ("Gap+pulse period distribution:", hgp), ("Timing distribution:", ht)):
        out.append(title)
        histogram_print(h, rate, out)
    out.append("Level estimates [high, low]: %6i, %6i" % (levels["high"], levels["low"]))
    out.append("RSSI: %.1f dB SNR: %.1f dB Noise: %.1f dB" % (levels["rssi"], levels["snr"], levels["noise"]))
    out.append("Frequency offsets [F1, F2]:  %6i, %6i\t(%+.1f kHz, %+.1f kHz)" % (
        levels["f1"], levels["f2"], float(F32(levels["f1"]) / F32(32767)) * (rate / 2.0 / 1000.0),
        float(F32(levels["f2"]) / F32(32767)) * (rate / 2.0 / 1000.0)))
    _bubble(hp, "mean")                                                 # :349-354
    _bubble(hg, "mean")
    if hp.bins and hp.bins[0]["mean"] == 0:
        del hp.bins[0]
    dev = dict(modulation=0, short_width=F32(0), long_width=F32(0), reset_limit=F32(0), gap_limit=F32(0),
               sync_width=F32(0), tolerance=F32(0))
    fsk = package_type == 2
    P, G = hp.bins, hg.bins
    biggest_gap = (G[-1]["max"] if G else 0) + 1
    if num == 1:                                                        # :357-359
        guess = "Single pulse detected. Probably Frequency Shift Keying or just noise..."
    elif len(P) == 1 and len(G) == 1:
        guess = "Un-modulated signal. Maybe a preamble..."
    elif len(P) == 1 and len(G) > 1:                                    # :363-370
        guess = "Pulse Position Modulation with fixed pulse width"
        dev.update(modulation=MOD["OOK_PULSE_PPM"], short_width=F32(to_us * G[0]["mean"]), long_width=F32(to_us * G[1]["mean"]),
                   gap_limit=F32(to_us * (G[1]["max"] + 1)), reset_limit=F32(to_us * biggest_gap))
    elif (len(P) == 2 and len(G) == 1) or (len(P) == 2 and len(G) == 2 and len(hpg.bins) == 1):  # :371-386
        guess = "Pulse Width Modulation with fixed gap" if len(G) == 1 else "Pulse Width Modulation with fixed period"
        s, l = F32(to_us * P[0]["mean"]), F32(to_us * P[1]["mean"])
        dev.update(modulation=MOD["FSK_PULSE_PWM" if fsk else "OOK_PULSE_PWM"], short_width=s, long_width=l,
                   tolerance=F32(float(l - s) * 0.4), reset_limit=F32(to_us * biggest_gap))
    elif len(P) == 2 and len(G) == 2 and len(hpg.bins) == 3:            # :387-393
        guess = "Manchester coding"
        dev.update(modulation=MOD["FSK_PULSE_MANCHESTER_ZEROBIT" if fsk else "OOK_PULSE_MANCHESTER_ZEROBIT"],
                   short_width=F32(to_us * min(P[0]["mean"], P[1]["mean"])), long_width=F32(0), reset_limit=F32(to_us * biggest_gap))
    elif len(P) == 2 and len(G) >= 3:                                   # :394-402
        guess = "Pulse Width Modulation with multiple packets"
        s, l = F32(to_us * P[0]["mean"]), F32(to_us * P[1]["mean"])
        dev.update(modulation=MOD["FSK_PULSE_PWM" if fsk else "OOK_PULSE_PWM"], short_width=s, long_width=l,
                   gap_limit=F32(to_us * (G[1]["max"] + 1)), tolerance=F32(float(l - s) * 0.4), reset_limit=F32(to_us * biggest_gap))
    elif (len(P) >= 3 and len(G) >= 3                                   # :403-414
            and abs(P[1]["mean"] - 2 * P[0]["mean"]) <= _cdiv(P[0]["mean"], 8) and abs(P[2]["mean"] - 3 * P[0]["mean"]) <= _cdiv(P[0]["mean"], 8)
            and abs(G[0]["mean"] - P[0]["mean"]) <= _cdiv(P[0]["mean"], 8) and abs(G[1]["mean"] - 2 * P[0]["mean"]) <= _cdiv(P[0]["mean"], 8)
            and abs(G[2]["mean"] - 3 * P[0]["mean"]) <= _cdiv(P[0]["mean"], 8)):
        guess = "Non Return to Zero coding (Pulse Code)"
        dev.update(modulation=MOD["FSK_PULSE_PCM" if fsk else "OOK_PULSE_PCM"], short_width=F32(to_us * P[0]["mean"]),
                   long_width=F32(to_us * P[0]["mean"]), reset_limit=F32(to_us * P[0]["mean"] * 1024))
    elif len(P) == 3:                                                   # :415-426
        guess = "Pulse Width Modulation with sync/delimiter"
        _bubble(hp, "count")
        p1, p2 = hp.bins[1]["mean"], hp.bins[2]["mean"]
        dev.update(modulation=MOD["FSK_PULSE_PWM" if fsk else "OOK_PULSE_PWM"], short_width=F32(to_us * min(p1, p2)),
                   long_width=F32(to_us * max(p1, p2)), sync_width=F32(to_us * hp.bins[0]["mean"]), reset_limit=F32(to_us * biggest_gap))
    else:
        guess = "No clue..."
    out.append("Guessing modulation: " + guess)
    # RfRaw line, :432-513
    if len(ht.bins) <= 8:
        def words():
            b = []
            for bn in ht.bins:
                w = max(0, bn["mean"] * to_us)
                w = int(w) if w < 65535 else 65535
                b += [w >> 8, w & 0xFF]
            return b
        if len(G) <= 2:
            hx = [0xAA, 0xB1, len(ht.bins)] + words()
            for i in range(num):
                hx.append(0x80 | (histogram_find_bin_index(ht, pulse[i]) << 4) | histogram_find_bin_index(ht, gap[i]))
            hx.append(0x55)
            out.append("view at https://triq.org/pdv/#" + "".join("%02X" % (v & 0xFF) for v in hx[:1024]))
        else:
            limit = G[min(3, len(G) - 1)]["min"]
            strs = []
            i = 0
            while i < num and len(strs) < 32:
                hx = [0xAA, 0xB0, 0, len(ht.bins), 1] + words()
                while i < num:
                    hx.append(0x80 | (histogram_find_bin_index(ht, pulse[i]) << 4) | histogram_find_bin_index(ht, gap[i]))
                    i += 1
                    if gap[i - 1] >= limit:
                        break
                hx.append(0x55)
                hx = hx[:1024]                                           # hexstr_push_byte drops what does not fit, :189-194
                hx[2] = len(hx) - 4 if len(hx) - 4 <= 255 else 0
                if strs and len(strs[-1]) == len(hx) and strs[-1][5:] == hx[5:]:
                    strs[-1][4] = (strs[-1][4] + 1) & 0xFF
                else:
                    strs.append(hx)
            out.append("view at https://triq.org/pdv/#" + "+".join("".join("%02X" % (v & 0xFF) for v in s) for s in strs))
            if len(strs) >= 32:
                out.append("Too many pulse groups (%u pulses missed in rfraw)" % (num - i))
    if dev["modulation"]:                                               # :516-556
        out.append("Attempting demodulation... short_width: %.0f, long_width: %.0f, reset_limit: %.0f, sync_width: %.0f" % (
            dev["short_width"], dev["long_width"], dev["reset_limit"], dev["sync_width"]))
        m = dev["modulation"]
        if m == MOD["FSK_PULSE_PCM"]:
            out.append("Use a flex decoder with -X 'n=name,m=FSK_PCM,s=%.0f,l=%.0f,r=%.0f'" % (dev["short_width"], dev["long_width"], dev["reset_limit"]))
        elif m == MOD["OOK_PULSE_PPM"]:
            out.append("Use a flex decoder with -X 'n=name,m=OOK_PPM,s=%.0f,l=%.0f,g=%.0f,r=%.0f'" % (
                dev["short_width"], dev["long_width"], dev["gap_limit"], dev["reset_limit"]))
        elif m in (MOD["OOK_PULSE_PWM"], MOD["FSK_PULSE_PWM"]):
            out.append("Use a flex decoder with -X 'n=name,m=%s,s=%.0f,l=%.0f,r=%.0f,g=%.0f,t=%.0f,y=%.0f'" % (
                "OOK_PWM" if m == MOD["OOK_PULSE_PWM"] else "FSK_PWM", dev["short_width"], dev["long_width"], dev["reset_limit"],
                dev["gap_limit"], dev["tolerance"], dev["sync_width"]))
        elif m == MOD["OOK_PULSE_MANCHESTER_ZEROBIT"]:
            out.append("Use a flex decoder with -X 'n=name,m=OOK_MC_ZEROBIT,s=%.0f,l=%.0f,r=%.0f'" % (
                dev["short_width"], dev["long_width"], dev["reset_limit"]))
        else:
            out.append("Unsupported")
    return out, dev
