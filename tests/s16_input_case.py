"""am.s16 / fm.s16 input files (file_info S16_AM / S16_FM): shared body of the emulator and GPU tests.

The reference reads such a file as 2-byte samples (src/rtl_433.c:1735-1739), runs its whole flow on the bytes as if they were
cu8 pairs and then copies the int16 words over am_buf / buf.fm in front of the pulse detector (src/r_flow.c:212-225).  The
checker is the UNMODIFIED reference (oracle/_ref/libr433ref.so) told that the capture is such a file."""
import numpy as np

from oracle import pyoracle as po
from rtl_433_amd import synth


def captures():
    """(format, list of int16 arrays): demodulated samples the reference itself produced (its -W am.s16 / fm.s16 dumps are the
    am_buf / buf.fm taps), plus words no filter would produce: negative AM, full-scale steps, a length that is not whole tiles."""
    ref = po.Ref(record=False)
    ook = [synth.ook_stream(900 + i, 40000 + 3000 * i)[0] for i in range(3)]
    fsk = [synth.fsk_stream_cu8(910 + i, 50001 + 777 * i) for i in range(2)]
    am = [ref.run(a, 2, 250000, 433920000, fpdm=0, taps=True)["am"].copy() for a in ook + fsk]
    fm = [ref.run(a, 2, 250000, 433920000, fpdm=0, taps=True)["fm"].copy() for a in fsk + ook[:1]]
    ref.close()
    rng = np.random.default_rng(920)
    wild = am[0].astype(np.int32) - 3000                          # an AM dump with a DC offset below zero
    wild[5000:5200] = rng.integers(-32768, 32768, 200)            # and a burst of anything
    am.append(np.clip(wild, -32768, 32767).astype(np.int16))
    am.append(rng.integers(-200, 200, 70001).astype(np.int16))    # noise around zero, odd length
    am.append(np.zeros(0, dtype=np.int16))
    fm.append(rng.integers(-32768, 32768, 33333).astype(np.int16))
    return [(1, am), (2, fm)]


def reference_records(fmt, words, enable_fm=1):
    ref = po.Ref(record=True)
    ref.set_enable_fm(enable_fm)
    ref.set_load_format(fmt)
    taps = []
    for s, w in enumerate(words):
        if w.size == 0:
            taps.append(None)
            continue
        r = ref.run(w.view(np.uint8), 2, 250000, 433920000, fpdm=0, stream_index=s, taps=True)
        taps.append((r["am"], r["fm"]))
    pk, npk = ref.packages()
    ev, nev = ref.events()
    devs = ref.devices()[0]
    ref.close()
    return devs, pk, npk, ev, nev, taps


def check(run):
    """run(list of uint8 arrays, devs, input_format, enable_fm) -> dict with packages / events / taps like emu_run / _gpu_run"""
    for fmt, words in captures():
        for enable_fm in ((1, 0) if fmt == 2 else (1,)):
            devs, pk, npk, ev, nev, taps = reference_records(fmt, words, enable_fm)
            g = run([w.view(np.uint8) for w in words], devs, 2 + fmt, enable_fm)
            assert g["packages"][1] == npk and npk > 0, (fmt, enable_fm)
            assert po.strip_ret_pos(g["packages"][0]) == pk, (fmt, enable_fm)
            assert g["events"][1] == nev
            assert po.events_normalize(g["events"][0]) == po.events_normalize(po.canonical_events(ev))
            for s, w in enumerate(words):
                if w.size:
                    assert np.array_equal(g["taps"][1][s, :w.size], taps[s][0]), ("am", fmt, s)
                    if enable_fm or fmt == 2:
                        assert np.array_equal(g["taps"][2][s, :w.size], taps[s][1]), ("fm", fmt, s)
