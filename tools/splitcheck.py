"""Split captures against the same captures walked by one wavefront: long FSK / OOK streams, both FSK detectors.
    python tools/splitcheck.py        (GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtl_433_amd import synth
from rtl_433_amd.engine import BatchEngine, flow_cfg

def run(iq, ss, rate, fpdm, split):
    nb = iq.nbytes
    stride = (nb + 15) // 16 * 16
    host = np.zeros((1, stride), dtype=np.uint8)
    host[0, :nb] = iq.view(np.uint8)
    d = torch.from_numpy(host).cuda()
    eng = BatchEngine(flow_cfg(ss, rate, fpdm=fpdm), None, profiling=True)
    eng.set_split(split)
    n = eng.run(d, np.array([nb], dtype=np.uint32))
    out = (n, bytes(eng.packages()[0]), eng.split_stats(), eng.timing()["detect_ms"])
    eng.close()
    return out

bad = 0
for seed in range(6):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1200000, 2400000))
    for name, ss, rate, make in (
        ("fsk_cu8", 2, 250000, lambda: synth.fsk_stream_cu8(seed, n, n_bursts=int(rng.integers(20, 60)), nbits=int(rng.integers(32, 400)),
                                                             gap=int(rng.integers(3000, 40000)), sigma=float(rng.choice([0, 1, 3])))),
        ("fsk_cs16", 4, 1024000, lambda: np.asarray(synth.fsk_stream_cs16(seed, n, n_bursts=int(rng.integers(20, 60)),
                                                                           sigma=float(rng.choice([0.0, 0.01, 0.03])))))):
        iq = make()
        for fpdm in (0, 1):
            a = run(iq, ss, rate, fpdm, 0)
            b = run(iq, ss, rate, fpdm, 1)
            c = run(iq, ss, rate, fpdm, 65536)
            ok = a[1] == b[1] == c[1]
            bad += not ok
            print(f"{name} seed {seed} fpdm {fpdm}: {a[0]} packages, one wavefront {a[3]:.1f} ms, auto split {b[3]:.2f} ms {b[2]}, 64k {c[3]:.2f} ms {c[2]} -> {'identical' if ok else 'DIFFERENT'}")
print("mismatches:", bad)
