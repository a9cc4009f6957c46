"""Kernel-only timing of the detection kernel on the bench workload (HIP events inside the library).
    python tools/kbench.py [--streams N] [--samples M] [--reps R]
--debug FLAGS: R433_DEBUG_* of include/r433_hip.h (256 / 512 skip phases, 1024 per-phase shader clocks)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rtl_433_amd import synth
from rtl_433_amd.engine import BatchEngine, flow_cfg, load_device_table

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=1024)
ap.add_argument("--samples", type=int, default=65536)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--nodevs", action="store_true")
ap.add_argument("--seed0", type=int, default=0)
ap.add_argument("--rotate", type=int, default=1, help="distinct input batches taken in turn (no launch re-reads the batch before it)")
ap.add_argument("--split", type=int, default=0)
ap.add_argument("--debug", type=lambda x: int(x, 0), default=0)
ap.add_argument("--lib", default=None, help="a development variant of the library to time instead of the product")
ap.add_argument("--sigma-only", action="store_true", help="only the captures with noise (sigma > 0): no closed-form silence")
ap.add_argument("--fsk-cu8", action="store_true", help="250 kS/s cu8 FSK bursts at 433.92 MHz: the classic FSK detector")
ap.add_argument("--analyze", action="store_true", help="also time the pulse analyzer (-A) over the packages of the run")
ap.add_argument("--bench-batch", action="store_true", help="the captures of bench.py's configs[1] batch (every third one a protocol transmission): 1024 distinct, tiled")
ap.add_argument("--make-batch-only", action="store_true", help="with --bench-batch: make the cached batch and leave (before a profiler run)")
ap.add_argument("--cs16", action="store_true", help="config 3 style: 1024 kS/s cs16 FSK Manchester bursts, minmax detector")
a = ap.parse_args()
if a.cs16:
    host = np.stack([synth.fsk_stream_cs16(s, a.samples) for s in range(min(a.streams, 64))])
    host = np.tile(host, ((a.streams + len(host) - 1) // len(host), 1))[: a.streams]
    cfg = flow_cfg(4, 1024000, fpdm=1, center_frequency=868000000)
elif a.fsk_cu8:
    host = np.stack([synth.fsk_stream_cu8(s, a.samples, n_bursts=4, nbits=512, gap=3000) for s in range(min(a.streams, 64))])
    host = np.tile(host, ((a.streams + len(host) - 1) // len(host), 1))[: a.streams]
    cfg = flow_cfg(2, 250000, fpdm=0)
elif a.bench_batch:
    # (kept in /tmp between the runs of a visit: under rocprofv3 the worker processes that make it take minutes)
    cache = f"/tmp/r433_bench_batch_{a.seed0}_{min(a.streams, 1024)}.npy"
    if os.path.exists(cache):
        host = np.load(cache)
    else:
        import bench
        host = bench.ook_batches(a.seed0, min(a.streams, 1024), procs=8)
        np.save(cache + ".tmp.npy", host)
        os.replace(cache + ".tmp.npy", cache)
    if a.make_batch_only:
        sys.exit(0)
    host = np.tile(host, ((a.streams + len(host) - 1) // len(host), 1))[: a.streams]
    cfg = flow_cfg(2, 250000)
else:
    if a.sigma_only:  # the bench recipe draws sigma from {0, 1, 2}: keep the two thirds a real receiver could produce
        rows, seed = [], 0
        while len(rows) < a.streams:
            iq, meta = synth.ook_stream(seed, a.samples, 250000)
            if meta["sigma"] > 0:
                rows.append(iq)
            seed += 1
        host = np.stack(rows)
    else:
        host = synth.ook_batch(a.streams, a.samples, 250000, seed0=a.seed0)
    cfg = flow_cfg(2, 250000)
d = torch.from_numpy(host).cuda()
ds = [d] + [torch.from_numpy(np.roll(host, k + 1, axis=0).copy()).cuda() for k in range(a.rotate - 1)]
devs = None if a.nodevs else load_device_table()[0]
lib = None
if a.lib:
    import ctypes
    from rtl_433_amd import _lib
    lib = _lib.bind(ctypes.CDLL(os.path.abspath(a.lib)))
elif a.debug & 1024:  # the per-phase clocks live in the development build of the library only
    import ctypes
    from rtl_433_amd import _lib, build
    lib = _lib.bind(ctypes.CDLL(build.build(timing=True)))
eng = BatchEngine(cfg, devs, profiling=True, library=lib)
if a.split:
    eng.set_split(a.split)
if a.debug:
    eng.set_debug(a.debug)
ts = []
for r in range(a.reps):
    n = eng.run(ds[r % len(ds)])
    ts.append(eng.timing())
best = min(ts, key=lambda t: t["detect_ms"])
import zlib
print("detect_ms per rep:", [round(t["detect_ms"], 3) for t in ts], "packages crc %08x" % zlib.crc32(eng.packages()[0]), eng.split_stats())
def _pulses(blob):  # sum of num_pulses over the package records (include/r433_records.h)
    b, at, tot = bytes(blob), 0, 0
    while at + 64 <= len(b):
        tot += int.from_bytes(b[at + 12:at + 16], "little")
        at += int.from_bytes(b[at:at + 4], "little")
    return tot
print(f"flags={a.debug} streams={a.streams} samples={a.samples} pkgs={n} pulses={_pulses(eng.packages()[0])} " +
      " ".join(f"{k}={v:.3f}" for k, v in best.items()) + (f" split={eng.split_stats()}" if a.split else ""))

if a.analyze:
    import time
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for r in range(5):
        t0 = time.perf_counter()
        res = eng.analyze()
        best = min(best, time.perf_counter() - t0)
    hist = {}
    for x in res:
        hist[x.guess] = hist.get(x.guess, 0) + 1
    print(f"analyze: {len(res)} packages in {best * 1e3:.3f} ms (kernel + {len(res) * 1668 / 1e6:.1f} MB D2H + sync), guesses {dict(sorted(hist.items()))}")

if a.debug & 1024:
    import ctypes as C
    rec = 34 * 4  # sizeof(StreamState) upper bound; the real size comes back from the call
    buf = np.zeros(a.streams * 64, dtype=np.int32)
    sz = eng.L.r433_batch_debug_state(eng.h, C.c_void_p(buf.ctypes.data), buf.nbytes)
    st = buf[: a.streams * sz // 4].reshape(a.streams, sz // 4)
    names = ["A+B", "idle", "gap", "pulse", "gapstart", "general", "resolve", "iters"]
    cols = st[:, [0, 1, 2, 3, 4, 5, 6, 7]].astype(np.int64)
    cols[:, :7] *= 64
    tot = cols[:, :7].sum(axis=1)
    order = np.argsort(tot)
    print("ticks per capture (shader clock), mean / max-capture:")
    worst = order[-1]
    for j, nm in enumerate(names):
        print(f"  {nm:9s} mean={cols[:, j].mean():12.0f}  slowest-capture={cols[worst, j]:12d}")
    print(f"  total     mean={tot.mean():12.0f}  slowest={tot[worst]} (capture {worst})  fastest={tot[order[0]]}")
    # inside the train engine (StreamState slots max_pulse, lead_in, low, high, f_state, f_f1, f_f2, f_vmax)
    en = ["window loads", "pulse pro/epilogue", "averages", "candidate check", "debounce legs", "gap legs", "block end + chunk skip", "legs (count)"]
    ecols = st[:, [8, 9, 10, 11, 13, 14, 15, 16]].astype(np.int64)
    ecols[:, :7] *= 64
    for j, nm in enumerate(en):
        print(f"  engine {nm:24s} mean={ecols[:, j].mean():12.0f}  slowest-capture={ecols[worst, j]:12d}")
    metas = [synth.ook_stream(int(s))[1] for s in order[-5:]]
    for s, m in zip(order[-5:], metas):
        print("  slow capture", int(s), {k: m[k] for k in ("family", "nbits", "short", "amp", "sigma", "repeats", "used")})
