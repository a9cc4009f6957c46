"""Where a step of the software pipeline goes (GPU box): the bench's Pipeline (bench.py) over resident inputs with the
library's stage stamps (R433_TRACE_LEGS=1: enter / turn / detected / sliced / mirrored per pass and engine) and the host's
own (submit, result, replay begin / end), merged into one timeline on the monotonic clock.
    python tools/leg_timeline.py [steps] [engines] [exclusive level] [threads] [real: 0 | 1]
real = 1: the reference's real decoders (dropin/_build/libr433plugins.so) behind the ordered replay, pre-filter on -- the
bench's headline pipeline; 0: the checksum decode_fn behind the plain multi-threaded replay.
Prints per stage the mean duration over the steps and, for the last three steps, every stamp in order."""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["R433_TRACE_LEGS"] = "1"
err = tempfile.TemporaryFile(mode="w+b")
keep = os.dup(2)
os.dup2(err.fileno(), 2)  # the library's stamps go to stderr: collected here, shown at the end
import numpy as np, torch
import bench
from rtl_433_amd import _lib, synth
from rtl_433_amd.engine import digest_plugin_addr, flow_cfg, load_device_table, make_rdevices

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_eng = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bench.EXCLUSIVE = int(sys.argv[3]) if len(sys.argv) > 3 else 2
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 32
real = len(sys.argv) > 5 and int(sys.argv[5]) != 0
host = np.tile(synth.ook_batch(1024, 65536, 250000, seed0=0), (8, 1))
batches = [torch.from_numpy(np.roll(host, 341 * k, axis=0).copy()).cuda() for k in range(3)]
devs, protocols, names = load_device_table()
ctx = _lib.DigestCtx(0, 0)
rdev_arr, rdev_objs = make_rdevices(devs, digest_plugin_addr(), C.addressof(ctx), names, protocols)
if real:
    from rtl_433_amd import plugins
    plug = plugins.Plugins()
    pipe = bench.Pipeline(lambda: flow_cfg(2, 250000), devs, plug.devices, threads, n_eng, 0, on_host_leg=lambda k, e, n: plug.take(), ordered=True,
                          hooks=plug.hooks() if hasattr(plug, "hooks") else None)
    for e in pipe.engines:
        e.set_stateless(plug.stateless())
        e.probe_prefilter(plug.devices, helper=plug.helper_probe())
else:
    pipe = bench.Pipeline(lambda: flow_cfg(2, 250000), devs, rdev_arr, threads, n_eng, 0)
stamps = []  # (ms, who, what)


def now():
    return time.monotonic() * 1e3


def leg(k):
    stamps.append((now(), f"leg{k}", "thread-begin"))
    out = pipe.gpu_leg(k, src=batches[k % 3])
    stamps.append((now(), f"leg{k}", "thread-end"))
    return out


def run(n):
    from collections import deque
    futs, nxt = deque(), 0
    while nxt < min(n, n_eng - 1):
        stamps.append((now(), "main", f"submit leg{nxt}"))
        futs.append(pipe.pool.submit(leg, nxt))
        nxt += 1
    for k in range(n):
        stamps.append((now(), "main", f"wait leg{k}"))
        n_pkgs, _ = futs.popleft().result()
        stamps.append((now(), "main", f"got leg{k}"))
        if nxt < n:
            stamps.append((now(), "main", f"submit leg{nxt}"))
            futs.append(pipe.pool.submit(leg, nxt))
            nxt += 1
        stamps.append((now(), "main", f"replay{k} begin"))
        pipe.host_leg(k, n_pkgs)
        stamps.append((now(), "main", f"replay{k} end"))


for k in range(n_eng):  # prime the engines
    pipe.host_leg(k, pipe.gpu_leg(k, src=batches[k % 3])[0])
run(4)
stamps.clear()
torch.cuda.synchronize()
mark = os.lseek(err.fileno(), 0, os.SEEK_END)  # (fd 2 shares this file's offset: what the timed run writes begins here)
t0 = now()
try:
    run(steps)
    torch.cuda.synchronize()
finally:
    os.dup2(keep, 2)
t1 = now()
os.lseek(err.fileno(), mark, os.SEEK_SET)
text = b""
while True:
    chunk = os.read(err.fileno(), 1 << 20)
    if not chunk:
        break
    text += chunk
engines = {}
order = {}  # engine handle -> passes of the timed run seen so far
for line in text.decode(errors="replace").splitlines():
    if line.startswith("r433-leg "):
        _, h, what, ms = line.split()
        e = engines.setdefault(h, len(engines))
        if what == "enter":
            order[h] = order.get(h, -1) + 1
        stamps.append((float(ms), f"eng{e}", f"{what} (pass {order.get(h, 0)} of this engine)"))
stamps.sort()
print(f"{steps} steps over {n_eng} engines, turn level {bench.EXCLUSIVE}, {threads} replay threads, {'real decoders (ordered, pre-filter)' if real else 'checksum decode_fn'}: {(t1 - t0) / steps:.2f} ms per step")
# mean stage durations from the library's stamps
per = {}
last = {}
for ms, who, what in stamps:
    if who.startswith("eng"):
        key = what.split()[0]
        if key != "enter" and who in last:
            per.setdefault(f"{last[who][1]} -> {key}", []).append(ms - last[who][0])
        last[who] = (ms, key)
for k, v in per.items():
    print(f"  {k:24s} mean {np.mean(v):7.2f} ms  (min {min(v):.2f}, max {max(v):.2f}, n {len(v)})")
main = [x for x in stamps if x[1] == "main"]
waits = [b[0] - a[0] for a, b in zip(main, main[1:]) if a[2].startswith("wait") and b[2].startswith("got")]
plays = [b[0] - a[0] for a, b in zip(main, main[1:]) if a[2].endswith("begin") and b[2].endswith("end")]
print(f"  main thread: waiting for a leg mean {np.mean(waits):.2f} ms, replay mean {np.mean(plays):.2f} ms")
cut = stamps[-1][0] - 3 * (t1 - t0) / steps
print("timeline of the last three steps (ms from the start of the timed run):")
for ms, who, what in stamps:
    if ms >= cut:
        print(f"  {ms - t0:9.2f}  {who:6s} {what}")
pipe.close()
