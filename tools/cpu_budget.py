"""Where the CPU quota of the GPU box goes (16 CPUs of 256, cpu.max): process CPU time per step of the headline pipeline,
of its GPU legs alone (does waiting for the GPU burn CPU?), and of the replay alone.
    python tools/cpu_budget.py [steps] [threads]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from rtl_433_amd import plugins, synth
from rtl_433_amd.engine import flow_cfg, load_device_table

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
threads = int(sys.argv[2]) if len(sys.argv) > 2 else bench.replay_threads(1)
host = np.tile(synth.ook_batch(1024, 65536, 250000, seed0=0), (8, 1))
batches = [torch.from_numpy(np.roll(host, 341 * k, axis=0).copy()).cuda() for k in range(3)]
devs, protocols, names = load_device_table()
plug = plugins.Plugins()
pipe = bench.Pipeline(lambda: flow_cfg(2, 250000), devs, plug.devices, threads, 3, 0, on_host_leg=lambda k, e, n: plug.take(), ordered=True, hooks=plug.hooks())
for e in pipe.engines:
    e.set_stateless(plug.stateless())
    e.probe_prefilter(plug.devices, helper=plug.helper_probe())


def cgroup():
    d = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            d[k] = int(v)
    except Exception:
        pass
    return d


def measure(label, fn, n):
    torch.cuda.synchronize()
    c0, t0, w0 = cgroup(), os.times(), time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    c1, t1, w1 = cgroup(), os.times(), time.perf_counter()
    wall = (w1 - w0) / n * 1e3
    user, sys_ = (t1.user - t0.user) / n * 1e3, (t1.system - t0.system) / n * 1e3
    print(f"{label:46s} wall {wall:6.2f} ms/step, cpu user {user:6.1f} + sys {sys_:6.1f} = {user + sys_:6.1f} ms/step "
          f"({(user + sys_) / wall:4.1f} CPUs), throttled periods +{c1.get('nr_throttled', 0) - c0.get('nr_throttled', 0)}, "
          f"throttled {((c1.get('throttled_usec', 0) - c0.get('throttled_usec', 0)) / 1e3 / n):.2f} ms/step", flush=True)


la = lambda k: dict(src=batches[k % 3])
for k in range(3):
    pipe.host_leg(k, pipe.gpu_leg(k, **la(k))[0])
pipe.run(4, la)

measure("GPU legs alone, one after the other", lambda n: [pipe.gpu_leg(k, **la(k)) for k in range(n)], steps)
real_host_leg = pipe.host_leg
pipe.host_leg = lambda k, n_pkgs: time.sleep(0.015)
measure("GPU legs in the pipeline, host leg = sleep 15 ms", lambda n: pipe.run(n, la), steps)
pipe.host_leg = real_host_leg
pipe.gpu_leg(0, **la(0))
measure("replay alone (engine 0's records, again and again)", lambda n: [pipe.host_leg(0, 0) for _ in range(n)], steps)
measure("the pipeline", lambda n: pipe.run(n, la), steps)
measure("the pipeline", lambda n: pipe.run(n, la), steps)
