"""Does waiting for the GPU burn CPU on this box?  A ~40 ms chain of kernels, then the wait: process CPU time against wall time,
for a spinning event, a blocking event (hipEventBlockingSync) and stream synchronize."""
import os, sys, time
import torch
print({k: v for k, v in os.environ.items() if any(t in k for t in ("HSA", "HIP", "ROC", "AMD_", "GPU"))})
x = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
y = torch.empty_like(x)


def work():
    for _ in range(12):
        torch.mm(x, x, out=y)


for label, fn in (("spinning event", lambda: (e := torch.cuda.Event(blocking=False), e.record(), e.synchronize())),
                  ("blocking event", lambda: (e := torch.cuda.Event(blocking=True), e.record(), e.synchronize())),
                  ("stream synchronize", lambda: torch.cuda.current_stream().synchronize())):
    work(); torch.cuda.synchronize()
    cpu, wall = [], []
    for _ in range(5):
        work()
        t0, w0 = os.times(), time.perf_counter()
        fn()
        t1, w1 = os.times(), time.perf_counter()
        cpu.append((t1.user - t0.user + t1.system - t0.system) * 1e3)
        wall.append((w1 - w0) * 1e3)
    print(f"{label:20s} wall {sum(wall) / 5:6.1f} ms per wait, cpu {sum(cpu) / 5:6.1f} ms per wait")
