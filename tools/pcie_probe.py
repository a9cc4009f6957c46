"""What the PCIe link of the box gives: pinned host -> HBM copies of the size of one bench step (1 GiB), alone, as two
halves on two streams, and with a device -> host copy of 200 MiB running against it.  (The headline `value` is bound by it.)"""
import time

import torch

n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(200 << 20, dtype=torch.uint8).pin_memory()
d2 = torch.empty(200 << 20, dtype=torch.uint8, device="cuda")
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def one():
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)


def halves():
    with torch.cuda.stream(s1):
        d[: n // 2].copy_(h[: n // 2], non_blocking=True)
    with torch.cuda.stream(s2):
        d[n // 2:].copy_(h[n // 2:], non_blocking=True)


def quarters():
    for k, st in enumerate((s1, s2, s1, s2)):
        with torch.cuda.stream(st):
            d[k * n // 4:(k + 1) * n // 4].copy_(h[k * n // 4:(k + 1) * n // 4], non_blocking=True)


def with_d2h():
    with torch.cuda.stream(s3):
        h2.copy_(d2, non_blocking=True)
    one()


def d2h_only():
    with torch.cuda.stream(s3):
        h2.copy_(d2, non_blocking=True)


for name, fn, nbytes in (("h2d 1 GiB one stream", one, n), ("h2d two halves on two streams", halves, n), ("h2d four quarters on two streams", quarters, n),
                         ("h2d 1 GiB + d2h 200 MiB", with_d2h, n), ("d2h 200 MiB", d2h_only, 200 << 20)):
    t = timed(fn)
    print(f"{name}: {t * 1e3:.2f} ms, {nbytes / t / 1e9:.1f} GB/s")
