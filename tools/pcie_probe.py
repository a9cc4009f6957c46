"""What the PCIe link of the box gives in the setting of bench.py: pinned host -> HBM copies of the size of one bench step (1 GiB)
with a device -> host copy of 200 MiB running against it -- on torch streams (what bench.py hands the library), on raw HIP streams
wrapped as torch ExternalStreams, with the D2H on a high-priority stream, and through hipMemcpyAsync called directly.
tools/ubench/pcie_duplex.hip is the plain-HIP form of the same question (full duplex there: 18.7 ms for both at once)."""
import ctypes as C
import time

import torch

hip = C.CDLL("libamdhip64.so")
n = 1 << 30
m = 200 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(m, dtype=torch.uint8).pin_memory()
d2 = torch.empty(m, dtype=torch.uint8, device="cuda")
pool = [torch.cuda.Stream() for _ in range(6)]
hi = torch.cuda.Stream(priority=-1)


def raw_stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0  # hipStreamNonBlocking
    return torch.cuda.ExternalStream(s.value)


raw = [raw_stream() for _ in range(2)]


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def pair(s_in, s_out, d2h_first=True):
    def go():
        if d2h_first:
            with torch.cuda.stream(s_out):
                h2.copy_(d2, non_blocking=True)
        with torch.cuda.stream(s_in):
            d.copy_(h, non_blocking=True)
        if not d2h_first:
            with torch.cuda.stream(s_out):
                h2.copy_(d2, non_blocking=True)
    return go


def direct(s_in, s_out):
    def go():
        assert hip.hipMemcpyAsync(C.c_void_p(h2.data_ptr()), C.c_void_p(d2.data_ptr()), C.c_size_t(m), 2, C.c_void_p(s_out.cuda_stream)) == 0
        assert hip.hipMemcpyAsync(C.c_void_p(d.data_ptr()), C.c_void_p(h.data_ptr()), C.c_size_t(n), 1, C.c_void_p(s_in.cuda_stream)) == 0
    return go


cases = [("h2d alone (torch stream)", lambda: pair(pool[0], pool[1])() if False else pool_only()), ]


def pool_only():
    with torch.cuda.stream(pool[0]):
        d.copy_(h, non_blocking=True)


def d2h_only():
    with torch.cuda.stream(pool[2]):
        h2.copy_(d2, non_blocking=True)


print(f"h2d 1 GiB alone: {timed(pool_only) * 1e3:.2f} ms;  d2h 200 MiB alone: {timed(d2h_only) * 1e3:.2f} ms")
for k in range(1, 6):
    print(f"torch pool streams 0 (h2d) + {k} (d2h): {timed(pair(pool[0], pool[k])) * 1e3:.2f} ms")
print(f"torch pool 0 (h2d) + 2 (d2h), h2d issued first: {timed(pair(pool[0], pool[2], False)) * 1e3:.2f} ms")
print(f"torch pool 0 (h2d) + high-priority stream (d2h): {timed(pair(pool[0], hi)) * 1e3:.2f} ms")
print(f"raw HIP streams (ExternalStream) h2d + d2h: {timed(pair(raw[0], raw[1])) * 1e3:.2f} ms")
print(f"torch pool 0 (h2d) + raw HIP stream (d2h): {timed(pair(pool[0], raw[1])) * 1e3:.2f} ms")
print(f"hipMemcpyAsync called directly, torch pool 0 + 2: {timed(direct(pool[0], pool[2])) * 1e3:.2f} ms")
print(f"hipMemcpyAsync called directly, raw streams: {timed(direct(raw[0], raw[1])) * 1e3:.2f} ms")
print(f"h2d on the null stream + d2h on pool 2: {timed(pair(torch.cuda.default_stream(), pool[2])) * 1e3:.2f} ms")
