"""The wave emulator itself: a workgroup of two wavefronts that run different code between barriers (the shape of a
producer/consumer kernel) computes what plain Python computes."""
import ctypes as C

from tests.emu import build_emu


def _expected(steps):
    acc = [0] * 64
    for t in range(1, steps + 1):
        s = t - 1
        v = [lane * 3 + s for lane in range(64)]
        for k in range(s % 3 + 1):
            v = [v[lane] + v[lane ^ (1 << k)] for lane in range(64)]
        ones = sum(x & 1 for x in v)
        for lane in range(64):
            acc[lane] += v[lane] + ones + (v[7] if t & 1 else 0)
    return acc


def test_two_wavefronts_with_different_code_between_barriers():
    L = C.CDLL(build_emu.build())
    for steps in (1, 2, 7):
        out = (C.c_int * 64)()
        assert L.r433emu_selftest_two_waves(out, steps) == 0
        assert list(out) == _expected(steps), steps
