// dsp_device.hpp -- per-sample baseband primitives for gfx950 lanes.
//
// Integer-only restatements of the reference's envelope / FM discriminator / first-order
// low-pass steps (reference src/baseband.c), written as register-resident step functions so
// one lane can carry a whole capture.  All arithmetic follows C on x86-64: truncating
// division, arithmetic shift of negatives, two's-complement narrowing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace r433 {

// reference src/baseband.c:36-45: (127-I)^2 + (127-Q)^2, 0..32768
__device__ __forceinline__ uint32_t env_amp_cu8(uint32_t i, uint32_t q)
{
    int di = 127 - (int)i, dq = 127 - (int)q;
    return (uint32_t)(di * di + dq * dq) & 0xffffu;
}

// reference src/baseband.c:65-79: 122*max + 51*min of |I-128|, |Q-128|
__device__ __forceinline__ uint32_t env_mag_cu8(uint32_t i, uint32_t q)
{
    int a = abs((int)i - 128), b = abs((int)q - 128);
    return (uint32_t)(122 * max(a, b) + 51 * min(a, b)) & 0xffffu;
}

// reference src/baseband.c:96-110: (122*max + 51*min) >> 8 of |I|, |Q|
__device__ __forceinline__ uint32_t env_mag_cs16(int i, int q)
{
    uint32_t a = (uint32_t)abs(i), b = (uint32_t)abs(q);
    return ((122u * max(a, b) + 51u * min(a, b)) >> 8) & 0xffffu;
}

// reference src/baseband.c:161-163: y = (13993*y1 + 1195*(x + x1)) >> 14 narrowed to int16
constexpr int kLpfA = 13993; // FIX(0.85408) >> 1
constexpr int kLpfB = 1195;  // FIX(0.07296) >> 1
__device__ __forceinline__ int lpf_step(int y1, int x, int x1)
{
    return (int)(int16_t)((kLpfA * y1 + kLpfB * (x + x1)) >> 14);
}

// n / d with C semantics for d in [1, 65536], |n| <= 8191 * d (what atan2_q15 divides): one float reciprocal
// lands within one of the quotient (cvt, rcp and mul are each good to 2^-22 relative, the quotient stays below
// 2^13), the remainder settles it.  A third of the instructions of the generic 32-bit division.
__device__ __forceinline__ int div_q15(int n, int d)
{
    int const an = abs(n);
    int qq = (int)((float)an * __builtin_amdgcn_rcpf((float)d));
    int r = an - __mul24(qq, d);
    if (r < 0) {
        qq -= 1;
        r += d;
    }
    if (r >= d)
        qq += 1;
    return n < 0 ? -qq : qq;
}

// reference src/baseband.c:181-202, pi == 32767
__device__ __forceinline__ int atan2_q15(int y, int x)
{
    int const q = 8191, q3 = 24575;
    int ay = abs(y);
    if ((x | y) == 0)
        return 0;
    int num, den, base;
    if (x >= 0) {
        den = ay + x;
        num = x - ay;
        base = q;
    }
    else {
        den = ay - x;
        num = x + ay;
        base = q3;
    }
    if (den == 0)
        den = 1;
    int ang = base - div_q15(q * num, den);
    return (int)(int16_t)(y < 0 ? -ang : ang);
}

// reference src/baseband.c:281-300; arguments already truncated to 32 bit, pi == INT32_MAX
__device__ __forceinline__ int atan2_q31(int y, int x)
{
    long long const q = 2147483647ll / 4, q3 = 3ll * 2147483647ll / 4;
    long long ay = (int)(y < 0 ? 0u - (uint32_t)y : (uint32_t)y); // 32-bit abs() as the reference calls it (wraps for INT32_MIN)
    long long num, den, base;
    if (x >= 0) {
        den = ay + x;
        num = x - ay;
        base = q;
    }
    else {
        den = ay - x;
        num = x + ay;
        base = q3;
    }
    if (den == 0)
        den = 1;
    long long ang = base - q * num / den;
    if (y < 0)
        ang = -ang;
    return (int)ang;
}

// FM discriminator + low-pass state of one capture (reference demodfm_state_t)
struct FmLane {
    int xr, xi, xf, yf;
};

// reference src/baseband.c:242-265: one cu8 sample -> filtered frequency (int16 range)
__device__ __forceinline__ int fm_step_cu8(FmLane &s, uint32_t bi, uint32_t bq, int a16, int b16)
{
    int r = (int)bi - 128, i = (int)bq - 128;
    int dot = r * s.xr + i * s.xi;
    int crs = i * s.xr - r * s.xi;
    int f = atan2_q15(crs, dot);
    int y = (int)(int16_t)((a16 * s.yf + b16 * (f + s.xf)) >> 14);
    s.xr = r;
    s.xi = i;
    s.xf = f;
    s.yf = y;
    return y;
}

// reference src/baseband.c:335-359: one cs16 sample -> filtered frequency (>>16 of the Q30 value)
__device__ __forceinline__ int fm_step_cs16(FmLane &s, int r, int i, long long a32, long long b32)
{
    long long dot = (long long)r * s.xr + (long long)i * s.xi;
    long long crs = (long long)i * s.xr - (long long)r * s.xi;
    int f = atan2_q31((int)crs, (int)dot);
    int y = (int)((a32 * s.yf + b32 * ((long long)f + s.xf)) >> 30);
    s.xr = r;
    s.xi = i;
    s.xf = f;
    s.yf = y;
    return (int)(int16_t)(y >> 16);
}

} // namespace r433
