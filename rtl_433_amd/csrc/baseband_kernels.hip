// baseband_kernels.hip -- the stateless baseband maps as stand-alone, HBM-bound kernels.
//
// envelope_detect / magnitude_est_cu8 / magnitude_est_cs16 (reference src/baseband.c:36-45,
// 65-79, 96-110): one 16-byte load per lane (8 cu8 or 4 cs16 samples), one 16/8-byte store,
// per-wave shuffle reduction and a single atomicAdd per block for the frame sum (the reference's
// uint32 accumulator wraps; addition mod 2^32 is associative, so the parallel sum is exact).
#include <algorithm>

#include "dsp_device.hpp"
#include "r433_hip.h"
#include "r433_internal.hpp"

namespace r433 {

namespace {

template <int KIND> __device__ __forceinline__ uint32_t env_one(uint32_t pr)
{
    if (KIND == ENV_AMP_CU8)
        return env_amp_cu8(pr & 0xffu, (pr >> 8) & 0xffu);
    if (KIND == ENV_MAG_CU8)
        return env_mag_cu8(pr & 0xffu, (pr >> 8) & 0xffu);
    if (KIND == ENV_TRUE_CU8) { // magnitude_true_cu8, src/baseband.c:82-93: (uint16_t)(sqrtf(x*x + y*y) * 128.0f), IEEE sqrt
        int const x = (int)(pr & 0xffu) - 128, y = (int)((pr >> 8) & 0xffu) - 128;
        return (uint32_t)(uint16_t)(int)__fmul_rn(sqrtf((float)(x * x + y * y)), 128.0f);
    }
    if (KIND == ENV_TRUE_CS16) { // magnitude_true_cs16, :113-124: (int)sqrtf((float)(x*x + y*y)) >> 1 with int32 x*x + y*y
        int const x = (int)(int16_t)(pr & 0xffffu), y = (int)(int16_t)(pr >> 16);
        int const ss = (int)((uint32_t)(x * x) + (uint32_t)(y * y)); // wraps like the C expression at (-32768, -32768)
        float const r = sqrtf((float)ss); // IEEE: the build sets -fhip-fp32-correctly-rounded-divide-sqrt (__fsqrt_rn is the native approximation here)
        int const ri = (r != r || r >= 2147483648.0f || r < -2147483648.0f) ? (int)0x80000000 : (int)r; // x86 cvttss2si
        return (uint32_t)(uint16_t)(ri >> 1);
    }
    return env_mag_cs16((int)(int16_t)(pr & 0xffffu), (int)(int16_t)(pr >> 16));
}

template <int KIND> __global__ __launch_bounds__(256) void k_envelope(uint8_t const *iq, uint16_t *env, uint32_t n,
        uint32_t *sum)
{
    constexpr int SS = (KIND == ENV_MAG_CS16 || KIND == ENV_TRUE_CS16) ? 4 : 2;
    constexpr int SPV = 16 / SS; // samples per 16-byte vector
    uint32_t const n_vec = n / SPV;
    uint32_t acc = 0;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n_vec; v += gridDim.x * blockDim.x) {
        uint4 w = ((uint4 const *)iq)[v];
        if (SS == 2) {
            uint32_t e0 = env_one<KIND>(w.x & 0xffffu), e1 = env_one<KIND>(w.x >> 16);
            uint32_t e2 = env_one<KIND>(w.y & 0xffffu), e3 = env_one<KIND>(w.y >> 16);
            uint32_t e4 = env_one<KIND>(w.z & 0xffffu), e5 = env_one<KIND>(w.z >> 16);
            uint32_t e6 = env_one<KIND>(w.w & 0xffffu), e7 = env_one<KIND>(w.w >> 16);
            acc += e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
            ((uint4 *)env)[v] = make_uint4(e0 | (e1 << 16), e2 | (e3 << 16), e4 | (e5 << 16), e6 | (e7 << 16));
        }
        else {
            uint32_t e0 = env_one<KIND>(w.x), e1 = env_one<KIND>(w.y), e2 = env_one<KIND>(w.z), e3 = env_one<KIND>(w.w);
            acc += e0 + e1 + e2 + e3;
            ((uint2 *)env)[v] = make_uint2(e0 | (e1 << 16), e2 | (e3 << 16));
        }
    }
    // ragged tail (< one vector), handled by the first lanes of block 0
    if (blockIdx.x == 0) {
        uint32_t i = n_vec * SPV + threadIdx.x;
        if (i < n) {
            uint32_t pr = SS == 2 ? ((uint16_t const *)iq)[i] : ((uint32_t const *)iq)[i];
            uint32_t e = env_one<KIND>(pr);
            env[i] = (uint16_t)e;
            acc += e;
        }
    }
    for (int o = 32; o > 0; o >>= 1)
        acc += (uint32_t)__shfl_down((int)acc, o, 64);
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && sum)
        atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}

// Per (capture, frame) envelope sums only: what -Y autolevel needs before the detector may run
// (reference src/r_flow.c:150-189: avg_db of the frame decides the detection level of that same frame).
// One block per (capture, frame); HBM-bound, the IQ stream is read once and nothing is written back.
template <int KIND> __global__ __launch_bounds__(256) void k_frame_sums(uint8_t const *iq, uint64_t stride_bytes,
        uint32_t const *stream_bytes, uint32_t uniform_bytes, uint32_t frame_samples, uint32_t frames_cap, uint32_t *sums)
{
    constexpr int SS = KIND == ENV_MAG_CS16 ? 4 : 2;
    constexpr int SPV = 16 / SS;
    uint32_t const s = blockIdx.x / frames_cap, f = blockIdx.x % frames_cap;
    uint32_t const my_n = (stream_bytes ? stream_bytes[s] : uniform_bytes) / SS;
    uint64_t const start = (uint64_t)f * frame_samples;
    uint32_t acc = 0;
    if (start < my_n) {
        uint32_t const cnt = (uint32_t)min((uint64_t)frame_samples, (uint64_t)my_n - start);
        uint8_t const *base = iq + (uint64_t)s * stride_bytes + start * SS; // 16-byte aligned: frames are multiples of 64 samples
        uint32_t const n_vec = cnt / SPV;
        for (uint32_t v = threadIdx.x; v < n_vec; v += 256) {
            uint4 w = ((uint4 const *)base)[v];
            if (SS == 2)
                acc += env_one<KIND>(w.x & 0xffffu) + env_one<KIND>(w.x >> 16) + env_one<KIND>(w.y & 0xffffu) + env_one<KIND>(w.y >> 16)
                        + env_one<KIND>(w.z & 0xffffu) + env_one<KIND>(w.z >> 16) + env_one<KIND>(w.w & 0xffffu) + env_one<KIND>(w.w >> 16);
            else
                acc += env_one<KIND>(w.x) + env_one<KIND>(w.y) + env_one<KIND>(w.z) + env_one<KIND>(w.w);
        }
        uint32_t const i = n_vec * SPV + threadIdx.x;
        if (i < cnt)
            acc += env_one<KIND>(SS == 2 ? ((uint16_t const *)base)[i] : ((uint32_t const *)base)[i]);
    }
    for (int o = 32; o > 0; o >>= 1)
        acc += (uint32_t)__shfl_down((int)acc, o, 64);
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// The reference converts two file formats while loading (src/rtl_433.c:1811-1834); here they are
// HBM-bound maps in front of the detection kernel.  Row-wise so that padded capture strides work.
__global__ __launch_bounds__(256) void k_cs8_to_cu8(uint8_t const *in, uint64_t in_stride, uint8_t *out, uint64_t out_stride,
        uint64_t row_bytes, uint32_t bx)
{
    uint32_t const row = blockIdx.x / bx, part = blockIdx.x % bx; // bx blocks per capture
    uint8_t const *src = in + (uint64_t)row * in_stride;
    uint8_t *dst = out + (uint64_t)row * out_stride;
    uint64_t const n_vec = row_bytes / 16;
    for (uint64_t v = (uint64_t)part * 256 + threadIdx.x; v < n_vec; v += (uint64_t)bx * 256) {
        uint4 w = ((uint4 const *)src)[v];
        ((uint4 *)dst)[v] = make_uint4(w.x ^ 0x80808080u, w.y ^ 0x80808080u, w.z ^ 0x80808080u, w.w ^ 0x80808080u); // int8 + 128
    }
    if (part == 0)
        for (uint64_t i = n_vec * 16 + threadIdx.x; i < row_bytes; i += 256)
            dst[i] = (uint8_t)(src[i] ^ 0x80u);
}

__device__ __forceinline__ int cf32_to_s16(float f)
{
    float const p = __fmul_rn(f, 32767.0f);
    // C converts out-of-range and NaN products to INT_MIN on x86 (cvttss2si); everything then clamps to -32767
    int s = (p >= 2147483648.0f || p < -2147483648.0f || p != p) ? INT32_MIN : (int)p;
    return s < -32767 ? -32767 : (s > 32767 ? 32767 : s);
}

__global__ __launch_bounds__(256) void k_cf32_to_cs16(float const *in, uint64_t in_stride_bytes, int16_t *out, uint64_t out_stride_bytes,
        uint64_t row_components, uint32_t bx)
{
    uint32_t const row = blockIdx.x / bx, part = blockIdx.x % bx;
    float const *src = (float const *)((uint8_t const *)in + (uint64_t)row * in_stride_bytes);
    int16_t *dst = (int16_t *)((uint8_t *)out + (uint64_t)row * out_stride_bytes);
    uint64_t const n_vec = row_components / 4;
    for (uint64_t v = (uint64_t)part * 256 + threadIdx.x; v < n_vec; v += (uint64_t)bx * 256) {
        uint4 const raw = ((uint4 const *)src)[v];
        float fx, fy, fz, fw;
        __builtin_memcpy(&fx, &raw.x, 4), __builtin_memcpy(&fy, &raw.y, 4), __builtin_memcpy(&fz, &raw.z, 4), __builtin_memcpy(&fw, &raw.w, 4);
        uint32_t const a = (uint32_t)cf32_to_s16(fx) & 0xffffu, b = (uint32_t)cf32_to_s16(fy) & 0xffffu;
        uint32_t const c = (uint32_t)cf32_to_s16(fz) & 0xffffu, d = (uint32_t)cf32_to_s16(fw) & 0xffffu;
        ((uint2 *)dst)[v] = make_uint2(a | (b << 16), c | (d << 16));
    }
    if (part == 0)
        for (uint64_t i = n_vec * 4 + threadIdx.x; i < row_components; i += 256)
            dst[i] = (int16_t)cf32_to_s16(src[i]);
}

// Mean raw envelope of every 2048-sample tile (as a sum), ESTIMATED from an eighth of it: where a long capture may be cut
// into independently processed segments -- tiles that carry no more energy than the noise floor -- and how heavy a
// segment is.  A heuristic only (every cut is verified after the fact), so it must not cost a second pass over the
// stream: four 128-byte lines per tile (one full-width request each), evenly spread from its first line to its last, eight lanes to a
// line, scaled back to the whole tile.
template <int KIND> __global__ __launch_bounds__(64) void k_tile_max(uint8_t const *iq, uint64_t stride_bytes,
        uint32_t const *stream_bytes, uint32_t uniform_bytes, uint32_t tiles_cap, uint32_t n_items, uint32_t *tile_sum)
{
    constexpr int SS = KIND == ENV_MAG_CS16 ? 4 : 2;
    constexpr uint32_t kLines = 2048u * SS / 128u; // 128-byte lines of a tile
    constexpr uint32_t kTaken = 4;
    // two tiles per workgroup: lanes 0..31 the first, 32..63 the second; eight lanes to a line (one 128-byte request)
    uint32_t const item = min(blockIdx.x * 2u + (threadIdx.x >> 5), n_items - 1u); // (an odd count: the last half workgroup repeats the last tile)
    uint32_t const s = item / tiles_cap, t = item % tiles_cap;
    uint32_t const lane = threadIdx.x & 31u;
    uint32_t const my_n = (stream_bytes ? stream_bytes[s] : uniform_bytes) / SS;
    uint64_t const start = (uint64_t)t * 2048u;
    uint32_t acc = 0;
    bool const whole = start + 2048u <= my_n; // whole tiles only: nobody cuts next to the ragged end of a capture
    if (whole) {
        uint8_t const *base = iq + (uint64_t)s * stride_bytes + start * SS;
        // the first and the last line of the tile are among the four: whatever crosses into a tile shows in it (a burst that
        // starts behind the last line taken would otherwise leave the tile looking quiet, and a cut behind such a tile cannot
        // be started from: the new piece establishes its carries and its floor on that tile)
        uint4 const w = *(uint4 const *)(base + (uint64_t)((lane >> 3) * (kLines - 1u) / (kTaken - 1u)) * 128u + (lane & 7u) * 16u);
        if (SS == 2)
            acc = env_one<KIND>(w.x & 0xffffu) + env_one<KIND>(w.x >> 16) + env_one<KIND>(w.y & 0xffffu) + env_one<KIND>(w.y >> 16)
                    + env_one<KIND>(w.z & 0xffffu) + env_one<KIND>(w.z >> 16) + env_one<KIND>(w.w & 0xffffu) + env_one<KIND>(w.w >> 16);
        else
            acc = env_one<KIND>(w.x) + env_one<KIND>(w.y) + env_one<KIND>(w.z) + env_one<KIND>(w.w);
        acc *= kLines / kTaken;
    }
    for (int o = 16; o > 0; o >>= 1)
        acc += (uint32_t)__shfl_xor((int)acc, o, 64);
    if (lane == 0)
        tile_sum[item] = whole ? acc : 0xffffffffu; // a ragged tile counts as loud
}


// ---- heaviest captures first (grids of several rounds of workgroups) ----
// A grid of more captures than the chip holds at once is handed out in index order; what is still running when the list
// runs dry is the tail.  Handing out the heavy captures first shortens it (8192 bench captures: 5.66 ms as they come, 5.38 ms
// heaviest first, 5.87 ms lightest first; tools/order_probe.py).  The weight of a capture is guessed from a few looks at it
// (a 16-byte look every 4096 samples): how many of them are well above the quietest one.  A guess only: whatever order comes out,
// every capture is walked in full and lands in its own slot.
template <int KIND> __global__ __launch_bounds__(64) void k_capture_weight(uint8_t const *iq, uint64_t stride_bytes,
        uint32_t const *stream_bytes, uint32_t uniform_bytes, uint32_t *weight)
{
    constexpr int SS = KIND == ENV_MAG_CS16 ? 4 : 2;
    uint32_t const s = blockIdx.x, lane = threadIdx.x;
    uint32_t const my_n = (stream_bytes ? stream_bytes[s] : uniform_bytes) / SS;
    constexpr uint32_t kApart = 4096; // samples between two looks (a look pulls a whole 128-byte line: 3 % of a cu8 capture this way)
    uint32_t const looks = my_n / kApart;
    uint8_t const *base = iq + (uint64_t)s * stride_bytes;
    uint32_t lo = 0xffffffffu;
    for (uint32_t k = lane; k < looks; k += 64) {
        uint4 const w = *(uint4 const *)(base + (uint64_t)k * kApart * SS);
        uint32_t e = SS == 2 ? env_one<KIND>(w.x & 0xffffu) + env_one<KIND>(w.x >> 16) + env_one<KIND>(w.y & 0xffffu) + env_one<KIND>(w.y >> 16)
                             : env_one<KIND>(w.x) + env_one<KIND>(w.y);
        lo = min(lo, e);
    }
    for (int o = 32; o > 0; o >>= 1)
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
    uint32_t heavy = 0;
    for (uint32_t k = lane; k < looks; k += 64) { // (the same lines again: they sit in the cache)
        uint4 const w = *(uint4 const *)(base + (uint64_t)k * kApart * SS);
        uint32_t e = SS == 2 ? env_one<KIND>(w.x & 0xffffu) + env_one<KIND>(w.x >> 16) + env_one<KIND>(w.y & 0xffffu) + env_one<KIND>(w.y >> 16)
                             : env_one<KIND>(w.x) + env_one<KIND>(w.y);
        heavy += e > 4u * lo + 64u ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1)
        heavy += (uint32_t)__shfl_xor((int)heavy, o, 64);
    if (lane == 0)
        weight[s] = min(heavy * 4u + (looks >> 2), 255u); // (a long quiet capture still outweighs a short one)
}

// counting sort of the captures by weight, heaviest first: one workgroup (the list is a few thousand entries)
__global__ __launch_bounds__(256) void k_order_by_weight(uint32_t const *weight, uint32_t n, uint32_t *order)
{
    __shared__ uint32_t count[256], first[256];
    count[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        atomicAdd(&count[weight[i] & 255u], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 255; w >= 0; --w) {
            first[w] = run;
            run += count[w];
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        order[atomicAdd(&first[weight[i] & 255u], 1u)] = i;
}

// ---- -w dump formats (reference src/r_flow.c:385-489): what the reference writes next to its input, as
// HBM-bound maps.  A group is what one lane turns out per step: 8 values of one or two bytes, or 4 floats (one
// 16-byte store per lane, so every store instruction of a wavefront covers one contiguous kilobyte), made
// from as many components of the IQ stream (or of an am/fm s16 stream) -- twice as many for the I (Q) halves.
// IN16: the input components are int16 (cs16 IQ, am.s16, fm.s16), else uint8.
template <int NW> __device__ __forceinline__ void load_words(void const *in, uint64_t idx, uint32_t (&w)[NW])
{
    if (NW == 1) {
        w[0] = ((uint32_t const *)in)[idx];
    }
    else if (NW == 2) {
        uint2 const t = ((uint2 const *)in)[idx];
        w[0] = t.x;
        w[NW > 1 ? 1 : 0] = t.y;
    }
    else {
        uint4 const t = ((uint4 const *)in)[idx];
        w[0] = t.x;
        w[NW > 1 ? 1 : 0] = t.y;
        w[NW > 2 ? 2 : 0] = t.z;
        w[NW > 3 ? 3 : 0] = t.w;
    }
}

template <int FMT> struct DumpGeom {
    static constexpr bool kFloat = FMT == R433_DUMP_CF32_IQ || FMT == R433_DUMP_F32_AM || FMT == R433_DUMP_F32_FM
            || FMT == R433_DUMP_F32_I || FMT == R433_DUMP_F32_Q;
    static constexpr bool kHalf = FMT == R433_DUMP_F32_I || FMT == R433_DUMP_F32_Q; // one float per IQ pair
    static constexpr int kOut = kFloat ? 4 : 8;                                     // values per group
    static constexpr int kIn = kHalf ? 2 * kOut : kOut;                             // components consumed per group
};

template <int FMT, bool IN16> __device__ __forceinline__ void dump_group(void const *in, void *out, uint64_t g, uint32_t cnt)
{
    using G = DumpGeom<FMT>;
    constexpr int kIn = G::kIn, kOut = G::kOut;
    int v[kIn];
    if (cnt == (uint32_t)kOut) { // whole group: one vector load
        constexpr int NW = kIn * (IN16 ? 2 : 1) / 4;
        uint32_t w[NW];
        load_words<NW>(in, g, w);
#pragma unroll
        for (int k = 0; k < kIn; ++k)
            v[k] = IN16 ? (int)(int16_t)((w[k >> 1] >> (16 * (k & 1))) & 0xffffu) : (int)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
    else { // the ragged tail
        for (int k = 0; k < kIn; ++k) {
            uint64_t const idx = g * kIn + (uint64_t)k;
            bool const ok = (uint32_t)(G::kHalf ? k / 2 : k) < cnt;
            v[k] = !ok ? 0 : IN16 ? (int)((int16_t const *)in)[idx] : (int)((uint8_t const *)in)[idx];
        }
    }
    if (FMT == R433_DUMP_CU8_IQ || FMT == R433_DUMP_CS8_IQ) { // one byte per component
        uint32_t o[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int b;
            if (FMT == R433_DUMP_CU8_IQ)
                b = v[k % kIn] / 256 + 128;                      // cs16 -> cu8, r_flow.c:397-400 (C division)
            else
                b = IN16 ? v[k % kIn] >> 8 : v[k % kIn] - 128;   // -> cs8, r_flow.c:412-419
            o[k >> 2] |= ((uint32_t)b & 0xffu) << (8 * (k & 3));
        }
        if (cnt == 8)
            ((uint2 *)out)[g] = make_uint2(o[0], o[1]);
        else
            for (uint32_t k = 0; k < cnt; ++k)
                ((uint8_t *)out)[g * 8 + k] = (uint8_t)(o[k >> 2] >> (8 * (k & 3)));
    }
    else if (FMT == R433_DUMP_CS16_IQ) { // cu8 -> cs16, r_flow.c:404-408
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = ((uint32_t)(v[(2 * k) % kIn] * 256 - 32768) & 0xffffu) | ((uint32_t)(v[(2 * k + 1) % kIn] * 256 - 32768) << 16);
        if (cnt == 8)
            ((uint4 *)out)[g] = make_uint4(o[0], o[1], o[2], o[3]);
        else
            for (uint32_t k = 0; k < cnt; ++k)
                ((int16_t *)out)[g * 8 + k] = (int16_t)(v[k % kIn] * 256 - 32768);
    }
    else { // float outputs, 4 per group
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (FMT == R433_DUMP_CF32_IQ)
                f[k] = IN16 ? (float)v[k % kIn] / 32768.0f : (float)(v[k % kIn] - 128) / 128.0f; // r_flow.c:424-431
            else if (FMT == R433_DUMP_F32_AM || FMT == R433_DUMP_F32_FM)
                f[k] = (float)v[k % kIn] * (1.0f / 0x8000);                                       // r_flow.c:444-455
            else { // F32_I / F32_Q, r_flow.c:456-479
                int const c = v[(2 * k + (FMT == R433_DUMP_F32_Q ? 1 : 0)) % kIn];
                f[k] = IN16 ? (float)c * (1.0f / 0x8000) : (float)(c - 128) * (1.0f / 0x80);
            }
        }
        if (cnt == 4)
            ((float4 *)out)[g] = make_float4(f[0], f[1], f[2], f[3]);
        else
            for (uint32_t k = 0; k < cnt; ++k)
                ((float *)out)[g * 4 + k] = f[k];
    }
}

template <int FMT, bool IN16> __global__ __launch_bounds__(256) void k_dump(void const *in, void *out, uint64_t n_out)
{
    constexpr uint64_t kOut = DumpGeom<FMT>::kOut;
    uint64_t const groups = (n_out + kOut - 1) / kOut;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * 256) {
        uint64_t const left = n_out - g * kOut;
        dump_group<FMT, IN16>(in, out, g, left >= kOut ? (uint32_t)kOut : (uint32_t)left);
    }
}

template <int FMT> void launch_dump_fmt(bool in16, void const *d_in, void *d_out, uint64_t n_out, hipStream_t st)
{
    uint64_t const groups = (n_out + DumpGeom<FMT>::kOut - 1) / DumpGeom<FMT>::kOut;
    uint32_t const blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(8192, (groups + 255) / 256)); // 256 CUs x 32 + grid stride
    if (in16)
        hipLaunchKernelGGL((k_dump<FMT, true>), dim3(blocks), dim3(256), 0, st, d_in, d_out, n_out);
    else
        hipLaunchKernelGGL((k_dump<FMT, false>), dim3(blocks), dim3(256), 0, st, d_in, d_out, n_out);
}

} // namespace

void launch_convert(int input_format, void const *d_in, uint64_t in_stride_bytes, void *d_out, uint64_t out_stride_bytes,
        uint64_t row_in_bytes, uint32_t n_rows, hipStream_t st)
{
    if (n_rows == 0 || row_in_bytes == 0)
        return;
    uint32_t bx = (uint32_t)std::min<uint64_t>(1024, (row_in_bytes / 16 + 255) / 256 + 1);
    dim3 grid(bx * n_rows), block(256);
    if (input_format == 1)
        hipLaunchKernelGGL(k_cs8_to_cu8, grid, block, 0, st, (uint8_t const *)d_in, in_stride_bytes, (uint8_t *)d_out, out_stride_bytes,
                row_in_bytes, bx);
    else
        hipLaunchKernelGGL(k_cf32_to_cs16, grid, block, 0, st, (float const *)d_in, in_stride_bytes, (int16_t *)d_out, out_stride_bytes,
                row_in_bytes / 4, bx);
}

void launch_tile_max(int kind, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes, uint32_t uniform_bytes,
        uint32_t n_streams, uint32_t tiles_cap, uint32_t *tile_max, hipStream_t st)
{
    dim3 grid((n_streams * tiles_cap + 1) / 2), block(64); // two tiles per workgroup (k_tile_max); the buffer holds an even count
    uint8_t const *iq = (uint8_t const *)d_iq;
    if (kind == ENV_AMP_CU8)
        hipLaunchKernelGGL(k_tile_max<ENV_AMP_CU8>, grid, block, 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, tiles_cap, n_streams * tiles_cap, tile_max);
    else if (kind == ENV_MAG_CU8)
        hipLaunchKernelGGL(k_tile_max<ENV_MAG_CU8>, grid, block, 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, tiles_cap, n_streams * tiles_cap, tile_max);
    else
        hipLaunchKernelGGL(k_tile_max<ENV_MAG_CS16>, grid, block, 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, tiles_cap, n_streams * tiles_cap, tile_max);
}

void launch_capture_order(int kind, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes, uint32_t uniform_bytes,
        uint32_t n_streams, uint32_t *weight, uint32_t *order, hipStream_t st)
{
    uint8_t const *iq = (uint8_t const *)d_iq;
    if (kind == ENV_AMP_CU8)
        hipLaunchKernelGGL(k_capture_weight<ENV_AMP_CU8>, dim3(n_streams), dim3(64), 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, weight);
    else if (kind == ENV_MAG_CU8)
        hipLaunchKernelGGL(k_capture_weight<ENV_MAG_CU8>, dim3(n_streams), dim3(64), 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, weight);
    else
        hipLaunchKernelGGL(k_capture_weight<ENV_MAG_CS16>, dim3(n_streams), dim3(64), 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, weight);
    hipLaunchKernelGGL(k_order_by_weight, dim3(1), dim3(256), 0, st, weight, n_streams, order);
}

void launch_frame_sums(int kind, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes, uint32_t uniform_bytes,
        uint32_t n_streams, uint32_t frame_samples, uint32_t frames_cap, uint32_t *sums, hipStream_t st)
{
    dim3 grid(n_streams * frames_cap), block(256);
    uint8_t const *iq = (uint8_t const *)d_iq;
    if (kind == ENV_AMP_CU8)
        hipLaunchKernelGGL(k_frame_sums<ENV_AMP_CU8>, grid, block, 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, frame_samples, frames_cap, sums);
    else if (kind == ENV_MAG_CU8)
        hipLaunchKernelGGL(k_frame_sums<ENV_MAG_CU8>, grid, block, 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, frame_samples, frames_cap, sums);
    else
        hipLaunchKernelGGL(k_frame_sums<ENV_MAG_CS16>, grid, block, 0, st, iq, stride_bytes, stream_bytes, uniform_bytes, frame_samples, frames_cap, sums);
}

void launch_envelope(int kind, void const *d_iq, uint16_t *d_env, uint32_t n, uint32_t *d_sum, hipStream_t st)
{
    uint32_t spv = (kind == ENV_MAG_CS16 || kind == ENV_TRUE_CS16) ? 4 : 8;
    uint32_t vecs = n / spv;
    uint32_t blocks = (vecs + 255) / 256;
    if (blocks < 1)
        blocks = 1;
    if (blocks > 4096) // 256 CUs x 8 blocks + grid stride, cdna guide G11
        blocks = 4096;
    uint8_t const *iq = (uint8_t const *)d_iq;
    if (kind == ENV_AMP_CU8)
        hipLaunchKernelGGL(k_envelope<ENV_AMP_CU8>, dim3(blocks), dim3(256), 0, st, iq, d_env, n, d_sum);
    else if (kind == ENV_MAG_CU8)
        hipLaunchKernelGGL(k_envelope<ENV_MAG_CU8>, dim3(blocks), dim3(256), 0, st, iq, d_env, n, d_sum);
    else if (kind == ENV_TRUE_CU8)
        hipLaunchKernelGGL(k_envelope<ENV_TRUE_CU8>, dim3(blocks), dim3(256), 0, st, iq, d_env, n, d_sum);
    else if (kind == ENV_TRUE_CS16)
        hipLaunchKernelGGL(k_envelope<ENV_TRUE_CS16>, dim3(blocks), dim3(256), 0, st, iq, d_env, n, d_sum);
    else
        hipLaunchKernelGGL(k_envelope<ENV_MAG_CS16>, dim3(blocks), dim3(256), 0, st, iq, d_env, n, d_sum);
}

int launch_dump(int format, uint32_t sample_size, void const *d_in, void *d_out, uint64_t n_out, hipStream_t st)
{
    bool const in16 = sample_size == 4 || format == R433_DUMP_F32_AM || format == R433_DUMP_F32_FM;
    switch (format) {
    case R433_DUMP_CU8_IQ:
        launch_dump_fmt<R433_DUMP_CU8_IQ>(true, d_in, d_out, n_out, st);
        return 0;
    case R433_DUMP_CS16_IQ:
        launch_dump_fmt<R433_DUMP_CS16_IQ>(false, d_in, d_out, n_out, st);
        return 0;
    case R433_DUMP_CS8_IQ:
        launch_dump_fmt<R433_DUMP_CS8_IQ>(in16, d_in, d_out, n_out, st);
        return 0;
    case R433_DUMP_CF32_IQ:
        launch_dump_fmt<R433_DUMP_CF32_IQ>(in16, d_in, d_out, n_out, st);
        return 0;
    case R433_DUMP_F32_AM:
        launch_dump_fmt<R433_DUMP_F32_AM>(true, d_in, d_out, n_out, st);
        return 0;
    case R433_DUMP_F32_FM:
        launch_dump_fmt<R433_DUMP_F32_FM>(true, d_in, d_out, n_out, st);
        return 0;
    case R433_DUMP_F32_I:
        launch_dump_fmt<R433_DUMP_F32_I>(in16, d_in, d_out, n_out, st);
        return 0;
    case R433_DUMP_F32_Q:
        launch_dump_fmt<R433_DUMP_F32_Q>(in16, d_in, d_out, n_out, st);
        return 0;
    default:
        return -1;
    }
}

} // namespace r433
