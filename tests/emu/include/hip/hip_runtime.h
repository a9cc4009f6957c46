// hip_runtime.h (EMULATOR) -- TEST INFRASTRUCTURE, never part of the product.
//
// A minimal stand-in for <hip/hip_runtime.h> that lets the kernels in rtl_433_amd/csrc be compiled
// with plain g++ and executed on the CPU in lock step: every HIP thread of a workgroup is a fiber,
// every cross-lane primitive (__syncthreads, __shfl*, __ballot, readlane ...) is a rendezvous of all
// live fibers of the workgroup.  The point is to run the *same kernel source* against the oracle in
// the CPU test suite (tests/test_emu_parity.py) where no GPU exists.  The product library
// (librtl433hip.so) is built by hipcc from the same sources and contains none of this.
//
// Supported: 1-D grids and blocks, static __shared__ (function-local statics: one workgroup runs at
// a time), the runtime calls host_api.cpp uses, wave64 cross-lane builtins called from wave-uniform
// control flow (a call-site tag check aborts on divergent use).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define R433_EMU 1
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ static __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

using std::max;
using std::min;
static inline uint32_t min(uint32_t a, int b) { return a < (uint32_t)b ? a : (uint32_t)b; }
static inline uint32_t min(int a, uint32_t b) { return (uint32_t)a < b ? (uint32_t)a : b; }
static inline uint32_t max(uint32_t a, int b) { return a > (uint32_t)b ? a : (uint32_t)b; }
static inline uint32_t max(int a, uint32_t b) { return (uint32_t)a > b ? (uint32_t)a : b; }

namespace emu {

struct Idx { unsigned x, y, z; };
struct Fiber;
struct Block;
Fiber &cur();
Block &blk();
unsigned tid();
unsigned nthreads();
unsigned bid();
unsigned nblocks();
// Rendezvous of all live fibers of the workgroup.  Stores `bytes` at this fiber's slot, returns the
// slot array base (stride kSlot bytes) valid until this fiber's next rendezvous, and `live` = a
// per-thread flag array telling which threads took part in this rendezvous.
constexpr unsigned kSlot = 16;
uint8_t *rendezvous(void const *val, unsigned bytes, unsigned tag, uint8_t const **live);
void launch(dim3 grid, dim3 block, std::function<void()> const &body);
bool all_at_barrier(); // did every live fiber of the workgroup deposit the barrier tag in the round just finished?

template <typename T> struct Exch {
    uint8_t *base;
    uint8_t const *live;
    T at(unsigned t) const
    {
        T v;
        memcpy(&v, base + (size_t)t * kSlot, sizeof(T));
        return v;
    }
};
template <typename T> inline Exch<T> exchange(T v, unsigned tag)
{
    static_assert(sizeof(T) <= kSlot, "slot too small");
    Exch<T> e;
    e.base = rendezvous(&v, sizeof(T), tag, &e.live);
    return e;
}

} // namespace emu

#define threadIdx (emu::Idx{emu::tid(), 0u, 0u})
#define blockIdx (emu::Idx{emu::bid(), 0u, 0u})
#define blockDim (emu::Idx{emu::nthreads(), 1u, 1u})
#define gridDim (emu::Idx{emu::nblocks(), 1u, 1u})
#define warpSize 64

// A workgroup barrier holds every fiber until all live fibers of the workgroup are at a barrier in the same round:
// wavefronts that run different code (a producer and a consumer wavefront, say) reach it after different numbers of
// cross-lane primitives.
static inline void __syncthreads()
{
    do {
        (void)emu::exchange<int>(0, 1);
    } while (!emu::all_at_barrier());
}

template <typename T> static inline T emu_shfl_abs(T v, unsigned tag, int rel_kind, int arg)
{
    auto e = emu::exchange<T>(v, tag);
    unsigned t = emu::tid(), lane = t & 63u, base = t & ~63u;
    int src;
    switch (rel_kind) {
    case 0: src = arg & 63; break;                    // absolute
    case 1: src = (int)lane - arg; break;             // up
    case 2: src = (int)lane + arg; break;             // down
    default: src = (int)(lane ^ (unsigned)arg); break; // xor
    }
    if (src < 0 || src > 63 || base + (unsigned)src >= emu::nthreads() || !e.live[base + (unsigned)src])
        return v;
    return e.at(base + (unsigned)src);
}
static inline int __shfl(int v, int src, int = 64) { return emu_shfl_abs<int>(v, 2, 0, src); }
static inline int __shfl_up(int v, unsigned d, int = 64) { return emu_shfl_abs<int>(v, 3, 1, (int)d); }
static inline int __shfl_down(int v, unsigned d, int = 64) { return emu_shfl_abs<int>(v, 4, 2, (int)d); }
static inline int __shfl_xor(int v, int m, int = 64) { return emu_shfl_abs<int>(v, 5, 3, m); }

static inline unsigned long long __ballot(int pred)
{
    auto e = emu::exchange<int>(pred ? 1 : 0, 6);
    unsigned t = emu::tid(), base = t & ~63u;
    unsigned long long m = 0;
    for (unsigned i = 0; i < 64 && base + i < emu::nthreads(); ++i)
        if (e.live[base + i] && e.at(base + i))
            m |= 1ull << i;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred)
{
    auto e = emu::exchange<int>(pred ? 1 : 0, 7);
    unsigned t = emu::tid(), base = t & ~63u;
    for (unsigned i = 0; i < 64 && base + i < emu::nthreads(); ++i)
        if (e.live[base + i] && !e.at(base + i))
            return 0;
    return 1;
}
static inline int __builtin_amdgcn_readlane(int v, int lane) { return emu_shfl_abs<int>(v, 8, 0, lane); }
static inline int __builtin_amdgcn_readfirstlane(int v)
{
    auto e = emu::exchange<int>(v, 9);
    unsigned t = emu::tid(), base = t & ~63u;
    for (unsigned i = 0; i < 64 && base + i < emu::nthreads(); ++i)
        if (e.live[base + i])
            return e.at(base + i);
    return v;
}
static inline int __builtin_amdgcn_ds_bpermute(int addr, int v) { return emu_shfl_abs<int>(v, 10, 0, (addr >> 2) & 63); }
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u));
}
static inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t add)
{
    unsigned lane = emu::tid() & 63u;
    uint32_t m = lane >= 32 ? mask : (mask & ((1u << lane) - 1u));
    return add + (uint32_t)__builtin_popcount(m);
}
static inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t add)
{
    unsigned lane = emu::tid() & 63u;
    uint32_t m = lane <= 32 ? 0u : (mask & ((1u << (lane - 32)) - 1u));
    return add + (uint32_t)__builtin_popcount(m);
}
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline int __mul24(int a, int b) { return (int)((uint32_t)((int)((uint32_t)a << 8) >> 8) * (uint32_t)((int)((uint32_t)b << 8) >> 8)); }
static inline long long clock64() { return 0; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned __lane_id() { return emu::tid() & 63u; }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}

// IEEE single ops that must not be contracted (g++ -ffp-contract=off keeps them separate)
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fsqrt_rn(float a) { volatile float r = __builtin_sqrtf(a); return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }

template <typename T> static inline T atomicAdd(T *p, T v)
{
    T o = *p;
    *p = o + v;
    return o;
}
template <typename T> static inline T atomicMax(T *p, T v)
{
    T o = *p;
    if (v > o)
        *p = v;
    return o;
}
template <typename T> static inline T atomicOr(T *p, T v)
{
    T o = *p;
    *p = o | v;
    return o;
}

#define __ATOMIC_RELAXED_HIP 0
#define __HIP_MEMORY_SCOPE_AGENT 4
template <typename T> static inline T __hip_atomic_load(T const *p, int, int) { return *p; }
static inline void __threadfence() {}

// ---- runtime API subset ----
typedef int hipError_t;
typedef void *hipStream_t;
struct EmuEvent { double t; };
typedef EmuEvent *hipEvent_t;
enum { hipSuccess = 0, hipErrorNoDevice = 100, hipErrorInvalidDevice = 101, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0 };

hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
enum { hipHostRegisterPortable = 1 };
hipError_t hipHostRegister(void *p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void *p);
hipError_t hipMemcpy(void *d, void const *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, void const *s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes);
enum { hipStreamNonBlocking = 1 };
hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipDeviceSynchronize();
hipError_t hipGetDeviceCount(int *n); /* R433_EMU_DEVICES pretend devices (default 1): one address space, the current one is thread-local */
hipError_t hipSetDevice(int device);
hipError_t hipGetDevice(int *device);
enum { hipHostMallocPortable = 1 };
hipError_t hipGetLastError();
char const *hipGetErrorString(hipError_t e);
enum { hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned flags);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e); /* (launches are synchronous here: always complete) */
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                                                      \
    emu::launch((grid), (block), [&]() { kern(__VA_ARGS__); })
