// slicer_kernels.hip -- decoder fan-out: every pulse package against every registered r_device.
//
// Work item = (package, chunk of 64 devices); one wavefront per item.  The package's (pulse, gap)
// pairs are staged once in LDS (<= 9.6 KB, coalesced 8-byte loads) and every lane runs the slicer
// of its own device over them (LDS broadcast reads).  Devices are grouped by modulation and every
// group is padded to whole wavefronts, so a wavefront executes exactly one slicer: the slowest lane
// of a mixed wavefront used to pay for every slicer present in it.  Two passes over the same code (COUNT, WRITE) around an
// exclusive scan give a dense event stream in canonical (package, device, event) order without
// atomics on the payload.
//
// Replaces run_ook_demods / run_fsk_demods (reference src/r_api.c:438-550) and the ten
// pulse_slicer_* functions (src/pulse_slicer.c) up to, not including, the decode_fn call, which
// stays on the host behind the r_device ABI.
#include "r433_internal.hpp"
#include "slicer_device.hpp"

namespace r433 {

namespace {

constexpr uint32_t kMaxDevs = 2048;

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    uint32_t x = v;
    int const lane = (int)(threadIdx.x & 63);
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
        if (lane >= o)
            x += y;
    }
    total = (uint32_t)__shfl((int)x, 63, 64);
    return x - v;
}

template <bool WRITE> __global__ __launch_bounds__(64) void k_slice(SliceParams p)
{
    __shared__ int2 pairs[R433_PD_MAX_PULSES];
    __shared__ uint32_t prefix[WRITE ? kMaxDevs : 1];

    uint32_t const n_pkgs = min(*p.n_pkgs, p.max_pkgs);
    uint32_t const chunks = p.n_rows / 64;
    uint32_t const lane = threadIdx.x;

    for (uint32_t work = blockIdx.x; work < n_pkgs * chunks; work += gridDim.x) {
        uint32_t const pkg = work / chunks;
        uint32_t const chunk = work - pkg * chunks;
        uint8_t const *rec = p.arena + (uint64_t)p.dir_stream[pkg] * p.arena_stride + p.dir_off[pkg];
        uint32_t const type = ((uint32_t const *)rec)[2];
        uint32_t const num = min(((uint32_t const *)rec)[3], (uint32_t)R433_PD_MAX_PULSES);
        int2 const *src = (int2 const *)(rec + sizeof(r433_pkg_rec));
        __syncthreads(); // previous item done with LDS
        for (uint32_t i = lane; i < num; i += 64)
            pairs[i] = src[i];
        if (WRITE) { // exclusive prefix of this package's per-device sizes, in registration order
            uint32_t carry = 0;
            for (uint32_t b = 0; b < p.n_devs; b += 64) {
                uint32_t i = b + lane;
                uint32_t v = i < p.n_devs ? p.sizes[(uint64_t)pkg * p.n_devs + i] : 0u;
                uint32_t tot;
                uint32_t ex = wave_excl_scan(v, tot);
                if (i < p.n_devs)
                    prefix[i] = carry + ex;
                carry += tot;
            }
        }
        __syncthreads();

        uint32_t const di = chunk * 64 + lane;
        uint32_t my_bytes = 0;
        DevRow const t = p.devs[di];
        if (t.orig >= 0) { // not a padding row
            bool const run = t.valid && (t.is_fsk != 0) == (type == R433_PKG_FSK);
            BitSink<WRITE> sink;
            uint8_t *out = nullptr;
            uint32_t limit = 0;
            bool fits = true;
            if (WRITE) {
                uint32_t base = p.pkg_off[pkg] + prefix[t.orig];
                limit = p.sizes[(uint64_t)pkg * p.n_devs + t.orig];
                fits = limit > 0 && (uint64_t)base + limit <= p.events_cap;
                out = p.events + base;
            }
            if (run && fits) {
                PulseView pv{pairs, num};
                sink.begin(out, limit, pkg, (uint32_t)t.orig);
                slice_dispatch<WRITE>(pv, t, sink);
                my_bytes = sink.off;
            }
            if (!WRITE)
                p.sizes[(uint64_t)pkg * p.n_devs + t.orig] = my_bytes;
        }
        if (!WRITE) {
            uint32_t tot;
            wave_excl_scan(my_bytes, tot);
            if (lane == 0 && tot)
                atomicAdd(&p.pkg_bytes[pkg], tot);
        }
    }
}

__global__ __launch_bounds__(1024) void k_scan_u32(uint32_t const *in, uint32_t *out, uint32_t const *n_ptr,
        uint32_t n_cap, uint32_t *total)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    uint32_t const n = min(*n_ptr, n_cap);
    int const tid = (int)threadIdx.x;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + (uint32_t)tid;
        uint32_t v = i < n ? in[i] : 0u;
        part[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t add = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (i < n)
            out[i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023)
            carry += part[1023];
        __syncthreads();
    }
    if (tid == 0)
        *total = carry;
}

uint32_t slice_grid(uint32_t grid_pkgs, uint32_t n_rows)
{
    uint64_t items = (uint64_t)grid_pkgs * (n_rows / 64);
    if (items < 1)
        items = 1;
    // 256 CUs x 8 wavefront slots per SIMD pair is plenty; the kernel grid-strides beyond this
    return (uint32_t)(items < 16384 ? items : 16384);
}

} // namespace

void launch_slice_count(SliceParams const &p, uint32_t grid_pkgs, hipStream_t st)
{
    hipLaunchKernelGGL(k_slice<false>, dim3(slice_grid(grid_pkgs, p.n_rows)), dim3(64), 0, st, p);
}

void launch_scan_u32(uint32_t const *in, uint32_t *out, uint32_t const *n_ptr, uint32_t n_cap, uint32_t *total,
        hipStream_t st)
{
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, in, out, n_ptr, n_cap, total);
}

void launch_slice_write(SliceParams const &p, uint32_t grid_pkgs, hipStream_t st)
{
    hipLaunchKernelGGL(k_slice<true>, dim3(slice_grid(grid_pkgs, p.n_rows)), dim3(64), 0, st, p);
}

} // namespace r433
