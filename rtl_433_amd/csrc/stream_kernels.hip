// stream_kernels.hip -- the IQ -> pulse-package kernel: one capture per lane.
//
// A wavefront owns 64 captures.  Per tile it pulls 64 samples of each of its captures from HBM
// with 16-byte-per-lane loads (one or two full 128-byte lines per capture and tile, nothing
// fetched twice), parks them transposed in LDS, and then every lane walks its own capture
// serially: envelope -> first-order low-pass -> FM discriminator + low-pass -> OOK/FSK pulse
// detector, all carried in registers with the reference's exact integer semantics (the three
// recurrences truncate, so they cannot be re-associated -- see DESIGN.md).  Packages leave as
// r433_pkg_rec records in a per-capture arena.  The next tile's loads are issued before the
// current tile is consumed so HBM latency hides behind the ~100 VALU ops per sample.
//
// Replaces, for file input: envelope_detect / magnitude_est_* (reference src/baseband.c:36-110),
// baseband_low_pass_filter (:145-169), baseband_demod_FM(_cs16) (:210-366), the frame loop of
// push_sdr_flow (src/r_flow.c:149-244) and pulse_detect_package (src/pulse_detect.c:199-483).
#include "dsp_device.hpp"
#include "r433_internal.hpp"

namespace r433 {

namespace {

constexpr int kTile = 64; // samples per capture per tile

template <int SS> struct TileGeom {
    static constexpr int row_bytes = kTile * SS;        // 128 (cu8) / 256 (cs16)
    static constexpr int row_pitch = row_bytes + 16;    // +16 B: conflict-free ds_read_b128 down a column
    static constexpr int lanes_per_row = row_bytes / 16;
    static constexpr int rows_per_load = 64 / lanes_per_row;
    static constexpr int n_loads = 64 / rows_per_load;  // 8 (cu8) / 16 (cs16)
    static constexpr int vecs_per_row = row_bytes / 16;
    static constexpr int samples_per_vec = 16 / SS;
};

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1)
        v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

struct LaneCtx {
    DetLane det;
    DetCfg cfg;
    FmLane fm;
    int lpf_y, lpf_x;
    uint64_t input_pos;
    uint32_t frame;
    uint32_t fsum;
    int dc;   // data_counter within the current frame
    int flen; // length of the current frame
};

template <int SS>
__device__ __forceinline__ void process_sample(LaneCtx &L, StreamParams const &p, uint32_t s, uint32_t idx,
        uint32_t my_n, int vi, int vq)
{
    if (L.dc == 0) { // a new frame == a new push_sdr_flow call
        uint32_t remaining = my_n - idx;
        L.flen = (int)min(remaining, p.frame_samples);
        if (p.frame_min_high)
            L.cfg.min_high = p.frame_min_high[(uint64_t)s * p.frames_cap + min(L.frame, p.frames_cap - 1)];
        L.lpf_x = (int)(int16_t)L.lpf_x; // the filter state keeps x[-1] in an int16 slot (baseband.c:167)
        L.fsum = 0;
        det_call_entry(L.det, L.cfg, L.flen, 0);
    }
    uint32_t env;
    if (SS == 2)
        env = p.use_mag ? env_mag_cu8((uint32_t)vi, (uint32_t)vq) : env_amp_cu8((uint32_t)vi, (uint32_t)vq);
    else
        env = env_mag_cs16(vi, vq);
    L.fsum += env;
    int am = lpf_step(L.lpf_y, (int)env, L.lpf_x);
    L.lpf_y = am;
    L.lpf_x = (int)env;
    int fm;
    if (p.enable_fm)
        fm = SS == 2 ? fm_step_cu8(L.fm, (uint32_t)vi, (uint32_t)vq, p.a16, p.b16) : fm_step_cs16(L.fm, vi, vq, p.a32, p.b32);
    else
        fm = (int)(int16_t)env; // buf.fm aliases the raw envelope (reference include/r_private.h:32-36)

    if (p.tap_am) {
        uint64_t o = (uint64_t)s * p.tap_stride + L.input_pos + (uint64_t)L.dc;
        p.tap_env[o] = (uint16_t)env;
        p.tap_am[o] = (int16_t)am;
        p.tap_fm[o] = (int16_t)fm;
    }

    int r = det_step(L.det, L.cfg, am, fm, L.flen, L.dc, L.input_pos, L.frame);
    if (r) { // package returned: the next call starts at the same sample, in the idle state
        det_call_entry(L.det, L.cfg, L.flen, L.dc);
        det_idle(L.det, L.cfg, am, L.flen, L.dc, L.input_pos);
    }
    L.dc += 1;
    if (L.dc == L.flen) {
        if (p.frame_sums && L.frame < p.frames_cap)
            p.frame_sums[(uint64_t)s * p.frames_cap + L.frame] = L.fsum;
        L.input_pos += (uint64_t)L.flen;
        L.frame += 1;
        L.dc = 0;
    }
}

template <int SS> __global__ __launch_bounds__(64) void k_stream(StreamParams p)
{
    using G = TileGeom<SS>;
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * G::row_pitch];

    int const lane = (int)threadIdx.x;
    uint32_t const s0 = blockIdx.x * 64u;
    uint32_t const s = s0 + (uint32_t)lane;
    bool const active = s < p.n_streams;
    uint32_t const my_bytes = active ? (p.stream_bytes ? p.stream_bytes[s] : p.uniform_bytes) : 0u;
    uint32_t const my_n = my_bytes / SS;
    uint32_t const n_tiles = (wave_max_u32(my_n) + kTile - 1) / kTile;

    // ---- lane state ----
    LaneCtx L;
    L.cfg = p.det;
    L.det.arena = p.arena + (uint64_t)(active ? s : 0) * p.arena_stride;
    L.det.fsk_ring = p.fsk_ring + (uint64_t)(active ? s : 0) * R433_PD_MAX_PULSES;
    L.det.arena_cap = p.arena_stride;
    L.det.stream = s;
    L.dc = 0;
    L.flen = 0;
    L.fsum = 0;
    if (active && (p.flags & RUN_CONTINUE)) {
        StreamState const &S = p.state[s];
        L.lpf_y = S.lpf_y;
        L.lpf_x = S.lpf_x;
        L.fm = FmLane{S.fm_xr, S.fm_xi, S.fm_xf, S.fm_yf};
        L.det.state = S.state;
        L.det.run = S.run;
        L.det.max_pulse = S.max_pulse;
        L.det.lead_in = S.lead_in;
        L.det.low = S.low;
        L.det.high = S.high;
        L.det.f_run = S.f_run;
        L.det.f_state = S.f_state;
        L.det.f_f1 = S.f_f1;
        L.det.f_f2 = S.f_f2;
        L.det.f_vmax = S.f_vmax;
        L.det.f_vmin = S.f_vmin;
        L.det.f_skip = S.f_skip;
        L.det.ook_num = S.ook_num;
        L.det.cur_pulse = S.cur_pulse;
        L.det.ook_f1 = S.ook_f1;
        L.det.fsk_num = S.fsk_num;
        L.det.start_ago = S.start_ago;
        L.det.offset = S.offset;
        L.det.fsk_offset = S.fsk_offset;
        L.input_pos = S.input_pos;
        L.frame = S.frame;
        L.det.cursor = S.cursor;
        L.det.n_pkgs = S.n_pkgs;
        L.det.overflow = S.overflow;
        L.det.eop_spurious = 0;
    }
    else {
        det_reset(L.det);
        L.lpf_y = L.lpf_x = 0;
        L.fm = FmLane{0, 0, 0, 0};
        L.input_pos = 0;
        L.frame = 0;
        L.det.cursor = 0;
        L.det.n_pkgs = 0;
        L.det.overflow = 0;
    }

    // ---- cooperative tile loads: instruction k covers rows_per_load captures x row_bytes ----
    int const ld_row = lane / G::lanes_per_row;
    int const ld_col = (lane % G::lanes_per_row) * 16;
    uint4 pf[G::n_loads];
    auto issue_loads = [&](uint32_t t) {
#pragma unroll
        for (int k = 0; k < G::n_loads; ++k) {
            uint32_t row = (uint32_t)(k * G::rows_per_load + ld_row);
            uint32_t rs = s0 + row;
            uint64_t off = (uint64_t)t * G::row_bytes + (uint64_t)ld_col;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (rs < p.n_streams && off + 16 <= p.stride_bytes)
                v = *(uint4 const *)(p.iq + (uint64_t)rs * p.stride_bytes + off);
            pf[k] = v;
        }
    };

    if (n_tiles > 0)
        issue_loads(0);
    for (uint32_t t = 0; t < n_tiles; ++t) {
#pragma unroll
        for (int k = 0; k < G::n_loads; ++k) {
            int row = k * G::rows_per_load + ld_row;
            *(uint4 *)(tile + row * G::row_pitch + ld_col) = pf[k];
        }
        __syncthreads();
        if (t + 1 < n_tiles)
            issue_loads(t + 1);

        uint32_t const base = t * kTile;
        if (base < my_n) {
#pragma unroll 1
            for (int v = 0; v < G::vecs_per_row; ++v) {
                uint4 w = *(uint4 const *)(tile + lane * G::row_pitch + v * 16);
#pragma unroll 1
                for (int j = 0; j < G::samples_per_vec; ++j) {
                    uint32_t idx = base + (uint32_t)(v * G::samples_per_vec + j);
                    int vi, vq;
                    if (SS == 2) { // one sample = low 16 bits; then shift the 128-bit vector down
                        vi = (int)(w.x & 0xffu);
                        vq = (int)((w.x >> 8) & 0xffu);
                        w.x = __builtin_amdgcn_alignbit(w.y, w.x, 16);
                        w.y = __builtin_amdgcn_alignbit(w.z, w.y, 16);
                        w.z = __builtin_amdgcn_alignbit(w.w, w.z, 16);
                        w.w >>= 16;
                    }
                    else {
                        vi = (int)(int16_t)(w.x & 0xffffu);
                        vq = (int)(int16_t)(w.x >> 16);
                        w.x = w.y;
                        w.y = w.z;
                        w.z = w.w;
                    }
                    if (idx < my_n)
                        process_sample<SS>(L, p, s, idx, my_n, vi, vq);
                }
            }
        }
        __syncthreads();
    }

    if (!active)
        return;
    if (!(p.flags & RUN_NOFLUSH))
        det_flush(L.det, L.cfg, L.frame);

    StreamState &S = p.state[s];
    S.lpf_y = L.lpf_y;
    S.lpf_x = L.lpf_x;
    S.fm_xr = L.fm.xr;
    S.fm_xi = L.fm.xi;
    S.fm_xf = L.fm.xf;
    S.fm_yf = L.fm.yf;
    S.state = L.det.state;
    S.run = L.det.run;
    S.max_pulse = L.det.max_pulse;
    S.lead_in = L.det.lead_in;
    S.low = L.det.low;
    S.high = L.det.high;
    S.f_run = L.det.f_run;
    S.f_state = L.det.f_state;
    S.f_f1 = L.det.f_f1;
    S.f_f2 = L.det.f_f2;
    S.f_vmax = L.det.f_vmax;
    S.f_vmin = L.det.f_vmin;
    S.f_skip = L.det.f_skip;
    S.ook_num = L.det.ook_num;
    S.cur_pulse = L.det.cur_pulse;
    S.ook_f1 = L.det.ook_f1;
    S.fsk_num = L.det.fsk_num;
    S.start_ago = L.det.start_ago;
    S.offset = L.det.offset;
    S.fsk_offset = L.det.fsk_offset;
    S.input_pos = L.input_pos;
    S.frame = L.frame;
    S.cursor = L.det.cursor;
    S.n_pkgs = L.det.n_pkgs;
    S.overflow = L.det.overflow;
}

// ---- package directory: canonical (capture, detection order) numbering ----

__global__ __launch_bounds__(1024) void k_pkg_scan(StreamState const *state, uint32_t n_streams, uint32_t *pkg_base,
        uint32_t *scal)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    __shared__ uint32_t any_overflow;
    int const tid = (int)threadIdx.x;
    if (tid == 0) {
        carry = 0;
        any_overflow = 0;
    }
    __syncthreads();
    for (uint32_t base = 0; base < n_streams; base += 1024) {
        uint32_t i = base + (uint32_t)tid;
        uint32_t v = i < n_streams ? state[i].n_pkgs : 0u;
        if (i < n_streams && state[i].overflow)
            any_overflow = 1;
        part[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) { // Hillis-Steele inclusive scan
            uint32_t add = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (i < n_streams)
            pkg_base[i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023)
            carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) {
        scal[0] = carry;
        scal[1] = any_overflow;
    }
}

__global__ __launch_bounds__(256) void k_pkg_directory(uint8_t const *arena, uint32_t arena_stride,
        StreamState const *state, uint32_t n_streams, uint32_t const *pkg_base, uint32_t *dir_stream,
        uint32_t *dir_off, uint32_t *rec_bytes, uint32_t max_pkgs)
{
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams)
        return;
    uint32_t n = state[s].n_pkgs;
    uint32_t at = 0;
    uint8_t const *a = arena + (uint64_t)s * arena_stride;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t g = pkg_base[s] + i;
        uint32_t sz = *(uint32_t const *)(a + at);
        if (g < max_pkgs) {
            dir_stream[g] = s;
            dir_off[g] = at;
            rec_bytes[g] = sz;
        }
        at += sz;
    }
}

// one wavefront per package: records are multiples of 8 bytes and 8-byte aligned on both sides
__global__ __launch_bounds__(64) void k_gather_packages(uint8_t const *arena, uint32_t arena_stride,
        uint32_t const *dir_stream, uint32_t const *dir_off, uint32_t const *rec_off, uint32_t const *n_pkgs,
        uint32_t max_pkgs, uint8_t *dst, uint32_t dst_cap)
{
    uint32_t const n = min(*n_pkgs, max_pkgs);
    for (uint32_t g = blockIdx.x; g < n; g += gridDim.x) {
        uint8_t const *src = arena + (uint64_t)dir_stream[g] * arena_stride + dir_off[g];
        uint32_t bytes = *(uint32_t const *)src;
        uint32_t off = rec_off[g];
        if ((uint64_t)off + bytes > dst_cap)
            continue;
        uint2 const *s2 = (uint2 const *)src;
        uint2 *d2 = (uint2 *)(dst + off);
        for (uint32_t i = threadIdx.x; i < bytes / 8; i += 64)
            d2[i] = s2[i];
    }
}

} // namespace

void launch_stream(StreamParams const &p, uint32_t sample_size, hipStream_t st)
{
    if (p.n_streams == 0)
        return;
    dim3 grid((p.n_streams + 63) / 64), block(64);
    if (sample_size == 2)
        hipLaunchKernelGGL(k_stream<2>, grid, block, 0, st, p);
    else
        hipLaunchKernelGGL(k_stream<4>, grid, block, 0, st, p);
}

void launch_pkg_scan(StreamState const *state, uint32_t n_streams, uint32_t *pkg_base, uint32_t *scal, hipStream_t st)
{
    hipLaunchKernelGGL(k_pkg_scan, dim3(1), dim3(1024), 0, st, state, n_streams, pkg_base, scal);
}

void launch_directory(uint8_t const *arena, uint32_t arena_stride, StreamState const *state, uint32_t n_streams,
        uint32_t const *pkg_base, uint32_t *dir_stream, uint32_t *dir_off, uint32_t *rec_bytes, uint32_t max_pkgs,
        hipStream_t st)
{
    hipLaunchKernelGGL(k_pkg_directory, dim3((n_streams + 255) / 256), dim3(256), 0, st, arena, arena_stride, state,
            n_streams, pkg_base, dir_stream, dir_off, rec_bytes, max_pkgs);
}

void launch_gather_packages(uint8_t const *arena, uint32_t arena_stride, uint32_t const *dir_stream,
        uint32_t const *dir_off, uint32_t const *rec_off, uint32_t const *n_pkgs, uint32_t max_pkgs, uint8_t *dst,
        uint32_t dst_cap, uint32_t grid_pkgs, hipStream_t st)
{
    uint32_t grid = grid_pkgs < 1 ? 1 : (grid_pkgs > 8192 ? 8192 : grid_pkgs);
    hipLaunchKernelGGL(k_gather_packages, dim3(grid), dim3(64), 0, st, arena, arena_stride, dir_stream, dir_off, rec_off,
            n_pkgs, max_pkgs, dst, dst_cap);
}

} // namespace r433
