// detect_seam.hip -- r433_detector_*: the reference's pulse_detect_package() contract (include/pulse_detect.h:37-71,
// src/pulse_detect.c:199-483) behind the C ABI, for the function-level seam (csrc/ref_seam.cpp exports it under the
// reference's own name).  One call = one visit of a buffer of filtered envelope / discriminator samples: the detector
// walks on from where the last call stopped, returns at the first package that ends (the next call re-examines the same
// sample, as the reference does by returning without advancing data_counter) or at the end of the buffer, and is
// resumable across buffers; a call with len == 0 flushes.  The state machine is the exact general step the detection
// kernel uses for everything irregular (csrc/detect_device.hpp), run by one wavefront sample by sample: this entry point
// is about the contract, not about speed -- the fast path is push_sdr_flow / r433_batch_run, where the same state machine
// sits fused behind the filters with its skip-ahead legs (k_wave).  Nothing is computed on the host.
#include "host_common.hpp"

#include "detect_device.hpp"

namespace r433 {

// what survives between calls (the reference keeps part of it in pulse_detect_t, part in the caller's two pulse_data_t)
struct DetSeamState {
    int state, run, max_pulse, lead_in, low, high;
    uint32_t f_run;
    int f_state, f_f1, f_f2, f_vmax, f_vmin, f_skip;
    uint32_t ook_num;
    int cur_pulse, ook_f1;
    uint32_t fsk_num;
    uint64_t offset, fsk_offset;
    uint32_t start_ago;
    int data_counter; // next sample of the current buffer
    int ret;          // result of the last call: 0, R433_PKG_OOK, R433_PKG_FSK
    uint32_t overflow;
    uint32_t starts;  // packages begun so far (the reference clears the caller's two structs whenever one begins)
};

namespace {

constexpr uint32_t kSeamArena = (uint32_t)sizeof(r433_pkg_rec) + 8u * (R433_PD_MAX_PULSES + 8);

__global__ __launch_bounds__(64) void k_detect_call(int16_t const *am, int16_t const *fm, int len, uint64_t input_pos, DetCfg cfg,
        DetSeamState *st, uint8_t *arena, int2 *ring)
{
    int const lane = (int)threadIdx.x;
    DetLane d;
    d.state = st->state, d.run = st->run, d.max_pulse = st->max_pulse, d.lead_in = st->lead_in, d.low = st->low, d.high = st->high;
    d.f_run = st->f_run, d.f_state = st->f_state, d.f_f1 = st->f_f1, d.f_f2 = st->f_f2, d.f_vmax = st->f_vmax, d.f_vmin = st->f_vmin;
    d.f_skip = st->f_skip;
    d.ook_num = st->ook_num, d.cur_pulse = st->cur_pulse, d.ook_f1 = st->ook_f1, d.fsk_num = st->fsk_num;
    d.offset = st->offset, d.fsk_offset = st->fsk_offset, d.start_ago = st->start_ago;
    d.eop_spurious = 0;
    d.arena = arena;
    d.fsk_ring = ring;
    d.writer = lane == 0;
    d.arena_cap = kSeamArena;
    d.cursor = 0;   // one package at a time: a returned record sits at the start of the arena, the pairs of the open
    d.ook_base = 0; // package right behind where its header will go
    d.n_pkgs = 0;
    d.overflow = 0;
    d.stream = 0;
    int pos = st->data_counter;
    uint32_t starts = st->starts;
    int ret = 0;
    __syncthreads(); // every lane has read the state before lane 0 writes it back
    if (len == 0) {
        ret = det_flush(d, cfg, 0);
    }
    else {
        det_call_entry(d, cfg, len, pos);
        while (pos < len && !ret) {
            int const base = pos & ~63;
            int const il = base + lane;
            int const a = il < len ? (int)am[il] : 0, f = il < len ? (int)fm[il] : 0;
            int const e = min(base + 64, len);
            while (pos < e) {
                bool const was_idle = d.state == ST_IDLE;
                ret = det_step(d, cfg, __builtin_amdgcn_readlane(a, pos - base), __builtin_amdgcn_readlane(f, pos - base), len, pos,
                        input_pos, 0);
                if (was_idle && d.state != ST_IDLE)
                    starts += 1;
                if (ret)
                    break; // the same sample is looked at again by the next call
                ++pos;
            }
        }
        if (!ret)
            pos = 0; // "out of data": the next buffer starts at its first sample
    }
    if (lane == 0) {
        st->state = d.state, st->run = d.run, st->max_pulse = d.max_pulse, st->lead_in = d.lead_in, st->low = d.low, st->high = d.high;
        st->f_run = d.f_run, st->f_state = d.f_state, st->f_f1 = d.f_f1, st->f_f2 = d.f_f2, st->f_vmax = d.f_vmax, st->f_vmin = d.f_vmin;
        st->f_skip = d.f_skip;
        st->ook_num = d.ook_num, st->cur_pulse = d.cur_pulse, st->ook_f1 = d.ook_f1, st->fsk_num = d.fsk_num;
        st->offset = d.offset, st->fsk_offset = d.fsk_offset, st->start_ago = d.start_ago;
        st->data_counter = pos;
        st->ret = ret;
        st->overflow = d.overflow;
        st->starts = starts;
    }
}

void reset_state(DetSeamState &s)
{
    memset(&s, 0, sizeof(s));
    s.f_vmax = -32768; // fsk_reset / det_reset of detect_device.hpp
    s.f_vmin = 32767;
    s.f_skip = 40;
}

} // namespace

} // namespace r433

using namespace r433;

struct r433_detector {
    int use_mag = 0;
    float fixed_db = 0.0f, min_db = -12.1442f, ratio_db = 9.0f; // src/pulse_detect.c:56-67
    DetSeamState host;
    uint32_t seen_starts = 0; // package starts whose clearing of the caller's structs has been done
    bool dirty = true; // `host` is newer than the device copy
    DevBuf<int16_t> d_am, d_fm;
    DevBuf<DetSeamState> d_state;
    DevBuf<uint8_t> d_arena;
    DevBuf<int2> d_ring;
    std::vector<uint8_t> h_arena;
    std::vector<int2> h_ring;
};

extern "C" {

r433_detector *r433_detector_create(void)
{
    r433_detector *d = new (std::nothrow) r433_detector();
    if (!d)
        return nullptr;
    reset_state(d->host);
    d->h_arena.resize(kSeamArena);
    d->h_ring.resize(R433_PD_MAX_PULSES);
    return d;
}

void r433_detector_destroy(r433_detector *d)
{
    if (!d)
        return;
    d->d_am.release();
    d->d_fm.release();
    d->d_state.release();
    d->d_arena.release();
    d->d_ring.release();
    delete d;
}

void r433_detector_reset(r433_detector *d)
{
    if (!d)
        return;
    reset_state(d->host);
    d->seen_starts = 0;
    d->dirty = true;
}

void r433_detector_set_levels(r433_detector *d, int use_mag_est, float fixed_high_level, float min_high_level, float high_low_ratio)
{
    if (!d)
        return;
    d->use_mag = use_mag_est;
    d->fixed_db = fixed_high_level;
    d->min_db = min_high_level;
    d->ratio_db = high_low_ratio;
}

// the detector-owned fields of one pulse_data_t from a returned package record
static void fill_from_record(r433_pulse_data *p, uint8_t const *rec)
{
    uint32_t h[16];
    memcpy(h, rec, sizeof(h));
    uint32_t const num = h[3] < R433_MAX_PULSES ? h[3] : R433_MAX_PULSES;
    p->offset = (uint64_t)h[6] | ((uint64_t)h[7] << 32);
    p->sample_rate = h[14];
    p->start_ago = h[8];
    p->end_ago = h[9];
    p->num_pulses = num;
    int2 const *pairs = (int2 const *)(rec + sizeof(r433_pkg_rec));
    for (uint32_t i = 0; i < num; ++i) {
        p->pulse[i] = pairs[i].x;
        p->gap[i] = pairs[i].y;
    }
    p->ook_low_estimate = (int)h[10];
    p->ook_high_estimate = (int)h[11];
    p->fsk_f1_est = (int)h[12];
    p->fsk_f2_est = (int)h[13];
}

int r433_detector_package(r433_detector *d, int16_t const *envelope, int16_t const *fm, int len, uint32_t samp_rate, uint64_t sample_offset,
        r433_pulse_data *pulses, r433_pulse_data *fsk_pulses, unsigned fpdm)
{
    if (!d || !pulses || !fsk_pulses || len < 0 || (len > 0 && (!envelope || !fm)))
        return fail(R433_EINVAL, "r433_detector_package: null argument or negative length");
    if (r433_device_count() < 0)
        return R433_ENODEV;
    int rc;
    if ((rc = d->d_state.ensure(1)) || (rc = d->d_arena.ensure(kSeamArena)) || (rc = d->d_ring.ensure(R433_PD_MAX_PULSES))
            || (rc = d->d_am.ensure((size_t)len + 64)) || (rc = d->d_fm.ensure((size_t)len + 64)))
        return rc;
    if (d->dirty) {
        HIP_TRY(hipMemcpy(d->d_state.p, &d->host, sizeof(d->host), hipMemcpyHostToDevice));
        d->dirty = false;
    }
    if (len > 0) { // every call: the reference reads whatever buffer it is handed, at its data_counter
        HIP_TRY(hipMemcpy(d->d_am.p, envelope, (size_t)len * sizeof(int16_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->d_fm.p, fm, (size_t)len * sizeof(int16_t), hipMemcpyHostToDevice));
    }
    DetCfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    levels_from_db(cfg, d->use_mag, d->fixed_db, d->min_db, d->ratio_db);
    cfg.per_ms = (int)(samp_rate / 1000);
    cfg.rate = samp_rate;
    cfg.fpdm = fpdm ? 1 : 0; // FSK_PULSE_DETECT_OLD = 0: classic; anything else: min/max (include/pulse_detect.h:26-33)
    hipLaunchKernelGGL(k_detect_call, dim3(1), dim3(64), 0, 0, d->d_am.p, d->d_fm.p, len, sample_offset, cfg, d->d_state.p, d->d_arena.p,
            d->d_ring.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(&d->host, d->d_state.p, sizeof(d->host), hipMemcpyDeviceToHost));
    DetSeamState const &s = d->host;
    if (s.overflow)
        return fail(R433_EOVERFLOW, "r433_detector_package: package arena overflow");
    // What the reference's two structs hold now.  It clears both whenever a package begins (src/pulse_detect.c:311-318), builds
    // the package's list in one and the FSK candidate in the other as it goes, and adds the levels and end_ago when it returns
    // one; an idle detector leaves them alone apart from their age.  The lists live on the device (arena, candidate ring).
    HIP_TRY(hipMemcpy(d->h_arena.data(), d->d_arena.p, kSeamArena, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(d->h_ring.data(), d->d_ring.p, R433_PD_MAX_PULSES * sizeof(int2), hipMemcpyDeviceToHost));
    if (s.starts != d->seen_starts) {
        d->seen_starts = s.starts;
        memset(pulses, 0, sizeof(*pulses));
        memset(fsk_pulses, 0, sizeof(*fsk_pulses));
    }
    if (s.starts) {
        pulses->sample_rate = fsk_pulses->sample_rate = samp_rate;
        int2 const *pairs = (int2 const *)(d->h_arena.data() + sizeof(r433_pkg_rec));
        // (an FSK package ends inside the first pulse: its record went over the place of the pairs, of which there are none)
        uint32_t const n = s.ret == R433_PKG_FSK ? 0u : s.ook_num < R433_MAX_PULSES ? s.ook_num : R433_MAX_PULSES;
        pulses->num_pulses = n;
        for (uint32_t i = 0; i < n; ++i) {
            pulses->pulse[i] = pairs[i].x;
            pulses->gap[i] = pairs[i].y;
        }
        if (n < R433_MAX_PULSES)
            pulses->pulse[n] = s.cur_pulse; // the pulse whose gap is running
        pulses->fsk_f1_est = s.ook_f1;
        pulses->offset = s.offset;
        uint32_t const fn = s.fsk_num < R433_MAX_PULSES ? s.fsk_num : R433_MAX_PULSES;
        fsk_pulses->num_pulses = fn;
        for (uint32_t i = 0; i < fn + 1 && i < R433_MAX_PULSES; ++i) { // complete pairs and the one in the making
            fsk_pulses->pulse[i] = d->h_ring[i].x;
            fsk_pulses->gap[i] = d->h_ring[i].y;
        }
        fsk_pulses->offset = s.fsk_offset;
    }
    pulses->start_ago = fsk_pulses->start_ago = s.start_ago;
    if (s.ret == R433_PKG_OOK) {
        fill_from_record(pulses, d->h_arena.data());
    }
    else if (s.ret == R433_PKG_FSK) {
        fill_from_record(fsk_pulses, d->h_arena.data());
        pulses->end_ago = fsk_pulses->end_ago; // src/pulse_detect.c:250,397
    }
    return s.ret;
}

} // extern "C"

// ---- the FSK sub-detectors on their own: pulse_detect_fsk_classic / _minmax / _wrap_up (reference
// include/pulse_detect_fsk.h:46-75, src/pulse_detect_fsk.c:34-221).  In the reference only pulse_detect_package calls them,
// one sample at a time; here they are the device functions the detection kernel runs (detect_device.hpp: fsk_classic,
// fsk_minmax, the wrap-up at the head of emit_fsk).  Exported for the completeness of the function-level seam: one call
// is one sample through one wavefront, with the caller's pulse list carried to the device and back -- about the contract,
// not about speed. ----

namespace r433 {
namespace {

struct FskSeamIO {
    uint32_t f_run;
    int f_state, f_f1, f_f2, f_vmax, f_vmin, f_skip;
    uint32_t fsk_num;
    uint64_t fsk_offset;
};

__global__ __launch_bounds__(64) void k_fsk_step(int op, int fm, FskSeamIO *io, int2 *ring)
{
    DetLane d;
    det_reset(d);
    d.f_run = io->f_run, d.f_state = io->f_state, d.f_f1 = io->f_f1, d.f_f2 = io->f_f2, d.f_vmax = io->f_vmax, d.f_vmin = io->f_vmin;
    d.f_skip = io->f_skip;
    d.fsk_num = io->fsk_num;
    d.fsk_offset = io->fsk_offset;
    d.fsk_ring = ring;
    d.writer = threadIdx.x == 0;
    d.arena = nullptr;
    d.arena_cap = 0;
    d.cursor = d.ook_base = d.n_pkgs = d.overflow = d.stream = 0;
    __syncthreads(); // every lane has read the state before lane 0 writes it back
    if (op == R433_FSK_CLASSIC) {
        fsk_classic(d, fm);
    }
    else if (op == R433_FSK_MINMAX) {
        fsk_minmax(d, fm);
    }
    else if (d.fsk_num < R433_PD_MAX_PULSES) { // wrap-up, src/pulse_detect_fsk.c:143-156
        d.f_run += 1;
        if (d.f_state == 1) {
            ring_set(d, 2 * d.fsk_num, (int)d.f_run);
            ring_set(d, 2 * d.fsk_num + 1, 0);
        }
        else {
            ring_set(d, 2 * d.fsk_num + 1, (int)d.f_run);
        }
        d.fsk_num += 1;
    }
    if (threadIdx.x == 0) {
        io->f_run = d.f_run, io->f_state = d.f_state, io->f_f1 = d.f_f1, io->f_f2 = d.f_f2, io->f_vmax = d.f_vmax, io->f_vmin = d.f_vmin;
        io->f_skip = d.f_skip;
        io->fsk_num = d.fsk_num;
        io->fsk_offset = d.fsk_offset;
    }
}

struct FskSeamBufs {
    FskSeamIO *d_io = nullptr;
    int2 *d_ring = nullptr;
    FskSeamIO *h_io = nullptr; // pinned
    int2 *h_ring = nullptr;
    std::mutex m;
} g_fsk;

} // namespace
} // namespace r433

extern "C" int r433_fsk_step(int op, r433_fsk_state *s, int fm, r433_pulse_data *fsk_pulses)
{
    if (!s || !fsk_pulses || op < R433_FSK_CLASSIC || op > R433_FSK_WRAP_UP)
        return fail(R433_EINVAL, "r433_fsk_step: bad argument");
    std::lock_guard<std::mutex> g(g_fsk.m);
    if (!g_fsk.d_io) {
        HIP_TRY(hipMalloc((void **)&g_fsk.d_io, sizeof(FskSeamIO)));
        HIP_TRY(hipMalloc((void **)&g_fsk.d_ring, sizeof(int2) * R433_PD_MAX_PULSES));
        HIP_TRY(hipHostMalloc((void **)&g_fsk.h_io, sizeof(FskSeamIO), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void **)&g_fsk.h_ring, sizeof(int2) * R433_PD_MAX_PULSES, hipHostMallocDefault));
    }
    FskSeamIO &io = *g_fsk.h_io;
    io.f_run = s->fsk_pulse_length;
    io.f_state = (int)s->fsk_state;
    io.f_f1 = s->fm_f1_est, io.f_f2 = s->fm_f2_est;
    io.f_vmax = s->var_test_max, io.f_vmin = s->var_test_min;
    io.f_skip = s->skip_samples;
    io.fsk_num = fsk_pulses->num_pulses;
    io.fsk_offset = fsk_pulses->offset;
    for (unsigned k = 0; k < R433_PD_MAX_PULSES; ++k)
        g_fsk.h_ring[k] = make_int2(fsk_pulses->pulse[k], fsk_pulses->gap[k]);
    HIP_TRY(hipMemcpyAsync(g_fsk.d_io, g_fsk.h_io, sizeof(FskSeamIO), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(g_fsk.d_ring, g_fsk.h_ring, sizeof(int2) * R433_PD_MAX_PULSES, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_fsk_step, dim3(1), dim3(64), 0, nullptr, op, fm, g_fsk.d_io, g_fsk.d_ring);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(g_fsk.h_io, g_fsk.d_io, sizeof(FskSeamIO), hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(g_fsk.h_ring, g_fsk.d_ring, sizeof(int2) * R433_PD_MAX_PULSES, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    s->fsk_pulse_length = io.f_run;
    s->fsk_state = (unsigned)io.f_state;
    s->fm_f1_est = io.f_f1, s->fm_f2_est = io.f_f2;
    s->var_test_max = (int16_t)io.f_vmax, s->var_test_min = (int16_t)io.f_vmin;
    s->skip_samples = io.f_skip;
    fsk_pulses->num_pulses = io.fsk_num;
    fsk_pulses->offset = io.fsk_offset;
    for (unsigned k = 0; k < R433_PD_MAX_PULSES; ++k) {
        fsk_pulses->pulse[k] = g_fsk.h_ring[k].x;
        fsk_pulses->gap[k] = g_fsk.h_ring[k].y;
    }
    return 0;
}
