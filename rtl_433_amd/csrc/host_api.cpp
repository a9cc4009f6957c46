// host_api.cpp -- the C ABI of librtl433hip.so (include/r433_hip.h): buffer management, kernel
// sequencing, record mirroring and the host-side decoder dispatch that mirrors the reference's
// run_ook_demods / run_fsk_demods + account_event (src/r_api.c:438-550, src/pulse_slicer.c:26-66).
//
// There is no CPU implementation of the hot path in here: if HIP is unusable every compute entry
// point fails with R433_ENODEV.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "r433_hip.h"
#include "r433_internal.hpp"

using namespace r433;

namespace {

thread_local std::string g_err;

int fail(int code, char const *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? R433_ENODEV : R433_EHIP, "%s: %s",   \
                    #expr, hipGetErrorString(e_));                                                                     \
    } while (0)

// ---- host-side scalar math the reference also does on the host (same libm) ----

// DB_TO_AMP / DB_TO_MAG / DB_TO_AMP_F / DB_TO_MAG_F, reference include/baseband.h:44-47;
// pulse_detect_set_levels, src/pulse_detect.c:86-105; OOK_MAX_HIGH_LEVEL, :24
void levels_from_db(DetCfg &c, int use_mag, float fixed_db, float min_db, float ratio_db)
{
    if (use_mag) {
        c.fixed_high = fixed_db < 0.0 ? (int)powf(10, (fixed_db + 84.2884f) / 20.0f) : 0;
        c.min_high = (int)powf(10, (min_db + 84.2884f) / 20.0f);
        c.ratio = (int)(0.5 + powf(10, ratio_db / 20.0f));
    }
    else {
        c.fixed_high = fixed_db < 0.0 ? (int)powf(10, (fixed_db + 42.1442f) / 10.0f) : 0;
        c.min_high = (int)powf(10, (min_db + 42.1442f) / 10.0f);
        c.ratio = (int)(0.5 + powf(10, ratio_db / 10.0f));
    }
    c.max_high = (int)powf(10, (0 + 42.1442f) / 10.0f);
}

// coefficient derivation of baseband_demod_FM(_cs16), reference src/baseband.c:217-232, 310-325
void fm_coeffs(float low_pass, uint32_t rate, int &a16, int &b16, long long &a32, long long &b32)
{
    if (low_pass > 1e4f)
        low_pass = low_pass / rate;
    else if (low_pass >= 1.0f)
        low_pass = 1e6f / low_pass / rate;
    double ita = 1.0 / tan(M_PI_2 * low_pass);
    double g16 = 1.0 / (1.0 + ita) / 2;
    double g32 = 1.0 / (1.0 + ita);
    a16 = (int)((ita - 1.0) * g16 * 32768);
    b16 = (int)(g16 * 32768);
    a32 = (int)((ita - 1.0) * g32 * 1073741824);
    b32 = (int)(g32 * 1073741824);
}

// integer timing of a device at a sample rate, reference src/pulse_slicer.c:70-99 (same float ops)
DevRow resolve_timing(r433_dev_timing const &d, uint32_t rate, int orig)
{
    DevRow r;
    memset(&r, 0, sizeof(r));
    volatile float us = rate / 1.0e6f;
    r.modulation = (int)d.modulation;
    r.orig = orig;
    r.is_fsk = d.modulation >= 16;
    volatile float v;
    v = d.short_width * us;
    r.s_short = (int)v;
    v = d.long_width * us;
    r.s_long = (int)v;
    v = d.reset_limit * us;
    r.s_reset = (int)v;
    v = d.gap_limit * us;
    r.s_gap = (int)v;
    v = d.sync_width * us;
    r.s_sync = (int)v;
    v = d.tolerance * us;
    r.s_tol = (int)v;
    bool ok = !((d.short_width > 0 && r.s_short <= 0) || (d.long_width > 0 && r.s_long <= 0)
            || (d.reset_limit > 0 && r.s_reset <= 0));
    if (d.modulation != 13) // pulse_slicer_rzi only checks short/long/reset, src/pulse_slicer.c:876-882
        ok = ok
                && !((d.gap_limit > 0 && r.s_gap <= 0) || (d.sync_width > 0 && r.s_sync <= 0)
                        || (d.tolerance > 0 && r.s_tol <= 0));
    volatile float ps = d.short_width * us, pl = d.long_width * us;
    r.f_short = d.short_width > 0.0f ? 1.0f / ps : 0;
    r.f_long = d.long_width > 0.0f ? 1.0f / pl : 0;
    switch (d.modulation) {
    case 3: case 4: case 5: case 6: case 8: case 9: case 10: case 11: case 12: case 13: case 16: case 17: case 18:
        break;
    default:
        ok = false; // "Unknown modulation" in the reference's switch
    }
    r.valid = ok ? 1 : 0;
    return r;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0; // elements
    int ensure(size_t n)
    {
        if (n <= cap)
            return 0;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 16;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess)
            return fail(R433_ENOMEM, "hipMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T> struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap)
            return 0;
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 16;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess)
            return fail(R433_ENOMEM, "hipHostMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};


// Persistent host workers for the decoder dispatch (spawning 32 threads per batch costs more than
// dispatching a small batch).
class Pool {
  public:
    ~Pool() { stop(); }
    void run(unsigned n, std::function<void(unsigned)> const &job)
    {
        if (n <= 1) {
            job(0);
            return;
        }
        grow(n - 1);
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &job;
            want_ = n - 1;
            pending_ = n - 1;
            ++epoch_;
        }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void grow(unsigned n)
    {
        while (threads_.size() < n) {
            unsigned id = (unsigned)threads_.size();
            threads_.emplace_back([this, id] { loop(id); });
        }
    }
    void loop(unsigned id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(unsigned)> const *job = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return quit_ || (epoch_ != seen && id < want_); });
                if (quit_)
                    return;
                seen = epoch_;
                job = job_;
            }
            (*job)(id + 1);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0)
                    done_.notify_all();
            }
        }
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_)
            t.join();
        threads_.clear();
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    std::function<void(unsigned)> const *job_ = nullptr;
    unsigned want_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool quit_ = false;
};

// The checksum plugin accumulates per thread and publishes once per dispatch: a shared counter hit by
// every bitbuffer from 32 threads is a cache-line ping-pong that costs more than the decoding.
struct DigestLocal {
    r433_digest_ctx *ctx = nullptr;
    uint64_t sum = 0, events = 0;
};
thread_local DigestLocal g_digest;

void digest_publish()
{
    if (g_digest.ctx && g_digest.events) {
        __atomic_fetch_add(&g_digest.ctx->sum, g_digest.sum, __ATOMIC_RELAXED);
        __atomic_fetch_add(&g_digest.ctx->events, g_digest.events, __ATOMIC_RELAXED);
    }
    g_digest = DigestLocal();
}

} // namespace

struct r433_batch;
static hipError_t stream_wait(r433_batch *b, hipStream_t st);

struct r433_batch {
    r433_flow_cfg cfg;
    DetCfg det;
    int a16 = 0, b16 = 0;
    long long a32 = 0, b32 = 0;
    std::vector<r433_dev_timing> timing; // registration order
    std::vector<DevRow> rows;            // sorted for the fan-out
    std::vector<uint32_t> prio_levels;   // distinct priorities ascending

    DevBuf<DevRow> d_rows;
    DevBuf<uint8_t> d_arena;
    DevBuf<int2> d_ring;
    DevBuf<StreamState> d_state;
    DevBuf<uint32_t> d_frame_sums, d_stream_bytes, d_pkg_base, d_scal;
    DevBuf<int> d_frame_min_high;
    std::vector<int> h_frame_min_high;
    // split captures (r433_batch_set_split)
    uint32_t split_samples = R433_SPLIT_AUTO;
    DevBuf<uint32_t> d_tile_max, d_order;
    DevBuf<SegDesc> d_segs;
    PinBuf<uint32_t> h_tile_max;
    PinBuf<StreamState> h_state;
    uint32_t last_segments = 0, last_redone = 0;
    DevBuf<uint32_t> d_dir_stream, d_dir_off, d_rec_bytes, d_rec_off, d_sizes, d_pkg_bytes, d_pkg_off;
    DevBuf<uint8_t> d_pkg_blob, d_events, d_stage, d_converted;
    DevBuf<r433_analysis> d_analysis;
    std::vector<uint32_t> conv_bytes;
    PinBuf<uint32_t> h_scal, h_frame_sums;
    PinBuf<uint8_t> h_pkg_blob, h_events, h_arena_stage;
    PinBuf<uint32_t> h_pkg_off, h_rec_off; // per package: byte offset of its first event / of its record

    uint32_t arena_stride = 0;
    uint32_t frames_cap = 0;
    uint32_t n_streams = 0;
    uint32_t n_pkgs = 0, n_events = 0;
    size_t pkg_bytes = 0, evt_bytes = 0;
    bool events_counted = false;

    void *tap_env = nullptr, *tap_am = nullptr, *tap_fm = nullptr;
    uint64_t tap_stride = 0;

    hipEvent_t sync_ev = nullptr; // blocking (sleeping) wait: host threads of other pipeline stages need the cores
    bool profiling = false;
    hipEvent_t ev[8] = {};
    bool ev_made = false;
    r433_batch_timing last_timing = {};

    // dispatch scratch
    r433_bitbuffer *bits = nullptr;
    r433_pulse_data *pulses = nullptr;
    Pool pool;
};

static hipError_t stream_wait(r433_batch *b, hipStream_t st)
{
    if (!b->sync_ev) {
        hipError_t e = hipEventCreateWithFlags(&b->sync_ev, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess)
            return e;
    }
    hipError_t e = hipEventRecord(b->sync_ev, st);
    return e != hipSuccess ? e : hipEventSynchronize(b->sync_ev);
}

// ---- r433_batch_run, stage by stage --------------------------------------------------------------
namespace {

// What one r433_batch_run call carries from stage to stage.
struct RunCtx {
    r433_batch *b;
    hipStream_t st;
    uint32_t ss;                  // bytes per sample: 2 = cu8, 4 = cs16
    void const *d_iq;             // the captures as the detector sees them (after input conversion)
    uint64_t stride_bytes;
    uint32_t const *stream_bytes; // host, per capture; null = every capture fills the stride
    uint32_t n_streams;
    uint32_t max_samples = 0, frames_cap = 0, want_stride = 0;
    int const *d_min_high = nullptr; // per-frame detection level (-Y autolevel), device
    // plan: one wavefront per capture, or several per long capture (speculative cuts)
    bool split = false;
    std::vector<SegDesc> segs;
    std::vector<uint32_t> seg_first_of; // segs of capture c: [seg_first_of[c], seg_first_of[c+1])
    std::vector<uint32_t> cap_n;        // samples per capture
    uint32_t max_seg_samples = 0, n_planned = 0, n_slots = 0;
    // detection result
    uint32_t n_order = 0;              // slots that make up the result, in capture order
    uint32_t const *d_order = nullptr; // null = slot i is capture i
    std::vector<uint32_t> order;
    uint32_t total_pkgs = 0;

    uint32_t const *d_lens() const { return stream_bytes ? b->d_stream_bytes.p : nullptr; }
    int env_kind() const { return ss == 4 ? ENV_MAG_CS16 : b->cfg.use_mag_est ? ENV_MAG_CU8 : ENV_AMP_CU8; }
};


// cs8 / cf32 input -> cu8 / cs16 in an internal buffer
int run_convert_input(RunCtx &r)
{
    r433_batch *const b = r.b;
    if (b->cfg.input_format == R433_IN_NATIVE)
        return 0;
    // The reference converts these formats while it loads a file (src/rtl_433.c:1811-1834): one HBM-bound
    // map into an internal buffer, then everything below sees cu8 / cs16 like the reference's flow does.
    uint32_t const shrink = b->cfg.input_format == R433_IN_CF32 ? 2 : 1; // 8 B -> 4 B per sample
    uint64_t in_max = 0;
    b->conv_bytes.resize(r.n_streams);
    for (uint32_t i = 0; i < r.n_streams; ++i) {
        uint64_t const nb = r.stream_bytes ? r.stream_bytes[i] : r.stride_bytes;
        if (nb > r.stride_bytes)
            return fail(R433_EINVAL, "capture %u is longer than the stride", i);
        in_max = std::max(in_max, nb);
        b->conv_bytes[i] = (uint32_t)(nb / (shrink * r.ss) * r.ss); // whole samples
    }
    uint64_t const out_stride = ((in_max / shrink) + 15) & ~15ull;
    if (int rc = b->d_converted.ensure((size_t)r.n_streams * out_stride + 16))
        return rc;
    launch_convert((int)b->cfg.input_format, r.d_iq, r.stride_bytes, b->d_converted.p, out_stride, in_max, r.n_streams, r.st);
    HIP_TRY(hipGetLastError());
    r.d_iq = b->d_converted.p;
    r.stride_bytes = out_stride;
    r.stream_bytes = b->conv_bytes.data();
    return 0;
}

// capture lengths to the device, per-capture scratch, first guess of the package arena
int run_size_buffers(RunCtx &r)
{
    r433_batch *const b = r.b;
    uint32_t max_bytes = 0;
    if (r.stream_bytes) {
        for (uint32_t i = 0; i < r.n_streams; ++i) {
            if (r.stream_bytes[i] > r.stride_bytes)
                return fail(R433_EINVAL, "capture %u is longer than the stride", i);
            max_bytes = std::max(max_bytes, r.stream_bytes[i]);
        }
    }
    else {
        max_bytes = (uint32_t)r.stride_bytes;
    }
    r.max_samples = max_bytes / r.ss;
    r.frames_cap = r.max_samples / b->cfg.frame_samples + 2;

    int rc;
    if ((rc = b->d_ring.ensure((size_t)r.n_streams * R433_PD_MAX_PULSES)) || (rc = b->d_state.ensure(r.n_streams))
            || (rc = b->d_frame_sums.ensure((size_t)r.n_streams * r.frames_cap)) || (rc = b->d_pkg_base.ensure(r.n_streams)))
        return rc;
    if (r.stream_bytes) {
        if ((rc = b->d_stream_bytes.ensure(r.n_streams)))
            return rc;
        HIP_TRY(hipMemcpyAsync(b->d_stream_bytes.p, r.stream_bytes, r.n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, r.st));
    }
    b->frames_cap = r.frames_cap;
    b->n_streams = r.n_streams;

    // arena: worst case is one (pulse, gap) pair per 20 samples plus headers; start at ~1 B/sample
    r.want_stride = std::max<uint32_t>(16384u, ((r.max_samples + 4096u) + 15u) & ~15u);
    if (b->arena_stride < r.want_stride)
        b->arena_stride = r.want_stride;

    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[0], r.st));
    return 0;
}

// -Y autolevel (reference src/r_flow.c:166-186): the detection level of a frame follows the noise
// estimate, which follows the mean envelope of the frames so far -- a short recurrence in host
// floats (same libm as the reference) over per-frame sums that one HBM-bound pass provides.
int run_autolevel(RunCtx &r)
{
    r433_batch *const b = r.b;
    if (!(b->cfg.auto_level > 0))
        return 0;
    int rc;
    if ((rc = b->d_frame_min_high.ensure((size_t)r.n_streams * r.frames_cap)) || (rc = b->h_frame_sums.ensure((size_t)r.n_streams * r.frames_cap)))
        return rc;
    launch_frame_sums(r.env_kind(), r.d_iq, r.stride_bytes, r.d_lens(), (uint32_t)r.stride_bytes, r.n_streams,
            b->cfg.frame_samples, r.frames_cap, b->d_frame_sums.p, r.st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_frame_sums.p, b->d_frame_sums.p, (size_t)r.n_streams * r.frames_cap * sizeof(uint32_t),
            hipMemcpyDeviceToHost, r.st));
    HIP_TRY(stream_wait(b, r.st));
    b->h_frame_min_high.assign((size_t)r.n_streams * r.frames_cap, b->det.min_high);
    int const is_mag = r.ss == 4 || b->cfg.use_mag_est;
    for (uint32_t s = 0; s < r.n_streams; ++s) {
        uint32_t const n = (r.stream_bytes ? r.stream_bytes[s] : (uint32_t)r.stride_bytes) / r.ss;
        float noise_level = 0.0f, min_level_auto = 0.0f;
        DetCfg lv = b->det;
        for (uint32_t f = 0; f < r.frames_cap; ++f) {
            uint64_t const start = (uint64_t)f * b->cfg.frame_samples;
            if (start < n) {
                uint32_t const cnt = (uint32_t)std::min<uint64_t>(b->cfg.frame_samples, n - start);
                float const avg_db = r433_level_db(b->h_frame_sums.p[(size_t)s * r.frames_cap + f], cnt, is_mag);
                if (min_level_auto == 0.0f)
                    min_level_auto = b->cfg.min_level_db;
                if (noise_level == 0.0f)
                    noise_level = min_level_auto - 3.0f;
                if (avg_db < noise_level + 3.0f) {
                    noise_level = (noise_level * 7 + avg_db) / 8;
                    if (noise_level < b->cfg.min_level_db - 3.0f && fabsf(min_level_auto - noise_level - 3.0f) > 1.0f) {
                        min_level_auto = noise_level + 3.0f;
                        levels_from_db(lv, (int)b->cfg.use_mag_est, b->cfg.level_limit_db, min_level_auto, b->cfg.min_snr_db);
                    }
                }
                else {
                    noise_level = (noise_level * 31 + avg_db) / 32;
                }
            }
            b->h_frame_min_high[(size_t)s * r.frames_cap + f] = lv.min_high;
        }
    }
    HIP_TRY(hipMemcpyAsync(b->d_frame_min_high.p, b->h_frame_min_high.data(), b->h_frame_min_high.size() * sizeof(int),
            hipMemcpyHostToDevice, r.st));
    r.d_min_high = b->d_frame_min_high.p;
    return 0;
}

// Where long captures may be cut: at the end of 12.5 ms of quiet, about every split_samples samples.
int run_plan_split(RunCtx &r, uint32_t split_samples)
{
    r433_batch *const b = r.b;
    int rc;
    constexpr uint32_t kTileS = 2048;
    uint32_t const tiles_cap = r.max_samples / kTileS + 1;
    if ((rc = b->d_tile_max.ensure((size_t)r.n_streams * tiles_cap)) || (rc = b->h_tile_max.ensure((size_t)r.n_streams * tiles_cap)))
        return rc;
    launch_tile_max(r.env_kind(), r.d_iq, r.stride_bytes, r.d_lens(), (uint32_t)r.stride_bytes, r.n_streams,
            tiles_cap, b->d_tile_max.p, r.st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_tile_max.p, b->d_tile_max.p, (size_t)r.n_streams * tiles_cap * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
    HIP_TRY(stream_wait(b, r.st));
    // A tile is quiet when it carries no more energy than the noise floor: mean envelope at most 1.5x
    // the capture's median tile, or -- for captures that are mostly signal -- below half the
    // falling-edge level of the lowest threshold the detector can have (pulse_detect.c:300-304).
    int thr = (-1 + std::min(b->det.min_high, b->det.max_high)) / 2;
    if (b->det.fixed_high)
        thr = b->det.fixed_high;
    uint64_t const abs_quiet = (uint64_t)std::max(1, (thr - thr / 8) / 2) * 2048u;
    bool const blind = getenv("R433_SPLIT_BLIND") != nullptr; // tests: cut anywhere, let the verification sort it out
    uint32_t const seg_len = (split_samples + kTileS - 1) / kTileS * kTileS;
    // a package stays open until its last gap exceeds 10 pulse widths and 10 ms (pulse_detect.c:446-450):
    // ask for 12.5 ms of quiet before a cut (pulses up to 1.25 ms; the stitch catches the rest)
    uint32_t const quiet_tiles = std::max<uint32_t>(2u, (b->cfg.samp_rate / 80u + kTileS - 1) / kTileS);
    r.max_seg_samples = 0;
    for (uint32_t c = 0; c < r.n_streams; ++c) {
        r.seg_first_of[c] = (uint32_t)r.segs.size();
        uint32_t const n = r.cap_n[c];
        uint32_t const *tm = b->h_tile_max.p + (size_t)c * tiles_cap;
        uint64_t quiet_below = abs_quiet; // on tile sums
        if (n >= 2 * kTileS) {
            std::vector<uint32_t> med(tm, tm + n / kTileS);
            std::nth_element(med.begin(), med.begin() + med.size() / 2, med.end());
            quiet_below = std::max<uint64_t>(abs_quiet, (uint64_t)med[med.size() / 2] * 3 / 2);
        }
        std::vector<uint32_t> cuts;
        uint32_t pos = seg_len;
        while (n > seg_len && pos + seg_len / 2 < n) {
            uint32_t cut = 0;
            for (uint32_t P = pos; P < std::min(n, pos + seg_len) && P + kTileS <= n; P += kTileS) {
                bool quiet = P / kTileS >= quiet_tiles;
                for (uint32_t q = 1; quiet && q <= quiet_tiles; ++q)
                    quiet = tm[P / kTileS - q] < quiet_below;
                if (blind || quiet) {
                    cut = P;
                    break;
                }
            }
            if (cut) {
                cuts.push_back(cut);
                pos = cut + seg_len;
            }
            else {
                pos += seg_len;
            }
        }
        uint32_t from = 0;
        for (size_t k = 0; k <= cuts.size(); ++k) {
            uint32_t const to = k < cuts.size() ? cuts[k] : n;
            uint32_t const last = k == cuts.size() ? SEG_LAST : 0u;
            if (k == 0) {
                r.segs.push_back(SegDesc{c, 0u, to, SEG_FIRST | SEG_PRIMARY | last});
            }
            else { // both parities of the noise floor
                r.segs.push_back(SegDesc{c, from, to, SEG_PRIMARY | last});
                r.segs.push_back(SegDesc{c, from, to, SEG_ODD | last});
            }
            r.max_seg_samples = std::max(r.max_seg_samples, to - from + kTileS);
            from = to;
        }
    }
    r.seg_first_of[r.n_streams] = (uint32_t)r.segs.size();
    return 0;
}

// one wavefront per capture, or several per long capture
int run_plan(RunCtx &r)
{
    r433_batch *const b = r.b;
    int rc;
    r.seg_first_of.assign(r.n_streams + 1, 0);
    r.cap_n.resize(r.n_streams);
    for (uint32_t c = 0; c < r.n_streams; ++c)
        r.cap_n[c] = (r.stream_bytes ? r.stream_bytes[c] : (uint32_t)r.stride_bytes) / r.ss;
    // automatic: only where one wavefront per capture would leave the chip empty -- few, long captures.
    // Aim at ~4096 segments, at least 32 Ki samples each.
    uint32_t split_samples = b->split_samples;
    if (split_samples == R433_SPLIT_AUTO) {
        uint64_t total = 0;
        for (uint32_t c = 0; c < r.n_streams; ++c)
            total += r.cap_n[c];
        split_samples = (r.n_streams <= 64 && r.max_samples >= (1u << 20)) ? (uint32_t)std::max<uint64_t>(32768, total / 4096) : 0u;
    }
    r.split = split_samples > 0;
    r.max_seg_samples = r.max_samples;
    r.n_order = r.n_streams;
    if (r.split && (rc = run_plan_split(r, split_samples)))
        return rc;
    r.n_planned = r.split ? (uint32_t)r.segs.size() : r.n_streams;
    r.n_slots = r.split ? 3 * r.n_planned : r.n_streams; // + re-run slots for cuts that have to be dropped
    if (r.split && b->arena_stride == r.want_stride) // sized for whole captures above: segments need less
        b->arena_stride = std::max<uint32_t>(16384u, ((r.max_seg_samples + 4096u) + 15u) & ~15u);
    if ((rc = b->d_ring.ensure((size_t)r.n_slots * R433_PD_MAX_PULSES)) || (rc = b->d_state.ensure(r.n_slots))
            || (rc = b->d_pkg_base.ensure(r.n_slots)) || (rc = b->h_state.ensure(r.n_slots)) || (rc = b->d_order.ensure(r.n_slots))
            || (rc = b->d_segs.ensure(r.n_slots)))
        return rc;
    return 0;
}

// what every launch of the detection kernel in this run shares
StreamParams stream_params(RunCtx const &r)
{
    r433_batch *const b = r.b;
    StreamParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.iq = (uint8_t const *)r.d_iq;
    sp.stride_bytes = r.stride_bytes;
    sp.stream_bytes = r.d_lens();
    sp.uniform_bytes = (uint32_t)r.stride_bytes;
    sp.n_streams = r.n_planned;
    sp.frame_samples = b->cfg.frame_samples;
    sp.flags = 0;
    if (char const *dbg = getenv("R433_DEBUG_FLAGS")) // phase timing experiments only (results are then incomplete)
        sp.flags |= (uint32_t)strtoul(dbg, nullptr, 0) & (RUN_DBG_SKIP_DETECT | RUN_DBG_SKIP_FILTERS | RUN_DBG_TIMING);
    sp.det = b->det;
    sp.use_mag = (int)b->cfg.use_mag_est;
    sp.enable_fm = (int)b->cfg.enable_fm;
    sp.a16 = b->a16;
    sp.b16 = b->b16;
    sp.a32 = b->a32;
    sp.b32 = b->b32;
    sp.arena = b->d_arena.p;
    sp.arena_stride = b->arena_stride;
    sp.fsk_ring = b->d_ring.p;
    sp.state = b->d_state.p;
    sp.frame_sums = b->d_frame_sums.p;
    sp.frames_cap = r.frames_cap;
    sp.frame_min_high = r.d_min_high;
    sp.tap_env = (uint16_t *)b->tap_env;
    sp.tap_am = (int16_t *)b->tap_am;
    sp.tap_fm = (int16_t *)b->tap_fm;
    sp.tap_stride = b->tap_stride;
    return sp;
}

// Stitch.  Per capture an ordered list of pieces; every piece but the first exists in two
// parity variants (two slots).  Walk the pieces in order; at each cut keep the variant that
// assumed exactly the floor (and the level estimate that an idle step leaves behind: a spurious
// short pulse returns to idle without one) the piece before it really ended with, and require
// that piece to have ended idle with the lead-in saturated -- that is the detector's whole
// state between packages (everything else is reset when a pulse starts).  A cut that does not verify is
// dropped: the piece before it is run again through to the end of the next piece, and the walk resumes from there.  Every
// round removes at least one cut per capture that still has a problem, so this terminates.
int run_stitch(RunCtx &r, StreamParams const &sp)
{
    r433_batch *const b = r.b;
    int rc;
    struct Piece {
        SegDesc seg;      // flags without SEG_ODD / SEG_PRIMARY
        uint32_t slot[2]; // even / odd parity variant (the first piece of a capture: slot[0] only)
    };
    std::vector<std::vector<Piece>> pieces(r.n_streams);
    for (uint32_t c = 0; c < r.n_streams; ++c)
        for (uint32_t k = r.seg_first_of[c]; k < r.seg_first_of[c + 1]; k += (k == r.seg_first_of[c] ? 1 : 2)) {
            Piece pc;
            pc.seg = r.segs[k];
            pc.seg.flags &= ~(uint32_t)(SEG_ODD | SEG_PRIMARY);
            pc.slot[0] = k;
            pc.slot[1] = k == r.seg_first_of[c] ? k : k + 1;
            pieces[c].push_back(pc);
        }
    uint32_t n_have = r.n_planned; // slots whose state is on the host
    std::vector<SegDesc> slot_seg(r.segs), launch_list;
    auto new_slot = [&](SegDesc const &d) {
        launch_list.push_back(d);
        slot_seg.push_back(d);
        return n_have + (uint32_t)launch_list.size() - 1;
    };
    auto run_launch_list = [&]() -> int {
        if (launch_list.empty())
            return 0;
        if (n_have + launch_list.size() > r.n_slots)
            return fail(R433_EHIP, "r.split bookkeeping ran out of slots");
        b->last_redone += (uint32_t)launch_list.size();
        HIP_TRY(hipMemcpyAsync(b->d_segs.p + n_have, launch_list.data(), launch_list.size() * sizeof(SegDesc), hipMemcpyHostToDevice, r.st));
        StreamParams sr = sp;
        sr.n_streams = (uint32_t)launch_list.size();
        sr.segs = b->d_segs.p + n_have;
        sr.arena = b->d_arena.p + (size_t)n_have * b->arena_stride;
        sr.fsk_ring = b->d_ring.p + (size_t)n_have * R433_PD_MAX_PULSES;
        sr.state = b->d_state.p + n_have;
        launch_stream(sr, r.ss, r.st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(b->h_state.p + n_have, b->d_state.p + n_have, launch_list.size() * sizeof(StreamState), hipMemcpyDeviceToHost, r.st));
        HIP_TRY(stream_wait(b, r.st));
        n_have += (uint32_t)launch_list.size();
        launch_list.clear();
        return 0;
    };
    HIP_TRY(hipMemcpyAsync(b->h_state.p, b->d_state.p, (size_t)r.n_planned * sizeof(StreamState), hipMemcpyDeviceToHost, r.st));
    HIP_TRY(stream_wait(b, r.st));
    // (1) cuts neither variant could start from (no provable filter carry or floor: digital
    // silence does that) are known now, all at once: merge across them in one extra launch
    for (uint32_t c = 0; c < r.n_streams; ++c) {
        std::vector<Piece> merged;
        for (Piece const &pc : pieces[c]) {
            bool const unstartable = !merged.empty() && b->h_state.p[pc.slot[0]].seg_fail && b->h_state.p[pc.slot[1]].seg_fail;
            if (!unstartable) {
                merged.push_back(pc);
                continue;
            }
            Piece &m = merged.back();
            m.seg.end = pc.seg.end;
            m.seg.flags |= pc.seg.flags & SEG_LAST;
            m.slot[0] = m.slot[1] = UINT32_MAX; // to be run again
        }
        for (Piece &m : merged)
            if (m.slot[0] == UINT32_MAX) {
                SegDesc d = m.seg;
                d.flags |= SEG_PRIMARY;
                m.slot[0] = new_slot(d);
                m.slot[1] = m.slot[0];
                if (!(d.flags & SEG_FIRST)) {
                    d.flags = (d.flags & ~(uint32_t)SEG_PRIMARY) | SEG_ODD;
                    m.slot[1] = new_slot(d);
                }
            }
        pieces[c].swap(merged);
    }
    if ((rc = run_launch_list()))
        return rc;
    // (2) the walk proper
    std::vector<std::vector<uint32_t>> chosen(r.n_streams);
    std::vector<size_t> at(r.n_streams, 1);
    std::vector<uint32_t> dropped(r.n_streams, 0);
    for (uint32_t c = 0; c < r.n_streams; ++c)
        chosen[c].push_back(pieces[c][0].slot[0]);
    for (;;) {
        for (uint32_t c = 0; c < r.n_streams; ++c) {
            std::vector<Piece> &pcs = pieces[c];
            while (at[c] < pcs.size()) {
                uint32_t const cur = chosen[c].back();
                StreamState const &P = b->h_state.p[cur];
                Piece const &nx = pcs[at[c]];
                uint32_t pick = UINT32_MAX;
                if (P.seg_end_state == ST_IDLE && P.seg_end_lead == 1025)
                    for (int v = 0; v < 2; ++v)
                        if (!b->h_state.p[nx.slot[v]].seg_fail && b->h_state.p[nx.slot[v]].seg_init_low == P.seg_end_low
                                && b->h_state.p[nx.slot[v]].seg_init_high == P.seg_end_high)
                            pick = nx.slot[v];
                if (pick != UINT32_MAX) {
                    chosen[c].push_back(pick);
                    at[c] += 1;
                    dropped[c] = 0;
                    continue;
                }
                // drop this cut: the standing piece continues through the next one.  (Three cuts in a
                // row that fail are not worth a fourth try: the piece then runs to the capture's end.)
                bool const give_up = ++dropped[c] >= 3;
                if (getenv("R433_SPLIT_DEBUG"))
                    fprintf(stderr, "r.split: capture %u cut at %u dropped (end state %d, floor %d vs %d/%d, fail %d/%d)\n", c, nx.seg.start,
                            P.seg_end_state, P.seg_end_low, b->h_state.p[nx.slot[0]].seg_init_low, b->h_state.p[nx.slot[1]].seg_init_low,
                            b->h_state.p[nx.slot[0]].seg_fail, b->h_state.p[nx.slot[1]].seg_fail);
                SegDesc d = slot_seg[cur];
                d.end = give_up ? r.cap_n[c] : nx.seg.end;
                d.flags = (d.flags & ~(uint32_t)SEG_LAST) | (give_up ? (uint32_t)SEG_LAST : (nx.seg.flags & SEG_LAST));
                chosen[c].back() = new_slot(d);
                at[c] = give_up ? pcs.size() : at[c] + 1;
                break; // its end state is not known yet: resume in the next round
            }
        }
        if (launch_list.empty())
            break;
        if ((rc = run_launch_list()))
            return rc;
    }
    r.order.clear();
    for (uint32_t c = 0; c < r.n_streams; ++c)
        r.order.insert(r.order.end(), chosen[c].begin(), chosen[c].end());
    r.n_order = (uint32_t)r.order.size();
    HIP_TRY(hipMemcpyAsync(b->d_order.p, r.order.data(), r.order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, r.st));
    r.d_order = b->d_order.p;
    return 0;
}

// envelope, filters, pulse detection: packages per slot in the arena (grown and repeated if it overflows)
int run_detect(RunCtx &r)
{
    r433_batch *const b = r.b;
    int rc;
    for (int attempt = 0;; ++attempt) {
        if ((rc = b->d_arena.ensure((size_t)r.n_slots * b->arena_stride)))
            return rc;
        StreamParams sp = stream_params(r);
        if (r.split) { // segments overlap in frames and may be re-run: the sums come from their own HBM-bound pass
            sp.frame_sums = nullptr;
            launch_frame_sums(r.env_kind(), r.d_iq, r.stride_bytes, r.d_lens(), (uint32_t)r.stride_bytes, r.n_streams,
                    b->cfg.frame_samples, r.frames_cap, b->d_frame_sums.p, r.st);
        }
        else {
            HIP_TRY(hipMemsetAsync(b->d_frame_sums.p, 0, (size_t)r.n_streams * r.frames_cap * sizeof(uint32_t), r.st));
        }
        r.d_order = nullptr;
        b->last_segments = r.n_planned;
        b->last_redone = 0;
        if (r.split) {
            HIP_TRY(hipMemcpyAsync(b->d_segs.p, r.segs.data(), r.segs.size() * sizeof(SegDesc), hipMemcpyHostToDevice, r.st));
            sp.segs = b->d_segs.p;
        }
        launch_stream(sp, r.ss, r.st);
        HIP_TRY(hipGetLastError());
        if (r.split && (rc = run_stitch(r, sp)))
            return rc;
        if (b->profiling && attempt == 0)
            HIP_TRY(hipEventRecord(b->ev[1], r.st));
        launch_pkg_scan(b->d_state.p, r.d_order, r.n_order, b->d_pkg_base.p, b->d_scal.p, r.st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(b->h_scal.p, b->d_scal.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
        HIP_TRY(stream_wait(b, r.st));
        r.total_pkgs = b->h_scal.p[0];
        if (!b->h_scal.p[1])
            break;
        if (attempt >= 6 || b->arena_stride > (1u << 28))
            return fail(R433_EOVERFLOW, "package arena overflow (stride %u)", b->arena_stride);
        b->arena_stride *= 4;
    }
    return 0;
}

// package directory, slicer fan-out, results to the host mirrors
int run_slice_and_mirror(RunCtx &r)
{
    r433_batch *const b = r.b;
    int rc;
    b->n_pkgs = r.total_pkgs;
    b->n_events = 0;
    b->pkg_bytes = b->evt_bytes = 0;
    uint32_t const n_devs = (uint32_t)b->timing.size();
    uint32_t const max_pkgs = std::max<uint32_t>(r.total_pkgs, 1);

    if ((rc = b->d_dir_stream.ensure(max_pkgs)) || (rc = b->d_dir_off.ensure(max_pkgs))
            || (rc = b->d_rec_bytes.ensure(max_pkgs)) || (rc = b->d_rec_off.ensure(max_pkgs))
            || (rc = b->d_pkg_bytes.ensure(max_pkgs)) || (rc = b->d_pkg_off.ensure(max_pkgs))
            || (rc = b->d_sizes.ensure((size_t)max_pkgs * std::max<uint32_t>(n_devs, 1))))
        return rc;

    launch_directory(b->d_arena.p, b->arena_stride, b->d_state.p, r.split ? b->d_order.p : nullptr, r.n_order, b->d_pkg_base.p,
            b->d_dir_stream.p, b->d_dir_off.p, b->d_rec_bytes.p, max_pkgs, r.st);
    launch_scan_u32(b->d_rec_bytes.p, b->d_rec_off.p, b->d_scal.p, max_pkgs, b->d_scal.p + 2, r.st);
    HIP_TRY(hipGetLastError());
    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[2], r.st));

    SliceParams lp;
    memset(&lp, 0, sizeof(lp));
    lp.arena = b->d_arena.p;
    lp.arena_stride = b->arena_stride;
    lp.dir_stream = b->d_dir_stream.p;
    lp.dir_off = b->d_dir_off.p;
    lp.n_pkgs = b->d_scal.p;
    lp.devs = b->d_rows.p;
    lp.n_rows = (uint32_t)b->rows.size();
    lp.n_devs = n_devs;
    lp.sizes = b->d_sizes.p;
    lp.pkg_bytes = b->d_pkg_bytes.p;
    lp.pkg_off = b->d_pkg_off.p;
    lp.max_pkgs = max_pkgs;
    if (n_devs && r.total_pkgs) {
        // One slicing pass into staging slots when they fit.  A default device set yields ~135 B per
        // (package, device) on average but the heavy PCM rows reach a few KB, and those are exactly the
        // slow lanes, so the slot is made as large as the arena budget allows (up to 4 KB); below 512 B
        // the classic count + write pair runs instead.
        constexpr size_t kStageMax = (size_t)6 << 30;
        uint32_t stage_cap = 4096;
        while (stage_cap >= 512 && (size_t)r.total_pkgs * b->rows.size() * stage_cap > kStageMax)
            stage_cap >>= 1;
        if (stage_cap >= 512 && !getenv("R433_TWO_PASS_SLICER")) {
            if ((rc = b->d_stage.ensure((size_t)r.total_pkgs * b->rows.size() * stage_cap)))
                return rc;
            lp.stage = b->d_stage.p;
            lp.stage_cap = stage_cap;
        }
        HIP_TRY(hipMemsetAsync(b->d_pkg_bytes.p, 0, (size_t)max_pkgs * sizeof(uint32_t), r.st));
        launch_slice_count(lp, r.total_pkgs, r.st);
        HIP_TRY(hipGetLastError());
        if (b->profiling)
            HIP_TRY(hipEventRecord(b->ev[3], r.st));
        launch_scan_u32(b->d_pkg_bytes.p, b->d_pkg_off.p, b->d_scal.p, max_pkgs, b->d_scal.p + 3, r.st);
    }
    else {
        HIP_TRY(hipMemsetAsync(b->d_scal.p + 3, 0, sizeof(uint32_t), r.st));
        if (b->profiling)
            HIP_TRY(hipEventRecord(b->ev[3], r.st));
    }
    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[4], r.st));
    HIP_TRY(hipMemcpyAsync(b->h_scal.p, b->d_scal.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
    HIP_TRY(stream_wait(b, r.st));
    size_t const pkg_bytes = b->h_scal.p[2];
    size_t const evt_bytes = b->h_scal.p[3];
    if (evt_bytes > 0xf0000000ull)
        return fail(R433_EOVERFLOW, "event stream exceeds 4 GiB; r.split the batch");

    if ((rc = b->d_pkg_blob.ensure(pkg_bytes + 16)) || (rc = b->h_pkg_blob.ensure(pkg_bytes + 16))
            || (rc = b->d_events.ensure(evt_bytes + 16)) || (rc = b->h_events.ensure(evt_bytes + 16))
            || (rc = b->h_frame_sums.ensure((size_t)r.n_streams * r.frames_cap)) || (rc = b->h_pkg_off.ensure(max_pkgs + 1))
            || (rc = b->h_rec_off.ensure(max_pkgs + 1)))
        return rc;
    if (r.total_pkgs) {
        launch_gather_packages(b->d_arena.p, b->arena_stride, b->d_dir_stream.p, b->d_dir_off.p, b->d_rec_off.p,
                b->d_scal.p, max_pkgs, b->d_pkg_blob.p, (uint32_t)std::min<size_t>(b->d_pkg_blob.cap, 0xffffffffu),
                r.total_pkgs, r.st);
        if (n_devs && evt_bytes) {
            lp.events = b->d_events.p;
            lp.events_cap = (uint32_t)std::min<size_t>(b->d_events.cap, 0xffffffffu);
            launch_slice_write(lp, r.total_pkgs, r.st);
        }
        HIP_TRY(hipGetLastError());
    }
    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[5], r.st));
    if (pkg_bytes)
        HIP_TRY(hipMemcpyAsync(b->h_pkg_blob.p, b->d_pkg_blob.p, pkg_bytes, hipMemcpyDeviceToHost, r.st));
    if (evt_bytes)
        HIP_TRY(hipMemcpyAsync(b->h_events.p, b->d_events.p, evt_bytes, hipMemcpyDeviceToHost, r.st));
    if (r.total_pkgs) {
        HIP_TRY(hipMemcpyAsync(b->h_rec_off.p, b->d_rec_off.p, r.total_pkgs * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
        if (n_devs)
            HIP_TRY(hipMemcpyAsync(b->h_pkg_off.p, b->d_pkg_off.p, r.total_pkgs * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
    }
    HIP_TRY(hipMemcpyAsync(b->h_frame_sums.p, b->d_frame_sums.p, (size_t)r.n_streams * r.frames_cap * sizeof(uint32_t),
            hipMemcpyDeviceToHost, r.st));
    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[6], r.st));
    HIP_TRY(stream_wait(b, r.st));
    b->pkg_bytes = pkg_bytes;
    b->evt_bytes = evt_bytes;
    b->events_counted = false;
    b->h_rec_off.p[r.total_pkgs] = (uint32_t)pkg_bytes;
    b->h_pkg_off.p[r.total_pkgs] = (uint32_t)evt_bytes;
    if (!n_devs)
        for (uint32_t i = 0; i < r.total_pkgs; ++i)
            b->h_pkg_off.p[i] = 0;

    if (b->profiling) {
        r433_batch_timing &t = b->last_timing;
        (void)hipEventElapsedTime(&t.detect_ms, b->ev[0], b->ev[1]);
        (void)hipEventElapsedTime(&t.dir_ms, b->ev[1], b->ev[2]);
        (void)hipEventElapsedTime(&t.count_ms, b->ev[2], b->ev[3]);
        (void)hipEventElapsedTime(&t.scan_ms, b->ev[3], b->ev[4]);
        (void)hipEventElapsedTime(&t.write_ms, b->ev[4], b->ev[5]);
        (void)hipEventElapsedTime(&t.d2h_ms, b->ev[5], b->ev[6]);
        (void)hipEventElapsedTime(&t.total_ms, b->ev[0], b->ev[6]);
    }
    return (int)r.total_pkgs;
}

} // namespace

extern "C" {

int r433_version(void)
{
    return 100; // 0.1.0
}

char const *r433_last_error(void)
{
    return g_err.c_str();
}

int r433_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(R433_ENODEV, "no usable HIP device: %s", hipGetErrorString(e));
    return n;
}

void r433_flow_cfg_default(r433_flow_cfg *cfg, uint32_t sample_size, uint32_t samp_rate)
{
    memset(cfg, 0, sizeof(*cfg));
    cfg->sample_size = sample_size;
    cfg->samp_rate = samp_rate;
    cfg->frame_samples = 0;
    cfg->fpdm = 0;
    cfg->enable_fm = 1;
    cfg->min_level_db = -12.1442f; // reference src/r_api.c:152-154
    cfg->min_snr_db = 9.0f;
    cfg->center_frequency = 433920000;
}

float r433_level_db(uint32_t sum, uint32_t n, int is_magnitude)
{
    float x = 1.0f;
    if (n > 0 && sum >= n)
        x = (float)sum / n;
    float lg = x > 0 ? log10f(x) : 0;
    return is_magnitude ? 20.0f * lg - 84.2884f : 10.0f * lg - 42.1442f;
}

r433_batch *r433_batch_create(r433_flow_cfg const *cfg, r433_dev_timing const *devs, uint32_t n_devs)
{
    if (!cfg || (cfg->sample_size != 2 && cfg->sample_size != 4) || cfg->samp_rate == 0) {
        fail(R433_EINVAL, "bad flow configuration");
        return nullptr;
    }
    if (n_devs > 2048) {
        fail(R433_EINVAL, "at most 2048 devices per batch engine");
        return nullptr;
    }
    if ((cfg->input_format == R433_IN_CS8 && cfg->sample_size != 2) || (cfg->input_format == R433_IN_CF32 && cfg->sample_size != 4)
            || cfg->input_format > R433_IN_CF32) {
        fail(R433_EINVAL, "input_format: cs8 goes with sample_size 2, cf32 with sample_size 4");
        return nullptr;
    }
    if (r433_device_count() < 0)
        return nullptr;
    r433_batch *b = new r433_batch();
    b->cfg = *cfg;
    if (b->cfg.frame_samples == 0)
        b->cfg.frame_samples = 262144u / cfg->sample_size; // DEFAULT_BUF_LENGTH, reference include/rtl_433.h:17
    if (b->cfg.frame_samples % 64 != 0) {
        fail(R433_EINVAL, "frame_samples must be a multiple of 64");
        delete b;
        return nullptr;
    }
    levels_from_db(b->det, (int)cfg->use_mag_est, cfg->level_limit_db, cfg->min_level_db, cfg->min_snr_db);
    b->det.per_ms = (int)(cfg->samp_rate / 1000);
    b->det.rate = cfg->samp_rate;
    b->det.fpdm = (int)cfg->fpdm;
    float lp = cfg->fm_low_pass != 0.0f ? cfg->fm_low_pass : cfg->fpdm ? 0.2f : 0.1f; // src/r_flow.c:204
    fm_coeffs(lp, cfg->samp_rate, b->a16, b->b16, b->a32, b->b32);

    b->timing.assign(devs, devs + n_devs);
    b->rows.reserve(n_devs);
    for (uint32_t i = 0; i < n_devs; ++i)
        b->rows.push_back(resolve_timing(devs[i], cfg->samp_rate, (int)i));
    std::stable_sort(b->rows.begin(), b->rows.end(), [](DevRow const &x, DevRow const &y) {
        if (x.is_fsk != y.is_fsk)
            return x.is_fsk < y.is_fsk;
        return x.modulation < y.modulation;
    });
    { // pad every (fsk, modulation) group to a multiple of 64 rows: one slicer per wavefront
        std::vector<DevRow> padded;
        DevRow pad;
        memset(&pad, 0, sizeof(pad));
        pad.orig = -1;
        for (size_t i = 0; i < b->rows.size(); ++i) {
            if (i > 0 && (b->rows[i].is_fsk != b->rows[i - 1].is_fsk || b->rows[i].modulation != b->rows[i - 1].modulation))
                while (padded.size() % 64)
                    padded.push_back(pad);
            padded.push_back(b->rows[i]);
        }
        while (padded.size() % 64)
            padded.push_back(pad);
        b->rows.swap(padded);
    }
    for (uint32_t i = 0; i < n_devs; ++i)
        b->prio_levels.push_back(devs[i].priority);
    std::sort(b->prio_levels.begin(), b->prio_levels.end());
    b->prio_levels.erase(std::unique(b->prio_levels.begin(), b->prio_levels.end()), b->prio_levels.end());
    if (n_devs) {
        if (b->d_rows.ensure(b->rows.size()) != 0
                || hipMemcpy(b->d_rows.p, b->rows.data(), b->rows.size() * sizeof(DevRow), hipMemcpyHostToDevice) != hipSuccess) {
            if (g_err.empty())
                fail(R433_EHIP, "device table upload failed");
            r433_batch_destroy(b);
            return nullptr;
        }
    }
    if (b->d_scal.ensure(16) != 0 || b->h_scal.ensure(16) != 0) {
        r433_batch_destroy(b);
        return nullptr;
    }
    return b;
}

void r433_batch_destroy(r433_batch *b)
{
    if (!b)
        return;
    b->d_rows.release();
    b->d_arena.release();
    b->d_ring.release();
    b->d_state.release();
    b->d_frame_sums.release();
    b->d_frame_min_high.release();
    b->d_tile_max.release();
    b->d_order.release();
    b->d_segs.release();
    b->h_tile_max.release();
    b->h_state.release();
    b->d_stream_bytes.release();
    b->d_pkg_base.release();
    b->d_scal.release();
    b->d_dir_stream.release();
    b->d_dir_off.release();
    b->d_rec_bytes.release();
    b->d_rec_off.release();
    b->d_sizes.release();
    b->d_pkg_bytes.release();
    b->d_pkg_off.release();
    b->d_pkg_blob.release();
    b->d_events.release();
    b->d_stage.release();
    b->d_converted.release();
    b->h_scal.release();
    b->h_frame_sums.release();
    b->h_pkg_blob.release();
    b->h_events.release();
    b->h_pkg_off.release();
    b->h_rec_off.release();
    if (b->ev_made)
        for (auto &e : b->ev)
            (void)hipEventDestroy(e);
    if (b->sync_ev)
        (void)hipEventDestroy(b->sync_ev);
    free(b->bits);
    free(b->pulses);
    delete b;
}

int r433_batch_set_taps(r433_batch *b, void *d_env, void *d_am, void *d_fm, uint64_t tap_stride)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if ((d_env || d_am || d_fm) && !(d_env && d_am && d_fm))
        return fail(R433_EINVAL, "taps come as a set of three");
    b->tap_env = d_env;
    b->tap_am = d_am;
    b->tap_fm = d_fm;
    b->tap_stride = tap_stride;
    return 0;
}

int r433_batch_set_split(r433_batch *b, uint32_t segment_samples)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (segment_samples > R433_SPLIT_AUTO && segment_samples < 4096)
        return fail(R433_EINVAL, "segments shorter than 4096 samples make no sense (the establishing tile alone is 2048)");
    b->split_samples = segment_samples;
    return 0;
}

int r433_batch_split_stats(r433_batch *b, uint32_t *segments, uint32_t *pieces_rerun)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (segments)
        *segments = b->last_segments;
    if (pieces_rerun)
        *pieces_rerun = b->last_redone;
    return 0;
}

int r433_batch_set_profiling(r433_batch *b, int on)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (on && !b->ev_made) {
        for (auto &e : b->ev)
            HIP_TRY(hipEventCreate(&e));
        b->ev_made = true;
    }
    b->profiling = on != 0;
    return 0;
}

int r433_batch_get_timing(r433_batch *b, r433_batch_timing *t)
{
    if (!b || !t)
        return fail(R433_EINVAL, "null argument");
    *t = b->last_timing;
    return 0;
}

int r433_batch_run(r433_batch *b, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes,
        uint32_t n_streams, void *stream)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (n_streams == 0) {
        b->n_streams = 0;
        b->n_pkgs = b->n_events = 0;
        b->pkg_bytes = b->evt_bytes = 0;
        return 0;
    }
    if (!d_iq || (stride_bytes & 15u) || ((uintptr_t)d_iq & 15u))
        return fail(R433_EINVAL, "capture base and stride must be 16-byte aligned");
    if (stride_bytes > 0xfffffff0ull)
        return fail(R433_EINVAL, "captures are limited to 4 GiB each");
    RunCtx r;
    r.b = b;
    r.st = (hipStream_t)stream;
    r.ss = b->cfg.sample_size;
    r.d_iq = d_iq;
    r.stride_bytes = stride_bytes;
    r.stream_bytes = stream_bytes;
    r.n_streams = n_streams;
    int rc;
    if ((rc = run_convert_input(r)) || (rc = run_size_buffers(r)) || (rc = run_autolevel(r)) || (rc = run_plan(r))
            || (rc = run_detect(r)))
        return rc;
    return run_slice_and_mirror(r);
}

int r433_batch_run_pulses(r433_batch *b, r433_pulse_data const *pulses, uint32_t n_packages, void *stream)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (n_packages == 0) {
        b->n_streams = 0;
        b->n_pkgs = b->n_events = 0;
        b->pkg_bytes = b->evt_bytes = 0;
        return 0;
    }
    if (!pulses)
        return fail(R433_EINVAL, "null pulse data");
    RunCtx r;
    r.b = b;
    r.st = (hipStream_t)stream;
    r.ss = b->cfg.sample_size;
    r.d_iq = nullptr;
    r.stride_bytes = 0;
    r.stream_bytes = nullptr;
    r.n_streams = n_packages;
    r.frames_cap = 1;
    r.n_order = n_packages;
    // one arena slot per package, laid out exactly as the detection kernel leaves a capture with one package
    uint32_t max_pulses = 0;
    for (uint32_t k = 0; k < n_packages; ++k) {
        if (pulses[k].num_pulses > R433_PD_MAX_PULSES)
            return fail(R433_EINVAL, "package %u has %u pulses (at most %u)", k, pulses[k].num_pulses, (unsigned)R433_PD_MAX_PULSES);
        max_pulses = std::max(max_pulses, pulses[k].num_pulses);
    }
    uint32_t const stride = ((uint32_t)sizeof(r433_pkg_rec) + 8u * max_pulses + 15u) & ~15u;
    int rc;
    if ((rc = b->d_arena.ensure((size_t)n_packages * stride)) || (rc = b->d_state.ensure(n_packages))
            || (rc = b->d_pkg_base.ensure(n_packages)) || (rc = b->d_frame_sums.ensure(n_packages))
            || (rc = b->h_arena_stage.ensure((size_t)n_packages * stride)) || (rc = b->h_state.ensure(n_packages)))
        return rc;
    b->arena_stride = stride;
    b->frames_cap = 1;
    b->n_streams = n_packages;
    memset(b->h_arena_stage.p, 0, (size_t)n_packages * stride);
    memset(b->h_state.p, 0, (size_t)n_packages * sizeof(StreamState));
    for (uint32_t k = 0; k < n_packages; ++k) {
        r433_pulse_data const &pd = pulses[k];
        uint8_t *rec = b->h_arena_stage.p + (size_t)k * stride;
        r433_pkg_rec h;
        memset(&h, 0, sizeof(h));
        h.total_bytes = (uint32_t)sizeof(h) + 8u * pd.num_pulses;
        h.stream = k;
        h.type = pd.fsk_f2_est ? R433_PKG_FSK : R433_PKG_OOK; // as the reference decides, src/rtl_433.c:1774
        h.num_pulses = pd.num_pulses;
        h.offset = pd.offset;
        h.start_ago = pd.start_ago;
        h.end_ago = pd.end_ago;
        h.ook_low = pd.ook_low_estimate;
        h.ook_high = pd.ook_high_estimate;
        h.fsk_f1 = pd.fsk_f1_est;
        h.fsk_f2 = pd.fsk_f2_est;
        h.sample_rate = pd.sample_rate ? pd.sample_rate : b->cfg.samp_rate;
        memcpy(rec, &h, sizeof(h));
        int32_t *pairs = (int32_t *)(rec + sizeof(h));
        for (uint32_t i = 0; i < pd.num_pulses; ++i) {
            pairs[2 * i] = pd.pulse[i];
            pairs[2 * i + 1] = pd.gap[i];
        }
        b->h_state.p[k].n_pkgs = 1;
        b->h_state.p[k].cursor = h.total_bytes;
    }
    if (b->profiling)
        for (int e = 0; e < 2; ++e)
            HIP_TRY(hipEventRecord(b->ev[e], r.st));
    HIP_TRY(hipMemcpyAsync(b->d_arena.p, b->h_arena_stage.p, (size_t)n_packages * stride, hipMemcpyHostToDevice, r.st));
    HIP_TRY(hipMemcpyAsync(b->d_state.p, b->h_state.p, (size_t)n_packages * sizeof(StreamState), hipMemcpyHostToDevice, r.st));
    HIP_TRY(hipMemsetAsync(b->d_frame_sums.p, 0, (size_t)n_packages * sizeof(uint32_t), r.st));
    launch_pkg_scan(b->d_state.p, nullptr, n_packages, b->d_pkg_base.p, b->d_scal.p, r.st);
    HIP_TRY(hipGetLastError());
    r.total_pkgs = n_packages;
    return run_slice_and_mirror(r);
}

int r433_batch_packages(r433_batch *b, uint8_t const **blob, size_t *len, uint32_t *count)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (blob)
        *blob = b->h_pkg_blob.p;
    if (len)
        *len = b->pkg_bytes;
    if (count)
        *count = b->n_pkgs;
    return 0;
}

int r433_batch_events(r433_batch *b, uint8_t const **blob, size_t *len, uint32_t *count)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (blob)
        *blob = b->h_events.p;
    if (len)
        *len = b->evt_bytes;
    if (!b->events_counted) {
        uint32_t n = 0;
        size_t at = 0;
        while (at + sizeof(r433_evt_rec) <= b->evt_bytes) {
            uint32_t total;
            memcpy(&total, b->h_events.p + at, 4);
            if (total < sizeof(r433_evt_rec) || at + total > b->evt_bytes)
                return fail(R433_EHIP, "corrupt event stream at byte %zu", at);
            at += total;
            n++;
        }
        b->n_events = n;
        b->events_counted = true;
    }
    if (count)
        *count = b->n_events;
    return 0;
}

int r433_batch_frame_sums(r433_batch *b, uint32_t const **sums, uint32_t *frames_cap)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (sums)
        *sums = b->h_frame_sums.p;
    if (frames_cap)
        *frames_cap = b->frames_cap;
    return 0;
}

int r433_batch_debug_state(r433_batch *b, void *host_buf, size_t bytes)
{
    if (!b || !host_buf)
        return fail(R433_EINVAL, "null argument");
    size_t have = (size_t)b->n_streams * sizeof(StreamState);
    HIP_TRY(hipMemcpy(host_buf, b->d_state.p, bytes < have ? bytes : have, hipMemcpyDeviceToHost));
    return (int)sizeof(StreamState);
}

int r433_batch_device_events(r433_batch *b, void const **d_events, size_t *len)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (d_events)
        *d_events = b->d_events.p;
    if (len)
        *len = b->evt_bytes;
    return 0;
}

// ---- decoder dispatch ----

namespace {

// calc_rssi_snr, reference src/r_flow.c:35-64
void fill_levels(r433_flow_cfg const &cfg, r433_pulse_data &p)
{
    float hi = p.ook_high_estimate > 0 ? p.ook_high_estimate : 1;
    float lo = p.ook_low_estimate > 0 ? p.ook_low_estimate : 1;
    int const max_high = (int)powf(10, (0 + 42.1442f) / 10.0f);
    float mx = hi < max_high ? hi : max_high;
    float asnr = mx / lo;
    float f1 = (float)p.fsk_f1_est / INT16_MAX * cfg.samp_rate / 2.0f;
    float f2 = (float)p.fsk_f2_est / INT16_MAX * cfg.samp_rate / 2.0f;
    p.freq1_hz = f1 + cfg.center_frequency;
    p.freq2_hz = f2 + cfg.center_frequency;
    p.centerfreq_hz = cfg.center_frequency;
    p.depth_bits = cfg.sample_size * 4;
    if (cfg.sample_size == 2 && !cfg.use_mag_est) {
        p.range_db = 42.1442f;
        p.rssi_db = 10.0f * log10f(hi) - 42.1442f;
        p.noise_db = 10.0f * log10f(lo) - 42.1442f;
        p.snr_db = 10.0f * log10f(asnr);
    }
    else {
        p.range_db = 84.2884f;
        p.rssi_db = 20.0f * log10f(hi) - 84.2884f;
        p.noise_db = 20.0f * log10f(lo) - 84.2884f;
        p.snr_db = 20.0f * log10f(asnr);
    }
}

struct DevStats {
    unsigned events = 0, ok = 0, messages = 0, fails[5] = {0, 0, 0, 0, 0};
};

thread_local r433_dispatch_info g_current;

// Replays packages [p0, p1).  Returns decoded event count or a negative error code.
int dispatch_range(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices, r433_package_fn pkg_cb, void *user,
        uint32_t p0, uint32_t p1, std::vector<DevStats> &stats, std::string &err)
{
    r433_bitbuffer *bits = (r433_bitbuffer *)calloc(1, sizeof(r433_bitbuffer));
    r433_pulse_data *pd = pkg_cb ? (r433_pulse_data *)calloc(1, sizeof(r433_pulse_data)) : nullptr;
    uint8_t const *ev = b->h_events.p;
    uint8_t const *pk = b->h_pkg_blob.p;
    std::vector<uint32_t> first(n_devices, 0), count(n_devices, 0), touched, refs;
    int decoded = 0;
    int rc = 0;

    for (uint32_t pkg = p0; pkg < p1 && rc == 0; ++pkg) {
        r433_pkg_rec ph;
        memcpy(&ph, pk + b->h_rec_off.p[pkg], sizeof(ph));
        if (pkg_cb) {
            memset(pd, 0, sizeof(*pd));
            pd->offset = ph.offset;
            pd->sample_rate = ph.sample_rate;
            pd->start_ago = ph.start_ago;
            pd->end_ago = ph.end_ago;
            pd->num_pulses = ph.num_pulses;
            int32_t const *pairs = (int32_t const *)(pk + b->h_rec_off.p[pkg] + sizeof(ph));
            for (uint32_t i = 0; i < ph.num_pulses && i < R433_MAX_PULSES; ++i) {
                pd->pulse[i] = pairs[2 * i];
                pd->gap[i] = pairs[2 * i + 1];
            }
            pd->ook_low_estimate = ph.ook_low;
            pd->ook_high_estimate = ph.ook_high;
            pd->fsk_f1_est = ph.fsk_f1;
            pd->fsk_f2_est = ph.fsk_f2;
            fill_levels(b->cfg, *pd);
            pkg_cb(user, ph.stream, ph.type, pd);
        }

        // index this package's events by device (they arrive sorted by device, then ordinal)
        refs.clear();
        touched.clear();
        size_t eat = b->h_pkg_off.p[pkg];
        size_t const eend = b->h_pkg_off.p[pkg + 1];
        while (eat + sizeof(r433_evt_rec) <= eend) {
            r433_evt_rec eh;
            memcpy(&eh, ev + eat, sizeof(eh));
            if (eh.pkg != pkg || eh.dev >= n_devices || eh.total_bytes < sizeof(eh) || eat + eh.total_bytes > eend) {
                err = "corrupt event stream";
                rc = R433_EHIP;
                break;
            }
            if (count[eh.dev] == 0) {
                first[eh.dev] = (uint32_t)refs.size();
                touched.push_back(eh.dev);
            }
            count[eh.dev]++;
            refs.push_back((uint32_t)eat);
            eat += eh.total_bytes;
        }

        int p_events = 0;
        for (uint32_t level : b->prio_levels) { // src/r_api.c:442-451: next level only while nothing decoded
            if (p_events || rc)
                break;
            for (uint32_t dev : touched) {
                if (b->timing[dev].priority != level || rc)
                    continue;
                r433_r_device *rd = devices[dev];
                for (uint32_t k = 0; k < count[dev]; ++k) {
                    uint8_t const *rec = ev + refs[first[dev] + k];
                    r433_evt_rec eh;
                    memcpy(&eh, rec, sizeof(eh));
                    // inflate into the reference bitbuffer layout
                    bits->num_rows = eh.num_rows;
                    bits->free_row = eh.free_row;
                    uint8_t const *rp = rec + sizeof(eh);
                    for (uint32_t r = 0; r < eh.num_rows && r < R433_BITBUF_ROWS; ++r) {
                        r433_row_rec rr;
                        memcpy(&rr, rp, sizeof(rr));
                        bits->bits_per_row[r] = rr.bits;
                        bits->syncs_before_row[r] = rr.syncs;
                        size_t room = (size_t)(R433_BITBUF_ROWS - r) * R433_BITBUF_COLS;
                        memcpy(bits->bb[r], rp + sizeof(rr), rr.nbytes < room ? rr.nbytes : room);
                        rp += sizeof(rr) + ((rr.nbytes + 3u) & ~3u);
                    }
                    uint32_t used_rows = std::max<uint32_t>(eh.num_rows, eh.free_row);

                    g_current.stream = ph.stream;
                    g_current.package = pkg;
                    g_current.device = dev;
                    g_current.ordinal = eh.ordinal;
                    g_current.package_type = ph.type;
                    g_current.start_ago = ph.start_ago;
                    int ret = 0;
                    if (rd && rd->decode_fn)
                        ret = rd->decode_fn(rd, bits);
                    DevStats &ds = stats[dev]; // statistics, src/pulse_slicer.c:35-47
                    ds.events += 1;
                    if (ret > 0) {
                        ds.ok += 1;
                        ds.messages += (unsigned)ret;
                    }
                    else if (ret >= R433_DECODE_FAIL_SANITY) {
                        ds.fails[-ret] += 1;
                        ret = 0;
                    }
                    else {
                        char buf[200];
                        snprintf(buf, sizeof(buf), "decoder \"%s\" gave invalid return value %d",
                                rd && rd->name ? rd->name : "?", ret);
                        err = buf;
                        rc = R433_EDECODER;
                        break;
                    }
                    if (ret > 0)
                        p_events += ret;
                    // bitbuffer_clear: only what can be dirty (the decoder may have grown the buffer)
                    used_rows = std::max<uint32_t>(used_rows, std::max<uint32_t>(bits->num_rows, bits->free_row));
                    if (used_rows > R433_BITBUF_ROWS)
                        used_rows = R433_BITBUF_ROWS;
                    memset(bits->bb, 0, (size_t)used_rows * R433_BITBUF_COLS);
                    memset(bits, 0, offsetof(r433_bitbuffer, bb));
                }
            }
        }
        decoded += p_events;
        for (uint32_t dev : touched)
            count[dev] = 0;
    }
    free(bits);
    free(pd);
    return rc ? rc : decoded;
}

} // namespace

int r433_dispatch_current(r433_dispatch_info *info)
{
    if (!info)
        return fail(R433_EINVAL, "null argument");
    *info = g_current;
    return 0;
}

int r433_batch_dispatch_mt(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices, r433_package_fn pkg_cb,
        void *user, uint32_t n_threads)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (n_devices != b->timing.size())
        return fail(R433_EINVAL, "dispatch needs the %zu devices the engine was created with", b->timing.size());
    uint32_t const np = b->n_pkgs;
    if (n_threads < 1)
        n_threads = 1;
    if (n_threads > np)
        n_threads = np ? np : 1;
    std::vector<std::vector<DevStats>> stats(n_threads, std::vector<DevStats>(n_devices));
    std::vector<int> results(n_threads, 0);
    std::vector<std::string> errs(n_threads);
    if (n_threads == 1) {
        results[0] = dispatch_range(b, devices, n_devices, pkg_cb, user, 0, np, stats[0], errs[0]);
        digest_publish();
    }
    else {
        // packages are handed out in small contiguous runs from a shared cursor: event counts per
        // package vary by orders of magnitude, static ranges leave most workers idle at the end
        uint32_t const grain = std::max<uint32_t>(1, std::min<uint32_t>(16, np / (n_threads * 8)));
        std::atomic<uint32_t> cursor{0};
        b->pool.run(n_threads, [&](unsigned w) {
            for (;;) {
                uint32_t p0 = cursor.fetch_add(grain, std::memory_order_relaxed);
                if (p0 >= np || results[w] < 0)
                    break;
                int r = dispatch_range(b, devices, n_devices, pkg_cb, user, p0, std::min(np, p0 + grain), stats[w], errs[w]);
                results[w] = r < 0 ? r : results[w] + r;
            }
            digest_publish();
        });
    }
    int decoded = 0;
    for (uint32_t i = 0; i < n_threads; ++i) {
        if (results[i] < 0)
            return fail(results[i], "%s", errs[i].c_str());
        decoded += results[i];
    }
    for (uint32_t d = 0; d < n_devices; ++d) {
        r433_r_device *rd = devices[d];
        if (!rd)
            continue;
        for (uint32_t i = 0; i < n_threads; ++i) {
            DevStats const &ds = stats[i][d];
            rd->decode_events += ds.events;
            rd->decode_ok += ds.ok;
            rd->decode_messages += ds.messages;
            for (int k = 0; k < 5; ++k)
                rd->decode_fails[k] += ds.fails[k];
        }
    }
    return decoded;
}

int r433_batch_dispatch(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices, r433_package_fn pkg_cb,
        void *user)
{
    return r433_batch_dispatch_mt(b, devices, n_devices, pkg_cb, user, 1);
}

// A decode_fn with the reference plugin signature that folds every bitbuffer it is handed into an
// order-independent checksum (see r433_hip.h).  decode_ctx must point to a r433_digest_ctx.
int r433_plugin_digest_decode(r433_r_device *decoder, r433_bitbuffer *bits)
{
    r433_digest_ctx *ctx = (r433_digest_ctx *)decoder->decode_ctx;
    if (!ctx)
        return R433_DECODE_ABORT_EARLY;
    uint64_t x = 1469598103934665603ull;
    auto mix = [&x](void const *p, size_t n) {
        uint8_t const *q = (uint8_t const *)p;
        for (size_t i = 0; i < n; ++i)
            x = (x ^ q[i]) * 1099511628211ull;
    };
    uint32_t pkg = g_current.package;
    uint16_t dev = (uint16_t)g_current.device, ord = (uint16_t)g_current.ordinal;
    mix(&pkg, 4);
    mix(&dev, 2);
    mix(&ord, 2);
    mix(&bits->num_rows, 2);
    mix(&bits->free_row, 2);
    for (unsigned r = 0; r < bits->num_rows && r < R433_BITBUF_ROWS; ++r) {
        mix(&bits->bits_per_row[r], 2);
        mix(&bits->syncs_before_row[r], 2);
        mix(bits->bb[r], ((unsigned)bits->bits_per_row[r] + 7) / 8);
    }
    if (g_digest.ctx != ctx) {
        digest_publish();
        g_digest.ctx = ctx;
    }
    g_digest.sum += x; // published by the dispatcher when this thread is done with the batch
    g_digest.events += 1;
    return R433_DECODE_ABORT_LENGTH;
}

// ---- function-level seam ----

static int run_envelope(int kind, void const *d_iq, void *d_env, uint32_t n, uint32_t *d_sum, void *stream)
{
    if (r433_device_count() < 0)
        return R433_ENODEV;
    if (!d_iq || !d_env)
        return fail(R433_EINVAL, "null buffer");
    if (((uintptr_t)d_iq & 15u) || ((uintptr_t)d_env & 15u))
        return fail(R433_EINVAL, "buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (d_sum)
        HIP_TRY(hipMemsetAsync(d_sum, 0, sizeof(uint32_t), st));
    launch_envelope(kind, d_iq, (uint16_t *)d_env, n, d_sum, st);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int run_convert(int fmt, void const *d_in, void *d_out, uint64_t n, void *stream)
{
    if (r433_device_count() < 0)
        return R433_ENODEV;
    if (!d_in || !d_out)
        return fail(R433_EINVAL, "null buffer");
    if (((uintptr_t)d_in & 15u) || ((uintptr_t)d_out & 15u))
        return fail(R433_EINVAL, "buffers must be 16-byte aligned");
    uint64_t const in_bytes = fmt == 1 ? n : n * 4;
    launch_convert(fmt, d_in, in_bytes, d_out, fmt == 1 ? n : n * 2, in_bytes, 1, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int r433_convert_cs8_cu8(void const *d_in, void *d_out, uint64_t n, void *stream)
{
    return run_convert(1, d_in, d_out, n, stream);
}

int r433_convert_cf32_cs16(void const *d_in, void *d_out, uint64_t n, void *stream)
{
    return run_convert(2, d_in, d_out, n, stream);
}

int r433_dump_convert(int format, uint32_t sample_size, void const *d_in, void *d_out, uint64_t n_out, void *stream)
{
    if (format < R433_DUMP_CU8_IQ || format > R433_DUMP_F32_Q)
        return fail(R433_EINVAL, "unknown dump format %d", format);
    if (sample_size != 2 && sample_size != 4)
        return fail(R433_EINVAL, "sample_size must be 2 (cu8) or 4 (cs16)");
    if (n_out == 0)
        return 0;
    if (!d_in || !d_out || ((uintptr_t)d_in & 15u) || ((uintptr_t)d_out & 15u))
        return fail(R433_EINVAL, "dump buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    // the reference writes its own buffers for these (src/r_flow.c:396,403,436-443): a copy
    size_t same = 0;
    if (format == R433_DUMP_CU8_IQ && sample_size == 2)
        same = n_out;
    else if ((format == R433_DUMP_CS16_IQ && sample_size == 4) || format == R433_DUMP_S16_AM || format == R433_DUMP_S16_FM)
        same = n_out * 2;
    if (same) {
        HIP_TRY(hipMemcpyAsync(d_out, d_in, same, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (launch_dump(format, sample_size, d_in, d_out, n_out, st))
        return fail(R433_EINVAL, "dump format %d is not a conversion", format);
    HIP_TRY(hipGetLastError());
    return 0;
}

int r433_batch_analyze(r433_batch *b, r433_analysis *out, uint32_t max_packages, void *stream)
{
    if (!b || (!out && max_packages))
        return fail(R433_EINVAL, "null argument");
    uint32_t const n = std::min(b->n_pkgs, max_packages);
    if (n == 0)
        return 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = b->d_analysis.ensure(n)))
        return rc;
    // the arena and the package directory of the last run are still on the device
    launch_analyze(b->d_arena.p, b->arena_stride, b->d_dir_stream.p, b->d_dir_off.p, n, b->d_analysis.p, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, b->d_analysis.p, (size_t)n * sizeof(r433_analysis), hipMemcpyDeviceToHost, st));
    HIP_TRY(stream_wait(b, st));
    return (int)n;
}

namespace {

// histogram_find_bin_index, reference src/pulse_analyzer.c:157-165
int find_bin(r433_histogram const &h, int width)
{
    for (uint32_t n = 0; n < h.bins_count && n < R433_HIST_BINS; ++n)
        if (h.bins[n].min <= width && width <= h.bins[n].max)
            return (int)n;
    return -1;
}

// hexstr_t, reference src/pulse_analyzer.c:180-209
struct HexStr {
    uint8_t p[1024];
    unsigned idx = 0;
    void byte(uint8_t v)
    {
        if (idx < sizeof(p))
            p[idx++] = v;
    }
    void word(uint16_t v)
    {
        if (idx + 1 < sizeof(p)) {
            p[idx++] = (uint8_t)(v >> 8);
            p[idx++] = (uint8_t)(v & 0xff);
        }
    }
};

} // namespace

int r433_analysis_text(r433_batch *b, uint32_t pkg, r433_analysis const *a, char *buf, size_t cap)
{
    if (!b || !a || (!buf && cap))
        return fail(R433_EINVAL, "null argument");
    if (pkg >= b->n_pkgs)
        return fail(R433_EINVAL, "package %u of %u", pkg, b->n_pkgs);
    size_t len = 0;
#define PUT(...)                                                                                                     \
    do {                                                                                                             \
        int const n_ = snprintf(len < cap ? buf + len : nullptr, len < cap ? cap - len : 0, __VA_ARGS__);           \
        if (n_ > 0)                                                                                                  \
            len += (size_t)n_;                                                                                       \
    } while (0)
    if (a->num_pulses == 0) { // src/pulse_analyzer.c:281-284
        PUT("No pulses detected.\n");
        return (int)len;
    }
    uint8_t const *rec = b->h_pkg_blob.p + b->h_rec_off.p[pkg];
    r433_pkg_rec ph;
    memcpy(&ph, rec, sizeof(ph));
    int32_t const *pairs = (int32_t const *)(rec + sizeof(ph));
    uint32_t const num = std::min<uint32_t>(ph.num_pulses, R433_MAX_PULSES);
    uint32_t const rate = ph.sample_rate;
    double const to_ms = 1e3 / rate, to_us = 1e6 / rate;
    r433_pulse_data lv; // only the level fields are used
    lv.ook_low_estimate = ph.ook_low;
    lv.ook_high_estimate = ph.ook_high;
    lv.fsk_f1_est = ph.fsk_f1;
    lv.fsk_f2_est = ph.fsk_f2;
    fill_levels(b->cfg, lv);

    auto print_hist = [&](char const *title, r433_histogram const &h) { // histogram_print, :168-178
        PUT("%s\n", title);
        for (uint32_t n = 0; n < h.bins_count && n < R433_HIST_BINS; ++n)
            PUT(" [%2u] count: %4u,  width: %4.0f us [%.0f;%.0f]\t(%4i S)\n", n, h.bins[n].count, h.bins[n].mean * 1e6 / rate,
                    h.bins[n].min * 1e6 / rate, h.bins[n].max * 1e6 / rate, h.bins[n].mean);
    };
    PUT("Analyzing pulses...\n"); // :326-346
    PUT("Total count: %4u,  width: %4.2f ms\t\t(%5i S)\n", a->num_pulses, a->total_period * to_ms, a->total_period);
    print_hist("Pulse width distribution:", a->pulses);
    print_hist("Gap width distribution:", a->gaps);
    print_hist("Pulse+gap period distribution:", a->periods_pg);
    print_hist("Gap+pulse period distribution:", a->periods_gp);
    print_hist("Timing distribution:", a->timings);
    PUT("Level estimates [high, low]: %6i, %6i\n", ph.ook_high, ph.ook_low);
    PUT("RSSI: %.1f dB SNR: %.1f dB Noise: %.1f dB\n", (double)lv.rssi_db, (double)lv.snr_db, (double)lv.noise_db);
    PUT("Frequency offsets [F1, F2]:  %6i, %6i\t(%+.1f kHz, %+.1f kHz)\n", ph.fsk_f1, ph.fsk_f2,
            ((float)ph.fsk_f1 / INT16_MAX) * (rate / 2.0 / 1000.0), ((float)ph.fsk_f2 / INT16_MAX) * (rate / 2.0 / 1000.0));
    static char const *const kGuess[] = {"", "Single pulse detected. Probably Frequency Shift Keying or just noise...",
            "Un-modulated signal. Maybe a preamble...", "Pulse Position Modulation with fixed pulse width",
            "Pulse Width Modulation with fixed gap", "Pulse Width Modulation with fixed period", "Manchester coding",
            "Pulse Width Modulation with multiple packets", "Non Return to Zero coding (Pulse Code)",
            "Pulse Width Modulation with sync/delimiter", "No clue..."};
    PUT("Guessing modulation: %s\n", kGuess[a->guess <= R433_GUESS_NO_CLUE ? a->guess : 0]);

    // RfRaw line, :432-513 (the guess sorted only copies of the pulse / gap histograms; their bin counts did not change,
    // except that an FSK zero-bin left the pulse histogram, which this part does not look at)
    r433_histogram const &T = a->timings;
    if (T.bins_count <= 8) {
        // gap bins by ascending mean, as the reference has sorted them by now
        r433_hist_bin gs[R433_HIST_BINS];
        uint32_t const ng = std::min<uint32_t>(a->gaps.bins_count, R433_HIST_BINS);
        for (uint32_t k = 0; k < ng; ++k)
            gs[k] = a->gaps.bins[k];
        for (uint32_t n = 0; n + 1 < ng; ++n)
            for (uint32_t m = n + 1; m < ng; ++m)
                if (gs[m].mean < gs[n].mean)
                    std::swap(gs[m], gs[n]);
        auto push_bins = [&](HexStr &h) {
            for (uint32_t k = 0; k < T.bins_count; ++k) {
                double const w = std::max(0.0, T.bins[k].mean * to_us);
                h.word((uint16_t)(w < 65535 ? w : 65535));
            }
        };
        if (ng <= 2) {
            HexStr h;
            h.byte(0xaa);
            h.byte(0xb1);
            h.byte((uint8_t)T.bins_count);
            push_bins(h);
            for (uint32_t i = 0; i < num; ++i)
                h.byte((uint8_t)(0x80 | (find_bin(T, pairs[2 * i]) << 4) | find_bin(T, pairs[2 * i + 1])));
            h.byte(0x55);
            PUT("view at https://triq.org/pdv/#");
            for (unsigned k = 0; k < h.idx; ++k)
                PUT("%02X", h.p[k]);
            PUT("\n");
        }
        else {
            int const limit = gs[std::min<uint32_t>(3, ng - 1)].min;
            std::vector<HexStr> strs(32);
            unsigned cnt = 0;
            uint32_t i = 0;
            while (i < num && cnt < 32) {
                HexStr &h = strs[cnt];
                h.idx = 0;
                h.byte(0xaa);
                h.byte(0xb0);
                h.byte(0);
                h.byte((uint8_t)T.bins_count);
                h.byte(1);
                push_bins(h);
                for (; i < num; ++i) {
                    h.byte((uint8_t)(0x80 | (find_bin(T, pairs[2 * i]) << 4) | find_bin(T, pairs[2 * i + 1])));
                    if (pairs[2 * i + 1] >= limit) {
                        ++i;
                        break;
                    }
                }
                h.byte(0x55);
                h.p[2] = (uint8_t)(h.idx - 4 <= 255 ? h.idx - 4 : 0);
                if (cnt > 0 && strs[cnt - 1].idx == h.idx && !memcmp(&strs[cnt - 1].p[5], &h.p[5], h.idx - 5)) {
                    h.idx = 0;
                    strs[cnt - 1].p[4] += 1;
                }
                else {
                    cnt++;
                }
            }
            PUT("view at https://triq.org/pdv/#");
            for (unsigned j = 0; j < cnt; ++j) {
                if (j > 0)
                    PUT("+");
                for (unsigned k = 0; k < strs[j].idx; ++k)
                    PUT("%02X", strs[j].p[k]);
            }
            PUT("\n");
            if (cnt >= 32)
                PUT("Too many pulse groups (%u pulses missed in rfraw)\n", num - i);
        }
    }
    r433_dev_timing const &d = a->device;
    if (d.modulation) { // :516-556
        PUT("Attempting demodulation... short_width: %.0f, long_width: %.0f, reset_limit: %.0f, sync_width: %.0f\n", (double)d.short_width,
                (double)d.long_width, (double)d.reset_limit, (double)d.sync_width);
        switch (d.modulation) {
        case 16: // FSK_PULSE_PCM
            PUT("Use a flex decoder with -X 'n=name,m=FSK_PCM,s=%.0f,l=%.0f,r=%.0f'\n", (double)d.short_width, (double)d.long_width,
                    (double)d.reset_limit);
            break;
        case 5: // OOK_PULSE_PPM
            PUT("Use a flex decoder with -X 'n=name,m=OOK_PPM,s=%.0f,l=%.0f,g=%.0f,r=%.0f'\n", (double)d.short_width, (double)d.long_width,
                    (double)d.gap_limit, (double)d.reset_limit);
            break;
        case 6:  // OOK_PULSE_PWM
        case 17: // FSK_PULSE_PWM
            PUT("Use a flex decoder with -X 'n=name,m=%s,s=%.0f,l=%.0f,r=%.0f,g=%.0f,t=%.0f,y=%.0f'\n", d.modulation == 6 ? "OOK_PWM" : "FSK_PWM",
                    (double)d.short_width, (double)d.long_width, (double)d.reset_limit, (double)d.gap_limit, (double)d.tolerance,
                    (double)d.sync_width);
            break;
        case 3: // OOK_PULSE_MANCHESTER_ZEROBIT
            PUT("Use a flex decoder with -X 'n=name,m=OOK_MC_ZEROBIT,s=%.0f,l=%.0f,r=%.0f'\n", (double)d.short_width, (double)d.long_width,
                    (double)d.reset_limit);
            break;
        default:
            PUT("Unsupported\n");
        }
    }
#undef PUT
    return (int)len;
}

// pulse_data_load, reference src/pulse_data.c:122-176, over a text in memory: one call of the reference reads one
// package; the file loop calls it until a package comes back empty (src/rtl_433.c:1757-1761).
int r433_pulse_text_load(char const *text, size_t len, uint32_t sample_rate, r433_pulse_data *out, uint32_t max_packages)
{
    if ((!text && len) || (!out && max_packages))
        return fail(R433_EINVAL, "null argument");
    size_t at = 0;
    uint32_t n_out = 0;
    double const to_sample = sample_rate / 1e6;
    // fgets(s, 1024, file): at most 1023 characters, up to and including the newline
    auto next_line = [&](char *s) -> bool {
        if (at >= len)
            return false;
        size_t k = 0;
        while (k < 1023 && at < len) {
            char const c = text[at++];
            s[k++] = c;
            if (c == '\n')
                break;
        }
        s[k] = '\0';
        return true;
    };
    for (;;) {
        r433_pulse_data *data = n_out < max_packages ? &out[n_out] : nullptr;
        if (!data)
            break;
        memset(data, 0, sizeof(*data)); // pulse_data_clear
        data->sample_rate = sample_rate;
        char s[1024];
        int i = 0;
        while (i < R433_MAX_PULSES && next_line(s)) {
            if (!strncmp(s, ";freq1", 6))
                data->freq1_hz = (float)strtol(s + 6, nullptr, 10);
            if (!strncmp(s, ";freq2", 6))
                data->freq2_hz = (float)strtol(s + 6, nullptr, 10);
            if (*s == ';') {
                if (i)
                    break; // end or next header found
                continue;  // still reading a header
            }
            char const *p = s;
            char *endptr;
            long const mark = strtol(p, &endptr, 10);
            p = endptr + 1;
            long const space = strtol(p, &endptr, 10);
            if (mark < 0 || space < 0)
                continue; // the reference warns and skips the line
            data->pulse[i] = (int)(to_sample * mark);
            data->gap[i++] = (int)(to_sample * space);
        }
        data->num_pulses = (unsigned)i;
        if (i == 0)
            break; // the file loop stops at the first empty package
        n_out += 1;
    }
    return (int)n_out;
}

// pulse_data_dump, reference src/pulse_data.c:193-224.  Returns the length of the text (like snprintf: the text
// is cut if it does not fit cap, the full length is returned either way).
int r433_pulse_text_dump(r433_pulse_data const *data, char const *received, char *buf, size_t cap)
{
    if (!data || (!buf && cap))
        return fail(R433_EINVAL, "null argument");
    size_t len = 0;
#define PUT(...)                                                                                                     \
    do {                                                                                                             \
        int const n_ = snprintf(len < cap ? buf + len : nullptr, len < cap ? cap - len : 0, __VA_ARGS__);           \
        if (n_ > 0)                                                                                                  \
            len += (size_t)n_;                                                                                       \
    } while (0)
    if (received)
        PUT(";received %s\n", received);
    if (data->fsk_f2_est) {
        PUT(";fsk %u pulses\n", data->num_pulses);
        PUT(";freq1 %.0f\n", (double)data->freq1_hz);
        PUT(";freq2 %.0f\n", (double)data->freq2_hz);
    }
    else {
        PUT(";ook %u pulses\n", data->num_pulses);
        PUT(";freq1 %.0f\n", (double)data->freq1_hz);
    }
    PUT(";centerfreq %.0f Hz\n", (double)data->centerfreq_hz);
    PUT(";samplerate %u Hz\n", data->sample_rate);
    PUT(";sampledepth %u bits\n", data->depth_bits);
    PUT(";range %.1f dB\n", (double)data->range_db);
    PUT(";rssi %.1f dB\n", (double)data->rssi_db);
    PUT(";snr %.1f dB\n", (double)data->snr_db);
    PUT(";noise %.1f dB\n", (double)data->noise_db);
    double const to_us = 1e6 / data->sample_rate;
    for (unsigned i = 0; i < data->num_pulses && i < R433_MAX_PULSES; ++i)
        PUT("%.0f %.0f\n", data->pulse[i] * to_us, data->gap[i] * to_us);
    PUT(";end\n");
#undef PUT
    return (int)len;
}

int r433_envelope_detect(void const *d_iq, void *d_env, uint32_t n, uint32_t *d_sum, void *stream)
{
    return run_envelope(ENV_AMP_CU8, d_iq, d_env, n, d_sum, stream);
}

int r433_magnitude_est_cu8(void const *d_iq, void *d_env, uint32_t n, uint32_t *d_sum, void *stream)
{
    return run_envelope(ENV_MAG_CU8, d_iq, d_env, n, d_sum, stream);
}

int r433_magnitude_est_cs16(void const *d_iq, void *d_env, uint32_t n, uint32_t *d_sum, void *stream)
{
    return run_envelope(ENV_MAG_CS16, d_iq, d_env, n, d_sum, stream);
}

} // extern "C"
