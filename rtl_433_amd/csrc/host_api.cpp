// host_api.cpp -- the C ABI of librtl433hip.so (include/r433_hip.h): batch life cycle, configuration, result
// accessors and the stateless function-level entry points.  The pass itself is in batch_run.cpp, the decoder
// dispatch in dispatch.cpp, report / text formats in reports.cpp.
//
// There is no CPU implementation of the hot path in here: if HIP is unusable every compute entry
// point fails with R433_ENODEV.
#include "host_common.hpp"

using namespace r433;

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

namespace r433 {

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
    r.pf = -1;
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

} // namespace r433

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

r433_batch *r433_batch_create_on(int device, r433_flow_cfg const *cfg, r433_dev_timing const *devs, uint32_t n_devs);

// Streams and hardware queues.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by
// default), and two streams that share one run in submission order: a pipeline of three engines has six streams of its own (a main
// and a fork stream each, slicer_kernels.hip) beside the host's, and the sizing pass of one engine sat for 17 ms behind the
// 1 GiB input copy of another (every third step of bench.py's host-fed pipeline: 22.3 -> 20.4 ms per step with eight queues,
// profiles/r06_hw_queues.txt).  The runtime reads the variable when it initialises, i.e. at the process's first HIP call: a
// library that is loaded before that asks for eight, unless the user has said otherwise.  (A host that has already made HIP
// calls -- a Python process that imported torch and touched the device -- sets the variable itself: bench.py does.)
__attribute__((constructor)) static void ask_for_hardware_queues()
{
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

namespace {
__global__ void k_warm(int *p)
{
    if (p)
        *p = 1;
}
} // namespace

// What a process pays once before its first kernel has run -- opening the device (50-130 ms on the MI355X boxes), loading this
// library's code object (20-150 ms at the first launch) -- as a call of its own, so that a host can make it on a thread beside
// its file loop (dropin/r_flow_hip.c does).
int r433_warmup(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(R433_ENODEV, "no HIP device");
    }
    hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, nullptr, (int *)nullptr);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        return fail(R433_EHIP, "the library's kernels do not load on this device");
    }
    return 0;
}

r433_batch *r433_batch_create(r433_flow_cfg const *cfg, r433_dev_timing const *devs, uint32_t n_devs)
{
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) {
        (void)hipGetLastError();
        cur = 0;
    }
    return r433_batch_create_on(cur, cfg, devs, n_devs);
}

int r433_batch_device(r433_batch *b)
{
    return b ? b->device : fail(R433_EINVAL, "null batch");
}

r433_batch *r433_batch_create_on(int device, r433_flow_cfg const *cfg, r433_dev_timing const *devs, uint32_t n_devs)
{
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
            (void)hipGetLastError();
            fail(R433_ENODEV, "no GPU %d (%d visible)", device, n);
            return nullptr;
        }
    }
    DeviceScope on_device(device);

    if (!cfg || (cfg->sample_size != 2 && cfg->sample_size != 4) || cfg->samp_rate == 0) {
        fail(R433_EINVAL, "bad flow configuration");
        return nullptr;
    }
    if (n_devs && !devs) {
        fail(R433_EINVAL, "null device table");
        return nullptr;
    }
    if (n_devs > 2048) {
        fail(R433_EINVAL, "at most 2048 devices per batch engine");
        return nullptr;
    }
    if ((cfg->input_format == R433_IN_CS8 && cfg->sample_size != 2) || (cfg->input_format == R433_IN_CF32 && cfg->sample_size != 4)
            || ((cfg->input_format == R433_IN_S16_AM || cfg->input_format == R433_IN_S16_FM) && cfg->sample_size != 2)
            || cfg->input_format > R433_IN_S16_FM) {
        fail(R433_EINVAL, "input_format: cs8, am.s16 and fm.s16 go with sample_size 2, cf32 with sample_size 4");
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
    b->device = device;
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
    { // Chunks of 64 rows, a wavefront's devices.  A line code with 64 decoders and more fills chunks of its own.  The rest --
      // most line codes have a handful of decoders -- share chunks with other line codes of their kind (OOK or FSK: a chunk draws
      // the packages of one kind, k_pkg_order): a wavefront then walks the package once per line code it holds, one after the
      // other, but an item's fixed cost (the draw, four dependent loads, staging the pulses, two barriers: 38 us of a wavefront's
      // life against 20-130 us for a walk, profiles/r04_slice_shares.txt) is paid once for all of them.  First fit, heaviest
      // first, by the walk's measured weight per line code; a chunk takes no more than 1.2 times the heaviest walk, so that the
      // longest item of the launch stays what it was.  (R433_SLICE_NO_PACK: a chunk per line code, development / A/B timing.)
        auto const weight = [](DevRow const &r) -> double {
            switch (r.modulation) { // mean wavefront microseconds of a walk over a bench package
            case 4: case 16: return 135; // PCM
            case 6: case 17: return 73;  // PWM
            case 3: case 18: return 72;  // Manchester
            case 11: return 46;          // PIWM DC
            case 5: return 44;           // PPM
            case 8: case 12: return 40;  // PIWM raw, NRZS (not in the default set: a guess)
            case 9: return 23;           // DMC
            case 13: return 18;
            case 10: return 1;           // Oregon v1
            default: return 40;
            }
        };
        struct Piece { size_t begin, n; double w; };
        struct Bin { std::vector<Piece> pieces; size_t rows = 0; double w = 0; int fsk = 0; };
        DevRow pad;
        memset(&pad, 0, sizeof(pad));
        pad.orig = -1;
        pad.pf = -1;
        std::vector<DevRow> packed;
        std::vector<Piece> rest[2];
        double heaviest = 0;
        for (size_t i = 0; i < b->rows.size();) {
            size_t j = i;
            while (j < b->rows.size() && b->rows[j].is_fsk == b->rows[i].is_fsk && b->rows[j].modulation == b->rows[i].modulation)
                ++j;
            for (; j - i >= 64; i += 64)
                packed.insert(packed.end(), b->rows.begin() + (long)i, b->rows.begin() + (long)i + 64);
            if (j > i) {
                rest[b->rows[i].is_fsk != 0].push_back({i, j - i, weight(b->rows[i])});
                heaviest = std::max(heaviest, weight(b->rows[i]));
            }
            i = j;
        }
        bool const no_pack = getenv("R433_SLICE_NO_PACK") != nullptr;
        for (int kind = 0; kind < 2; ++kind) {
            std::stable_sort(rest[kind].begin(), rest[kind].end(), [](Piece const &x, Piece const &y) { return x.w > y.w; });
            std::vector<Bin> bins;
            for (Piece const &pc : rest[kind]) {
                Bin *into = nullptr;
                if (!no_pack)
                    for (Bin &bin : bins)
                        if (bin.rows + pc.n <= 64 && bin.w + pc.w <= 1.2 * heaviest) {
                            into = &bin;
                            break;
                        }
                if (!into) {
                    bins.emplace_back();
                    into = &bins.back();
                }
                into->pieces.push_back(pc);
                into->rows += pc.n;
                into->w += pc.w;
            }
            for (Bin const &bin : bins) {
                for (Piece const &pc : bin.pieces)
                    packed.insert(packed.end(), b->rows.begin() + (long)pc.begin, b->rows.begin() + (long)(pc.begin + pc.n));
                while (packed.size() % 64)
                    packed.push_back(pad);
            }
        }
        b->rows.swap(packed);
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
    DeviceScope on_device(b->device);
    b->d_rows.release();
    b->d_arena.release();
    b->d_ring.release();
    b->d_state.release();
    b->d_frame_sums.release();
    b->d_frame_min_high.release();
    b->d_tile_max.release();
    b->d_order.release();
    b->d_wg.release();
    b->d_tile_store.release();
    b->d_tile_desc.release();
    b->d_tile_words.release();
    b->d_idx_cnt.release();
    b->d_slice_start.release();
    b->d_slices.release();
    b->h_slice_start.release();
    b->h_slices.release();
    b->d_pkg_order.release();
    b->d_slice_cursor.release();
    b->d_chunk_work.release();
    b->d_chunk_deal.release();
    b->h_chunk_work.release();
    b->d_pf_tables.release();
    b->d_pf_counts.release();
    b->h_pf_counts.release();
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
    b->d_dev_off.release();
    b->d_pkg_bytes.release();
    b->d_pkg_off.release();
    b->d_pkg_blob.release();
    b->d_events.release();
    b->d_stage.release();
    b->d_converted.release();
    b->d_analysis.release();
    b->h_arena_stage.release();
    b->d_input.release();
    b->d_logic.release();
    b->h_logic.release();
    if (b->own_stream)
        (void)hipStreamDestroy(b->own_stream);
    if (b->slice_stream)
        (void)hipStreamDestroy(b->slice_stream);
    if (b->slice_forked)
        (void)hipEventDestroy(b->slice_forked);
    if (b->slice_joined)
        (void)hipEventDestroy(b->slice_joined);
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

int r433_batch_enable_logic_dump(r433_batch *b, int on)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    DeviceScope on_device(b->device);
    b->logic_on = on != 0;
    return 0;
}

int r433_batch_logic_dump(r433_batch *b, uint8_t const **host, uint64_t *stride)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (!b->logic_on || !b->h_logic.p)
        return fail(R433_EINVAL, "no logic dump: r433_batch_enable_logic_dump before the run");
    if (host)
        *host = b->h_logic.p;
    if (stride)
        *stride = b->logic_stride;
    return 0;
}

int r433_batch_set_staging_slot(r433_batch *b, uint32_t bytes)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (bytes && (bytes < 512u || bytes > 8192u))
        return fail(R433_EINVAL, "staging slots are 512 .. 8192 bytes (0: the default, 8192)");
    b->stage_slot = bytes & ~511u;
    return 0;
}

int r433_batch_set_exclusive_detect(r433_batch *b, int on)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    b->exclusive_detect = on < 0 ? 0 : on > 3 ? 3 : on;
    return 0;
}

int r433_batch_set_debug(r433_batch *b, uint32_t flags)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    b->debug_flags = flags;
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

int r433_batch_detect_form(r433_batch *b)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    return b->last_roles ? 45 : 0;
}

int r433_batch_set_profiling(r433_batch *b, int on)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    DeviceScope on_device(b->device);
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
    DeviceScope on_device(b->device);
    *t = b->last_timing;
    return 0;
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
    if (count && !b->events_counted) { // (a walk over the whole stream: only for a caller that asks)
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
    DeviceScope on_device(b->device);
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
