// dispatch.cpp -- the host-side decoder dispatch that mirrors the reference's run_ook_demods / run_fsk_demods +
// account_event (src/r_api.c:438-550, src/pulse_slicer.c:26-66), single- and multi-threaded, and the checksum plugin.
#include "host_common.hpp"

#include <chrono>

using namespace r433;

namespace {

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

namespace r433 {

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

} // namespace r433

extern "C" {

// ---- decoder dispatch ----

namespace {

struct DevStats {
    unsigned events = 0, ok = 0, messages = 0, fails[5] = {0, 0, 0, 0, 0};
};

thread_local r433_dispatch_info g_current;

// Replays packages [p0, p1).  Returns decoded event count or a negative error code.
int dispatch_range(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices, r433_package_fn pkg_cb, void *user,
        uint32_t p0, uint32_t p1, std::vector<DevStats> &stats, std::string &err, r433_dispatch_hooks const *hooks = nullptr)
{
    r433_bitbuffer *bits = (r433_bitbuffer *)calloc(1, sizeof(r433_bitbuffer));
    bool const want_pd = pkg_cb || (hooks && hooks->package_begin);
    r433_pulse_data *pd = want_pd ? (r433_pulse_data *)calloc(1, sizeof(r433_pulse_data)) : nullptr;
    uint8_t const *ev = b->h_events.p;
    uint8_t const *pk = b->h_pkg_blob.p;
    std::vector<uint32_t> first(n_devices, 0), count(n_devices, 0), touched, refs;
    int decoded = 0;
    int rc = 0;

    for (uint32_t pkg = p0; pkg < p1 && rc == 0; ++pkg) {
        r433_pkg_rec ph;
        memcpy(&ph, pk + b->h_rec_off.p[pkg], sizeof(ph));
        if (hooks && hooks->package_filter && !hooks->package_filter(hooks->user, &ph))
            continue; // not this caller's business (e.g. a frame it has replayed before): no decoder sees it
        if (want_pd) {
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
            if (pkg_cb)
                pkg_cb(user, ph.stream, ph.type, pd);
            // (r433_dispatch_current inside the package hooks: which package of the run this is; device / ordinal as the last event left them)
            g_current.stream = ph.stream;
            g_current.package = pkg;
            g_current.package_type = ph.type;
            g_current.start_ago = ph.start_ago;
            if (hooks && hooks->package_begin)
                hooks->package_begin(hooks->user, &ph, pd);
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
                    if (eh.num_rows == kPfStubRows) { // a bitbuffer the pre-filter proved refused (a decoder of a later level: kPfStub)
                        unsigned const code = std::min<unsigned>(eh.free_row, 4u);
                        if (hooks) { // (the hooks form moves the r_device's counters as it goes)
                            if (rd) {
                                rd->decode_events += 1;
                                rd->decode_fails[code] += 1;
                            }
                        }
                        else {
                            stats[dev].events += 1;
                            stats[dev].fails[code] += 1;
                        }
                        continue;
                    }
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
                    if (hooks) { // single-threaded: the r_device's counters move right here, as in account_event
                        if (rd) {
                            rd->decode_events += ds.events;
                            rd->decode_ok += ds.ok;
                            rd->decode_messages += ds.messages;
                            for (int f = 0; f < 5; ++f)
                                rd->decode_fails[f] += ds.fails[f];
                        }
                        ds = DevStats();
                        if (hooks->event_done)
                            hooks->event_done(hooks->user, rd, ret, bits);
                    }
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
        if (hooks && hooks->package_end && rc == 0)
            hooks->package_end(hooks->user, &ph, p_events);
        if (pkg < b->pkg_decoded.size())
            b->pkg_decoded[pkg] = p_events; // for the sample grabber's "known / unknown" modes
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
    b->pkg_decoded.assign(b->n_pkgs, 0);
    b->dispatched = true;
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
    apply_prefilter_counts(b, devices, n_devices);
    return decoded;
}

int r433_batch_dispatch_hooks(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices,
        r433_dispatch_hooks const *hooks)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    b->pkg_decoded.assign(b->n_pkgs, 0);
    b->dispatched = true;
    if (n_devices != b->timing.size())
        return fail(R433_EINVAL, "dispatch needs the %zu devices the engine was created with", b->timing.size());
    if (b->pf_ran && hooks && (hooks->event_done || hooks->package_filter))
        return fail(R433_EINVAL, "the last run dropped records on the device (r433_batch_probe_prefilter): no event_done hook or package_filter can see them");
    std::vector<DevStats> stats(n_devices);
    std::string err;
    static r433_dispatch_hooks const none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int const decoded = dispatch_range(b, devices, n_devices, nullptr, nullptr, 0, b->n_pkgs, stats, err, hooks ? hooks : &none);
    digest_publish();
    if (decoded < 0)
        return fail(decoded, "%s", err.c_str());
    apply_prefilter_counts(b, devices, n_devices);
    return decoded;
}

// ---- ordered multi-threaded replay: threads own decoders, outputs are committed in reference order ----

namespace {

struct Captured {
    uint32_t pkg, level_rank, dev, ordinal, seq;
    int is_log, log_level;
    struct data *payload;
};

struct CaptureCtx {
    std::vector<Captured> *out = nullptr;
    uint32_t level_rank = 0, seq = 0;
    void *(*render)(void *user, r433_r_device *device, void *data) = nullptr; // r433_dispatch_hooks::output_render
    void *render_user = nullptr;
};
thread_local CaptureCtx g_capture;

void capture_output(r433_r_device *decoder, struct data *payload)
{
    if (g_capture.out && g_capture.render) // the host renders what the decoder reported right here, beside the other decoders
        payload = (struct data *)g_capture.render(g_capture.render_user, decoder, payload);
    if (g_capture.out)
        g_capture.out->push_back({g_current.package, g_capture.level_rank, g_current.device, g_current.ordinal, g_capture.seq++, 0, 0, payload});
}

void capture_log(r433_r_device *decoder, int level, struct data *payload)
{
    (void)decoder;
    if (g_capture.out)
        g_capture.out->push_back({g_current.package, g_capture.level_rank, g_current.device, g_current.ordinal, g_capture.seq++, 1, level, payload});
}

void inflate_bits(r433_bitbuffer *bits, uint8_t const *rec, r433_evt_rec const &eh)
{
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
}

} // namespace

int r433_batch_dispatch_ordered(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices,
        r433_dispatch_hooks const *hooks, uint32_t n_threads)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (hooks && hooks->event_done)
        return fail(R433_EINVAL, "the ordered replay has no event_done hook: use r433_batch_dispatch_hooks");
    if (n_devices != b->timing.size())
        return fail(R433_EINVAL, "dispatch needs the %zu devices the engine was created with", b->timing.size());
    if (b->pf_ran && hooks && hooks->package_filter)
        return fail(R433_EINVAL, "the last run dropped records on the device (r433_batch_probe_prefilter): a package_filter cannot take their counts back");
    uint32_t const np = b->n_pkgs;
    b->pkg_decoded.assign(np, 0);
    b->dispatched = true;
    if (n_threads < 1)
        n_threads = 1;
    uint8_t const *ev = b->h_events.p;
    uint8_t const *pk = b->h_pkg_blob.p;
    bool const trace = (b->debug_flags & R433_DEBUG_DISPATCH_TRACE) != 0;
    auto const t_begin = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    struct CallEnd { // (declared first: runs after every other local has been given back)
        bool on;
        std::chrono::steady_clock::time_point t0;
        ~CallEnd()
        {
            if (on)
                fprintf(stderr, "r.dispatch: call returns after %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    } call_end{trace, t_begin};
    std::vector<double> dev_ms(trace ? n_devices : 0, 0.0);

    size_t const end = b->evt_bytes;
    uint32_t const *pkg_off = b->h_pkg_off.p;
    std::atomic<size_t> corrupt_at{SIZE_MAX};
    if (np && pkg_off[0] != 0)
        corrupt_at.store(0);
    // one record of the stream at `at`, checked: false = the stream is corrupt there
    auto record_at = [&](size_t at, size_t stop, r433_evt_rec &eh) -> bool {
        if (at + sizeof(eh) > stop) {
            corrupt_at.store(at);
            return false;
        }
        memcpy(&eh, ev + at, sizeof(eh));
        if (eh.dev >= n_devices || eh.pkg >= np || eh.total_bytes < sizeof(eh) || at + eh.total_bytes > stop) {
            corrupt_at.store(at);
            return false;
        }
        return true;
    };
    std::vector<uint32_t> pkg_stream(np), pkg_type(np), pkg_start_ago(np);
    std::vector<uint8_t> skipped(np, 0); // hooks->package_filter said no: no decoder sees the package, no hook is called for it
    for (uint32_t p = 0; p < np; ++p) {
        r433_pkg_rec ph;
        memcpy(&ph, pk + b->h_rec_off.p[p], sizeof(ph));
        pkg_stream[p] = ph.stream;
        pkg_type[p] = ph.type;
        pkg_start_ago[p] = ph.start_ago;
        if (hooks && hooks->package_filter && !hooks->package_filter(hooks->user, &ph))
            skipped[p] = 1;
    }

    // outputs go to the capture while the threads run
    std::vector<void (*)(r433_r_device *, struct data *)> keep_out(n_devices);
    std::vector<void (*)(r433_r_device *, int, struct data *)> keep_log(n_devices);
    for (uint32_t d = 0; d < n_devices; ++d) {
        if (!devices[d])
            continue;
        keep_out[d] = devices[d]->output_fn;
        keep_log[d] = devices[d]->log_fn;
        // only where the caller listens: a decoder checks these pointers before it builds a payload, and a payload
        // nobody takes at commit time would have no owner (data_t belongs to whoever output_fn hands it to)
        if (keep_out[d])
            devices[d]->output_fn = capture_output;
        if (keep_log[d])
            devices[d]->log_fn = capture_log;
    }
    struct PutBack { // whatever way this call ends, the decoders get their own output_fn / log_fn back
        std::function<void()> fn;
        bool done = false;
        void now()
        {
            if (!done)
                fn();
            done = true;
        }
        ~PutBack() { now(); }
    } put_back{[&]() {
        for (uint32_t d = 0; d < n_devices; ++d) {
            if (!devices[d])
                continue;
            devices[d]->output_fn = keep_out[d];
            devices[d]->log_fn = keep_log[d];
        }
    }};
    std::vector<std::vector<Captured>> captured(n_threads);
    std::vector<std::atomic<int>> p_events(np);
    for (auto &x : p_events)
        x.store(0, std::memory_order_relaxed);
    std::atomic<int> failed{0};
    std::string err;
    std::mutex err_m;

    // one bitbuffer to one decoder: what account_event does around decode_fn (src/pulse_slicer.c:26-66), on whichever thread
    struct Tally {
        unsigned n_ev = 0, n_ok = 0, n_msg = 0, fails[5] = {0, 0, 0, 0, 0};
    };
    auto call_one = [&](uint32_t dev, uint8_t const *rec, r433_evt_rec const &eh, r433_bitbuffer *bits, Tally &t) -> bool {
        r433_r_device *rd = devices[dev];
        if (eh.num_rows == kPfStubRows) { // a bitbuffer the pre-filter proved refused, of a decoder on a later level (kPfStub)
            t.n_ev += 1;
            t.fails[std::min<unsigned>(eh.free_row, 4u)] += 1;
            return true;
        }
        inflate_bits(bits, rec, eh);
        uint32_t used_rows = std::max<uint32_t>(eh.num_rows, eh.free_row);
        g_current.stream = pkg_stream[eh.pkg];
        g_current.package = eh.pkg;
        g_current.device = dev;
        g_current.ordinal = eh.ordinal;
        g_current.package_type = pkg_type[eh.pkg];
        g_current.start_ago = pkg_start_ago[eh.pkg];
        g_capture.seq = 0;
        int ret = 0;
        if (rd && rd->decode_fn)
            ret = rd->decode_fn(rd, bits);
        t.n_ev += 1;
        bool ok = true;
        if (ret > 0) {
            t.n_ok += 1;
            t.n_msg += (unsigned)ret;
            p_events[eh.pkg].fetch_add(ret, std::memory_order_relaxed);
        }
        else if (ret >= R433_DECODE_FAIL_SANITY) {
            t.fails[-ret] += 1;
        }
        else {
            std::lock_guard<std::mutex> g(err_m);
            char buf[200];
            snprintf(buf, sizeof(buf), "decoder \"%s\" gave invalid return value %d", rd && rd->name ? rd->name : "?", ret);
            err = buf;
            failed.store(1);
            ok = false;
        }
        used_rows = std::max<uint32_t>(used_rows, std::max<uint32_t>(bits->num_rows, bits->free_row));
        if (used_rows > R433_BITBUF_ROWS)
            used_rows = R433_BITBUF_ROWS;
        memset(bits->bb, 0, (size_t)used_rows * R433_BITBUF_COLS);
        memset(bits, 0, offsetof(r433_bitbuffer, bb));
        return ok;
    };
    auto book = [&](uint32_t dev, Tally const &t) { // (atomically: a stateless decoder's calls end on several threads)
        r433_r_device *rd = devices[dev];
        if (!rd || !t.n_ev)
            return;
        __atomic_fetch_add(&rd->decode_events, t.n_ev, __ATOMIC_RELAXED);
        __atomic_fetch_add(&rd->decode_ok, t.n_ok, __ATOMIC_RELAXED);
        __atomic_fetch_add(&rd->decode_messages, t.n_msg, __ATOMIC_RELAXED);
        for (int f = 0; f < 5; ++f)
            __atomic_fetch_add(&rd->decode_fails[f], t.fails[f], __ATOMIC_RELAXED);
    };

    // (Tried and dropped: serving the stateless decoders in ONE pass over the stream, every thread a stretch of packages,
    // every record to its decoder as it comes -- sequential reads instead of a cache miss per record.  It was twice as slow per
    // call, 171 ns against 75-100: a package's records go to 277 different decoders in turn, half a megabyte of decoder
    // code cycling through a 32 KB instruction cache, where the walk per decoder keeps one decoder's code and branch history
    // hot.  So: the walk per decoder, in stretches for the stateless ones, with the next records prefetched.)
    // Where a decoder's records lie.  With the slice index of the run (slicer_kernels.hip k_index_*: per decoder the (offset,
    // bytes) of its non-empty slices, made on the device) nothing is searched here: dev_count counts slices and an item walks
    // the records of its slices.  Without it (runs that made none) the host indexes every record itself.
    bool const by_slice = b->slices_valid && n_devices == b->timing.size();
    static uint32_t const pf_dist = [] { char const *e = getenv("R433_REPLAY_PF_DIST"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 24u; }(); // (development: A/B -- 6 / 12 / 24 / 40 slices ahead: 75 / 78 / 80.4 / 70 GS/s, profiles/r06_replay_prefetch.txt)
    static bool const pf_lines_all = [] { char const *e = getenv("R433_REPLAY_PREFETCH"); return !e || atoi(e) != 1; }(); // (development: 1 = the first line only, A/B)
    uint2 const *const slices = b->h_slices.p;
    std::vector<uint32_t> dev_count(n_devices + 1, 0);
    std::vector<uint32_t> ev_off;
    if (by_slice) {
        for (uint32_t d = 0; d <= n_devices; ++d)
            dev_count[d] = b->h_slice_start.p[d];
        if (dev_count[n_devices] != b->n_slices)
            return fail(R433_EHIP, "slice index: %u entries listed, %u made", dev_count[n_devices], b->n_slices);
    }
    else {
    // index: the events of every device, in package order (the stream is sorted by package, device, ordinal).  Built by
    // the pool over package ranges of about equal bytes (h_pkg_off delimits a package's events): every thread counts its
    // range per device, the counts are laid out device-major / range-minor, every thread fills its own slices.
    unsigned const n_parts = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<uint32_t>(n_threads, 64), end / (256u << 10)));
    std::vector<uint32_t> part_first(n_parts + 1, np);
    part_first[0] = 0;
    for (unsigned t = 1; t < n_parts; ++t) // the first package at or behind byte t / n_parts of the stream
        part_first[t] = (uint32_t)(std::lower_bound(pkg_off, pkg_off + np, (uint32_t)(end / n_parts * t)) - pkg_off);
    std::vector<uint32_t> part_count((size_t)n_parts * n_devices, 0);
    auto walk = [&](unsigned t, auto &&visit) {
        size_t at = part_first[t] < np ? pkg_off[part_first[t]] : end;
        size_t const stop = part_first[t + 1] < np ? pkg_off[part_first[t + 1]] : end;
        while (at < stop) {
            r433_evt_rec eh;
            if (!record_at(at, stop, eh))
                return;
            visit(eh, at);
            at += eh.total_bytes;
        }
    };
    b->pool.run(n_parts, [&](unsigned t) {
        uint32_t *mine = part_count.data() + (size_t)t * n_devices;
        walk(t, [&](r433_evt_rec const &eh, size_t) { mine[eh.dev]++; });
    });
    if (corrupt_at.load() != SIZE_MAX || (np == 0 && end != 0))
        return fail(R433_EHIP, "corrupt event stream at byte %zu", corrupt_at.load() == SIZE_MAX ? (size_t)0 : corrupt_at.load());
    {
        uint32_t run = 0;
        for (uint32_t d = 0; d < n_devices; ++d) {
            dev_count[d] = run;
            for (unsigned t = 0; t < n_parts; ++t) {
                uint32_t const c = part_count[(size_t)t * n_devices + d];
                part_count[(size_t)t * n_devices + d] = run; // from a count to the slice's first slot
                run += c;
            }
        }
        dev_count[n_devices] = run;
    }
    ev_off.resize(dev_count[n_devices]);
    b->pool.run(n_parts, [&](unsigned t) {
        uint32_t *fill = part_count.data() + (size_t)t * n_devices;
        walk(t, [&](r433_evt_rec const &eh, size_t at) { ev_off[fill[eh.dev]++] = (uint32_t)at; });
    });
    }
    if (!by_slice) {
        b->n_events = dev_count[n_devices]; // (the index has walked and checked the whole stream: r433_batch_events need not count again)
        b->events_counted = true;
    }
    if (trace)
        fprintf(stderr, "r.dispatch: engine %p index of %u %s over %u packages %.3f ms\n", (void *)b, dev_count[n_devices], by_slice ? "slices (made on the device)" : "records",
                np, since(t_begin));
    std::atomic<uint64_t> walked{0}; // records met on the way through the slices (every decoder's, whatever the levels let through)
    for (uint32_t li = 0; li < b->prio_levels.size() && !failed.load(); ++li) {
        auto const t_level = std::chrono::steady_clock::now();
        uint32_t const level = b->prio_levels[li];
        std::vector<uint32_t> devs_of_level;
        for (uint32_t d = 0; d < n_devices; ++d)
            if (b->timing[d].priority == level && dev_count[d + 1] > dev_count[d])
                devs_of_level.push_back(d);
        // Work items: a decoder's records in package order -- all of them for a decoder that may keep state between calls,
        // stretches of them for one the host declared stateless (r433_batch_set_stateless).  Heaviest first: the pass ends
        // when the last thread does.
        struct Item {
            uint32_t dev, first, last;
        };
        std::vector<Item> items;
        // records per item of a stateless decoder: a third of a millisecond of a cheap decoder -- small enough that the pass ends
        // with every thread busy (tests: R433_DEBUG_SMALL_STRETCH, a handful)
        uint32_t const kStretch = (b->debug_flags & R433_DEBUG_SMALL_STRETCH) ? 5u : by_slice ? 1536u : 6144u; // (a slice holds ~4 records)
        for (uint32_t d : devs_of_level) {
            uint32_t const first = dev_count[d], last = dev_count[d + 1];
            bool const split = d < b->stateless.size() && b->stateless[d] == 1 && last - first > kStretch + kStretch / 2;
            if (!split) {
                items.push_back({d, first, last});
                continue;
            }
            uint32_t const parts = (last - first + kStretch - 1) / kStretch;
            for (uint32_t k = 0; k < parts; ++k)
                items.push_back({d, first + (uint32_t)((uint64_t)(last - first) * k / parts), first + (uint32_t)((uint64_t)(last - first) * (k + 1) / parts)});
        }
        std::stable_sort(items.begin(), items.end(), [](Item const &x, Item const &y) { return x.last - x.first > y.last - y.first; });
        // a package whose lower levels produced an event is closed for this level (src/r_api.c:442)
        // (the lowest level sees every package: its own decoders' events do not close it)
        std::vector<uint8_t> open(np);
        for (uint32_t p = 0; p < np; ++p)
            open[p] = !skipped[p] && (li == 0 || p_events[p].load(std::memory_order_relaxed) == 0);
        std::atomic<uint32_t> cursor{0};
        uint32_t const nt = std::max<uint32_t>(1, std::min<uint32_t>(n_threads, (uint32_t)items.size()));
        b->pool.run(nt, [&](unsigned w) {
            r433_bitbuffer *bits = (r433_bitbuffer *)calloc(1, sizeof(r433_bitbuffer));
            g_capture.out = &captured[w];
            g_capture.level_rank = li;
            g_capture.render = hooks ? hooks->output_render : nullptr;
            g_capture.render_user = hooks ? hooks->user : nullptr;
            for (;;) {
                uint32_t const k = cursor.fetch_add(1, std::memory_order_relaxed);
                if (k >= items.size() || failed.load(std::memory_order_relaxed))
                    break;
                uint32_t const dev = items[k].dev;
                auto const t_dev = std::chrono::steady_clock::now();
                Tally t;
                bool go_on = true;
                uint32_t walked_here = 0;
                for (uint32_t e = items[k].first; e < items[k].last && go_on; ++e) {
                    // a decoder's records lie a package's worth of other decoders' apart: every slice a cache miss
                    // (... and a slice is 4-5 records, half a kilobyte: every line of it, not the first alone)
                    if (e + pf_dist < items[k].last) {
                        if (by_slice) {
                            uint2 const nx = slices[e + pf_dist];
                            uint8_t const *const q = ev + nx.x;
                            uint32_t const nb = pf_lines_all ? std::min<uint32_t>(nx.y, 768u) : 1u;
                            for (uint32_t o = 0; o < nb; o += 64)
                                __builtin_prefetch(q + o);
                        }
                        else
                            __builtin_prefetch(ev + ev_off[e + pf_dist]);
                    }
                    if (!by_slice) {
                        uint8_t const *rec = ev + ev_off[e];
                        r433_evt_rec eh;
                        memcpy(&eh, rec, sizeof(eh));
                        if (open[eh.pkg])
                            go_on = call_one(dev, rec, eh, bits, t);
                        continue;
                    }
                    size_t at = slices[e].x;
                    size_t const stop = std::min<size_t>(at + slices[e].y, end);
                    while (at < stop && go_on) { // the records of one (package, decoder) slice
                        walked_here += 1;
                        r433_evt_rec eh;
                        if (!record_at(at, stop, eh) || eh.dev != dev) {
                            corrupt_at.store(at);
                            go_on = false;
                            break;
                        }
                        if (open[eh.pkg])
                            go_on = call_one(dev, ev + at, eh, bits, t);
                        at += eh.total_bytes;
                    }
                }
                if (trace) {
                    std::lock_guard<std::mutex> g(err_m);
                    dev_ms[dev] += since(t_dev);
                }
                walked.fetch_add(walked_here, std::memory_order_relaxed);
                book(dev, t);
            }
            g_capture.out = nullptr;
            g_capture.render = nullptr;
            digest_publish();
            free(bits);
        });
        if (corrupt_at.load() != SIZE_MAX)
            break; // (decoders have run: what they handed out is committed below, in order, before the call reports the stream)
        if (trace)
            fprintf(stderr, "r.dispatch: level %u, %zu decoders in %zu items on %u threads %.3f ms\n", level, devs_of_level.size(), items.size(), nt, since(t_level));
    }
    if (trace) {
        std::vector<uint32_t> by(n_devices);
        double sum = 0;
        for (uint32_t d = 0; d < n_devices; ++d)
            by[d] = d, sum += dev_ms[d];
        std::sort(by.begin(), by.end(), [&](uint32_t x, uint32_t y) { return dev_ms[x] > dev_ms[y]; });
        fprintf(stderr, "r.dispatch: decoder time in all %.3f ms; slowest:", sum);
        for (uint32_t k = 0; k < std::min<uint32_t>(8, n_devices); ++k)
            fprintf(stderr, " %s %.3f ms / %u calls;", devices[by[k]] && devices[by[k]]->name ? devices[by[k]]->name : "?", dev_ms[by[k]],
                    dev_count[by[k] + 1] - dev_count[by[k]]);
        fprintf(stderr, "\n");
    }
    if (by_slice && !failed.load() && corrupt_at.load() == SIZE_MAX) { // every slice has been walked once: that was every record of the stream
        b->n_events = (uint32_t)walked.load();
        b->events_counted = true;
    }
    auto const t_commit = std::chrono::steady_clock::now();
    put_back.now();

    // commit: what the decoders handed out, in the order the single-threaded replay produces it
    std::vector<Captured> all;
    for (auto &c : captured)
        all.insert(all.end(), c.begin(), c.end());
    std::sort(all.begin(), all.end(), [](Captured const &x, Captured const &y) {
        if (x.pkg != y.pkg) return x.pkg < y.pkg;
        if (x.level_rank != y.level_rank) return x.level_rank < y.level_rank;
        if (x.dev != y.dev) return x.dev < y.dev;
        if (x.ordinal != y.ordinal) return x.ordinal < y.ordinal;
        return x.seq < y.seq;
    });
    bool const want_pkgs = hooks && (hooks->package_begin || hooks->package_end);
    r433_pulse_data *pd = want_pkgs && hooks->package_begin ? (r433_pulse_data *)calloc(1, sizeof(r433_pulse_data)) : nullptr;
    size_t ci = 0;
    int decoded = 0;
    for (uint32_t p = 0; p < np; ++p) {
        int const pe = p_events[p].load(std::memory_order_relaxed);
        b->pkg_decoded[p] = pe;
        decoded += pe;
        bool const has_out = ci < all.size() && all[ci].pkg == p;
        if (skipped[p] || (!want_pkgs && !has_out))
            continue;
        r433_pkg_rec ph;
        memcpy(&ph, pk + b->h_rec_off.p[p], sizeof(ph));
        if (pd) {
            memset(pd, 0, sizeof(*pd));
            pd->offset = ph.offset;
            pd->sample_rate = ph.sample_rate;
            pd->start_ago = ph.start_ago;
            pd->end_ago = ph.end_ago;
            pd->num_pulses = ph.num_pulses;
            int32_t const *pairs = (int32_t const *)(pk + b->h_rec_off.p[p] + sizeof(ph));
            for (uint32_t i = 0; i < ph.num_pulses && i < R433_MAX_PULSES; ++i) {
                pd->pulse[i] = pairs[2 * i];
                pd->gap[i] = pairs[2 * i + 1];
            }
            pd->ook_low_estimate = ph.ook_low;
            pd->ook_high_estimate = ph.ook_high;
            pd->fsk_f1_est = ph.fsk_f1;
            pd->fsk_f2_est = ph.fsk_f2;
            fill_levels(b->cfg, *pd);
            g_current.stream = ph.stream;
            g_current.package = p;
            g_current.package_type = ph.type;
            g_current.start_ago = ph.start_ago;
            hooks->package_begin(hooks->user, &ph, pd);
        }
        for (; ci < all.size() && all[ci].pkg == p; ++ci) {
            Captured const &c = all[ci];
            r433_r_device *rd = devices[c.dev];
            g_current.stream = ph.stream;
            g_current.package = p;
            g_current.device = c.dev;
            g_current.ordinal = c.ordinal;
            g_current.package_type = ph.type;
            g_current.start_ago = ph.start_ago;
            if (c.is_log) {
                if (rd && rd->log_fn)
                    rd->log_fn(rd, c.log_level, c.payload);
            }
            else if (rd && rd->output_fn) {
                rd->output_fn(rd, c.payload);
            }
        }
        if (hooks && hooks->package_end && !failed.load()) {
            g_current.stream = ph.stream;
            g_current.package = p;
            g_current.package_type = ph.type;
            g_current.start_ago = ph.start_ago;
            hooks->package_end(hooks->user, &ph, pe);
        }
    }
    free(pd);
    if (trace)
        fprintf(stderr, "r.dispatch: commit %.3f ms, whole replay %.3f ms\n", since(t_commit), since(t_begin));
    if (failed.load())
        return fail(R433_EDECODER, "%s", err.c_str());
    if (corrupt_at.load() != SIZE_MAX) // (nothing a decoder made is left behind: its outputs went out above, its statistics are booked)
        return fail(R433_EHIP, "corrupt event stream at byte %zu", corrupt_at.load());
    apply_prefilter_counts(b, devices, n_devices);
    return decoded;
}

int r433_batch_set_stateless(r433_batch *b, uint8_t const *stateless, uint32_t n_devices)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (!stateless) {
        b->stateless.clear();
        return 0;
    }
    if (n_devices != b->timing.size())
        return fail(R433_EINVAL, "one flag for each of the %zu devices the engine was created with", b->timing.size());
    b->stateless.assign(stateless, stateless + n_devices);
    return 0;
}

int r433_batch_decoded(r433_batch *b, int const **per_package, uint32_t *count)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (!b->dispatched)
        return fail(R433_EINVAL, "no dispatch since the last run");
    if (per_package)
        *per_package = b->pkg_decoded.data();
    if (count)
        *count = (uint32_t)b->pkg_decoded.size();
    return 0;
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

} // extern "C"
