// reports.cpp -- host-side text and report formats around the path: the pulse analyzer's report (`-A`) over the
// device-computed analysis, and the `.ook` pulse-data writer (the reader is pulse_text.cpp).
#include "host_common.hpp"

using namespace r433;

extern "C" {

int r433_batch_analyze(r433_batch *b, r433_analysis *out, uint32_t max_packages, void *stream)
{
    if (!b || (!out && max_packages))
        return fail(R433_EINVAL, "null argument");
    DeviceScope on_device(b->device);
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

// The sample grabber's bookkeeping (`-S`, reference src/r_flow.c:136-147, 246-252, 342-362 and samp_grab_write,
// src/samp_grab.c:100-165) replayed over the package records of the last run: which byte ranges of which capture the
// reference would have written to its g###_<freq>_<rate> files.  The ring buffer is taken per capture (every capture
// starts from r_init_cfg state like everything else in a batch); where the reference would read ring memory that was
// never written (a window reaching before the start of the input) the range is clipped and flagged.
int r433_batch_grab_plan(r433_batch *b, int grab_mode, r433_grab *out, uint32_t max_grabs)
{
    if (!b || (!out && max_grabs))
        return fail(R433_EINVAL, "null argument");
    DeviceScope on_device(b->device);
    if (grab_mode < 1 || grab_mode > 4)
        return fail(R433_EINVAL, "grab mode must be 1 (all), 2 (unknown), 3 (known) or 4 (undecoded)");
    if (grab_mode == 4 && b->pkg_quality.size() != b->n_pkgs)
        return fail(R433_EINVAL, "grab mode 4 needs the analyzer's verdict on every package: r433_batch_set_package_quality first");
    if (grab_mode != 1 && !b->dispatched)
        return fail(R433_EINVAL, "grab modes 2 and 3 need the decode results: dispatch first");
    if (b->stream_samples.size() != b->n_streams)
        return fail(R433_EINVAL, "the grabber follows a detection run (r433_batch_run), not a pulse-data run");
    uint32_t const ss = b->cfg.sample_size, F = b->cfg.frame_samples;
    constexpr uint64_t kRing = 12ull * 262144ull; // SIGNAL_GRABBER_BUFFER, include/rtl_433.h:22
    constexpr uint64_t kBlock = 128 * 1024;       // src/samp_grab.c:98
    uint32_t n_out = 0, counter = 1;
    uint32_t pkg = 0;
    for (uint32_t c = 0; c < b->n_streams; ++c) {
        uint64_t const n_total = b->stream_samples.size() > c ? b->stream_samples[c] : 0;
        uint32_t const data_calls = (uint32_t)((n_total + F - 1) / F);
        unsigned start_ago = 0, end_ago = 0, event_count = 0;
        int quality = 0; // demod->frame_quality: the best analyzer verdict among the frame's packages nobody decoded
        uint64_t pushed = 0; // bytes
        for (uint32_t call = 0; call <= data_calls; ++call) { // the last one is the flush (len 0)
            uint32_t const n = call < data_calls ? (uint32_t)std::min<uint64_t>(F, n_total - (uint64_t)call * F) : 0u;
            pushed += (uint64_t)n * ss;
            if (start_ago)
                start_ago += n;
            if (end_ago)
                end_ago += n;
            for (; pkg < b->n_pkgs; ++pkg) { // packages are in (capture, detection order)
                r433_pkg_rec ph;
                memcpy(&ph, b->h_pkg_blob.p + b->h_rec_off.p[pkg], sizeof(ph));
                bool const flushed = ph.ret_pos == R433_RET_FLUSH;
                if (ph.stream != c || (flushed ? call < data_calls : ph.frame != call))
                    break;
                if (!start_ago)
                    start_ago = ph.start_ago;
                end_ago = ph.end_ago;
                if (pkg < b->pkg_decoded.size())
                    event_count += (unsigned)b->pkg_decoded[pkg];
                if (grab_mode == 4 && b->pkg_decoded[pkg] == 0) // src/r_flow.c:290-294,308-312
                    quality = std::max(quality, (int)b->pkg_quality[pkg]);
            }
            if (start_ago && end_ago > n) { // the frame is older than a whole buffer: it is over
                if (grab_mode == 1 || (grab_mode == 2 && event_count == 0) || (grab_mode == 3 && event_count > 0)
                        || (grab_mode == 4 && event_count == 0 && quality > 0)) {
                    unsigned const pad = n / 8;
                    unsigned const start_padded = start_ago + pad, end_padded = end_ago - pad;
                    unsigned const len_padded = start_padded - end_padded;
                    uint64_t bsize = (uint64_t)ss * len_padded;
                    bsize += kBlock - bsize % kBlock;
                    bsize = std::min<uint64_t>(bsize, std::min<uint64_t>(pushed, kRing));
                    uint64_t const end_byte = pushed - std::min<uint64_t>(pushed, (uint64_t)ss * end_padded);
                    uint64_t const want_start = end_byte >= bsize ? end_byte - bsize : 0;
                    if (n_out < max_grabs) {
                        r433_grab &g = out[n_out];
                        g.stream = c;
                        g.counter = counter;
                        g.byte_offset = want_start;
                        g.byte_len = end_byte - want_start;
                        g.n_samples = len_padded;
                        g.clipped = end_byte < bsize ? 1u : 0u;
                        g.pushed = pushed;
                    }
                    n_out += 1;
                    counter += 1;
                }
                start_ago = 0;
                event_count = 0;
                quality = 0;
            }
        }
    }
    return (int)n_out;
}

int r433_batch_set_package_quality(r433_batch *b, int32_t const *quality, uint32_t n_packages)
{
    if (!b || (!quality && n_packages))
        return fail(R433_EINVAL, "null argument");
    if (n_packages != b->n_pkgs)
        return fail(R433_EINVAL, "one verdict per package of the last run (%u given, %u packages)", n_packages, b->n_pkgs);
    b->pkg_quality.assign(quality, quality + n_packages);
    return 0;
}

namespace {

// one 512-byte tar member header as microtar writes it (reference src/microtar.c:140-170, 388-398): mode 0664, owner,
// group, mtime and device numbers zero, type '0', "ustar" "00", checksum over the header with its own field as spaces
void tar_header(uint8_t *h, char const *name, uint64_t size)
{
    memset(h, 0, 512);
    snprintf((char *)h, 100, "%s", name);
    snprintf((char *)h + 100, 8, "%07o", 0664);
    snprintf((char *)h + 108, 8, "%07o", 0);
    snprintf((char *)h + 116, 8, "%07o", 0);
    snprintf((char *)h + 124, 12, "%011o", (unsigned)size);
    snprintf((char *)h + 136, 12, "%011o", 0u);
    h[156] = '0';
    memcpy(h + 257, "ustar", 6);
    memcpy(h + 263, "00", 2);
    snprintf((char *)h + 329, 8, "%07o", 0);
    snprintf((char *)h + 337, 8, "%07o", 0);
    unsigned sum = 256;
    for (int k = 0; k < 148; ++k)
        sum += h[k];
    for (int k = 156; k < 512; ++k)
        sum += h[k];
    snprintf((char *)h + 148, 8, "%07o", sum);
    h[155] = ' ';
}

} // namespace

// The SigMF container of a grabbed signal (`-S sigmf:...`, reference src/samp_grab.c:166-232, src/sigmf.c:290-325,
// 441-494): everything that precedes the data -- the archive's meta member and the header of the data member.
int r433_sigmf_prefix(uint32_t sample_size, uint32_t sample_rate, uint32_t frequency, uint64_t data_len, uint8_t *buf, size_t cap)
{
    if (!buf && cap)
        return fail(R433_EINVAL, "null argument");
    if (sample_size != 2 && sample_size != 4)
        return fail(R433_EINVAL, "sample_size must be 2 (cu8) or 4 (cs16)");
    char json[1024] = {0};
    snprintf(json, sizeof(json),
            "{    \"global\" : {        \"core:datatype\" : \"%s\",        \"core:sample_rate\" : %u,        \"core:recorder\" : \"%s\","
            "        \"core:version\" : \"1.0.0\"    },    \"captures\" : [        {            \"core:sample_start\" : %u,"
            "            \"core:frequency\" : %u        }    ],    \"annotations\" : []}",
            sample_size == 2 ? "cu8" : "ci16_le", sample_rate, "rtl_433", 0u, frequency);
    size_t const json_len = strlen(json);
    size_t const meta_padded = (json_len + 511) / 512 * 512;
    size_t const total = 512 + meta_padded + 512;
    if (cap >= total) {
        memset(buf, 0, total);
        tar_header(buf, "foobar.sigmf-meta", json_len);
        memcpy(buf + 512, json, json_len);
        tar_header(buf + 512 + meta_padded, "foobar.sigmf-data", data_len);
    }
    return (int)total;
}

// ... and what follows it: padding to the next 512-byte record and the two null records that end the archive.
int r433_sigmf_trailer(uint64_t data_len, uint8_t *buf, size_t cap)
{
    if (!buf && cap)
        return fail(R433_EINVAL, "null argument");
    size_t const total = (size_t)((512 - data_len % 512) % 512) + 1024;
    if (cap >= total)
        memset(buf, 0, total);
    return (int)total;
}

// sigmf_reader_open, reference src/sigmf.c:336-434, over an archive in memory: the first stream's meta data (the keys
// json_parse reads, :127-286) and where its samples sit.  Like the reference, whatever `core:datatype` says, the file
// loop then treats the samples as cu8 (src/rtl_433.c:1719).
int r433_sigmf_probe(uint8_t const *buf, size_t len, r433_sigmf_info *info)
{
    if (!buf || !info)
        return fail(R433_EINVAL, "null argument");
    memset(info, 0, sizeof(*info));
    auto octal = [](uint8_t const *p, size_t n) {
        uint64_t v = 0;
        for (size_t k = 0; k < n && p[k] >= '0' && p[k] <= '7'; ++k)
            v = v * 8 + (uint64_t)(p[k] - '0');
        return v;
    };
    auto has_ext = [](char const *name, char const *ext) {
        size_t const a = strlen(name), b = strlen(ext);
        return a >= b && !strcmp(name + a - b, ext);
    };
    char stream[101] = {0};
    size_t pos = 0;
    bool found_data = false;
    while (pos + 512 <= len && buf[pos] != 0) { // a null record ends the archive
        uint8_t const *h = buf + pos;
        char name[101] = {0};
        memcpy(name, h, 100);
        uint64_t const size = octal(h + 124, 11);
        size_t const data_at = pos + 512;
        if (data_at + size > len)
            return fail(R433_EINVAL, "SigMF archive is cut short");
        bool const regular = h[156] == '0' || h[156] == 0;
        if (regular && has_ext(name, ".sigmf-meta") && !stream[0]) {
            snprintf(stream, sizeof(stream), "%s", name);
            std::string const json((char const *)buf + data_at, (size_t)size);
            auto value_after = [&](char const *key) -> char const * {
                size_t const k = json.find(std::string("\"") + key + "\"");
                if (k == std::string::npos)
                    return nullptr;
                size_t c = json.find(':', k + strlen(key) + 2);
                if (c == std::string::npos)
                    return nullptr;
                c += 1;
                while (c < json.size() && (json[c] == ' ' || json[c] == '\t' || json[c] == '\n' || json[c] == '\r'))
                    c += 1;
                return json.c_str() + c;
            };
            if (char const *v = value_after("core:datatype")) {
                if (*v == '"') {
                    size_t n = 0;
                    for (++v; v[n] && v[n] != '"' && n + 1 < sizeof(info->datatype); ++n)
                        info->datatype[n] = v[n];
                }
            }
            if (char const *v = value_after("core:sample_rate"))
                info->sample_rate = (uint32_t)strtod(v, nullptr);
            if (char const *v = value_after("core:frequency"))
                info->frequency = (uint32_t)strtod(v, nullptr);
            if (char const *v = value_after("core:sample_start"))
                info->sample_start = (uint32_t)strtod(v, nullptr);
        }
        else if (regular && stream[0] && !found_data && has_ext(name, ".sigmf-data")
                && !strncmp(name, stream, strlen(stream) - 4)) { // "<stream>.sigmf-" + "data"
            info->data_offset = data_at;
            info->data_len = size;
            found_data = true;
        }
        pos = data_at + (size_t)((size + 511) / 512 * 512);
    }
    if (!stream[0])
        return fail(R433_EINVAL, "SigMF input with no streams");
    if (!found_data)
        return fail(R433_EINVAL, "SigMF input with no stream data");
    return 0;
}

// pulse_data_print_vcd_header, reference src/pulse_data.c:77-100 (nice_freq: src/r_util.c:290-307)
int r433_pulse_vcd_header(uint32_t sample_rate, char const *date, char *buf, size_t cap)
{
    if (!buf && cap)
        return fail(R433_EINVAL, "null argument");
    char freq[30];
    double const f = sample_rate;
    if (f >= 1E9)
        snprintf(freq, sizeof(freq), "%.3fGHz", f / 1E9);
    else if (f >= 1E6)
        snprintf(freq, sizeof(freq), "%.3fMHz", f / 1E6);
    else if (f >= 1E3)
        snprintf(freq, sizeof(freq), "%.3fkHz", f / 1E3);
    else
        snprintf(freq, sizeof(freq), "%f", f);
    int const n = snprintf(buf, cap,
            "$date %s $end\n$version rtl_433 0.1.0 $end\n$comment Acquisition at %s Hz $end\n$timescale %s $end\n"
            "$scope module rtl_433 $end\n$var wire 1 / FRAME $end\n$var wire 1 ' AM $end\n$var wire 1 \" FM $end\n"
            "$upscope $end\n$enddefinitions $end\n#0 0/ 0' 0\"\n",
            date ? date : "", freq, sample_rate <= 500000 ? "1 us" : "100 ns");
    return n;
}

// pulse_data_print_vcd, reference src/pulse_data.c:102-120; ch_id is '\'' for an OOK package, '"' for an FSK one
int r433_pulse_vcd(r433_pulse_data const *data, int ch_id, char *buf, size_t cap)
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
    float scale; // (sic) integer division, then float
    if (data->sample_rate <= 500000)
        scale = (float)(1000000 / data->sample_rate);
    else
        scale = (float)(10000000 / data->sample_rate);
    uint64_t pos = data->offset;
    for (unsigned n = 0; n < data->num_pulses && n < R433_MAX_PULSES; ++n) {
        if (n == 0)
            PUT("#%.f 1/ 1%c\n", pos * scale, ch_id);
        else
            PUT("#%.f 1%c\n", pos * scale, ch_id);
        pos += (uint64_t)data->pulse[n];
        PUT("#%.f 0%c\n", pos * scale, ch_id);
        pos += (uint64_t)data->gap[n];
    }
    if (data->num_pulses > 0)
        PUT("#%.f 0/\n", pos * scale);
#undef PUT
    return (int)len;
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

} // extern "C"
