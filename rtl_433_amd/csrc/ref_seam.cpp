// ref_seam.cpp -- librtl433seam.so: the reference's OWN function names and prototypes for the hot path's
// function-level seam, so that rtl_433 can be linked with src/baseband.c, src/pulse_slicer.c (and src/pulse_detect*.c)
// left out (dropin/Makefile SEAM=1) and the reference's tests/baseband-test.c links unchanged:
//
//   include/baseband.h:27-143      envelope_detect, envelope_detect_nolut, magnitude_est_cu8 / _cs16, magnitude_true_cu8 /
//                                  _cs16, baseband_low_pass_filter(_reset), baseband_demod_FM(_cs16)(_reset), baseband_init
//   include/pulse_slicer.h:38-184  the ten pulse_slicer_* and pulse_slicer_string
//   include/pulse_detect.h:37-52   pulse_detect_create / _free / _reset / _set_levels (the object rtl_433 creates at
//                                  start-up and configures from -Y options)
//   include/pulse_detect.h:71      pulse_detect_package: the reference's resumable one-package-per-call contract over
//                                  r433_detector_package (the exact state machine on the device, one wavefront, sample by
//                                  sample: it completes the seam -- the reference's own src/r_flow.c links and decodes
//                                  over it, `make -C dropin refflow` -- and is no fast path; that is push_sdr_flow)
//
//   include/r_api.h:50-52          run_ook_demods / run_fsk_demods (src/r_api.c:438-550): the decoder fan-out of ONE package as ONE
//                                  launch -- every registered decoder's slicer over the package (r433_batch_run_pulses) and
//                                  the replay in the reference's order (priority levels, registration order, account_event:
//                                  r433_batch_dispatch_hooks) -- where the reference's own loop over the ten pulse_slicer_*
//                                  exports above is one launch and one round trip per DECODER
//
// Every function is a thin host wrapper: host pointers in, the work on the GPU through librtl433hip.so's C ABI
// (r433_envelope_host, r433_filter_frame, r433_batch_run_pulses + r433_batch_dispatch), host pointers out.  Nothing is
// computed here.  A per-call round trip over PCIe makes these slower than the CPU code they replace -- they exist so the
// seam is complete; the fast path is the batch entry (dropin/r_flow_hip.c).
//
//   include/pulse_detect_fsk.h:46-75  pulse_detect_fsk_init / _classic / _minmax / _wrap_up over r433_fsk_step: one sample,
//                                  one wavefront, one round trip per call (in the reference only pulse_detect_package calls
//                                  them; exported so that the four headers are complete)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "r433_hip.h"

extern "C" {

// ---- the reference's types, as far as these prototypes need them (layouts: include/baseband.h:91-107) ----
typedef struct filter_state {
    int16_t y[1];
    int16_t x[1];
} filter_state_t;

typedef struct demodfm_state {
    int32_t xr, xi, xf, yf;
    uint32_t rate;
    int32_t alp_16[2], blp_16[2];
    int64_t alp_32[2], blp_32[2];
} demodfm_state_t;

typedef r433_pulse_data pulse_data_t; // include/pulse_data.h:30-50 (sizes asserted in tests/test_abi.py)
typedef r433_fsk_state pulse_detect_fsk_t; // include/pulse_detect_fsk.h:23-41
typedef r433_r_device r_device;       // include/r_device.h:59-92
typedef r433_bitbuffer bitbuffer_t;   // include/bitbuffer.h:34-40

// reference functions this library calls back when the host program has them (weak: tests/baseband-test.c has neither)
void decoder_log_bitbuffer(r_device *decoder, int level, char const *func, const bitbuffer_t *bitbuffer, char const *msg) __attribute__((weak));
void bitbuffer_parse(bitbuffer_t *bits, const char *code) __attribute__((weak));
void print_logf(int level, char const *src, char const *fmt, ...) __attribute__((weak)); // include/logger.h:66 (log_level_t is an int-sized enum)

static void die(char const *what)
{
    fprintf(stderr, "librtl433seam: %s: %s\n", what, r433_last_error());
    exit(1);
}

// ---- include/baseband.h ----

void baseband_init(void)
{
    // the reference fills its (127 - i)^2 table here (src/baseband.c:22-32,368-371); the kernels evaluate it
}

static float envelope(uint32_t kind, void const *iq, uint16_t *y, uint32_t len, int is_mag)
{
    uint32_t sum = 0;
    if (r433_envelope_host(kind, iq, y, len, &sum) < 0)
        die("envelope");
    return r433_level_db(sum, len, is_mag);
}

float envelope_detect(uint8_t const *iq_buf, uint16_t *y_buf, uint32_t len)
{
    return envelope(R433_ENV_AMP_CU8, iq_buf, y_buf, len, 0);
}

float envelope_detect_nolut(uint8_t const *iq_buf, uint16_t *y_buf, uint32_t len)
{
    return envelope(R433_ENV_AMP_CU8, iq_buf, y_buf, len, 0); // the same values without the table (src/baseband.c:50-61)
}

float magnitude_est_cu8(uint8_t const *iq_buf, uint16_t *y_buf, uint32_t len)
{
    return envelope(R433_ENV_MAG_CU8, iq_buf, y_buf, len, 1);
}

float magnitude_true_cu8(uint8_t const *iq_buf, uint16_t *y_buf, uint32_t len)
{
    return envelope(R433_ENV_TRUE_CU8, iq_buf, y_buf, len, 1);
}

float magnitude_est_cs16(int16_t const *iq_buf, uint16_t *y_buf, uint32_t len)
{
    return envelope(R433_ENV_MAG_CS16, iq_buf, y_buf, len, 1);
}

float magnitude_true_cs16(int16_t const *iq_buf, uint16_t *y_buf, uint32_t len)
{
    return envelope(R433_ENV_TRUE_CS16, iq_buf, y_buf, len, 1);
}

void baseband_low_pass_filter_reset(filter_state_t *lowpass_filter)
{
    memset(lowpass_filter, 0, sizeof(*lowpass_filter));
}

void baseband_low_pass_filter(filter_state_t *state, uint16_t const *x_buf, int16_t *y_buf, uint32_t len)
{
    if (len < 1)
        return;
    r433_filter_carry c;
    memset(&c, 0, sizeof(c));
    c.am_y = state->y[0];
    c.am_x = state->x[0];
    if (r433_filter_frame(R433_FILTER_AM, x_buf, len, y_buf, &c, 0, 0, 0, 0) < 0)
        die("baseband_low_pass_filter");
    state->y[0] = (int16_t)c.am_y;
    state->x[0] = (int16_t)c.am_x; // the u16 envelope through an int16 slot, like the reference's memcpy (src/baseband.c:166-168)
}

void baseband_demod_FM_reset(demodfm_state_t *demod_fm)
{
    memset(demod_fm, 0, sizeof(*demod_fm));
}

void baseband_demod_FM(demodfm_state_t *state, uint8_t const *x_buf, int16_t *y_buf, unsigned long num_samples, uint32_t samp_rate, float low_pass)
{
    if (state->rate != samp_rate) { // coefficient selection, src/baseband.c:217-232 (host doubles, same libm)
        if (low_pass > 1e4f)
            low_pass = low_pass / samp_rate;
        else if (low_pass >= 1.0f)
            low_pass = 1e6f / low_pass / samp_rate;
        if (print_logf) // what a -vv user is told at this point (src/baseband.c:222-223), LOG_NOTICE = 5
            print_logf(5, "Baseband", "FM low pass filter for %u Hz at cutoff %.0f Hz, %.1f us", samp_rate, samp_rate * (double)low_pass,
                    1e6 / (samp_rate * (double)low_pass));
        double ita = 1.0 / tan(M_PI_2 * low_pass);
        double gain = 1.0 / (1.0 + ita) / 2;
        state->alp_16[0] = (int)(1.0 * 32768);
        state->alp_16[1] = (int)((ita - 1.0) * gain * 32768);
        state->blp_16[0] = (int)(gain * 32768);
        state->blp_16[1] = (int)(gain * 32768);
        state->rate = samp_rate;
    }
    if (num_samples == 0)
        return;
    r433_filter_carry c;
    memset(&c, 0, sizeof(c));
    c.last_i = (int16_t)state->xr; // the reference pre-feeds its state through int16 locals (:237-240)
    c.last_q = (int16_t)state->xi;
    c.fm_x = (int16_t)state->xf;
    c.fm_y = (int16_t)state->yf;
    if (r433_filter_frame(R433_FILTER_FM_CU8, x_buf, (uint32_t)num_samples, y_buf, &c, state->alp_16[1], state->blp_16[0], 0, 0) < 0)
        die("baseband_demod_FM");
    state->xr = c.last_i;
    state->xi = c.last_q;
    state->xf = (int16_t)c.fm_x;
    state->yf = (int16_t)c.fm_y;
}

void baseband_demod_FM_cs16(demodfm_state_t *state, int16_t const *x_buf, int16_t *y_buf, unsigned long num_samples, uint32_t samp_rate, float low_pass)
{
    if (state->rate != samp_rate) { // src/baseband.c:310-325
        if (low_pass > 1e4f)
            low_pass = low_pass / samp_rate;
        else if (low_pass >= 1.0f)
            low_pass = 1e6f / low_pass / samp_rate;
        if (print_logf) // src/baseband.c:315-316
            print_logf(5, "Baseband", "low pass filter for %u Hz at cutoff %.0f Hz, %.1f us", samp_rate, samp_rate * (double)low_pass,
                    1e6 / (samp_rate * (double)low_pass));
        double ita = 1.0 / tan(M_PI_2 * low_pass);
        double gain = 1.0 / (1.0 + ita);
        state->alp_32[0] = (int)(1.0 * 1073741824);
        state->alp_32[1] = (int)((ita - 1.0) * gain * 1073741824);
        state->blp_32[0] = (int)(gain * 1073741824);
        state->blp_32[1] = (int)(gain * 1073741824);
        state->rate = samp_rate;
    }
    if (num_samples == 0)
        return;
    r433_filter_carry c;
    memset(&c, 0, sizeof(c));
    c.last_i = state->xr;
    c.last_q = state->xi;
    c.fm_x = state->xf;
    c.fm_y = state->yf;
    if (r433_filter_frame(R433_FILTER_FM_CS16, x_buf, (uint32_t)num_samples, y_buf, &c, 0, 0, state->alp_32[1], state->blp_32[0]) < 0)
        die("baseband_demod_FM_cs16");
    state->xr = c.last_i;
    state->xi = c.last_q;
    state->xf = c.fm_x;
    state->yf = c.fm_y;
}

// ---- include/pulse_detect.h ----

typedef struct pulse_detect {
    int use_mag_est;
    float fixed_high_level, min_high_level, high_low_ratio;
    int verbosity;
    r433_detector *det; // made on the first pulse_detect_package call (the batch path never needs it)
} pulse_detect_t;

void pulse_detect_set_levels(pulse_detect_t *pulse_detect, int use_mag_est, float fixed_high_level, float min_high_level, float high_low_ratio, int verbosity)
{
    pulse_detect->use_mag_est = use_mag_est;
    pulse_detect->fixed_high_level = fixed_high_level;
    pulse_detect->min_high_level = min_high_level;
    pulse_detect->high_low_ratio = high_low_ratio;
    pulse_detect->verbosity = verbosity;
    if (pulse_detect->det)
        r433_detector_set_levels(pulse_detect->det, use_mag_est, fixed_high_level, min_high_level, high_low_ratio);
}

pulse_detect_t *pulse_detect_create(void)
{
    pulse_detect_t *p = (pulse_detect_t *)calloc(1, sizeof(*p));
    if (!p) {
        fprintf(stderr, "librtl433seam: out of memory\n");
        exit(1);
    }
    pulse_detect_set_levels(p, 0, 0.0f, -12.1442f, 9.0f, 0); // src/pulse_detect.c:56-67
    return p;
}

void pulse_detect_free(pulse_detect_t *pulse_detect)
{
    if (pulse_detect)
        r433_detector_destroy(pulse_detect->det);
    free(pulse_detect);
}

void pulse_detect_reset(pulse_detect_t *pulse_detect)
{
    if (pulse_detect->det)
        r433_detector_reset(pulse_detect->det);
}

// include/pulse_detect.h:71.  pulse_data_t is r433_pulse_data byte for byte (include/r433_abi.h, tests/test_abi.py).
int pulse_detect_package(pulse_detect_t *pulse_detect, int16_t const *envelope_data, int16_t const *fm_data, int len, uint32_t samp_rate,
        uint64_t sample_offset, r433_pulse_data *pulses, r433_pulse_data *fsk_pulses, unsigned fpdm)
{
    if (!pulse_detect->det) {
        pulse_detect->det = r433_detector_create();
        if (!pulse_detect->det)
            die("pulse_detect_package");
        r433_detector_set_levels(pulse_detect->det, pulse_detect->use_mag_est, pulse_detect->fixed_high_level, pulse_detect->min_high_level,
                pulse_detect->high_low_ratio);
    }
    int const r = r433_detector_package(pulse_detect->det, envelope_data, fm_data, len, samp_rate, sample_offset, pulses, fsk_pulses, fpdm);
    if (r < 0)
        die("pulse_detect_package");
    return r;
}

// ---- include/pulse_slicer.h ----

namespace {

struct SlicerEngine {
    uint32_t rate;
    r433_dev_timing row;
    r433_batch *b;
};
static std::vector<SlicerEngine> g_engines;

static r433_batch *engine_for(uint32_t rate, r433_dev_timing const &row)
{
    for (auto &e : g_engines)
        if (e.rate == rate && memcmp(&e.row, &row, sizeof(row)) == 0)
            return e.b;
    if (g_engines.size() >= 64) { // a host that sweeps timings (the analyzer's trial decoders): keep the table small
        r433_batch_destroy(g_engines.front().b);
        g_engines.erase(g_engines.begin());
    }
    r433_flow_cfg cfg;
    r433_flow_cfg_default(&cfg, 2, rate);
    r433_batch *b = r433_batch_create(&cfg, &row, 1);
    if (!b)
        die("r433_batch_create");
    g_engines.push_back({rate, row, b});
    return b;
}

struct EventHook {
    char const *func;
};

static void on_event(void *user, r433_r_device *dev, int ret, r433_bitbuffer const *bits)
{
    // account_event's debug printout, src/pulse_slicer.c:49-59
    unsigned max_bits = 0;
    for (int row = 0; row < bits->num_rows; ++row)
        if (bits->bits_per_row[row] > max_bits)
            max_bits = bits->bits_per_row[row];
    if (decoder_log_bitbuffer && (!dev->decode_fn || (dev->verbose && ret > 0) || (dev->verbose > 1 && max_bits > 16) || (dev->verbose > 2)))
        decoder_log_bitbuffer(dev, ret > 0 ? 1 : 2, ((EventHook *)user)->func, bits, dev->name);
}

// one package through one slicer: modulation = the slicer the caller named, timings = the device's
static int slice_one(pulse_data_t const *pulses, r_device *device, unsigned modulation, char const *func)
{
    uint32_t const rate = pulses->sample_rate;
    if (rate == 0)
        return 0;
    r433_dev_timing row;
    memset(&row, 0, sizeof(row));
    row.modulation = modulation;
    row.short_width = device->short_width;
    row.long_width = device->long_width;
    row.reset_limit = device->reset_limit;
    row.gap_limit = device->gap_limit;
    row.sync_width = device->sync_width;
    row.tolerance = device->tolerance;
    row.priority = 0;
    r433_batch *b = engine_for(rate, row);
    static pulse_data_t copy; // 9.6 KB; an OOK-numbered slicer must see the package whatever estimates it carries
    copy = *pulses;
    copy.fsk_f2_est = 0;
    if (r433_batch_run_pulses(b, &copy, 1, NULL) < 0)
        die("r433_batch_run_pulses");
    EventHook hook = {func};
    r433_dispatch_hooks hooks = {&hook, NULL, on_event, NULL, NULL, NULL};
    r433_r_device *devs[1] = {device};
    int const events = r433_batch_dispatch_hooks(b, devs, 1, &hooks);
    if (events == R433_EDECODER) {
        fprintf(stderr, "%s: %s: notify maintainer\n", func, r433_last_error()); // src/pulse_slicer.c:44-47
        exit(1);
    }
    if (events < 0)
        die("r433_batch_dispatch_hooks");
    return events;
}

} // namespace

// ---- include/r_api.h:50-52: the fan-out of one package over every registered decoder ----

typedef struct list { // include/list.h:18-22
    void **elems;
    size_t size;
    size_t len;
} list_t;

namespace {

struct FanoutEngine {
    uint32_t rate;
    std::vector<r433_dev_timing> rows; // what the decoders of the list looked like when the engine was made
    r433_batch *b;
};
static std::vector<FanoutEngine> g_fanouts;

// account_event names the slicer that made the bitbuffer (src/pulse_slicer.c:26-66 is handed __func__)
static char const *slicer_name(unsigned modulation)
{
    switch (modulation) {
    case 3: case 18: return "pulse_slicer_manchester_zerobit";
    case 4: case 16: return "pulse_slicer_pcm";
    case 5: return "pulse_slicer_ppm";
    case 6: case 17: return "pulse_slicer_pwm";
    case 8: return "pulse_slicer_piwm_raw";
    case 9: return "pulse_slicer_dmc";
    case 10: return "pulse_slicer_osv1";
    case 11: return "pulse_slicer_piwm_dc";
    case 12: return "pulse_slicer_nrzs";
    case 13: return "pulse_slicer_rzi";
    default: return "pulse_slicer";
    }
}

static void on_fanout_event(void *, r433_r_device *dev, int ret, r433_bitbuffer const *bits)
{
    EventHook hook = {slicer_name(dev->modulation)};
    on_event(&hook, dev, ret, bits);
}

static int run_demods(list_t *r_devs, pulse_data_t const *pulse_data, bool fsk, char const *func)
{
    uint32_t const rate = pulse_data->sample_rate;
    if (!r_devs || rate == 0)
        return 0;
    std::vector<r433_r_device *> devs;
    std::vector<r433_dev_timing> rows;
    for (void **iter = r_devs->elems; iter && *iter; ++iter) {
        r_device *d = (r_device *)*iter;
        // (the reference reports these from inside its loop, once per priority level it walks: src/r_api.c:494,546)
        bool const known = (d->modulation >= 3 && d->modulation <= 6) || (d->modulation >= 8 && d->modulation <= 13) || (d->modulation >= 16 && d->modulation <= 18);
        if (!known)
            fprintf(stderr, "Unknown modulation %u in protocol!\n", d->modulation);
        r433_dev_timing row;
        memset(&row, 0, sizeof(row));
        row.modulation = d->modulation;
        row.short_width = d->short_width;
        row.long_width = d->long_width;
        row.reset_limit = d->reset_limit;
        row.gap_limit = d->gap_limit;
        row.sync_width = d->sync_width;
        row.tolerance = d->tolerance;
        row.priority = d->priority;
        devs.push_back(d);
        rows.push_back(row);
    }
    if (devs.empty())
        return 0;
    r433_batch *b = nullptr;
    for (auto &e : g_fanouts)
        if (e.rate == rate && e.rows.size() == rows.size() && memcmp(e.rows.data(), rows.data(), rows.size() * sizeof(rows[0])) == 0)
            b = e.b;
    if (!b) {
        if (g_fanouts.size() >= 8) { // (a host that keeps changing its decoders: the oldest engine goes)
            r433_batch_destroy(g_fanouts.front().b);
            g_fanouts.erase(g_fanouts.begin());
        }
        r433_flow_cfg cfg;
        r433_flow_cfg_default(&cfg, 2, rate);
        b = r433_batch_create(&cfg, rows.data(), (uint32_t)rows.size());
        if (!b)
            die(func);
        g_fanouts.push_back({rate, rows, b});
    }
    // The package's kind is the caller's statement, not the package's (src/r_flow.c:292-309 hands its OOK list to
    // run_ook_demods and its FSK list to run_fsk_demods whatever estimates they carry): r433_batch_run_pulses reads it off
    // fsk_f2_est like the `.ook` file loop does (src/rtl_433.c:1774), so the copy says what the caller said.
    static pulse_data_t copy;
    copy = *pulse_data;
    if (!fsk)
        copy.fsk_f2_est = 0;
    else if (copy.fsk_f2_est == 0)
        copy.fsk_f2_est = 1;
    if (r433_batch_run_pulses(b, &copy, 1, NULL) < 0)
        die(func);
    r433_dispatch_hooks hooks = {NULL, NULL, on_fanout_event, NULL, NULL, NULL};
    int const events = r433_batch_dispatch_hooks(b, devs.data(), (uint32_t)devs.size(), &hooks);
    if (events == R433_EDECODER) {
        fprintf(stderr, "%s: %s: notify maintainer\n", func, r433_last_error()); // src/pulse_slicer.c:44-47
        exit(1);
    }
    if (events < 0)
        die(func);
    return events;
}

} // namespace

int run_ook_demods(list_t *r_devs, pulse_data_t *pulse_data) { return run_demods(r_devs, pulse_data, false, __func__); } // src/r_api.c:438-500
int run_fsk_demods(list_t *r_devs, pulse_data_t *fsk_pulse_data) { return run_demods(r_devs, fsk_pulse_data, true, __func__); } // src/r_api.c:502-550

// enum modulation_types, include/r_device.h:24-40 (the OOK number selects the slicer; FSK packages take the same code)
int pulse_slicer_pcm(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 4, __func__); }
int pulse_slicer_ppm(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 5, __func__); }
int pulse_slicer_pwm(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 6, __func__); }
int pulse_slicer_manchester_zerobit(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 3, __func__); }
int pulse_slicer_dmc(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 9, __func__); }
int pulse_slicer_piwm_raw(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 8, __func__); }
int pulse_slicer_piwm_dc(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 11, __func__); }
int pulse_slicer_nrzs(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 12, __func__); }
int pulse_slicer_osv1(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 10, __func__); }
int pulse_slicer_rzi(pulse_data_t const *pulses, r_device *device) { return slice_one(pulses, device, 13, __func__); }

// src/pulse_slicer.c:920-935: a bitbuffer given as text straight to the decoder (-y).  Host-only: no pulses involved.
int pulse_slicer_string(const char *code, r_device *device)
{
    if (!bitbuffer_parse) {
        fprintf(stderr, "librtl433seam: pulse_slicer_string needs the host program's bitbuffer_parse\n");
        exit(1);
    }
    static bitbuffer_t bits;
    memset(&bits, 0, sizeof(bits));
    bitbuffer_parse(&bits, code);
    int ret = device->decode_fn ? device->decode_fn(device, &bits) : 0;
    device->decode_events += 1; // account_event, src/pulse_slicer.c:26-66
    if (ret > 0) {
        device->decode_ok += 1;
        device->decode_messages += (unsigned)ret;
    }
    else if (ret >= R433_DECODE_FAIL_SANITY) {
        device->decode_fails[-ret] += 1;
        ret = 0;
    }
    else {
        fprintf(stderr, "%s: Decoder \"%s\" gave invalid return value %d: notify maintainer\n", __func__, device->name, ret);
        exit(1);
    }
    EventHook hook = {__func__};
    on_event(&hook, device, ret, &bits);
    return ret;
}


// ---- include/pulse_detect_fsk.h ----

void pulse_detect_fsk_init(pulse_detect_fsk_t *s) // src/pulse_detect_fsk.c:26-32: constants, nothing to compute
{
    memset(s, 0, sizeof(*s));
    s->var_test_max = INT16_MIN;
    s->var_test_min = INT16_MAX;
    s->skip_samples = 40;
}

void pulse_detect_fsk_classic(pulse_detect_fsk_t *s, int16_t fm_n, pulse_data_t *fsk_pulses)
{
    if (r433_fsk_step(R433_FSK_CLASSIC, s, fm_n, fsk_pulses) < 0)
        die("pulse_detect_fsk_classic");
}

void pulse_detect_fsk_minmax(pulse_detect_fsk_t *s, int16_t fm_n, pulse_data_t *fsk_pulses)
{
    if (r433_fsk_step(R433_FSK_MINMAX, s, fm_n, fsk_pulses) < 0)
        die("pulse_detect_fsk_minmax");
}

void pulse_detect_fsk_wrap_up(pulse_detect_fsk_t *s, pulse_data_t *fsk_pulses)
{
    if (r433_fsk_step(R433_FSK_WRAP_UP, s, 0, fsk_pulses) < 0)
        die("pulse_detect_fsk_wrap_up");
}

} // extern "C"
