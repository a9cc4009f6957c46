/* ref_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Thin driver around the UNMODIFIED rtl_433 reference sources (compiled where they
 * lie under /root/reference by oracle/Makefile into oracle/_ref/libr433ref.so).
 * It feeds captures through the reference's own push_sdr_flow()/flush_sdr_flow()
 * exactly as the `-r file` loop does (src/rtl_433.c:1796-1854) and records
 *   - every pulse package handed to run_ook_demods / run_fsk_demods,
 *   - every bitbuffer handed to a decoder (account_event, src/pulse_slicer.c:26-66),
 *   - the low-passed envelope / FM taps (cfg->demod->am_buf / buf.fm),
 *   - JSON events from the real decoders (optional).
 * Hook: r_flow.c is compiled with -Drun_ook_demods=refh_run_ook_demods and
 * -Drun_fsk_demods=refh_run_fsk_demods so the package hand-off passes through here;
 * nothing else in the reference is touched.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rtl_433.h"
#include "r_private.h"
#include "r_api.h"
#include "r_flow.h"
#include "r_device.h"
#include "rtl_433_devices.h"
#include "bitbuffer.h"
#include "pulse_data.h"
#include "pulse_detect.h"
#include "pulse_slicer.h"
#include "baseband.h"
#include "fileformat.h"
#include "list.h"
#include "data.h"
#include "output_file.h"
#include "logger.h"

#include "r433_records.h"

/* the real fan-out, src/r_api.c:438,502 (r_flow.c sees the renamed symbols below) */
int run_ook_demods(list_t *r_devs, pulse_data_t *pulse_data);
int run_fsk_demods(list_t *r_devs, pulse_data_t *fsk_pulse_data);

typedef struct blob {
    uint8_t *data;
    size_t len, cap;
    uint32_t count;
} blob;

static void blob_put(blob *b, void const *src, size_t n)
{
    if (b->len + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 1 << 16;
        while (nc < b->len + n)
            nc *= 2;
        b->data = realloc(b->data, nc);
        b->cap = nc;
    }
    if (src)
        memcpy(b->data + b->len, src, n);
    else
        memset(b->data + b->len, 0, n);
    b->len += n;
}

typedef struct wrapped_dev {
    r_device dev; /* must be first: decode_fn receives &dev */
    unsigned index;
    int (*real_decode)(r_device *, bitbuffer_t *);
    unsigned ordinal; /* within the current package */
} wrapped_dev;

typedef struct harness {
    r_cfg_t *cfg;
    int call_real; /* run the reference decoders after recording */
    int record;    /* 0: digest only, 1: keep records */
    int digest_mode; /* 0 none, 1 full 6604-byte image, 2 compact (header + payload bytes) */
    uint64_t digest2;
    blob packages, events, returns;
    uint32_t stream, frame, pkg_count;
    uint64_t digest;
    uint32_t digest_events;
    int16_t *tap_am, *tap_fm;
    int load_format; /* 0: what the sample size says; 1 / 2: am.s16 / fm.s16 input files (file_info S16_AM / S16_FM) */
    size_t tap_pos;
    uint32_t *frame_sums;
    float *frame_db;
} harness;

static harness *g_h; /* the reference API has no user pointer on this path */

/* ---- package hook ---- */

static void record_package(pulse_data_t const *p, int type)
{
    harness *h = g_h;
    if (h->record) {
        r433_pkg_rec r;
        memset(&r, 0, sizeof(r));
        r.total_bytes = (uint32_t)(sizeof(r) + 8u * p->num_pulses);
        r.stream = h->stream;
        r.type = (uint32_t)type;
        r.num_pulses = p->num_pulses;
        r.frame = h->frame;
        r.ret_pos = 0; /* not observable from outside the detector */
        r.offset = p->offset;
        r.start_ago = p->start_ago;
        r.end_ago = p->end_ago;
        r.ook_low = p->ook_low_estimate;
        r.ook_high = p->ook_high_estimate;
        r.fsk_f1 = p->fsk_f1_est;
        r.fsk_f2 = p->fsk_f2_est;
        r.sample_rate = p->sample_rate;
        blob_put(&h->packages, &r, sizeof(r));
        for (unsigned i = 0; i < p->num_pulses; ++i) {
            int32_t pair[2] = {p->pulse[i], p->gap[i]};
            blob_put(&h->packages, pair, 8);
        }
        h->packages.count++;
    }
    for (void **it = h->cfg->demod->r_devs.elems; it && *it; ++it)
        ((wrapped_dev *)*it)->ordinal = 0;
}

int refh_run_ook_demods(list_t *r_devs, pulse_data_t *pulse_data)
{
    record_package(pulse_data, R433_PKG_OOK);
    int r = run_ook_demods(r_devs, pulse_data);
    g_h->pkg_count++;
    return r;
}

int refh_run_fsk_demods(list_t *r_devs, pulse_data_t *pulse_data)
{
    record_package(pulse_data, R433_PKG_FSK);
    int r = run_fsk_demods(r_devs, pulse_data);
    g_h->pkg_count++;
    return r;
}

/* ---- decoder hook ---- */

static uint64_t fnv(uint64_t h, void const *p, size_t n)
{
    uint8_t const *b = p;
    for (size_t i = 0; i < n; ++i)
        h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

static int recording_decode(r_device *decoder, bitbuffer_t *bits)
{
    harness *h = g_h;
    wrapped_dev *w = (wrapped_dev *)decoder;
    uint32_t pkg = h->pkg_count;
    uint16_t dev = (uint16_t)w->index, ord = (uint16_t)w->ordinal++;

    /* checksum-of-checksums over (key, raw 6604-byte bitbuffer image) */
    uint8_t key[8];
    memcpy(key, &pkg, 4);
    memcpy(key + 4, &dev, 2);
    memcpy(key + 6, &ord, 2);
    /* per-event FNV-1a, summed mod 2^64 so the total does not depend on call order */
    if (h->digest_mode == 1)
        h->digest += fnv(fnv(1469598103934665603ull, key, 8), bits, sizeof(*bits));
    else if (h->digest_mode == 2) {
        uint64_t x = fnv(1469598103934665603ull, key, 8);
        x = fnv(x, &bits->num_rows, 2);
        x = fnv(x, &bits->free_row, 2);
        for (unsigned r = 0; r < bits->num_rows && r < BITBUF_ROWS; ++r) {
            x = fnv(x, &bits->bits_per_row[r], 2);
            x = fnv(x, &bits->syncs_before_row[r], 2);
            x = fnv(x, bits->bb[r], ((unsigned)bits->bits_per_row[r] + 7) / 8);
        }
        h->digest2 += x;
    }
    h->digest_events++;

    if (h->record) {
        size_t at = h->events.len;
        r433_evt_rec e = {0, pkg, dev, ord, bits->num_rows, bits->free_row};
        blob_put(&h->events, &e, sizeof(e));
        unsigned covered = 0;
        for (unsigned r = 0; r < bits->num_rows && r < BITBUF_ROWS; ++r) {
            unsigned span = bits->bits_per_row[r] ? (bits->bits_per_row[r] + BITBUF_COLS * 8 - 1) / (BITBUF_COLS * 8) : 1;
            unsigned end_row = r + span;
            if (r == (unsigned)bits->num_rows - 1 || end_row > BITBUF_ROWS)
                end_row = BITBUF_ROWS;
            unsigned nb = 0;
            if (r >= covered) {
                uint8_t const *base = bits->bb[r];
                nb = (end_row - r) * BITBUF_COLS;
                while (nb > 0 && base[nb - 1] == 0)
                    nb--;
                covered = end_row;
            }
            r433_row_rec rr = {bits->bits_per_row[r], bits->syncs_before_row[r], (uint16_t)nb, 0};
            blob_put(&h->events, &rr, sizeof(rr));
            blob_put(&h->events, bits->bb[r], nb);
            blob_put(&h->events, NULL, ((nb + 3u) & ~3u) - nb);
        }
        uint32_t total = (uint32_t)(h->events.len - at);
        memcpy(h->events.data + at, &total, 4);
        h->events.count++;
    }

    int ret = 0;
    if (h->call_real && w->real_decode)
        ret = w->real_decode(decoder, bits);
    if (h->record) {
        int32_t r32 = ret;
        blob_put(&h->returns, &r32, 4);
    }
    return ret;
}

/* wrap every registered device that is not wrapped yet so its decode_fn records first */
static void wrap_devices(r_cfg_t *cfg)
{
    unsigned idx = 0;
    for (void **it = cfg->demod->r_devs.elems; it && *it; ++it, ++idx) {
        r_device *p = *it;
        if (p->decode_fn == recording_decode)
            continue;
        wrapped_dev *w = calloc(1, sizeof(*w));
        w->dev = *p;
        w->index = idx;
        w->real_decode = p->decode_fn;
        w->dev.decode_fn = recording_decode;
        free(p);
        *it = &w->dev;
    }
}

/* Synthetic decoders: plain timing rows registered like any protocol (no decode_fn of their own), so that every
 * slicer -- also the ones no default-enabled protocol uses, OOK_PULSE_PIWM_RAW and OOK_PULSE_NRZS -- can be run
 * with arbitrary timings through the reference's own fan-out.  Appended after the devices registered so far. */
void refh_add_rows(void *hv, r433_dev_timing const *rows, int n_rows)
{
    harness *h = hv;
    for (int i = 0; i < n_rows; ++i) {
        r_device d;
        memset(&d, 0, sizeof(d));
        d.name = "probe";
        d.modulation = rows[i].modulation;
        d.short_width = rows[i].short_width;
        d.long_width = rows[i].long_width;
        d.reset_limit = rows[i].reset_limit;
        d.gap_limit = rows[i].gap_limit;
        d.sync_width = rows[i].sync_width;
        d.tolerance = rows[i].tolerance;
        d.priority = rows[i].priority;
        register_protocol(h->cfg, &d, NULL);
        if (d.modulation >= FSK_DEMOD_MIN_VAL)
            h->cfg->demod->enable_FM_demod = 1;
    }
    wrap_devices(h->cfg);
}

/* The registered decoders with their REAL decode_fn, as plain r_device objects: what a host hands to a dispatcher that
 * replays bitbuffers produced elsewhere (bench.py's "real decoders" leg hands them to r433_batch_dispatch).  Their
 * output goes to a sink that counts and frees it.  Copies: the harness's own (recording) list is untouched. */
static unsigned long g_sink_count;
static int g_text_on; /* refh_text_mode: what decoders report is kept as JSON lines (the reference's own printer) */
static blob g_text;
static void sink_output(r_device *decoder, data_t *data)
{
    (void)decoder;
    if (g_text_on) {
        char *line = data_print_jsons_dup(data);
        if (line) {
            blob_put(&g_text, line, strlen(line));
            blob_put(&g_text, "\n", 1);
            free(line);
        }
    }
    data_free(data);
    g_sink_count++;
}

/* on: every message a decoder hands out -- in the harness's own reference flow (call_real) and through the plain devices of
 * refh_plain_devices alike -- is printed with data_print_jsons (src/data.c) into one buffer, a line per message, instead of
 * going to the reference's output handlers: the text a host compares with what its own replay of the same decoders says. */
void refh_text_mode(void *hv, int on)
{
    harness *h = hv;
    g_text_on = on;
    g_text.len = 0;
    for (void **it = h->cfg->demod->r_devs.elems; it && *it; ++it)
        ((r_device *)*it)->output_fn = on ? sink_output : data_acquired_handler;
}

/* the lines since the last call (valid until the next message) */
size_t refh_take_text(char const **text)
{
    size_t const n = g_text.len;
    if (text)
        *text = g_text.data ? (char const *)g_text.data : "";
    g_text.len = 0;
    return n;
}

int refh_plain_devices(void *hv, r_device **out, int cap)
{
    harness *h = hv;
    int n = 0;
    for (void **it = h->cfg->demod->r_devs.elems; it && *it && n < cap; ++it, ++n) {
        wrapped_dev *w = (wrapped_dev *)*it;
        r_device *p = malloc(sizeof(*p)); /* lives as long as the process: a handful of KB */
        *p = w->dev;
        p->decode_fn = w->real_decode;
        p->output_fn = sink_output;
        p->decode_events = p->decode_ok = p->decode_messages = 0;
        memset(p->decode_fails, 0, sizeof(p->decode_fails));
        out[n] = p;
    }
    return n;
}

unsigned long refh_sink_count(void)
{
    return g_sink_count;
}

/* ---- lifecycle ---- */

static void quiet_log(log_level_t level, char const *src, char const *msg, void *userdata)
{
    (void)level;
    (void)src;
    (void)msg;
    (void)userdata;
}

/* protocols: NULL/0 => all default-enabled (src/rtl_433.c:1511-1513); else list of protocol numbers.
 * flex_specs: '\n' separated -X specs (may be NULL).  json_path: where real decoders print (may be NULL). */
void *refh_create(int const *protocols, int n_protocols, char const *flex_specs, int call_real, int record,
        char const *json_path, int report_meta_level, int report_protocol)
{
    harness *h = calloc(1, sizeof(*h));
    r_logger_set_log_handler(quiet_log, NULL);
    r_cfg_t *cfg = r_create_cfg();
    h->cfg = cfg;
    h->call_real = call_real;
    h->record = record;
    h->digest_mode = 1;
    h->digest = 0;
    cfg->report_time = REPORT_TIME_SAMPLES; /* src/rtl_433.c:1480-1483 for file input */
    cfg->report_meta = report_meta_level;
    cfg->report_protocol = report_protocol;
    cfg->verbosity = 0;

    if (n_protocols > 0) {
        for (int i = 0; i < n_protocols; ++i)
            if (protocols[i] >= 1 && protocols[i] <= cfg->num_r_devices)
                register_protocol(cfg, &cfg->devices[protocols[i] - 1], NULL);
    }
    else if (n_protocols == 0) {
        register_all_protocols(cfg, 0);
    }
    if (flex_specs && *flex_specs) {
        char *dup = strdup(flex_specs);
        for (char *s = strtok(dup, "\n"); s; s = strtok(NULL, "\n")) {
            char *spec = strdup(s); /* the flex parser keeps pointers into its argument */
            register_protocol(cfg, &flex_decoder, spec);
        }
        free(dup);
    }
    /* src/rtl_433.c:1515-1526 */
    for (void **it = cfg->demod->r_devs.elems; it && *it; ++it)
        if (((r_device *)*it)->modulation >= FSK_DEMOD_MIN_VAL)
            cfg->demod->enable_FM_demod = 1;

    wrap_devices(cfg);
    if (json_path && *json_path) {
        list_push(&cfg->output_handler, data_output_json_create(0, strdup(json_path))); /* appends, flushes per line */
    }
    return h;
}

void refh_destroy(void *hv)
{
    harness *h = hv;
    if (!h)
        return;
    /* decode_ctx of wrapped devices is owned by the wrappers now; r_free_cfg frees both */
    r_free_cfg(h->cfg);
    free(h->cfg);
    free(h->packages.data);
    free(h->events.data);
    free(h->returns.data);
    free(h);
}

int refh_num_devices(void *hv)
{
    harness *h = hv;
    return (int)h->cfg->demod->r_devs.len;
}

int refh_device_info(void *hv, int idx, r433_dev_timing *t, uint32_t *protocol_num, char *name, int name_cap)
{
    harness *h = hv;
    if (idx < 0 || (size_t)idx >= h->cfg->demod->r_devs.len)
        return -1;
    r_device *d = h->cfg->demod->r_devs.elems[idx];
    t->modulation = d->modulation;
    t->short_width = d->short_width;
    t->long_width = d->long_width;
    t->reset_limit = d->reset_limit;
    t->gap_limit = d->gap_limit;
    t->sync_width = d->sync_width;
    t->tolerance = d->tolerance;
    t->priority = d->priority;
    if (protocol_num)
        *protocol_num = d->protocol_num;
    if (name && name_cap > 0) {
        strncpy(name, d->name ? d->name : "", (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    return 0;
}

/* -Y style detector options; call before the first refh_run_capture */
void refh_set_levels(void *hv, int use_mag_est, float level_limit, float min_level, float min_snr, float auto_level,
        float squelch_offset, float fm_low_pass)
{
    harness *h = hv;
    struct dm_state *dm = h->cfg->demod;
    dm->use_mag_est = use_mag_est;
    dm->level_limit = level_limit;
    dm->min_level = min_level;
    dm->min_snr = min_snr;
    dm->auto_level = auto_level;
    dm->squelch_offset = squelch_offset;
    dm->fm_low_pass = fm_low_pass;
    /* src/rtl_433.c:1465 */
    pulse_detect_set_levels(dm->pulse_detect, dm->use_mag_est, dm->level_limit, dm->min_level, dm->min_snr, dm->detect_verbosity);
}

/* the next captures are am.s16 (1) / fm.s16 (2) files: src/rtl_433.c:1735-1739 reads them as 2-byte samples and
 * src/r_flow.c:212-225 puts their words in place of the demodulated buffers */
void refh_set_load_format(void *hv, int fmt)
{
    ((harness *)hv)->load_format = fmt;
}

void refh_set_enable_fm(void *hv, int on)
{
    ((harness *)hv)->cfg->demod->enable_FM_demod = on;
}

/* One capture = one `-r file`: frames of DEFAULT_BUF_LENGTH bytes, flush, reset
 * (src/rtl_433.c:1796-1854, process_sdr_frame :1084-1123).  sample_size 2 = cu8, 4 = cs16.
 * fpdm: 0 classic, 1 minmax, 2 auto-by-frequency.  Taps may be NULL. */
int refh_run_capture(void *hv, uint8_t const *iq, size_t n_bytes, uint32_t sample_size, uint32_t samp_rate,
        uint32_t center_freq, int fpdm, uint32_t stream_index, int16_t *tap_am, int16_t *tap_fm,
        uint32_t *frame_sums, float *frame_db)
{
    harness *h = hv;
    r_cfg_t *cfg = h->cfg;
    struct dm_state *dm = cfg->demod;
    g_h = h;
    h->stream = stream_index;
    h->frame = 0;

    cfg->samp_rate = samp_rate;
    cfg->center_frequency = center_freq;
    dm->sample_size = (int)sample_size;
    dm->load_info.format = h->load_format == 1 ? S16_AM : h->load_format == 2 ? S16_FM : sample_size == 2 ? CU8_IQ : CS16_IQ;
    dm->sample_file_pos = 0.0f;

    unsigned mode = (unsigned)fpdm;
    if (fpdm == 2)
        mode = center_freq > FSK_PULSE_DETECTOR_LIMIT ? FSK_PULSE_DETECT_NEW : FSK_PULSE_DETECT_OLD;

    int events = 0;
    size_t done = 0;
    int n_blocks = 0;
    uint8_t *frame_buf = malloc(DEFAULT_BUF_LENGTH);
    while (done < n_bytes) {
        size_t n = n_bytes - done < DEFAULT_BUF_LENGTH ? n_bytes - done : DEFAULT_BUF_LENGTH;
        memcpy(frame_buf, iq + done, n);
        dm->sample_file_pos = ((float)n_blocks * DEFAULT_BUF_LENGTH + n) / cfg->samp_rate / dm->sample_size;
        n_blocks++;
        /* process_sdr_frame's parameter hand-over */
        dm->raw_handler = &cfg->raw_handler;
        dm->fsk_pulse_detect_mode = (int)mode;
        dm->report_noise = 0;
        dm->verbosity = 0;
        dm->raw_mode = 0;
        dm->grab_mode = 0;
        if (dm->center_frequency != cfg->center_frequency || dm->samp_rate != cfg->samp_rate)
            events += flush_sdr_flow(cfg);
        dm->center_frequency = cfg->center_frequency;
        dm->samp_rate = cfg->samp_rate;

        size_t ns = n / sample_size;
        if (frame_sums || frame_db) { /* recompute the frame level through the reference function */
            uint16_t *tmp = malloc(sizeof(uint16_t) * (ns + 1));
            float db;
            if (sample_size == 2)
                db = dm->use_mag_est ? magnitude_est_cu8(frame_buf, tmp, (uint32_t)ns) : envelope_detect(frame_buf, tmp, (uint32_t)ns);
            else
                db = magnitude_est_cs16((int16_t *)frame_buf, tmp, (uint32_t)ns);
            uint32_t s = 0;
            for (size_t i = 0; i < ns; ++i)
                s += tmp[i];
            if (frame_sums)
                frame_sums[h->frame] = s;
            if (frame_db)
                frame_db[h->frame] = db;
            free(tmp);
        }
        events += push_sdr_flow(cfg, frame_buf, (uint32_t)n);
        if (tap_am)
            memcpy(tap_am + done / sample_size, dm->am_buf, ns * sizeof(int16_t));
        if (tap_fm)
            memcpy(tap_fm + done / sample_size, dm->buf.fm, ns * sizeof(int16_t));
        done += n;
        h->frame++;
    }
    free(frame_buf);
    events += flush_sdr_flow(cfg);
    reset_sdr_flow(cfg);
    /* batch semantics: every capture starts from a freshly initialised flow */
    dm->input_pos = 0;
    dm->frame_start_ago = 0;
    dm->frame_end_ago = 0;
    return events;
}

uint8_t const *refh_packages(void *hv, size_t *len, uint32_t *count)
{
    harness *h = hv;
    *len = h->packages.len;
    *count = h->packages.count;
    return h->packages.data;
}

uint8_t const *refh_events(void *hv, size_t *len, uint32_t *count)
{
    harness *h = hv;
    *len = h->events.len;
    *count = h->events.count;
    return h->events.data;
}

int32_t const *refh_returns(void *hv, size_t *count)
{
    harness *h = hv;
    *count = h->returns.len / 4;
    return (int32_t const *)h->returns.data;
}

uint64_t refh_digest(void *hv, uint32_t *n_events, uint32_t *n_packages)
{
    harness *h = hv;
    if (n_events)
        *n_events = h->digest_events;
    if (n_packages)
        *n_packages = h->pkg_count;
    return h->digest;
}

void refh_set_digest_mode(void *hv, int mode)
{
    ((harness *)hv)->digest_mode = mode;
}

uint64_t refh_digest2(void *hv)
{
    return ((harness *)hv)->digest2;
}

void refh_clear(void *hv)
{
    harness *h = hv;
    h->packages.len = h->events.len = h->returns.len = 0;
    h->packages.count = h->events.count = 0;
    h->pkg_count = 0;
    h->digest = 0;
    h->digest2 = 0;
    h->digest_events = 0;
}

/* sizes the product mirrors must agree with */
void refh_abi_sizes(uint32_t *out)
{
    out[0] = (uint32_t)sizeof(bitbuffer_t);
    out[1] = (uint32_t)sizeof(pulse_data_t);
    out[2] = (uint32_t)sizeof(r_device);
    out[3] = (uint32_t)offsetof(r_device, decode_fn);
    out[4] = (uint32_t)offsetof(r_device, priority);
    out[5] = (uint32_t)offsetof(r_device, decode_ctx);
    out[6] = (uint32_t)offsetof(pulse_data_t, pulse);
    out[7] = (uint32_t)offsetof(pulse_data_t, ook_low_estimate);
}
