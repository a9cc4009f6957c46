/** @file
    r_flow_hip.c -- the translation unit a maintainer adds to rtl_433 in place of src/r_flow.c: the same three entry
    points (include/r_flow.h:19-23)

        int  push_sdr_flow(struct r_cfg *cfg, unsigned char *iq_buf, uint32_t len);
        int  flush_sdr_flow(struct r_cfg *cfg);
        void reset_sdr_flow(struct r_cfg *cfg);

    served by librtl433hip.so (include/r433_hip.h).  Everything else of rtl_433 stays what it is: the CLI and its
    `-r` file loop (src/rtl_433.c:1703-1859), r_api.c with register_protocol / data_acquired_handler, the ~330
    decoders behind r_device.decode_fn, the output modules.  It is compiled against the reference's own headers and
    is C99 like the rest of that tree.

    What changes is WHEN the work happens.  The reference walks a file frame by frame on one core.  Here a frame
    that is pushed is only appended to the current capture in pinned host memory; a flush (the end of a `-r` file)
    closes the capture; and a *drain* hands every capture collected so far to the GPU in one pass
    (r433_batch_run_host: one wavefront per capture, then the slicer fan-out) and replays the resulting bitbuffers
    into the registered decoders in the reference's order (r433_batch_dispatch_hooks), with the pieces of
    struct dm_state / r_cfg that the output path reads (src/r_api.c:306-333,632-840: in_filename, sample_file_pos,
    pulse_data / fsk_pulse_data levels and start_ago, ...) set to what they were in the reference at that moment.
    So `rtl_433 -r a.cu8 -r b.cu8 ... -F json` prints the same lines in the same order.

    A drain happens
      - at every flush                      when batching is off (RTL433_HIP_BATCH=1, no end hook compiled in, or an output that
                                            also prints log messages -- -F kv, -F log -- so that the file loop's messages and the
                                            events of each file come out in the reference's order);
      - when RTL433_HIP_BATCH captures (default 4096) or 1 GiB of samples are waiting;
      - at hip_sdr_flow_drain(cfg), which the host calls once after its file loop and before close_dumpers()
        (one added line in src/rtl_433.c:1860; builds of the unmodified rtl_433.c get the same effect from
        -Dclose_dumpers=hip_sdr_flow_close_dumpers on that one file, see dropin/Makefile).
    Deferred work means push_sdr_flow returns 0 and the flush (or the drain) returns the events.  With -E quit / -E hop the
    file loop acts on the event count of every push (src/rtl_433.c:1136-1143): then nothing is deferred -- every push runs
    the capture as far as it has come and replays its newest packages (sync_step below), and the loop quits where the
    reference's does.

    S16_AM / S16_FM pseudo-IQ input files are served (the library's R433_IN_S16_AM / R433_IN_S16_FM); the raw rtl_tcp
    output's per-frame pacing is kept.  The sample grabber (-S all | unknown | known | undecoded, raw or SigMF) writes its files after
    the replay of a pass (write_grabs below).  Every -w / -W dumper is served: the input's own format is a
    copy, the other IQ formats are the library's dump kernel on each frame, am / fm dumps come from the detection pass's
    taps, .u8 is painted by the detection kernel, .ook / .vcd are written during the replay.  (One quirk of the reference
    is NOT reproduced: its IQ conversions write into a buffer that is a union with buf.fm, so an IQ dumper listed before
    an fm.s16 dumper clobbers what that one writes; here fm.s16 is always the discriminator.)
 */

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdbool.h>
#include <string.h>
#include <math.h>
#include <unistd.h>
#include <time.h>

#include "r_flow.h"
#include "rtl_433.h"
#include "r_private.h"
#include "r_device.h"
#include "r_api.h"
#include "pulse_analyzer.h"
#include "pulse_data.h"
#include "decoder_util.h"
#include "data.h"
#include "raw_output.h"
#include "r_util.h"
#include "logger.h"
#include "fatal.h"

#include "samp_grab.h"

#include "r433_hip.h"

r433_helper_probe *r433_host_helper_probe(int session); /* dropin/helper_wrap.c */

#include <pthread.h>
#include <sys/mman.h>

/* the three structs that cross the boundary are the reference's own (include/r433_abi.h mirrors them) */
_Static_assert(sizeof(r433_r_device) == sizeof(r_device), "r_device layout");
_Static_assert(sizeof(r433_pulse_data) == sizeof(pulse_data_t), "pulse_data_t layout");
_Static_assert(sizeof(r433_bitbuffer) == sizeof(bitbuffer_t), "bitbuffer_t layout");

int hip_sdr_flow_drain(struct r_cfg *cfg);
void hip_sdr_flow_close_dumpers(struct r_cfg *cfg);

/* one `-r` file (or one stretch of live input between two flushes) */
typedef struct hip_capture {
    char const *in_filename;
    file_info_t load_info;
    uint32_t samp_rate;
    uint32_t center_frequency;
    int sample_size;
    int fpdm;
    size_t offset; /* into the staging buffer */
    uint32_t bytes;
    uint32_t frame_bytes; /* length of every push but the last */
    uint32_t last_bytes;  /* length of the last push */
    uint32_t n_frames;
    uint32_t frames_cap;
    float *frame_pos;          /* dm_state.sample_file_pos at each push */
    struct timeval *frame_now; /* dm_state.now at each push */
    uint64_t input_pos;        /* dm_state.input_pos at the first push */
    int irregular;
} hip_capture;

/* An engine per flow configuration met so far (a file list may alternate between sample rates or formats): creating one
   costs device allocations, so the last few stay. */
#define HIP_ENGINES 4
typedef struct hip_engine {
    r433_batch *eng;
    r433_flow_cfg cfg;
    size_t devs;
    void *first_dev;
    int probed, tables; /* the decoder pre-filter of this engine: asked for, decoders with a table */
    int lane;           /* which of the two passes in flight it serves (a pass on the GPU while the one before is replayed) */
    unsigned long used; /* for the least recently used */
    int busy;           /* passes (in flight or owed a replay) that hold this engine: never the one to make room */
} hip_engine;

/* H.engines is looked at by the file loop's thread and by the thread of a pass in flight */
static pthread_mutex_t g_eng_mu = PTHREAD_MUTEX_INITIALIZER;

static struct {
    r433_batch *eng; /* the engine of the pass at hand (one of engines[]) */
    hip_engine engines[HIP_ENGINES];
    hip_engine *cur;
    unsigned long eng_clock;
    size_t probe_devs;  /* the decoder set the process-wide probe memory of the library was filled from */
    void *probe_first;
    uint8_t *stage; /* pinned */
    size_t stage_cap, stage_len;
    size_t staged_total; /* bytes of samples this process has taken so far (engine_prefilter) */
    hip_capture *caps;
    size_t n_caps, caps_cap;
    int open; /* the last capture is still being pushed to */
    int warned_dump;
    uint8_t *conv;      /* a converted frame / capture for the sample dumpers */
    size_t conv_cap;
    int16_t *taps[3];   /* pinned: raw envelope, filtered envelope, filtered discriminator of the captures of a pass (am / fm dumpers) */
    size_t taps_cap;
    uint8_t *hist;      /* the sample grabber's view of the past: the last 3 MiB pushed before the captures now queued */
    size_t hist_len;
    uint64_t pushed_before_queue; /* bytes pushed, ever, before the first capture of the queue */
    int warned_grab_mode;
    int32_t *quality;   /* -S undecoded: pulse_analyzer_check's verdict on every package of the pass nobody decoded */
    size_t quality_cap, quality_n;
    r433_analysis *analysis; /* -A: the pulse analyzer's histograms and guess for every package of the pass (r433_batch_analyze: one launch) */
    size_t analysis_cap, analysis_n;
    char *text;         /* what the library renders for a package: the analyzer's report, a `.ook` record, VCD lines */
    size_t text_cap;
    /* answering every push at once (-E: the file loop acts on the event count of a push) */
    int sync_active, sync_flush, warned_sync_grab;
    uint32_t sync_frame;            /* the frame just pushed */
    unsigned sync_count, sync_squelch; /* frames / noise-only frames the run before this one counted for the same capture */
    uint32_t fm_note_rate; /* the rate the "FM low pass filter" notice was last printed for (src/baseband.c:217,310) */
    /* replay context */
    r_cfg_t *cfg;
    hip_capture *group;
    uint32_t const *frame_sums;
    uint32_t sums_cap;
    uint32_t cur_stream;
    uint32_t cur_frames_done;
    /* two passes in flight (hip_sdr_flow_drain): the queue of one is on the GPU while the file loop fills the next */
    int lane;            /* the engines of the next pass to start */
    int lane_seen;       /* a pass has been handed over: fresh staging buffers start at the size of a pass */
    int final_drain;     /* the file loop is over: nothing is left in flight when the drain returns */
    uint8_t *spare_stage; /* the pinned buffer of the pass before the one in flight, for the queue after it */
    size_t spare_cap;
} H = {.cur_stream = UINT32_MAX};

/* a pass whose GPU leg runs (or has run) on a thread of its own and whose replay is owed */
typedef struct pending_pass {
    int active;
    pthread_t thread;
    r433_batch *eng;
    hip_engine *mine; /* made by pass_start already (the pre-filter's questions were due): held */
    r_cfg_t *cfg;
    r433_flow_cfg fc;
    int lane;
    hip_capture *caps;
    size_t n;
    uint8_t *stage;
    size_t stage_cap, stage_len;
    void const **ptrs;
    uint32_t *bytes;
    int n_pkgs;
    char err[256];
    double t_start;
} pending_pass;
static pending_pass P; /* the pass in flight */

static size_t batch_limit(r_cfg_t *cfg)
{
    char const *e = getenv("RTL433_HIP_BATCH");
    if (e && *e) {
        long v = atol(e);
        return v < 1 ? 1 : (size_t)v;
    }
    /* An output that also takes log messages (-F kv, -F log, -F json:v...) interleaves the file loop's own messages
       ("Test mode active. Reading samples from file: ...") with the events of each file: keep that order, one pass per file. */
    for (size_t i = 0; i < cfg->output_handler.len; ++i) {
        data_output_t *o = cfg->output_handler.elems[i];
        if (o && o->log_level > 0)
            return 1;
    }
    /* the am / fm sample dumpers are fed from per-sample taps of the detection pass (6 bytes per sample, pinned): per file */
    for (void **iter = cfg->demod->dumper.elems; iter && *iter; ++iter) {
        file_info_t const *dumper = *iter;
        if (dumper->file && (dumper->format == S16_AM || dumper->format == S16_FM || dumper->format == F32_AM || dumper->format == F32_FM))
            return 1;
    }
#ifdef R433_HIP_HAVE_DRAIN
    return 4096;
#else
    return 1;
#endif
}

/* host threads of the decoder replay: RTL433_HIP_THREADS, default min(16, online cores) */
static int replay_threads(void)
{
    char const *e = getenv("RTL433_HIP_THREADS");
    if (e && *e) {
        int v = atoi(e);
        return v < 1 ? 1 : v;
    }
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n < 1 ? 1 : n > 16 ? 16 : (int)n;
}

/* RTL433_HIP_TRACE=1: where a pass spends its time, on stderr */
static double trace_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static double g_t_loaded; /* when this translation unit was loaded: the process's start, near enough */
static void trace_exit(void);
__attribute__((constructor)) static void trace_loaded(void)
{
    g_t_loaded = trace_now();
}

static int trace_on(void)
{
    static int on = -1;
    if (on < 0) {
        char const *e = getenv("RTL433_HIP_TRACE");
        on            = e && *e && *e != '0';
        if (on)
            atexit(trace_exit);
    }
    return on;
}

static void trace_exit(void)
{
    fprintf(stderr, "hip flow: exit handlers begin %.1f ms after the program was loaded\n", trace_now() - g_t_loaded);
}

static void hip_fatal(char const *what)
{
    print_logf(LOG_FATAL, "HIP", "%s: %s", what, r433_last_error());
    exit(1);
}

/* a pass is run when this much has been queued (1920 captures of 128 KiB): large enough to fill the GPU, small enough
   that the buffers of a pass (pinned staging here, record mirrors in the library) are cheap to come by */
#define STAGE_PASS ((size_t)256 << 20)

/* ---- what a process pays once before its first kernel has run, off the file loop's thread --------------------------------
   Opening the GPU (50-130 ms) and loading the library's kernels (20-150 ms) used to happen inside the first staging
   allocation and the first engine, on the thread that reads the files -- a third of a run over 8192 captures.  They start on
   a thread of their own at the first push and are waited for where the first GPU call is made (the first pass -- on its own
   thread, too, when the list is long: the file loop never stops for them).  RTL433_HIP_WARM=0: as before, inline. */
static struct {
    pthread_t thread;
    int started, joined;
    double t0;
} W;

static void *warm_thread(void *arg)
{
    (void)arg;
    (void)r433_warmup(); /* (a failure shows at the first real call, with its message) */
    return NULL;
}

static void warm_join(void)
{
    if (W.started && !W.joined) {
        W.joined = 1;
        pthread_join(W.thread, NULL);
        if (trace_on())
            fprintf(stderr, "hip flow: [%.1f ms] GPU opened and kernels loaded on a thread of their own (started at the first push, %.1f ms)\n",
                    trace_now() - g_t_loaded, W.t0 - g_t_loaded);
    }
}

/* At exit() the GPU runtime's library destructors give back, piece by piece, what the kernel reclaims in one go when the
   process is gone: device memory of the engines, a gigabyte of page-locked host memory, queues, code objects -- 100-180 ms at
   the end of a 0.5 s run over 8192 files.  The program has written and closed its outputs by then (src/rtl_433.c:1861-1866:
   close_dumpers, r_free_cfg before main returns; the reference registers no exit handler of its own), so the flow's handler
   flushes what stdio still holds and leaves through _exit() with the program's own status.  RTL433_HIP_FAST_EXIT=0: the
   long way. */
static void flow_exit(int status, void *arg)
{
    (void)arg;
    warm_join(); /* (a process that leaves without a drain must not exit under a thread that is opening the device) */
    char const *e = getenv("RTL433_HIP_FAST_EXIT");
    if (e && *e == '0')
        return;
    if (trace_on())
        fprintf(stderr, "hip flow: [%.1f ms] leaving through _exit(%d)\n", trace_now() - g_t_loaded, status);
    fflush(NULL);
    _exit(status);
}

static void warm_start(void)
{
    static int exit_hooked;
    if (!exit_hooked) {
        exit_hooked = 1;
        on_exit(flow_exit, NULL);
    }
    char const *e = getenv("RTL433_HIP_WARM");
    if (W.started || (e && *e == '0'))
        return;
    W.t0 = trace_now();
    if (pthread_create(&W.thread, NULL, warm_thread, NULL) == 0)
        W.started = 1;
}

/* Staging buffers are plain memory: the file loop's thread fills them without a word to the GPU.  A buffer is REGISTERED with
   the device (page-locked in place: 13-15 ms per 256 MiB of touched memory, against 46-60 ms for allocating it pinned, and on
   the thread that starts the pass, not on the one that reads the files) the first time a pass reads from it, and stays so
   while it is reused.  tools/ubench/h2d_pageable.hip: 256 MiB from registered memory cross in 4.7 ms, from unregistered
   memory in 23-26 ms. */
#define STAGE_BUFFERS 8
static struct {
    uint8_t *p;
    size_t cap;
    int pinned;
} g_stage_buf[STAGE_BUFFERS];
static pthread_mutex_t g_stage_lock = PTHREAD_MUTEX_INITIALIZER;

static uint8_t *stage_alloc(size_t cap)
{
    void *p = NULL;
    if (posix_memalign(&p, (size_t)2 << 20, cap) != 0 || !p)
        FATAL_MALLOC("hip staging buffer");
#ifdef MADV_HUGEPAGE
    /* (fresh memory is faulted in by the file loop's own copies: 65 536 faults per 256 MiB in 4 KiB pages -- the first two
       passes' worth of files took 70 ms to read against 25 ms later --, 128 in 2 MiB pages where the system grants them) */
    (void)madvise(p, cap, MADV_HUGEPAGE);
#endif
    pthread_mutex_lock(&g_stage_lock);
    for (int k = 0; k < STAGE_BUFFERS; ++k)
        if (!g_stage_buf[k].p) {
            g_stage_buf[k].p      = p;
            g_stage_buf[k].cap    = cap;
            g_stage_buf[k].pinned = 0;
            break;
        }
    pthread_mutex_unlock(&g_stage_lock);
    return p;
}

static void stage_free(uint8_t *p)
{
    if (!p)
        return;
    pthread_mutex_lock(&g_stage_lock);
    for (int k = 0; k < STAGE_BUFFERS; ++k)
        if (g_stage_buf[k].p == p) {
            if (g_stage_buf[k].pinned)
                (void)r433_host_unregister(p);
            g_stage_buf[k].p = NULL;
        }
    pthread_mutex_unlock(&g_stage_lock);
    free(p);
}

/* before a pass reads from it (any thread; after warm_join) */
static void stage_pin(uint8_t *p)
{
    pthread_mutex_lock(&g_stage_lock);
    for (int k = 0; k < STAGE_BUFFERS; ++k)
        if (g_stage_buf[k].p == p && !g_stage_buf[k].pinned) {
            double const t0 = trace_now();
            if (r433_host_register(p, g_stage_buf[k].cap) == 0) /* (refused: the copy still works, from pageable memory) */
                g_stage_buf[k].pinned = 1;
            if (trace_on())
                fprintf(stderr, "hip flow: staging buffer of %zu MiB %s in %.1f ms\n", g_stage_buf[k].cap >> 20,
                        g_stage_buf[k].pinned ? "registered with the device" : "could not be registered (pageable copies)", trace_now() - t0);
        }
    pthread_mutex_unlock(&g_stage_lock);
}

static void stage_reserve(size_t need)
{
    if (need <= H.stage_cap)
        return;
    /* 8 MiB for the lone small file, then straight to the size a pass is cut at (STAGE_PASS below), doubling only for
       captures that are larger than that: a grown buffer has to be copied into (and registered again). */
    size_t cap = H.stage_cap ? H.stage_cap : H.lane_seen ? STAGE_PASS : (size_t)8 << 20; /* (a process that hands passes over has a list) */
    if (cap < need && cap < STAGE_PASS)
        cap = STAGE_PASS;
    while (cap < need)
        cap *= 2;
    uint8_t *p = stage_alloc(cap);
    if (H.stage_len)
        memcpy(p, H.stage, H.stage_len);
    stage_free(H.stage);
    H.stage     = p;
    H.stage_cap = cap;
}

static hip_capture *capture_open(r_cfg_t *cfg)
{
    struct dm_state *demod = cfg->demod;
    if (H.n_caps == H.caps_cap) {
        H.caps_cap = H.caps_cap ? H.caps_cap * 2 : 64;
        H.caps     = realloc(H.caps, H.caps_cap * sizeof(*H.caps));
        if (!H.caps)
            FATAL_REALLOC("hip captures");
    }
    warm_start(); /* (the first push of the process: the GPU is opened beside the file loop) */
    hip_capture *c = &H.caps[H.n_caps++];
    memset(c, 0, sizeof(*c));
    c->in_filename      = cfg->in_filename;
    c->load_info        = demod->load_info;
    c->samp_rate        = demod->samp_rate;
    c->center_frequency = demod->center_frequency;
    c->sample_size      = demod->sample_size;
    c->fpdm             = demod->fsk_pulse_detect_mode;
    c->input_pos        = demod->input_pos;
    H.stage_len         = (H.stage_len + 15) & ~(size_t)15;
    c->offset           = H.stage_len;
    H.open              = 1;
    return c;
}

static void capture_free(hip_capture *c)
{
    free(c->frame_pos);
    free(c->frame_now);
}

/* ---- replay: what the reference's frame loop does around its decoders (src/r_flow.c:166-189, 240-340) ---- */

/* squelch / auto level bookkeeping of the frames [cur_frames_done, upto) of the current capture */
static void account_frames(hip_capture *c, uint32_t stream, uint32_t upto)
{
    r_cfg_t *cfg           = H.cfg;
    struct dm_state *demod = cfg->demod;
    if (upto > c->n_frames)
        upto = c->n_frames;
    for (uint32_t f = H.cur_frames_done; f < upto; ++f) {
        uint32_t n_bytes   = f + 1 < c->n_frames ? c->frame_bytes : c->last_bytes;
        uint32_t n_samples = n_bytes / c->sample_size;
        uint32_t sum       = H.frame_sums && f < H.sums_cap ? H.frame_sums[(size_t)stream * H.sums_cap + f] : 0;
        float avg_db       = r433_level_db(sum, n_samples, c->sample_size == 4 || demod->use_mag_est);
        if (demod->min_level_auto == 0.0f) {
            demod->min_level_auto = demod->min_level;
        }
        if (demod->noise_level == 0.0f) {
            demod->noise_level = demod->min_level_auto - 3.0f;
        }
        int noise_only = avg_db < demod->noise_level + 3.0f;
        demod->total_frames_count += 1;
        if (noise_only) {
            demod->total_frames_squelch += 1;
            demod->noise_level = (demod->noise_level * 7 + avg_db) / 8;
            if (demod->auto_level > 0 && demod->noise_level < demod->min_level - 3.0f
                    && fabsf(demod->min_level_auto - demod->noise_level - 3.0f) > 1.0f) {
                demod->min_level_auto = demod->noise_level + 3.0f;
                if (!H.sync_active || (!H.sync_flush && f >= H.sync_frame)) /* (a frame replayed again for -E has said it already) */
                    print_logf(LOG_WARNING, "Auto Level", "Estimated noise level is %.1f dB, adjusting minimum detection level to %.1f dB",
                            demod->noise_level, demod->min_level_auto);
            }
        }
        else {
            demod->noise_level = (demod->noise_level * 31 + avg_db) / 32;
        }
    }
    if (upto > H.cur_frames_done)
        H.cur_frames_done = upto;
}

static void switch_to(uint32_t stream)
{
    r_cfg_t *cfg           = H.cfg;
    struct dm_state *demod = cfg->demod;
    hip_capture *c         = &H.group[stream];
    H.cur_stream           = stream;
    H.cur_frames_done      = 0;
    /* what reset_sdr_flow leaves behind (src/r_flow.c:79-97) and the file loop sets (src/rtl_433.c:1706-1711) */
    demod->min_level_auto        = 0.0f;
    demod->noise_level           = 0.0f;
    cfg->in_filename             = c->in_filename;
    demod->load_info             = c->load_info;
    cfg->samp_rate               = c->samp_rate;
    cfg->center_frequency        = c->center_frequency;
    demod->samp_rate             = c->samp_rate;
    demod->center_frequency      = c->center_frequency;
    demod->sample_size           = c->sample_size;
    demod->fsk_pulse_detect_mode = c->fpdm;
}

/* make `stream` the capture the replay is in; captures on the way (no packages) still count their frames */
static void enter_capture(uint32_t stream)
{
    if (H.cur_stream == stream)
        return;
    uint32_t s = 0;
    if (H.cur_stream != UINT32_MAX) {
        account_frames(&H.group[H.cur_stream], H.cur_stream, UINT32_MAX);
        s = H.cur_stream + 1;
    }
    for (; s < stream; ++s) {
        switch_to(s);
        account_frames(&H.group[s], s, UINT32_MAX);
    }
    switch_to(stream);
}

static void on_package_begin(void *user, r433_pkg_rec const *rec, r433_pulse_data const *pulses)
{
    (void)user;
    r_cfg_t *cfg           = H.cfg;
    struct dm_state *demod = cfg->demod;
    char time_str[LOCAL_TIME_BUFLEN];

    enter_capture(rec->stream);
    hip_capture *c = &H.group[rec->stream];
    /* the push_sdr_flow call the reference returned this package in (the flush call comes after the last frame) */
    uint32_t f = rec->frame < c->n_frames ? rec->frame : c->n_frames - 1;
    account_frames(c, rec->stream, f + 1);
    demod->sample_file_pos = c->frame_pos[f];
    demod->now             = c->frame_now[f];

    if (rec->type == R433_PKG_OOK) {
        memcpy(&demod->pulse_data, pulses, sizeof(pulse_data_t));
        demod->pulse_data.offset += c->input_pos;
        /* the FSK candidate next to it: cleared at the package start, no estimates stored (src/pulse_detect.c:313-324) */
        pulse_data_clear(&demod->fsk_pulse_data);
        demod->fsk_pulse_data.sample_rate = pulses->sample_rate;
        demod->fsk_pulse_data.start_ago   = pulses->start_ago;
        if (demod->analyze_pulses) {
            fprintf(stderr, "Detected OOK package\t%s\n", time_pos_str(cfg, demod->pulse_data.start_ago, time_str));
        }
    }
    else {
        memcpy(&demod->fsk_pulse_data, pulses, sizeof(pulse_data_t));
        demod->fsk_pulse_data.offset += c->input_pos;
        /* the OOK side is still in its first pulse; the handlers only look at its start_ago (src/r_api.c:820) */
        pulse_data_clear(&demod->pulse_data);
        demod->pulse_data.sample_rate = pulses->sample_rate;
        demod->pulse_data.offset      = demod->fsk_pulse_data.offset;
        demod->pulse_data.start_ago   = pulses->start_ago;
        demod->pulse_data.end_ago     = pulses->end_ago;
        if (demod->analyze_pulses) {
            fprintf(stderr, "Detected FSK package\t%s\n", time_pos_str(cfg, demod->fsk_pulse_data.start_ago, time_str));
        }
    }
}

static char const *slicer_name(unsigned modulation)
{
    switch (modulation) {
    case OOK_PULSE_PCM:
    case FSK_PULSE_PCM: return "pulse_slicer_pcm";
    case OOK_PULSE_PPM: return "pulse_slicer_ppm";
    case OOK_PULSE_PWM:
    case FSK_PULSE_PWM: return "pulse_slicer_pwm";
    case OOK_PULSE_MANCHESTER_ZEROBIT:
    case FSK_PULSE_MANCHESTER_ZEROBIT: return "pulse_slicer_manchester_zerobit";
    case OOK_PULSE_PIWM_RAW: return "pulse_slicer_piwm_raw";
    case OOK_PULSE_PIWM_DC: return "pulse_slicer_piwm_dc";
    case OOK_PULSE_DMC: return "pulse_slicer_dmc";
    case OOK_PULSE_PWM_OSV1: return "pulse_slicer_osv1";
    case OOK_PULSE_NRZS: return "pulse_slicer_nrzs";
    case OOK_PULSE_RZI: return "pulse_slicer_rzi";
    default: return "pulse_slicer";
    }
}

/* account_event's debug printout (src/pulse_slicer.c:49-59) */
static void on_event_done(void *user, r433_r_device *dev, int ret, r433_bitbuffer const *bb)
{
    (void)user;
    r_device *device        = (r_device *)dev;
    bitbuffer_t const *bits = (bitbuffer_t const *)bb;
    unsigned max_bits       = 0;
    for (int row = 0; row < bits->num_rows; ++row) {
        if (bits->bits_per_row[row] > max_bits) {
            max_bits = bits->bits_per_row[row];
        }
    }
    if (!device->decode_fn || (device->verbose && ret > 0) || (device->verbose > 1 && max_bits > 16) || (device->verbose > 2)) {
        decoder_log_bitbuffer(device, ret > 0 ? 1 : 2, slicer_name(device->modulation), bits, device->name);
    }
}

static char *text_reserve(size_t need)
{
    if (need > H.text_cap) {
        free(H.text);
        H.text_cap = need + (need >> 2) + 4096;
        H.text     = malloc(H.text_cap);
        if (!H.text)
            FATAL_MALLOC("hip report text");
    }
    return H.text;
}

/* `-w file.ook` / `-w file.vcd` (src/r_flow.c:276-287,318-329 -> pulse_data_dump / pulse_data_print_vcd, src/pulse_data.c:102-224):
   the package as the library renders it (r433_pulse_text_dump / r433_pulse_vcd), written where the reference writes it */
static void dump_package(file_info_t const *dumper, pulse_data_t const *pd, int is_ook)
{
    char time_str[LOCAL_TIME_BUFLEN];
    for (int pass = 0; pass < 2; ++pass) { /* (snprintf convention: the second time round the buffer is large enough) */
        int const n = dumper->format == VCD_LOGIC
                ? r433_pulse_vcd((r433_pulse_data const *)pd, is_ook ? '\'' : '"', H.text, H.text_cap)
                : r433_pulse_text_dump((r433_pulse_data const *)pd, usecs_time_str(time_str, NULL, 1, 0), H.text, H.text_cap);
        if (n < 0)
            hip_fatal("pulse text");
        if ((size_t)n < H.text_cap) {
            if (fwrite(H.text, 1, (size_t)n, dumper->file) != (size_t)n)
                print_log(LOG_ERROR, __func__, "Short write, disk full?");
            return;
        }
        text_reserve((size_t)n + 1);
    }
}

/* `-A` (src/r_flow.c:295-298,313-316 -> pulse_analyzer, src/pulse_analyzer.c:279-560): the histograms and the modulation
   guess of every package of the pass were made on the device in one launch (r433_batch_analyze, replay_group); here the
   report of one package is rendered (r433_analysis_text: "Analyzing pulses..." through the flex-decoder suggestion) and the
   trial demodulation the reference ends with is run -- the guessed timings through the slicer the guess names
   (pulse_slicer_* = librtl433seam.so -> the GPU library), whose bitbuffer the analyzer's device logs like any decoder without
   a decode_fn (src/pulse_slicer.c:49-59). */
static void analyzer_report(r_cfg_t *cfg, pulse_data_t *pd, uint32_t pkg)
{
    r433_analysis const *a = &H.analysis[pkg];
    for (int pass = 0; pass < 2; ++pass) {
        int const n = r433_analysis_text(H.eng, pkg, a, H.text, H.text_cap);
        if (n < 0)
            hip_fatal("r433_analysis_text");
        if ((size_t)n < H.text_cap) {
            fwrite(H.text, 1, (size_t)n, stderr);
            break;
        }
        text_reserve((size_t)n + 1);
    }
    if (a->num_pulses == 0)
        return; /* "No pulses detected." (src/pulse_analyzer.c:281-284) */
    if (a->device.modulation) {
        r_device device    = {.log_fn = log_device_handler, .output_ctx = cfg};
        device.name        = "Analyzer Device"; /* src/pulse_analyzer.c:348-349 */
        device.verbose     = 2;
        device.modulation  = a->device.modulation;
        device.short_width = a->device.short_width;
        device.long_width  = a->device.long_width;
        device.reset_limit = a->device.reset_limit;
        device.gap_limit   = a->device.gap_limit;
        device.sync_width  = a->device.sync_width;
        device.tolerance   = a->device.tolerance;
        double const to_us = 1e6 / pd->sample_rate;
        if (device.modulation != FSK_PULSE_PCM && pd->num_pulses)
            pd->gap[pd->num_pulses - 1] = device.reset_limit / to_us + 1; /* "Be sure to terminate package", :530-550 */
        switch (device.modulation) {
        case FSK_PULSE_PCM: pulse_slicer_pcm(pd, &device); break;
        case OOK_PULSE_PPM: pulse_slicer_ppm(pd, &device); break;
        case OOK_PULSE_PWM:
        case FSK_PULSE_PWM: pulse_slicer_pwm(pd, &device); break;
        case OOK_PULSE_MANCHESTER_ZEROBIT: pulse_slicer_manchester_zerobit(pd, &device); break;
        default: break; /* "Unsupported" has been said */
        }
    }
    fprintf(stderr, "\n");
}

static void on_package_end(void *user, r433_pkg_rec const *rec, int p_events)
{
    (void)user;
    r_cfg_t *cfg           = H.cfg;
    struct dm_state *demod = cfg->demod;
    int const is_ook       = rec->type == R433_PKG_OOK;
    pulse_data_t *pd       = is_ook ? &demod->pulse_data : &demod->fsk_pulse_data;

    if (is_ook) {
        demod->total_frames_ook += 1;
        demod->frames_ook += 1;
    }
    else {
        demod->total_frames_fsk += 1;
        demod->frames_fsk += 1;
    }
    demod->total_frames_events += p_events > 0;
    demod->frames_events += p_events > 0;
    if (demod->grab_mode == 4 && H.quality_n < H.quality_cap) { /* src/r_flow.c:290-294,308-312 */
        int q = 0;
        if (p_events == 0) {
            r_device device = {.log_fn = log_device_handler, .output_ctx = cfg};
            q               = pulse_analyzer_check(pd, is_ook ? PULSE_DATA_OOK : PULSE_DATA_FSK, &device);
        }
        H.quality[H.quality_n++] = q;
    }

    for (void **iter = demod->dumper.elems; iter && *iter; ++iter) {
        file_info_t const *dumper = *iter;
        if ((dumper->format == VCD_LOGIC || dumper->format == PULSE_OOK) && dumper->file)
            dump_package(dumper, pd, is_ook);
    }
    if (demod->verbosity >= LOG_TRACE) {
        pulse_data_print(pd);
    }
    if (demod->raw_mode == 1 || (demod->raw_mode == 2 && p_events == 0) || (demod->raw_mode == 3 && p_events > 0)) {
        data_t *data = pulse_data_print_data(pd);
        event_occurred_handler(cfg, data);
    }
    if (demod->analyze_pulses && (demod->grab_mode <= 1 || (demod->grab_mode == 2 && p_events == 0) || (demod->grab_mode == 3 && p_events > 0))) {
        r433_dispatch_info at;
        if (r433_dispatch_current(&at) < 0 || at.package >= H.analysis_n)
            hip_fatal("pulse analyzer: no result for this package");
        analyzer_report(cfg, pd, at.package);
    }
}

/* ---- the GPU pass over a group of captures that share one flow configuration ---- */

static uint32_t capture_frame_samples(hip_capture const *c)
{
    /* a capture that came in one push has no frame boundary inside: any frame at least that long will do */
    uint32_t dflt = DEFAULT_BUF_LENGTH / c->sample_size;
    if (c->n_frames <= 1)
        return c->bytes / c->sample_size <= dflt ? dflt : ((c->bytes / c->sample_size + 63u) & ~63u);
    return c->frame_bytes / c->sample_size;
}

/* am.s16 / fm.s16 input files reach push_sdr_flow as they are (src/rtl_433.c:1735-1739): the library takes their words for
   the demodulated buffers where the reference's flow copies them there (src/r_flow.c:212-225) */
static uint32_t capture_input_format(hip_capture const *c)
{
    return c->load_info.format == S16_AM ? R433_IN_S16_AM : c->load_info.format == S16_FM ? R433_IN_S16_FM : R433_IN_NATIVE;
}

static void engine_config(r_cfg_t *cfg, hip_capture const *c, r433_flow_cfg *fc)
{
    struct dm_state *demod = cfg->demod;
    r433_flow_cfg_default(fc, c->sample_size, c->samp_rate);
    fc->frame_samples    = capture_frame_samples(c);
    fc->fpdm             = c->fpdm;
    fc->use_mag_est      = demod->use_mag_est;
    fc->enable_fm        = demod->enable_FM_demod;
    fc->fm_low_pass      = demod->fm_low_pass;
    fc->level_limit_db   = demod->level_limit;
    fc->min_level_db     = demod->min_level;
    fc->min_snr_db       = demod->min_snr;
    fc->auto_level       = demod->auto_level;
    fc->center_frequency = c->center_frequency;
    fc->input_format     = capture_input_format(c);
}

/* the engine of a flow configuration and a lane: found or made.  Touches H.engines only (the thread of a pass in flight calls it
   for its own lane while the file loop's thread replays the pass before: H.cur / H.eng are the caller's to set) */
static hip_engine *engine_get_locked(r_cfg_t *cfg, r433_flow_cfg const *fc, int lane, int hold);

/* hold: the caller is a pass that keeps the engine beyond this call (engine_release when its replay is done) */
static hip_engine *engine_get_held(r_cfg_t *cfg, r433_flow_cfg const *fc, int lane, int hold)
{
    warm_join(); /* (the first GPU call of the process is made below) */
    pthread_mutex_lock(&g_eng_mu);
    hip_engine *const e = engine_get_locked(cfg, fc, lane, hold);
    pthread_mutex_unlock(&g_eng_mu);
    return e;
}

static hip_engine *engine_get(r_cfg_t *cfg, r433_flow_cfg const *fc, int lane)
{
    return engine_get_held(cfg, fc, lane, 0);
}

static void engine_release(r433_batch const *eng)
{
    pthread_mutex_lock(&g_eng_mu);
    for (int k = 0; k < HIP_ENGINES; ++k)
        if (H.engines[k].eng == eng && eng && H.engines[k].busy > 0)
            H.engines[k].busy -= 1;
    pthread_mutex_unlock(&g_eng_mu);
}

static hip_engine *engine_get_locked(r_cfg_t *cfg, r433_flow_cfg const *fc, int lane, int hold)
{
    struct dm_state *demod = cfg->demod;
    void *first            = demod->r_devs.len ? demod->r_devs.elems[0] : NULL;
    hip_engine *slot       = NULL;
    for (int k = 0; k < HIP_ENGINES; ++k) {
        hip_engine *e = &H.engines[k];
        if (e->eng && memcmp(fc, &e->cfg, sizeof(*fc)) == 0 && e->devs == demod->r_devs.len && e->first_dev == first && e->lane == lane) {
            e->used = ++H.eng_clock;
            e->busy += hold;
            return e;
        }
        /* where a new engine would go: the first empty place, else the one that rested longest -- never one that a pass in
           flight, a pass owed its replay or the file loop's own run (H.cur) still works with: at most three of the four */
        if (e->eng && (e->busy || e == H.cur))
            continue;
        if (!slot || (slot->eng && (!e->eng || e->used < slot->used)))
            slot = e;
    }
    if (!slot)
        hip_fatal("no engine slot free");
    r433_batch_destroy(slot->eng);
    memset(slot, 0, sizeof(*slot));
    /* another set of decoders than the engines so far were made for (decoders registered or freed in between): what the
       pre-filter probe remembers about decoder objects may describe objects that are gone */
    if (H.probe_devs != demod->r_devs.len || H.probe_first != first) {
        if (H.probe_first)
            r433_prefilter_forget();
        H.probe_devs  = demod->r_devs.len;
        H.probe_first = first;
    }
    size_t n              = demod->r_devs.len;
    r433_dev_timing *rows = calloc(n ? n : 1, sizeof(*rows));
    if (!rows)
        FATAL_CALLOC("hip device rows");
    for (size_t i = 0; i < n; ++i) {
        r_device const *d  = demod->r_devs.elems[i];
        rows[i].modulation  = d->modulation;
        rows[i].short_width = d->short_width;
        rows[i].long_width  = d->long_width;
        rows[i].reset_limit = d->reset_limit;
        rows[i].gap_limit   = d->gap_limit;
        rows[i].sync_width  = d->sync_width;
        rows[i].tolerance   = d->tolerance;
        rows[i].priority    = d->priority;
    }
    double const t_create = trace_now();
    slot->eng = r433_batch_create(fc, rows, (uint32_t)n);
    if (trace_on())
        fprintf(stderr, "hip flow: engine created in %.1f ms\n", trace_now() - t_create);
    free(rows);
    if (!slot->eng)
        hip_fatal("r433_batch_create");
    /* This process lives as long as one file list: staging slots of 2 KB instead of 8 (include/r433_hip.h; a record over 2 KB --
       a PCM row of more than ~15 000 bits, a bitbuffer of twenty long rows -- is sliced again by the placing pass, same bytes).
       What the engine allocates, and the driver frees when the process is gone, falls from 7 GB to under 2 per engine; back to
       back the CLI took 330-1030 ms per run over 8192 files with the large slots, 250-400 ms with these (RTL433_HIP_STAGE_SLOT). */
    {
        char const *e = getenv("RTL433_HIP_STAGE_SLOT");
        if (r433_batch_set_staging_slot(slot->eng, e ? (uint32_t)strtoul(e, NULL, 0) : 2048u) < 0)
            hip_fatal("r433_batch_set_staging_slot");
    }
    if (getenv("RTL433_HIP_DEBUG")) /* development: R433_DEBUG_* switches of include/r433_hip.h for every engine of the flow (32: the replay's own trace) */
        (void)r433_batch_set_debug(slot->eng, (uint32_t)strtoul(getenv("RTL433_HIP_DEBUG"), NULL, 0));
    /* Which of these decoders keep nothing between two calls: the ordered replay may then spread one decoder's calls over its
       threads instead of keeping each decoder on one (a busy TPMS decoder alone was most of a replay).  The program that
       registered the decoders is the one that knows: of the reference's, four keep state in file-scope statics
       (src/devices/secplus_v1.c:142-143, secplus_v2.c:260-266, ikea_sparsnas.c:92, arad_ms_meter.c:256), and everything made by
       a create_fn (flex decoders among them) or carrying a decode_ctx owns a context.  A name of the list that is not among
       the registered decoders while its neighbours are means the list is stale: then nobody is declared stateless. */
    {
        static char const *const stateful[] = {"Security+ (Keyfob)", "Security+ 2.0 (Keyfob)", "IKEA Sparsnas Energy Meter Monitor",
                "Arad/Master Meter Dialog3G water utility meter"};
        uint8_t *flags = calloc(n ? n : 1, 1);
        if (!flags)
            FATAL_CALLOC("hip stateless flags");
        for (size_t i = 0; i < n; ++i) {
            r_device const *d = demod->r_devs.elems[i];
            int keeps         = d->decode_ctx != NULL || d->create_fn != NULL || !d->decode_fn;
            int statics       = !d->decode_fn;
            for (size_t k = 0; k < sizeof(stateful) / sizeof(stateful[0]); ++k)
                statics |= d->name && strcmp(d->name, stateful[k]) == 0;
            /* 2 = R433_KEEPS_CONTEXT: its state is the context of its create_fn (blueline, vivint, flex decoders): one replay
               thread, asked by the pre-filter with that context out of reach (include/r433_hip.h) */
            flags[i] = statics ? 0 : keeps ? (d->decode_ctx ? 2 : 0) : 1;
        }
        if (r433_batch_set_stateless(slot->eng, flags, (uint32_t)n) < 0)
            hip_fatal("r433_batch_set_stateless");
        free(flags);
    }
    slot->cfg       = *fc;
    slot->devs      = n;
    slot->first_dev = first;
    slot->used      = ++H.eng_clock;
    slot->lane      = lane;
    slot->busy      = hold;
    return slot;
}

static void engine_ensure(r_cfg_t *cfg, r433_flow_cfg const *fc, int lane)
{
    H.cur = engine_get(cfg, fc, lane);
    H.eng = H.cur->eng;
}

/* The replay is quiet and spread over threads (see the dispatch below): then no hook looks at single decoder calls, and the
   bitbuffers a decoder refuses on their head alone can stay on the device (r433_batch_probe_prefilter; the statistics of
   -M stats come out the same).  RTL433_HIP_PREFILTER=0 in the environment keeps every record coming. */
static int replay_is_chatty(struct dm_state *demod)
{
    for (void **iter = demod->r_devs.elems; iter && *iter; ++iter) {
        r_device const *d = *iter;
        if (!d->decode_fn || d->verbose)
            return 1;
    }
    return 0;
}

/* from how many bytes of samples on the decoders are asked (RTL433_HIP_PREFILTER_FROM: tests bring the moment into a short list) */
static size_t prefilter_from(void)
{
    char const *e = getenv("RTL433_HIP_PREFILTER_FROM");
    return e ? (size_t)strtoull(e, NULL, 0) : (size_t)1 << 33;
}

/* may_probe = 0: the caller runs beside a replay (the thread of a pass in flight).  The questions call every decoder's
   decode_fn with the decoder's output_fn / log_fn swapped for swallowers and a process-wide SIGSEGV / SIGBUS handler in place
   -- the very objects the replay of the pass before is calling at that moment: they are only ever asked by the file loop's
   thread, between the join of one pass and the start of the next (drain_queue), where no decoder is running. */
static void engine_prefilter_of(r_cfg_t *cfg, hip_engine *cur, int may_probe)
{
    struct dm_state *demod = cfg->demod;
    char const *env        = getenv("RTL433_HIP_PREFILTER");
    /* Asking every decoder costs 0.2-0.3 s once per process (50 000 heads and up to 65 534 tiny rows each, eight threads); what
       it saves is two thirds of the records: of their copy and of the replay's calls -- 15-20 ms per GiB of samples as dense
       with signals as the bench's, a few ms per GiB of mostly idle captures.  Measured with the CLI on the MI355X box
       (tools/gpu_r4_cli.sh): 8192 captures of 128 KiB (1 GiB) 0.46-0.78 s without the questions, 0.59-0.90 s with them asked
       half-way.  So: from 8 GiB of samples on, or when told to (RTL433_HIP_PREFILTER=1; =0: never). */
    int const worth        = (env && env[0] == '1') || H.staged_total >= prefilter_from();
    int const want         = !(env && env[0] == '0') && worth && !H.sync_active && replay_threads() > 1 && !replay_is_chatty(demod) && demod->r_devs.len;
    if (want && !cur->probed && may_probe) {
        cur->probed = 1;
        double const t_ask = trace_now();
        /* this program's decoders reach bitbuffer_invert / _search / _find_repeated_* through dropin/helper_wrap.c (ld --wrap) */
        r433_prefilter_set_helper_probe(r433_host_helper_probe);
        int const t = r433_batch_probe_prefilter(cur->eng, (r433_r_device *const *)demod->r_devs.elems, (uint32_t)demod->r_devs.len);
        if (t < 0)
            hip_fatal("r433_batch_probe_prefilter");
        cur->tables = t;
        if (trace_on())
            fprintf(stderr, "hip flow: pre-filter questions asked on the file loop's thread in %.1f ms after %zu bytes of samples, pass in flight: %d, %d decoders with a table\n",
                    trace_now() - t_ask, H.staged_total, P.active, t);
    }
    if (cur->tables > 0 && r433_batch_set_prefilter(cur->eng, want) < 0)
        hip_fatal("r433_batch_set_prefilter");
}

static void engine_prefilter(r_cfg_t *cfg)
{
    engine_prefilter_of(cfg, H.cur, 1);
}

/* would engine_prefilter_of ask its questions for an engine that has not been asked for yet? (same test, no side effect) */
static int prefilter_questions_due(r_cfg_t *cfg)
{
    struct dm_state *demod = cfg->demod;
    char const *env        = getenv("RTL433_HIP_PREFILTER");
    int const worth        = (env && env[0] == '1') || H.staged_total >= prefilter_from();
    return !(env && env[0] == '0') && worth && !H.sync_active && replay_threads() > 1 && !replay_is_chatty(demod) && demod->r_devs.len;
}

/* -E quit / -E hop: src/rtl_433.c:1136-1143 acts on the events of each push, so a push cannot be left for later.  The capture as
   far as it has been pushed is run again from its first sample (the detector is causal: the packages of the earlier frames
   come out as before) and only the packages the reference would have returned from THIS call go through the decoders. */
static int sync_filter(void *user, r433_pkg_rec const *rec)
{
    (void)user;
    int const from_flush = rec->ret_pos == R433_RET_FLUSH;
    if (H.sync_flush)
        return from_flush;
    return !from_flush && rec->frame == H.sync_frame;
}

/* the first sample of a capture whose dump bytes have not been written yet (everything, unless frames are replayed again for -E) */
static size_t sync_first_sample(hip_capture const *c, size_t n_samples)
{
    if (!H.sync_active)
        return 0;
    if (H.sync_flush)
        return n_samples;
    size_t from = (size_t)H.sync_frame * (c->frame_bytes / (unsigned)c->sample_size);
    return from < n_samples ? from : n_samples;
}

static uint8_t *conv_reserve(size_t need)
{
    if (need > H.conv_cap) {
        free(H.conv);
        H.conv = malloc(need);
        if (!H.conv)
            FATAL_MALLOC("hip dump conversion buffer");
        H.conv_cap = need;
    }
    return H.conv;
}

/* the library's name of a dump format that is a conversion of the IQ frame (src/r_flow.c:396-432,456-479); 0 = not one */
static int iq_dump_format(file_info_t const *dumper, unsigned sample_size, int *values_per_sample, int *out_width)
{
    *values_per_sample = 2;
    switch (dumper->format) {
    case CU8_IQ: *out_width = 1; return sample_size == 4 ? R433_DUMP_CU8_IQ : 0;
    case CS16_IQ: *out_width = 2; return sample_size == 2 ? R433_DUMP_CS16_IQ : 0;
    case CS8_IQ: *out_width = 1; return R433_DUMP_CS8_IQ;
    case CF32_IQ: *out_width = 4; return R433_DUMP_CF32_IQ;
    case F32_I: *values_per_sample = 1; *out_width = 4; return R433_DUMP_F32_I;
    case F32_Q: *values_per_sample = 1; *out_width = 4; return R433_DUMP_F32_Q;
    default: return 0;
    }
}

/* ---- the sample grabber (-S all | unknown | known; src/r_flow.c:136-147,342-362, src/samp_grab.c:97-233) ----
   The reference keeps the last 3 MiB of everything pushed in a ring and, when a frame with signal is over, saves the padded
   stretch around it to g<counter>_<freq>M_<rate>k.<type> (or a SigMF container).  Here the captures of a pass are still in the
   staging buffer after the replay: r433_batch_grab_plan says which byte range of which capture each file holds (from the
   package records and the decode results), and what the ring would have held from before a capture's first byte -- the
   tails of the captures pushed before it, zeros before the first -- comes from the queue and a 3 MiB history of earlier passes. */
#define GRAB_RING (12u * 262144u)  /* SIGNAL_GRABBER_BUFFER, include/rtl_433.h:22 */
#define GRAB_BLOCK (128u * 1024u)  /* src/samp_grab.c:95 */

/* the `need` bytes pushed right before capture `idx` of the queue, oldest first */
static void grab_write_before(size_t idx, size_t need, FILE *fp)
{
    size_t from_caps = 0, k = idx;
    while (k > 0 && from_caps < need) { /* how far back into the queue */
        --k;
        from_caps += H.caps[k].bytes;
    }
    size_t from_hist  = from_caps < need ? need - from_caps : 0;
    size_t zeros      = 0;
    if (from_hist > H.hist_len) {
        zeros     = from_hist - H.hist_len;
        from_hist = H.hist_len;
    }
    static uint8_t const zero[4096];
    for (size_t z = zeros; z > 0;) {
        size_t w = z < sizeof(zero) ? z : sizeof(zero);
        fwrite(zero, 1, w, fp);
        z -= w;
    }
    if (from_hist)
        fwrite(H.hist + H.hist_len - from_hist, 1, from_hist, fp);
    size_t skip = from_caps > need ? from_caps - need : 0; /* of the oldest capture reached */
    for (; k < idx; ++k) {
        fwrite(H.stage + H.caps[k].offset + skip, 1, H.caps[k].bytes - skip, fp);
        skip = 0;
    }
}

static void write_grabs(r_cfg_t *cfg, hip_capture *group, size_t n)
{
    struct dm_state *demod = cfg->demod;
    samp_grab_t *g         = demod->samp_grab;
    if (!g || !demod->grab_mode)
        return;
    if (H.sync_active) {
        if (!H.warned_sync_grab) {
            H.warned_sync_grab = 1;
            print_log(LOG_WARNING, "HIP", "the sample grabber (-S) is not served together with -E");
        }
        return;
    }
    if (demod->grab_mode < 1 || demod->grab_mode > 4)
        return;
    if (demod->grab_mode == 4 && r433_batch_set_package_quality(H.eng, H.quality, (uint32_t)H.quality_n) < 0)
        hip_fatal("r433_batch_set_package_quality");
    int count = r433_batch_grab_plan(H.eng, demod->grab_mode, NULL, 0);
    if (count < 0)
        hip_fatal("r433_batch_grab_plan");
    if (count == 0)
        return;
    r433_grab *plan = calloc((size_t)count, sizeof(*plan));
    if (!plan)
        FATAL_CALLOC("hip grab plan");
    if (r433_batch_grab_plan(H.eng, demod->grab_mode, plan, (uint32_t)count) < 0)
        hip_fatal("r433_batch_grab_plan");
    for (int k = 0; k < count; ++k) {
        hip_capture const *c = &group[plan[k].stream];
        (void)n;
        unsigned ss           = (unsigned)c->sample_size;
        unsigned signal_bsize = ss * plan[k].n_samples;
        signal_bsize += GRAB_BLOCK - (signal_bsize % GRAB_BLOCK);
        /* the ring's fill when the reference writes this file: everything pushed so far, at most the ring (src/samp_grab.c:84-87) */
        uint64_t fill = H.pushed_before_queue + plan[k].pushed;
        for (hip_capture const *q = H.caps; q < c; ++q)
            fill += q->bytes;
        unsigned sg_len = fill > GRAB_RING ? GRAB_RING : (unsigned)fill;
        if (signal_bsize > sg_len) {
            fprintf(stderr, "Signal bigger than buffer, signal = %u > buffer %u !!\n", signal_bsize, sg_len);
            signal_bsize = sg_len;
        }
        double freq_mhz = c->center_frequency / 1000000.0;
        double rate_khz = c->samp_rate / 1000.0;
        char f_name[64] = {0};
        while (1) {
            snprintf(f_name, sizeof(f_name), "g%03u_%gM_%gk.%s", g->sg_counter, freq_mhz, rate_khz,
                    g->sg_fileformat ? "sigmf" : ss == 2 ? "cu8" : "cs16");
            g->sg_counter++;
            if (access(f_name, F_OK) == -1)
                break;
        }
        fprintf(stderr, "*** Saving signal to file %s (%u samples, %u bytes)\n", f_name, plan[k].n_samples, signal_bsize);
        FILE *fp = fopen(f_name, "wb");
        if (!fp) {
            fprintf(stderr, "Failed to open %s\n", f_name);
            continue;
        }
        uint8_t wrap[4096];
        if (g->sg_fileformat) { /* the reference's SigMF container around the same bytes (src/sigmf.c, microtar) */
            int w = r433_sigmf_prefix(ss, c->samp_rate, c->center_frequency, signal_bsize, wrap, sizeof(wrap));
            if (w < 0)
                hip_fatal("r433_sigmf_prefix");
            fwrite(wrap, 1, (size_t)w, fp);
        }
        uint64_t end_byte = plan[k].byte_offset + plan[k].byte_len; /* where the stretch ends inside the capture */
        size_t inside     = end_byte < signal_bsize ? (size_t)end_byte : signal_bsize;
        if (inside < signal_bsize)
            grab_write_before((size_t)(c - H.caps), signal_bsize - inside, fp);
        fwrite(H.stage + c->offset + (end_byte - inside), 1, inside, fp);
        if (g->sg_fileformat) {
            int w = r433_sigmf_trailer(signal_bsize, wrap, sizeof(wrap));
            if (w < 0)
                hip_fatal("r433_sigmf_trailer");
            fwrite(wrap, 1, (size_t)w, fp);
        }
        fclose(fp);
    }
    free(plan);
}

static int replay_group(r_cfg_t *cfg, hip_capture *group, size_t n, int n_pkgs);

static int run_group(r_cfg_t *cfg, hip_capture *group, size_t n)
{
    struct dm_state *demod = cfg->demod;
    r433_flow_cfg fc;
    engine_config(cfg, &group[0], &fc);
    engine_ensure(cfg, &fc, H.lane);

    void const **ptrs = malloc(n * sizeof(*ptrs));
    uint32_t *bytes   = malloc(n * sizeof(*bytes));
    if (!ptrs || !bytes)
        FATAL_MALLOC("hip capture list");
    for (size_t i = 0; i < n; ++i) {
        ptrs[i]  = H.stage + group[i].offset;
        bytes[i] = group[i].bytes;
    }
    /* a `-w file.u8` dumper: the detection kernel paints the logic bytes (src/r_flow.c:236-237,271-272,314-315,364-371) */
    int want_logic = 0;
    for (void **iter = demod->dumper.elems; iter && *iter; ++iter) {
        file_info_t const *dumper = *iter;
        if (dumper->format == U8_LOGIC && dumper->file)
            want_logic = 1;
    }
    r433_batch_enable_logic_dump(H.eng, want_logic);
    /* `-w file.am.s16 / .fm.s16 / .am.f32 / .fm.f32`: the filtered envelope and discriminator of every sample, left behind by
       the detection kernel in (pinned) tap buffers (src/r_flow.c:436-453 writes demod->am_buf / buf.fm of each frame) */
    int want_taps = 0;
    for (void **iter = demod->dumper.elems; iter && *iter; ++iter) {
        file_info_t const *dumper = *iter;
        if (dumper->file && (dumper->format == S16_AM || dumper->format == S16_FM || dumper->format == F32_AM || dumper->format == F32_FM))
            want_taps = 1;
    }
    size_t tap_stride = 0;
    if (want_taps) {
        for (size_t i = 0; i < n; ++i) {
            size_t ns = group[i].bytes / group[i].sample_size;
            tap_stride = ns > tap_stride ? ns : tap_stride;
        }
        tap_stride = (tap_stride + 63) & ~(size_t)63;
        if (n * tap_stride > H.taps_cap) {
            for (int k = 0; k < 3; ++k) {
                r433_host_free(H.taps[k]);
                H.taps[k] = r433_host_alloc(n * tap_stride * sizeof(int16_t) + 64);
                if (!H.taps[k])
                    hip_fatal("pinned tap buffers");
            }
            H.taps_cap = n * tap_stride;
        }
        if (r433_batch_set_taps(H.eng, H.taps[0], H.taps[1], H.taps[2], tap_stride) < 0)
            hip_fatal("r433_batch_set_taps");
    }
    else {
        r433_batch_set_taps(H.eng, NULL, NULL, NULL, 0);
    }
    double const t_probe = trace_now();
    engine_prefilter(cfg);
    stage_pin(H.stage);
    double const t_run = trace_now();
    int n_pkgs = r433_batch_run_host(H.eng, ptrs, bytes, (uint32_t)n);
    if (trace_on())
        fprintf(stderr, "hip flow: %zu captures, %.1f MiB: pre-filter %.1f ms, GPU pass (H2D, kernels, D2H) %.1f ms\n", n, H.stage_len / 1048576.0,
                t_run - t_probe, trace_now() - t_run);
    if (n_pkgs >= 0 && want_taps) {
        for (void **iter = demod->dumper.elems; iter && *iter; ++iter) {
            file_info_t const *dumper = *iter;
            if (!dumper->file)
                continue;
            int const is_am = dumper->format == S16_AM || dumper->format == F32_AM;
            int const is_f32 = dumper->format == F32_AM || dumper->format == F32_FM;
            if (!is_am && dumper->format != S16_FM && dumper->format != F32_FM)
                continue;
            for (size_t i = 0; i < n; ++i) {
                size_t all_samples = group[i].bytes / group[i].sample_size;
                size_t from        = sync_first_sample(&group[i], all_samples);
                size_t n_samples   = all_samples - from;
                int16_t const *tap = H.taps[is_am ? 1 : 2] + i * tap_stride + from;
                void const *out    = tap;
                size_t out_len     = n_samples * sizeof(int16_t);
                if (is_f32 && n_samples) { /* scale from Q0.15, src/r_flow.c:444-453 */
                    out = conv_reserve(n_samples * sizeof(float));
                    if (r433_dump_convert_host(is_am ? R433_DUMP_F32_AM : R433_DUMP_F32_FM, 2, tap, H.conv, n_samples) < 0)
                        hip_fatal("r433_dump_convert_host");
                    out_len = n_samples * sizeof(float);
                }
                if (fwrite(out, 1, out_len, dumper->file) != out_len)
                    print_log(LOG_ERROR, __func__, "Short write, samples lost, exiting!");
            }
        }
    }
    if (n_pkgs >= 0 && want_logic) {
        uint8_t const *logic = NULL;
        uint64_t stride      = 0;
        if (r433_batch_logic_dump(H.eng, &logic, &stride) < 0)
            hip_fatal("r433_batch_logic_dump");
        for (void **iter = demod->dumper.elems; iter && *iter; ++iter) {
            file_info_t const *dumper = *iter;
            if (dumper->format != U8_LOGIC || !dumper->file)
                continue;
            for (size_t i = 0; i < n; ++i) {
                size_t n_samples = group[i].bytes / group[i].sample_size;
                size_t from      = sync_first_sample(&group[i], n_samples);
                if (fwrite(logic + i * stride + from, 1, n_samples - from, dumper->file) != n_samples - from)
                    print_log(LOG_ERROR, __func__, "Short write, samples lost, exiting!");
            }
        }
    }
    free(ptrs);
    free(bytes);
    if (n_pkgs < 0)
        hip_fatal("r433_batch_run_host");
    return replay_group(cfg, group, n, n_pkgs);
}

/* the host half of a pass: the records of H.eng's last run into the decoders, the files that hang on them */
static int replay_group(r_cfg_t *cfg, hip_capture *group, size_t n, int n_pkgs)
{
    struct dm_state *demod = cfg->demod;
    H.cfg        = cfg;
    H.group      = group;
    H.cur_stream = UINT32_MAX;
    H.quality_n  = 0;
    if (demod->grab_mode == 4 && (size_t)n_pkgs > H.quality_cap) {
        free(H.quality);
        H.quality = calloc((size_t)n_pkgs, sizeof(*H.quality));
        if (!H.quality)
            FATAL_CALLOC("hip package verdicts");
        H.quality_cap = (size_t)n_pkgs;
    }
    r433_batch_frame_sums(H.eng, &H.frame_sums, &H.sums_cap);
    text_reserve(1 << 16);
    H.analysis_n = 0;
    if (demod->analyze_pulses && n_pkgs > 0) { /* -A: every package of the pass through the analyzer kernel, one launch */
        if ((size_t)n_pkgs > H.analysis_cap) {
            free(H.analysis);
            H.analysis = calloc((size_t)n_pkgs, sizeof(*H.analysis));
            if (!H.analysis)
                FATAL_CALLOC("hip analyzer results");
            H.analysis_cap = (size_t)n_pkgs;
        }
        int const got = r433_batch_analyze(H.eng, H.analysis, (uint32_t)n_pkgs, NULL);
        if (got < 0)
            hip_fatal("r433_batch_analyze");
        H.analysis_n = (size_t)got;
    }
    /* The replay.  account_event's debug printout (a decoder without decode_fn, -vv) needs every bitbuffer after its
       decoder ran: then everything stays on this thread.  Otherwise the decoders are spread over host threads -- each
       decoder on one thread, its calls in reference order -- and what they hand to output_fn is committed here in
       reference order (r433_batch_dispatch_ordered). */
    int chatty    = replay_is_chatty(demod);
    int n_threads = replay_threads();
    int events;
    double const t_replay = trace_now();
    if (chatty || n_threads <= 1) {
        r433_dispatch_hooks hooks = {NULL, on_package_begin, on_event_done, on_package_end, H.sync_active ? sync_filter : NULL, NULL};
        events = r433_batch_dispatch_hooks(H.eng, (r433_r_device *const *)demod->r_devs.elems, (uint32_t)demod->r_devs.len, &hooks);
    }
    else {
        r433_dispatch_hooks hooks = {NULL, on_package_begin, NULL, on_package_end, H.sync_active ? sync_filter : NULL, NULL};
        events = r433_batch_dispatch_ordered(H.eng, (r433_r_device *const *)demod->r_devs.elems, (uint32_t)demod->r_devs.len, &hooks, (uint32_t)n_threads);
    }
    if (events == R433_EDECODER) {
        /* src/pulse_slicer.c:44-47 */
        print_logf(LOG_ERROR, "pulse_slicer", "%s: notify maintainer", r433_last_error());
        exit(1);
    }
    if (events < 0)
        hip_fatal("r433_batch_dispatch_hooks");
    if (trace_on())
        fprintf(stderr, "hip flow: %d packages replayed into %zu decoders on %d thread(s) %.1f ms, %d events\n", n_pkgs, demod->r_devs.len,
                chatty ? 1 : n_threads, trace_now() - t_replay, events);
    write_grabs(cfg, group, n);
    /* captures without packages (and the frames after the last package) still count their frames */
    enter_capture((uint32_t)n - 1);
    account_frames(&group[n - 1], (uint32_t)n - 1, UINT32_MAX);
    H.cur_stream = UINT32_MAX;
    return events;
}

static int same_group(r_cfg_t *cfg, hip_capture const *a, hip_capture const *b)
{
    (void)cfg;
    return a->sample_size == b->sample_size && a->samp_rate == b->samp_rate && a->fpdm == b->fpdm
            && a->center_frequency == b->center_frequency && capture_frame_samples(a) == capture_frame_samples(b)
            && capture_input_format(a) == capture_input_format(b);
}

/* ---- two passes in flight ------------------------------------------------------------------------------------------
   The file loop does one thing at a time (src/rtl_433.c:1703-1859); with a list to get through, the GPU leg of a pass can run
   while the decoders work through the pass before it and the loop reads the files of the pass after it: the queue is handed to
   a thread (its own engine, its own pinned buffer), and its replay happens at the NEXT drain -- in list order, on the calling
   thread, as ever.  Only for the plain case: one flow configuration in the queue, no dumper, no sample grabber, no -E.
   RTL433_HIP_OVERLAP=0 turns it off, =1 takes every pass (tests); by default passes of 512 captures and more. */
static void *pass_thread(void *arg)
{
    (void)arg;
    /* Everything of the pass that talks to the GPU happens here, the wait for the device's opening included (the first pass of
       a process): the file loop's thread is back at its files meanwhile, and at the next drain it replays the pass before
       this one through H.eng / H.cur -- which this thread therefore leaves alone: its engine is in `mine` and P.eng. */
    hip_engine *const mine = P.mine ? P.mine : engine_get_held(P.cfg, &P.fc, P.lane, 1);
    r433_batch_enable_logic_dump(mine->eng, 0);
    r433_batch_set_taps(mine->eng, NULL, NULL, NULL, 0);
    engine_prefilter_of(P.cfg, mine, 0); /* (switches on what pass_start had asked; asks nothing itself: a replay may be running) */
    P.eng = mine->eng;
    stage_pin(P.stage);
    P.n_pkgs = r433_batch_run_host(P.eng, P.ptrs, P.bytes, (uint32_t)P.n);
    if (P.n_pkgs < 0)
        snprintf(P.err, sizeof(P.err), "%s", r433_last_error());
    return NULL;
}

static int pass_may_overlap(r_cfg_t *cfg, size_t n_run)
{
    struct dm_state *demod = cfg->demod;
    char const *e          = getenv("RTL433_HIP_OVERLAP");
    if (e && *e == '0')
        return 0;
    if (H.final_drain || H.open || H.sync_active || cfg->after_successful_events_flag || demod->samp_grab || n_run == 0)
        return 0;
    if (!(e && *e == '1') && n_run < 512)
        return 0;
    for (void **iter = demod->dumper.elems; iter && *iter; ++iter)
        if (((file_info_t const *)*iter)->file)
            return 0;
    for (size_t i = 0; i < n_run; ++i) {
        hip_capture const *c = &H.caps[i];
        if (c->irregular || c->bytes == 0 || c->n_frames == 0 || (c->n_frames > 1 && c->frame_bytes / c->sample_size % 64 != 0)
                || !same_group(cfg, &H.caps[0], c))
            return 0;
    }
    return 1;
}

/* the queue leaves for the GPU on a thread of its own; H gets an empty queue and the other pinned buffer */
static void pass_start(r_cfg_t *cfg, size_t n)
{
    engine_config(cfg, &H.caps[0], &P.fc);
    P.cfg   = cfg;
    P.lane  = H.lane; /* (the engine of this lane: made or found by the pass's own thread) */
    P.eng   = NULL;
    P.mine  = NULL;
    /* The decoder pre-filter's questions, when they are due for this lane's engine, are asked HERE: the pass before has been
       joined, its replay has not begun, no decoder is running anywhere.  (On the pass's thread they ran beside that replay,
       through the same r_device objects.)  Costs the file loop the engine's making in that one pass. */
    if (prefilter_questions_due(cfg)) {
        hip_engine *const e = engine_get_held(cfg, &P.fc, P.lane, 1);
        if (!e->probed)
            engine_prefilter_of(cfg, e, 1);
        P.mine = e;
    }
    P.ptrs  = malloc(n * sizeof(*P.ptrs));
    P.bytes = malloc(n * sizeof(*P.bytes));
    if (!P.ptrs || !P.bytes)
        FATAL_MALLOC("hip capture list");
    for (size_t i = 0; i < n; ++i) {
        P.ptrs[i]  = H.stage + H.caps[i].offset;
        P.bytes[i] = H.caps[i].bytes;
        H.pushed_before_queue += H.caps[i].bytes;
    }
    P.caps      = H.caps;
    P.n         = n;
    P.stage     = H.stage;
    P.stage_cap = H.stage_cap;
    P.stage_len = H.stage_len;
    P.n_pkgs    = 0;
    P.err[0]    = '\0';
    P.t_start   = trace_now();
    /* the file loop goes on into a queue and a buffer of its own */
    H.caps      = NULL;
    H.n_caps    = 0;
    H.caps_cap  = 0;
    H.stage     = H.spare_stage;
    H.stage_cap = H.spare_cap;
    H.stage_len = 0;
    H.spare_stage = NULL;
    H.spare_cap   = 0;
    H.lane ^= 1;
    H.lane_seen = 1;
    if (pthread_create(&P.thread, NULL, pass_thread, NULL) != 0) {
        print_log(LOG_FATAL, "HIP", "pthread_create failed");
        exit(1);
    }
    P.active = 1;
}

/* the replay that is owed, of a pass whose thread has been joined: the decoders, in list order, on this thread */
static int pass_replay(r_cfg_t *cfg, pending_pass *d)
{
    if (trace_on())
        fprintf(stderr, "hip flow: %zu captures, %.1f MiB: GPU pass on its own thread, taken %.1f ms after its start\n", d->n, d->stage_len / 1048576.0,
                trace_now() - d->t_start);
    free(d->ptrs);
    free(d->bytes);
    if (d->n_pkgs < 0) {
        print_logf(LOG_FATAL, "HIP", "r433_batch_run_host: %s", d->err);
        exit(1);
    }
    r433_batch *keep_eng = H.eng;
    H.eng                = d->eng;
    int const events     = replay_group(cfg, d->caps, d->n, d->n_pkgs);
    H.eng                = keep_eng;
    engine_release(d->eng);
    for (size_t i = 0; i < d->n; ++i)
        capture_free(&d->caps[i]);
    free(d->caps);
    /* its pinned buffer (if the queue after it has not taken it already) serves a later queue */
    if (d->stage) {
        if (H.spare_stage)
            stage_free(H.spare_stage);
        H.spare_stage = d->stage;
        H.spare_cap   = d->stage_cap;
    }
    memset(d, 0, sizeof(*d));
    return events;
}

/* One pass over what is queued.  From the flow's own batch-limit drain (push_sdr_flow) the pass may stay in flight: it is
   replayed at the next drain, beside the GPU leg of the pass after it.  The PUBLIC drain below never leaves anything behind. */
static int drain_queue(struct r_cfg *cfg)
{
    struct dm_state *demod = cfg->demod;
    if (!demod || (H.n_caps == 0 && !P.active))
        return 0;
    /* what the host set for the file it is working on right now: restored after the replay */
    char const *keep_filename  = cfg->in_filename;
    file_info_t keep_load_info = demod->load_info;
    uint32_t keep_rate = cfg->samp_rate, keep_freq = cfg->center_frequency;
    uint32_t keep_drate = demod->samp_rate, keep_dfreq = demod->center_frequency;
    int keep_ss = demod->sample_size, keep_fpdm = demod->fsk_pulse_detect_mode;
    float keep_pos = demod->sample_file_pos, keep_noise = demod->noise_level, keep_auto = demod->min_level_auto;
    struct timeval keep_now = demod->now;

    int events   = 0;
    size_t n_run = H.n_caps - (H.open ? 1 : 0); /* a capture still being pushed to stays queued */
    static double t_last_drain;
    double const t_drain = trace_now();
    if (trace_on())
        fprintf(stderr, "hip flow: [%.1f ms] %zu captures queued over %.1f ms (since the start / the pass before)\n", t_drain - g_t_loaded, n_run,
                t_last_drain ? t_drain - t_last_drain : 0.0);
    if (pass_may_overlap(cfg, n_run)) {
        /* this queue to the GPU, THEN the replay of the pass before it -- beside it */
        if (P.active)
            pthread_join(P.thread, NULL); /* (one GPU leg at a time: the pass before is off the device) */
        pending_pass done = P;            /* (after the join: its thread wrote the package count into P) */
        memset(&P, 0, sizeof(P));
        if (done.active && !H.spare_stage) {
            /* its samples are on the device and its replay reads none of them (no dumper, no grabber here): the buffer
               can take the queue after this one right away -- two pinned buffers in all, not three */
            H.spare_stage = done.stage;
            H.spare_cap   = done.stage_cap;
            done.stage    = NULL;
        }
        pass_start(cfg, n_run);
        if (done.active)
            events += pass_replay(cfg, &done);
        n_run = 0;
    }
    else if (P.active) { /* nothing may overtake the pass that is owed */
        pthread_join(P.thread, NULL);
        pending_pass done = P;
        memset(&P, 0, sizeof(P));
        events += pass_replay(cfg, &done);
    }
    for (size_t i = 0; i < n_run;) {
        hip_capture *c = &H.caps[i];
        if (c->irregular || c->bytes == 0 || c->n_frames == 0 || (c->n_frames > 1 && c->frame_bytes / c->sample_size % 64 != 0)) {
            if (c->bytes && c->n_frames)
                print_logf(LOG_ERROR, "HIP", "\"%s\": frames of unequal or odd length are not served by the HIP flow, capture skipped",
                        c->in_filename ? c->in_filename : "?");
            ++i;
            continue;
        }
        size_t j = i + 1;
        while (j < n_run && !H.caps[j].irregular && H.caps[j].bytes && same_group(cfg, c, &H.caps[j]))
            ++j;
        events += run_group(cfg, c, j - i);
        i = j;
    }

    /* the sample grabber's history: what was pushed last, for signals near the start of the captures of the next pass */
    if (demod->samp_grab && n_run) {
        uint8_t *h = malloc(GRAB_RING);
        if (!h)
            FATAL_MALLOC("hip grab history");
        size_t have = 0, k = n_run;
        while (k > 0 && have < GRAB_RING) { /* newest first, filled from the end */
            --k;
            size_t take = H.caps[k].bytes < GRAB_RING - have ? H.caps[k].bytes : GRAB_RING - have;
            memcpy(h + GRAB_RING - have - take, H.stage + H.caps[k].offset + (H.caps[k].bytes - take), take);
            have += take;
        }
        if (have < GRAB_RING && H.hist_len) {
            size_t take = H.hist_len < GRAB_RING - have ? H.hist_len : GRAB_RING - have;
            memcpy(h + GRAB_RING - have - take, H.hist + H.hist_len - take, take);
            have += take;
        }
        memmove(h, h + GRAB_RING - have, have);
        free(H.hist);
        H.hist     = h;
        H.hist_len = have;
    }
    for (size_t i = 0; i < n_run; ++i)
        H.pushed_before_queue += H.caps[i].bytes;
    /* keep a capture that is still open at the front of the queue */
    for (size_t i = 0; i < n_run; ++i)
        capture_free(&H.caps[i]);
    if (H.open) {
        hip_capture last = H.caps[H.n_caps - 1];
        memmove(H.stage, H.stage + last.offset, last.bytes);
        last.offset = 0;
        H.caps[0]   = last;
        H.n_caps    = 1;
        H.stage_len = last.bytes;
    }
    else {
        H.n_caps    = 0;
        H.stage_len = 0;
    }

    if (trace_on())
        fprintf(stderr, "hip flow: pass %.1f ms\n", trace_now() - t_drain);
    t_last_drain = trace_now();
    cfg->in_filename             = keep_filename;
    demod->load_info             = keep_load_info;
    cfg->samp_rate               = keep_rate;
    cfg->center_frequency        = keep_freq;
    demod->samp_rate             = keep_drate;
    demod->center_frequency      = keep_dfreq;
    demod->sample_size           = keep_ss;
    demod->fsk_pulse_detect_mode = keep_fpdm;
    demod->sample_file_pos       = keep_pos;
    demod->noise_level           = keep_noise;
    demod->min_level_auto        = keep_auto;
    demod->now                   = keep_now;
    return events;
}

/* For builds of the unmodified src/rtl_433.c: compiled with -Dclose_dumpers=hip_sdr_flow_close_dumpers the file loop's
   last act (src/rtl_433.c:1861) first drains what is still queued. */
/* The integration hook: called by the host once its file loop is over (or whenever it wants every event of
   what it has pushed so far).  Always complete: the pass in flight is joined and replayed, then the queue runs and is
   replayed, in that order -- when it returns no pass is in flight, no thread is running and every event has been delivered. */
int hip_sdr_flow_drain(struct r_cfg *cfg)
{
    int const nested = H.final_drain;
    H.final_drain    = 1; /* pass_may_overlap() says no: nothing is handed to a thread from here */
    int const events = drain_queue(cfg);
    warm_join(); /* (a list that was all empty files never made a GPU call: the opening thread is not left behind) */
    H.final_drain    = nested;
    return events;
}

void hip_sdr_flow_close_dumpers(struct r_cfg *cfg)
{
    hip_sdr_flow_drain(cfg);
    close_dumpers(cfg);
}

/* ---- the seam ---- */

int flush_sdr_flow(r_cfg_t *cfg)
{
    return push_sdr_flow(cfg, NULL, 0);
}

void reset_sdr_flow(r_cfg_t *cfg)
{
    struct dm_state *demod = cfg->demod;

    get_time_now(&demod->now);

    demod->frame_start_ago   = 0;
    demod->frame_end_ago     = 0;
    demod->frame_event_count = 0;
    demod->frame_quality     = 0;

    demod->min_level_auto = 0.0f;
    demod->noise_level    = 0.0f;

    /* filter, discriminator and detector state live on the device, per capture: a new capture starts clean */
    H.open = 0;
    H.fm_note_rate = 0; /* baseband_demod_FM_reset zeroes the rate the notice is keyed on */
}

/* one push (or the flush) of the open capture, answered now: the capture so far through the GPU, its newest packages replayed */
static int sync_step(r_cfg_t *cfg, int flush)
{
    struct dm_state *demod = cfg->demod;
    if (!H.n_caps)
        return 0;
    hip_capture *c = &H.caps[H.n_caps - 1];
    if (!c->n_frames || !c->bytes)
        return 0;
    if (c->irregular || (c->n_frames > 1 && c->frame_bytes / c->sample_size % 64 != 0)) {
        if (flush)
            print_logf(LOG_ERROR, "HIP", "\"%s\": frames of unequal or odd length are not served by the HIP flow, capture skipped",
                    c->in_filename ? c->in_filename : "?");
        return 0;
    }
    /* what the host has set for this file is what the replay wants: nothing to restore but the clocks of the push */
    float keep_pos = demod->sample_file_pos, keep_noise = demod->noise_level, keep_auto = demod->min_level_auto;
    struct timeval keep_now = demod->now;
    unsigned base_count = demod->total_frames_count, base_squelch = demod->total_frames_squelch;

    H.sync_active = 1;
    H.sync_flush  = flush;
    H.sync_frame  = c->n_frames - 1;
    int events    = run_group(cfg, c, 1);
    H.sync_active = 0;

    /* the frames of the capture were counted from its first one again: keep what this push adds */
    unsigned made_count = demod->total_frames_count - base_count, made_squelch = demod->total_frames_squelch - base_squelch;
    demod->total_frames_count   = base_count + (made_count - H.sync_count);
    demod->total_frames_squelch = base_squelch + (made_squelch - H.sync_squelch);
    H.sync_count                = made_count;
    H.sync_squelch              = made_squelch;
    (void)keep_noise;
    (void)keep_auto; /* noise_level / min_level_auto: as the replay of all frames so far leaves them, like the reference's */
    demod->sample_file_pos = keep_pos;
    demod->now             = keep_now;
    if (flush) { /* the capture is done: nothing of it is left for a drain */
        H.pushed_before_queue += c->bytes;
        H.stage_len = c->offset;
        capture_free(c);
        H.n_caps -= 1;
        H.sync_count = H.sync_squelch = 0;
    }
    return events;
}

int push_sdr_flow(r_cfg_t *cfg, unsigned char *iq_buf, uint32_t len)
{
    struct dm_state *demod = cfg->demod;

    if (!demod) {
        return 0; // might happen when the demod closed and we get a last data frame
    }

    if (!len) {
        /* flush: the capture is complete */
        if (cfg->after_successful_events_flag && H.open) {
            int ev = sync_step(cfg, 1);
            H.open = 0;
            return ev;
        }
        H.open = 0;
        if (H.n_caps >= batch_limit(cfg) || H.stage_len + STAGE_PASS / 16 >= STAGE_PASS)
            return drain_queue(cfg); /* (the only drain that may leave its pass in flight) */
        return 0;
    }

    unsigned long n_samples = len / demod->sample_size;
    if (n_samples * demod->sample_size != len) {
        print_log(LOG_WARNING, __func__, "Sample buffer length not aligned to sample size!");
    }
    /* A receiver never flushes: this flow would queue samples for ever and say nothing (captures are run when they end).
       Frames one by one are what the function seam is for (make -C dropin refflow: the reference's own flow over librtl433seam.so). */
    if (cfg->in_files.len == 0 && !demod->load_info.format) {
        print_log(LOG_FATAL, "HIP", "live input is not served by the batch flow (it decodes whole captures): use the function-seam build, dropin/_build/rtl_433_refflow_hip");
        exit(1);
    }

    // Feed data to all raw outputs (e.g. rtl_tcp)
    for (void **iter = demod->raw_handler ? demod->raw_handler->elems : NULL; iter && *iter; ++iter) {
        raw_output_t *output = *iter;
        raw_output_frame(output, iq_buf, len);
    }

    get_time_now(&demod->now);

    /* what baseband_demod_FM(_cs16) tells a -vv user when it derives its filter: on the first frame after a reset and on
       every change of the sample rate (src/baseband.c:217-223,310-316; same float arithmetic, same wording) */
    if (demod->enable_FM_demod && H.fm_note_rate != demod->samp_rate) {
        float low_pass = demod->fm_low_pass != 0.0f ? demod->fm_low_pass : demod->fsk_pulse_detect_mode ? 0.2f : 0.1f; /* src/r_flow.c:204 */
        if (low_pass > 1e4f)
            low_pass = low_pass / demod->samp_rate;
        else if (low_pass >= 1.0f)
            low_pass = 1e6f / low_pass / demod->samp_rate;
        print_logf(LOG_NOTICE, "Baseband", demod->sample_size == 2 ? "FM low pass filter for %u Hz at cutoff %.0f Hz, %.1f us" : "low pass filter for %u Hz at cutoff %.0f Hz, %.1f us",
                demod->samp_rate, demod->samp_rate * (double)low_pass, 1e6 / (demod->samp_rate * (double)low_pass));
        H.fm_note_rate = demod->samp_rate;
    }

    hip_capture *c = H.open && H.n_caps ? &H.caps[H.n_caps - 1] : capture_open(cfg);
    if (c->n_frames && (c->sample_size != demod->sample_size || c->samp_rate != demod->samp_rate)) {
        H.open = 0; /* the stream changed without a flush */
        c      = capture_open(cfg);
    }
    if ((uint64_t)c->bytes + len > 0xfffffff0ull) {
        print_log(LOG_ERROR, "HIP", "captures are limited to 4 GiB");
        return -1;
    }
    if (c->n_frames == c->frames_cap) {
        c->frames_cap = c->frames_cap ? c->frames_cap * 2 : 8;
        c->frame_pos  = realloc(c->frame_pos, c->frames_cap * sizeof(*c->frame_pos));
        c->frame_now  = realloc(c->frame_now, c->frames_cap * sizeof(*c->frame_now));
        if (!c->frame_pos || !c->frame_now)
            FATAL_REALLOC("hip frame table");
    }
    if (c->n_frames == 0)
        c->frame_bytes = len;
    else if (c->last_bytes != c->frame_bytes)
        c->irregular = 1; /* a short frame that was not the last */
    c->last_bytes             = len;
    c->frame_pos[c->n_frames] = demod->sample_file_pos;
    c->frame_now[c->n_frames] = demod->now;
    c->n_frames += 1;

    stage_reserve(c->offset + c->bytes + len + 16);
    memcpy(H.stage + c->offset + c->bytes, iq_buf, len);
    c->bytes += len;
    H.staged_total += len;
    H.stage_len = c->offset + c->bytes;

    /* sample dumpers: the input's own format is a plain copy (src/r_flow.c:396,403); the package dumpers
       (.ook, .vcd) are written during the replay */
    int d_events = 0;
    for (void **iter = demod->dumper.elems; iter && *iter; ++iter) {
        file_info_t const *dumper = *iter;
        if (!dumper->file || dumper->format == VCD_LOGIC || dumper->format == PULSE_OOK) {
            continue;
        }
        if (dumper->format == U8_LOGIC) {
            continue; /* written when the capture has been through the detection kernel */
        }
        if ((dumper->format == CU8_IQ && demod->sample_size == 2) || (dumper->format == CS16_IQ && demod->sample_size == 4)) {
            if (fwrite(iq_buf, 1, len, dumper->file) != len) {
                print_log(LOG_ERROR, __func__, "Short write, samples lost, exiting!");
                d_events = -1;
            }
        }
        else if (dumper->format == S16_AM || dumper->format == S16_FM || dumper->format == F32_AM || dumper->format == F32_FM) {
            continue; /* written when the capture has been through the detection kernel (taps) */
        }
        else {
            int per_sample = 2, width = 1;
            int fmt = iq_dump_format(dumper, demod->sample_size, &per_sample, &width);
            if (fmt) { /* a conversion of the IQ frame: the library's dump kernel on this frame */
                size_t n_out   = n_samples * (size_t)per_sample;
                size_t out_len = n_out * (size_t)width;
                conv_reserve(out_len + 16);
                if (n_out && r433_dump_convert_host(fmt, demod->sample_size, iq_buf, H.conv, n_out) < 0)
                    hip_fatal("r433_dump_convert_host");
                if (fwrite(H.conv, 1, out_len, dumper->file) != out_len) {
                    print_log(LOG_ERROR, __func__, "Short write, samples lost, exiting!");
                    d_events = -1;
                }
            }
            else if (!H.warned_dump) {
                H.warned_dump = 1;
                print_logf(LOG_WARNING, "HIP", "sample dumper \"%s\" is not served by the HIP flow", dumper->spec);
            }
        }
    }

    demod->input_pos += n_samples;

    if (cfg->after_successful_events_flag && d_events >= 0)
        return sync_step(cfg, 0);
    return d_events;
}
