/* pipeline_host.c -- a C host that keeps several engines of librtl433hip.so in flight over one list of capture files.
 *
 * The rtl_433 file loop reads a file, pushes its frames, and moves on (reference src/rtl_433.c:1703-1859): one thing at a
 * time.  A host that has a LIST to get through -- an archive of recordings, a service that decodes uploads -- can overlap
 * the three legs of a pass instead: while the decoders work through the bitbuffers of pass k on the host threads, the GPU
 * already detects and slices pass k + 1, and pass k + 2 is being read from disk into pinned memory.  This program is that
 * host, in C, over the C ABI of include/r433_hip.h and the reference's own decoders (dropin/_build/libr433plugins.so):
 *
 *     worker thread of engine e:   read the files of its pass -> r433_batch_run_host (H2D, kernels, D2H)       [per engine]
 *     main thread, in list order:  r433_batch_dispatch_ordered into the decoders' decode_fn -> JSON lines      [one at a time]
 *
 * The decoders see every bitbuffer in the order the reference would have produced it (a decoder that keeps state between
 * calls -- src/devices/secplus_v1.c:142 -- depends on that), what they report is printed by the reference's own
 * data_print_jsons, one line per message, in list order.  With -e 1 the same code runs one pass at a time: same output.
 *
 *     pipeline_host [-g gpus] [-e engines] [-b captures per pass] [-t replay threads] [-s sample rate] [-f frequency] [-p] [-o] file.cu8 ...
 *       -g: spread the engines over that many GPUs of the node (0 = all visible).  Engine e lives on GPU e mod gpus
 *           (r433_batch_create_on) and the passes go to the engines in turn, so consecutive passes run on different GPUs;
 *           the replay stays one ordered stream on the main thread, the output is the one of a single GPU.  (The list is NOT
 *           cut into one contiguous half per GPU: the decoders must see the list in order, so the second half's results would
 *           sit in their engines until the first half is through -- the second GPU would fill its engines and stop.)  No
 *           collective library: one process, host memory is the meeting point (the multi-PROCESS form, one rank per GPU with a
 *           gather of the ranks' JSON lines, is bench.py --config 4 over torch.distributed / RCCL).
 *       -p: ask the decoders which bitbuffer heads they refuse and leave those records on the device (r433_batch_probe_prefilter)
 *       -o: keep every decoder's calls on one thread (default: the decoders the plugin library declares stateless are spread
 *           over the replay threads, r433_batch_set_stateless)
 *
 * Own code; C99 + pthreads.  Compiled against include/ only (r433_abi.h mirrors r_device). */
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "r433_abi.h"
#include "r433_hip.h"

/* dropin/plugins_shim.c */
void *r433p_create(void);
int r433p_devices(void *h, r433_r_device **out, int cap);
size_t r433p_take(void *h, char const **text, unsigned long *messages);
int r433p_stateless(void *h, unsigned char *flags, int cap);
void r433p_destroy(void *h);
r433_helper_probe *r433_host_helper_probe(int session); /* dropin/helper_wrap.c, linked into the plugin library */
void *r433p_render(void *user, void *device, void *data); /* r433_dispatch_hooks.output_render: data_t -> its JSON line, on the replay threads */

typedef struct leg {
    r433_batch *eng;
    pthread_t thread;
    pthread_mutex_t lock;
    pthread_cond_t wake;
    int state; /* 0 idle, 1 a pass is wanted, 2 the pass is done, 3 quit */
    /* the pass */
    char *const *files;
    size_t n_files;
    uint8_t *stage; /* pinned */
    size_t stage_cap;
    void const **ptrs;
    uint32_t *bytes;
    int n_pkgs;
    char err[256];
    double read_ms, gpu_ms;
} leg;

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* the files of a pass into the engine's pinned buffer, back to back (16-byte aligned), then the GPU leg */
static void run_pass(leg *g)
{
    double const t0 = now_ms();
    size_t at = 0;
    g->err[0] = '\0';
    for (size_t i = 0; i < g->n_files; ++i) {
        FILE *f = fopen(g->files[i], "rb");
        if (!f) {
            snprintf(g->err, sizeof(g->err), "%s: %s", g->files[i], strerror(errno));
            g->n_pkgs = -1;
            return;
        }
        fseek(f, 0, SEEK_END);
        long const len = ftell(f);
        fseek(f, 0, SEEK_SET);
        at = (at + 15) & ~(size_t)15;
        if (len < 0 || at + (size_t)len > g->stage_cap || (size_t)len > 0xfffffff0u) {
            snprintf(g->err, sizeof(g->err), "%s: larger than the staging buffer of a pass", g->files[i]);
            fclose(f);
            g->n_pkgs = -1;
            return;
        }
        size_t const got = fread(g->stage + at, 1, (size_t)len, f);
        fclose(f);
        g->ptrs[i]  = g->stage + at;
        g->bytes[i] = (uint32_t)(got & ~(size_t)1); /* whole cu8 samples */
        at += got;
    }
    double const t1 = now_ms();
    g->n_pkgs       = r433_batch_run_host(g->eng, g->ptrs, g->bytes, (uint32_t)g->n_files);
    if (g->n_pkgs < 0)
        snprintf(g->err, sizeof(g->err), "r433_batch_run_host: %s", r433_last_error());
    g->read_ms += t1 - t0;
    g->gpu_ms += now_ms() - t1;
}

static void *leg_main(void *arg)
{
    leg *g = arg;
    pthread_mutex_lock(&g->lock);
    for (;;) {
        while (g->state != 1 && g->state != 3)
            pthread_cond_wait(&g->wake, &g->lock);
        if (g->state == 3)
            break;
        pthread_mutex_unlock(&g->lock);
        run_pass(g);
        pthread_mutex_lock(&g->lock);
        g->state = 2;
        pthread_cond_broadcast(&g->wake);
    }
    pthread_mutex_unlock(&g->lock);
    return NULL;
}

static void leg_start(leg *g, char *const *files, size_t n)
{
    pthread_mutex_lock(&g->lock);
    g->files   = files;
    g->n_files = n;
    g->state   = 1;
    pthread_cond_broadcast(&g->wake);
    pthread_mutex_unlock(&g->lock);
}

static void leg_wait(leg *g)
{
    pthread_mutex_lock(&g->lock);
    while (g->state != 2)
        pthread_cond_wait(&g->wake, &g->lock);
    g->state = 0;
    pthread_mutex_unlock(&g->lock);
}

int main(int argc, char **argv)
{
    unsigned n_eng = 0, per_pass = 1024, threads = 16, rate = 250000, freq = 433920000;
    int n_gpus = 1;
    size_t file_room = 0; /* bytes of staging per capture; 0 = the largest file of the list */
    int prefilter = 0, quiet = 0, one_thread_each = 0;
    int a         = 1;
    for (; a < argc && argv[a][0] == '-' && argv[a][1]; ++a) {
        char const o = argv[a][1];
        if (o == 'p') {
            prefilter = 1;
            continue;
        }
        if (o == 'q') {
            quiet = 1;
            continue;
        }
        if (o == 'o') {
            one_thread_each = 1;
            continue;
        }
        if (a + 1 >= argc)
            break;
        unsigned long const v = strtoul(argv[++a], NULL, 10);
        if (o == 'e')
            n_eng = v ? (unsigned)v : 1;
        else if (o == 'g')
            n_gpus = (int)v;
        else if (o == 'b')
            per_pass = v ? (unsigned)v : 1;
        else if (o == 't')
            threads = v ? (unsigned)v : 1;
        else if (o == 's')
            rate = (unsigned)v;
        else if (o == 'f')
            freq = (unsigned)v;
        else {
            fprintf(stderr, "unknown option -%c\n", o);
            return 2;
        }
    }
    size_t const n_files = (size_t)(argc - a);
    if (n_files == 0) {
        fprintf(stderr, "usage: %s [-g gpus] [-e engines] [-b captures per pass] [-t replay threads] [-s rate] [-f frequency] [-p] [-o] [-q] file.cu8 ...\n", argv[0]);
        return 2;
    }
    int const visible = r433_device_count();
    if (visible <= 0) {
        fprintf(stderr, "no GPU: %s\n", r433_last_error());
        return 1;
    }
    if (n_gpus <= 0 || n_gpus > visible)
        n_gpus = visible;
    if (n_eng == 0)
        n_eng = n_gpus > 1 ? 2u * (unsigned)n_gpus : 3u; /* two passes in flight per GPU, three on a single one */
    char *const *files = argv + a;
    for (size_t i = 0; i < n_files; ++i) {
        FILE *f = fopen(files[i], "rb");
        if (!f) {
            fprintf(stderr, "%s: %s\n", files[i], strerror(errno));
            return 1;
        }
        fseek(f, 0, SEEK_END);
        long const len = ftell(f);
        fclose(f);
        if (len > 0 && (size_t)len + 16 > file_room)
            file_room = (size_t)len + 16;
    }
    if (per_pass > n_files)
        per_pass = (unsigned)n_files;
    size_t const n_passes = (n_files + per_pass - 1) / per_pass;
    if (n_eng > n_passes)
        n_eng = (unsigned)n_passes;
    if ((unsigned)n_gpus > n_eng)
        n_gpus = (int)n_eng;

    /* the decoders: the reference's, registered the way the CLI registers them */
    void *plugins = r433p_create();
    int const n_dev = plugins ? r433p_devices(plugins, NULL, 0) : 0;
    if (n_dev <= 0) {
        fprintf(stderr, "no decoders (libr433plugins.so)\n");
        return 1;
    }
    r433_r_device **devs = calloc((size_t)n_dev, sizeof(*devs));
    r433_dev_timing *rows = calloc((size_t)n_dev, sizeof(*rows));
    if (!devs || !rows || r433p_devices(plugins, devs, n_dev) != n_dev)
        return 1;
    for (int d = 0; d < n_dev; ++d) { /* what the slicers need to know of an r_device (include/r_device.h:59-92) */
        rows[d].modulation  = devs[d]->modulation;
        rows[d].short_width = devs[d]->short_width;
        rows[d].long_width  = devs[d]->long_width;
        rows[d].reset_limit = devs[d]->reset_limit;
        rows[d].gap_limit   = devs[d]->gap_limit;
        rows[d].sync_width  = devs[d]->sync_width;
        rows[d].tolerance   = devs[d]->tolerance;
        rows[d].priority    = devs[d]->priority;
    }
    unsigned char *stateless = one_thread_each ? NULL : calloc((size_t)n_dev, 1);
    if (stateless && r433p_stateless(plugins, stateless, n_dev) != n_dev)
        return 1;
    r433_flow_cfg cfg;
    r433_flow_cfg_default(&cfg, 2, rate);
    cfg.center_frequency = freq;
    cfg.fpdm             = freq > 800000000u; /* FSK_PULSE_DETECTOR_LIMIT, src/rtl_433.c:1094-1102 */

    leg *legs = calloc(n_eng, sizeof(*legs));
    for (unsigned e = 0; e < n_eng; ++e) {
        leg *g  = &legs[e];
        g->eng  = r433_batch_create_on((int)(e % (unsigned)n_gpus), &cfg, rows, (uint32_t)n_dev);
        if (!g->eng) {
            fprintf(stderr, "r433_batch_create_on(%u): %s\n", e % (unsigned)n_gpus, r433_last_error());
            return 1;
        }
        if (n_eng > (unsigned)n_gpus)
            r433_batch_set_exclusive_detect(g->eng, 2); /* the engines of one GPU take turns on the kernels of a pass */
        r433_batch_set_staging_slot(g->eng, 2048); /* a process that lives as long as its file list: a quarter of the device memory to allocate and to leave behind (include/r433_hip.h) */
        if (stateless && r433_batch_set_stateless(g->eng, stateless, (uint32_t)n_dev) != 0) {
            fprintf(stderr, "r433_batch_set_stateless: %s\n", r433_last_error());
            return 1;
        }
        if (prefilter && e == 0) /* the plugin library's decoders reach four bitbuffer helpers through wrappers (dropin/helper_wrap.c) */
            r433_prefilter_set_helper_probe(r433_host_helper_probe);
        if (prefilter && r433_batch_probe_prefilter(g->eng, devs, (uint32_t)n_dev) < 0) {
            fprintf(stderr, "r433_batch_probe_prefilter: %s\n", r433_last_error());
            return 1;
        }
        g->stage_cap = (size_t)per_pass * ((file_room + 15) & ~(size_t)15) + 64;
        g->stage     = r433_host_alloc(g->stage_cap);
        g->ptrs      = calloc(per_pass, sizeof(*g->ptrs));
        g->bytes     = calloc(per_pass, sizeof(*g->bytes));
        if (!g->stage || !g->ptrs || !g->bytes) {
            fprintf(stderr, "staging memory: %s\n", r433_last_error());
            return 1;
        }
        pthread_mutex_init(&g->lock, NULL);
        pthread_cond_init(&g->wake, NULL);
        if (pthread_create(&g->thread, NULL, leg_main, g) != 0)
            return 1;
    }

    double const t_start = now_ms();
    double replay_ms = 0, wait_ms = 0;
    long events = 0;
    unsigned long messages = 0;
    int rc = 0;
    /* prologue: one pass per engine in flight */
    for (size_t k = 0; k < n_eng && k < n_passes; ++k) {
        size_t const first = k * per_pass;
        leg_start(&legs[k % n_eng], files + first, first + per_pass <= n_files ? per_pass : n_files - first);
    }
    for (size_t k = 0; k < n_passes && rc == 0; ++k) {
        leg *g = &legs[k % n_eng];
        double const t0 = now_ms();
        leg_wait(g);
        double const t1 = now_ms();
        wait_ms += t1 - t0;
        if (g->n_pkgs < 0) {
            fprintf(stderr, "pass %zu: %s\n", k, g->err);
            rc = 1;
            break;
        }
        /* the replay, in list order: the decoders of pass k run while the other engines' passes are on the GPU; what they report
           is rendered where they ran, this thread appends the lines in reference order */
        r433_dispatch_hooks const hooks = {plugins, NULL, NULL, NULL, NULL, (void *(*)(void *, r433_r_device *, void *))r433p_render};
        int const ev = r433_batch_dispatch_ordered(g->eng, devs, (uint32_t)n_dev, &hooks, threads);
        if (ev < 0) {
            fprintf(stderr, "r433_batch_dispatch_ordered: %s\n", r433_last_error());
            rc = 1;
            break;
        }
        events += ev;
        char const *text = NULL;
        unsigned long n_msg = 0;
        size_t const len = r433p_take(plugins, &text, &n_msg);
        if (len && !quiet)
            fwrite(text, 1, len, stdout);
        messages += n_msg;
        replay_ms += now_ms() - t1;
        /* this engine's next pass */
        size_t const nk = k + n_eng;
        if (nk < n_passes) {
            size_t const first = nk * per_pass;
            leg_start(g, files + first, first + per_pass <= n_files ? per_pass : n_files - first);
        }
    }
    double const total_ms = now_ms() - t_start;
    double read_ms = 0, gpu_ms = 0;
    for (unsigned e = 0; e < n_eng; ++e) {
        leg *g = &legs[e];
        pthread_mutex_lock(&g->lock);
        while (g->state == 1) /* (after an error: a pass may still be running) */
            pthread_cond_wait(&g->wake, &g->lock);
        g->state = 3;
        pthread_cond_broadcast(&g->wake);
        pthread_mutex_unlock(&g->lock);
        pthread_join(g->thread, NULL);
        read_ms += g->read_ms;
        gpu_ms += g->gpu_ms;
        r433_batch_destroy(g->eng);
        r433_host_free(g->stage);
        free(g->ptrs);
        free(g->bytes);
    }
    fprintf(stderr, "pipeline_host: %u engine(s) on %d of %d visible GPU(s)\n", n_eng, n_gpus, visible);
    fprintf(stderr, "pipeline_host: %zu captures in %zu passes over %u engine(s): %.1f ms (reading %.1f ms and GPU legs %.1f ms on the engines' threads, "
                    "replay %.1f ms and waiting for a pass %.1f ms on the main thread), %ld decoded events, %lu messages\n",
            n_files, n_passes, n_eng, total_ms, read_ms, gpu_ms, replay_ms, wait_ms, events, messages);
    free(legs);
    free(stateless);
    free(devs);
    free(rows);
    r433p_destroy(plugins);
    return rc;
}
