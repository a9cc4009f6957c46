/* plugins_shim.c -- the reference's protocol decoders as a shared library of plugins (dropin/_build/libr433plugins.so).
 *
 * The decoders are the CONSUMERS of the hot path: unchanged host C behind r_device.decode_fn (reference
 * include/r_device.h:59-92).  A host that is not the rtl_433 CLI -- bench.py's multi-GPU run, a service that decodes
 * uploaded captures -- needs them without the CLI around them: this file, linked with the reference's sources compiled
 * where they lie (dropin/Makefile `plugins`; everything but src/rtl_433.c, src/r_flow.c and the four DSP units, which
 * librtl433seam.so replaces), registers the default
 * protocols the way the CLI does (register_all_protocols, src/r_api.c) and hands out the r_device instances.  What the
 * decoders report (data_t) is printed with the reference's own JSON printer (data_print_jsons, src/data.c) into a
 * buffer the host takes, one line per message -- the payload of the final event gather of BASELINE.json configs[3].
 *
 * Own code: only this file.  Compiled against the reference's headers; C99. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "data.h"
#include "list.h"
#include "logger.h"
#include "r_api.h"
#include "r_device.h"
#include "r_private.h"
#include "rtl_433.h"
#include "rtl_433_devices.h" /* flex_decoder */

typedef struct r433p {
    r_cfg_t *cfg;
    char *text;
    size_t len, cap;
    unsigned long messages;
} r433p;

static void quiet_log(log_level_t level, char const *src, char const *msg, void *userdata)
{
    (void)level;
    (void)src;
    (void)msg;
    (void)userdata;
}

/* A rendered message: what r433p_render hands the replay in a data_t's place.  take_message tells the two apart by the first
 * word: a data_t begins with its `next` pointer (include/data.h:70-72: NULL or a malloc'ed data_t), never with this value. */
#define R433P_LINE_MAGIC ((uintptr_t)0x52343333u | 1u) /* "R433", odd: no allocator returns it */
typedef struct r433p_line {
    uintptr_t magic;
    char *text;
} r433p_line;

/* the rendering: a pure function of the data (any thread) */
static char *render_line(data_t *data)
{
    char *line = data_print_jsons_dup(data); /* grows until the whole document fits: never a truncated line */
    if (!line)
        abort();
    data_free(data);
    return line;
}

/* the ordered append (the committing thread): takes the line */
static void append_line(r433p *h, char *line)
{
    size_t const n = strlen(line);
    if (h->cap - h->len < n + 2) {
        while (h->cap - h->len < n + 2)
            h->cap = h->cap ? h->cap * 2 : 65536;
        h->text = realloc(h->text, h->cap);
        if (!h->text)
            abort();
    }
    memcpy(h->text + h->len, line, n);
    free(line);
    h->len += n;
    h->text[h->len++] = '\n';
    h->text[h->len]   = '\0';
    h->messages += 1;
}

/* r_device.output_fn: a data_t from a decoder -- or, behind a replay that was given r433p_render as its output_render hook
 * (include/r433_hip.h r433_dispatch_hooks), the line the data was rendered to on the replay thread that ran the decoder. */
static void take_message(r_device *decoder, data_t *data)
{
    r433p *h = decoder->output_ctx;
    r433p_line *rendered = (r433p_line *)data;
    if (rendered->magic == R433P_LINE_MAGIC) {
        append_line(h, rendered->text);
        free(rendered);
    }
    else {
        append_line(h, render_line(data));
    }
}

/* r433_dispatch_hooks.output_render for these plugins (thread-safe: it only touches the data it is given) */
void *r433p_render(void *user, void *device, void *data)
{
    (void)user;
    (void)device;
    r433p_line *rendered = malloc(sizeof(*rendered));
    if (!rendered)
        abort();
    rendered->magic = R433P_LINE_MAGIC;
    rendered->text  = render_line(data);
    return rendered;
}

static void drop_log(r_device *decoder, int level, data_t *data)
{
    (void)decoder;
    (void)level;
    data_free(data);
}

/* all default-enabled protocols, in registration order, then one flex decoder per line of flex_specs (`-X` of the CLI,
   src/rtl_433.c:847-851; NULL or "": none) */
void *r433p_create_with(char const *flex_specs)
{
    r433p *h = calloc(1, sizeof(*h));
    if (!h)
        return NULL;
    r_logger_set_log_handler(quiet_log, NULL);
    h->cfg = r_create_cfg();
    register_all_protocols(h->cfg, 0);
    if (flex_specs && *flex_specs) {
        char *dup = strdup(flex_specs);
        if (!dup)
            abort();
        for (char *s = strtok(dup, "\n"); s; s = strtok(NULL, "\n")) {
            char *spec = strdup(s); /* (the flex parser keeps pointers into its argument) */
            if (!spec)
                abort();
            register_protocol(h->cfg, &flex_decoder, spec);
        }
        free(dup);
    }
    for (void **it = h->cfg->demod->r_devs.elems; it && *it; ++it) {
        r_device *d   = *it;
        d->output_fn  = take_message;
        d->log_fn     = drop_log;
        d->output_ctx = h;
    }
    return h;
}

void *r433p_create(void)
{
    return r433p_create_with(NULL);
}

int r433p_devices(void *hv, r_device **out, int cap)
{
    r433p *h = hv;
    int n    = 0;
    for (void **it = h->cfg->demod->r_devs.elems; it && *it; ++it, ++n)
        if (out && n < cap)
            out[n] = *it;
    return n;
}

/* Which of these decoders keep nothing between calls: what a host passes to r433_batch_set_stateless (include/r433_hip.h) so
 * that the ordered replay may spread a decoder's calls over its threads.  The plugin library is the one that knows: of the
 * reference's decoders four keep state in file-scope statics (src/devices/secplus_v1.c:142-143, secplus_v2.c:260-266,
 * ikea_sparsnas.c:92, arad_ms_meter.c:256), and every decoder made by a create_fn owns a context (flex, blueline, vivint,
 * arad_ms_meter); all others are functions of the bitbuffer they are handed.  Returns the number of decoders, -1 if the list
 * of names below no longer matches the registered decoders. */
int r433p_stateless(void *hv, unsigned char *flags, int cap)
{
    static char const *const stateful[] = {"Security+ (Keyfob)", "Security+ 2.0 (Keyfob)", "IKEA Sparsnas Energy Meter Monitor",
            "Arad/Master Meter Dialog3G water utility meter"};
    r433p *h = hv;
    int n    = 0;
    unsigned matched = 0; /* bit k: stateful[k] is among the registered decoders */
    for (void **it = h->cfg->demod->r_devs.elems; it && *it; ++it, ++n) {
        r_device const *d = *it;
        int keeps         = d->decode_ctx != NULL || d->create_fn != NULL;
        int statics       = 0;
        for (size_t k = 0; k < sizeof(stateful) / sizeof(stateful[0]); ++k)
            if (d->name && strcmp(d->name, stateful[k]) == 0) {
                statics = 1;
                matched |= 1u << k;
            }
        /* 1 stateless; 0 keeps state in file-scope statics (never asked by the pre-filter); 2 (R433_KEEPS_CONTEXT) all of its
         * state sits in the context its create_fn allocated (decoder_create, src/decoder_util.c:19-45: blueline, vivint, flex --
         * none of them has a file-scope variable; arad_ms_meter has both and is in the list above): one replay thread, and asked
         * by the pre-filter with decode_ctx out of reach */
        if (flags && n < cap)
            flags[n] = statics ? 0 : keeps ? (d->decode_ctx ? 2 : 0) : 1;
    }
    /* fail closed: a name of the list that matches no registered decoder means the list is stale (a decoder was renamed) --
     * answering then would declare the renamed decoder stateless and let its statics race on the replay threads */
    if (matched != (1u << (sizeof(stateful) / sizeof(stateful[0]))) - 1)
        return -1;
    return n;
}

/* the JSON lines since the last call (valid until the next message arrives); *messages = how many */
size_t r433p_take(void *hv, char const **text, unsigned long *messages)
{
    r433p *h   = hv;
    size_t len = h->len;
    if (text)
        *text = h->text ? h->text : "";
    if (messages)
        *messages = h->messages;
    h->len      = 0;
    h->messages = 0;
    return len;
}

void r433p_destroy(void *hv)
{
    r433p *h = hv;
    if (!h)
        return;
    r_free_cfg(h->cfg);
    free(h->text);
    free(h);
}
