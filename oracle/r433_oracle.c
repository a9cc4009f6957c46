/* r433_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the rtl_433 hot path (see r433_oracle.h).  Integer
 * semantics deliberately follow C on x86-64/gcc as the reference is built:
 * truncating division, arithmetic right shift of negatives, two's-complement
 * narrowing, uint32 wrap, float products rounded to float before truncation.
 * Build with -ffp-contract=off.
 */
#include "r433_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))

/* ------------------------------------------------------------------ baseband */

/* reference src/baseband.c:22-45: LUT (127-b)^2 per component, summed; the LUT is
 * just arithmetic so it is evaluated directly. */
uint32_t orc_envelope_cu8(uint8_t const *iq, uint16_t *env, uint32_t n)
{
    uint32_t acc = 0;
    for (uint32_t k = 0; k < n; ++k) {
        int di = 127 - (int)iq[2 * k];
        int dq = 127 - (int)iq[2 * k + 1];
        uint16_t v = (uint16_t)(di * di + dq * dq);
        env[k] = v;
        acc += v;
    }
    return acc;
}

/* reference src/baseband.c:65-79 */
uint32_t orc_magest_cu8(uint8_t const *iq, uint16_t *env, uint32_t n)
{
    uint32_t acc = 0;
    for (uint32_t k = 0; k < n; ++k) {
        int a = abs((int)iq[2 * k] - 128);
        int b = abs((int)iq[2 * k + 1] - 128);
        int hi = IMAX(a, b), lo = IMIN(a, b);
        uint16_t v = (uint16_t)(122 * hi + 51 * lo);
        env[k] = v;
        acc += v;
    }
    return acc;
}

/* reference src/baseband.c:96-110 */
uint32_t orc_magest_cs16(int16_t const *iq, uint16_t *env, uint32_t n)
{
    uint32_t acc = 0;
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t a = (uint32_t)abs((int)iq[2 * k]);
        uint32_t b = (uint32_t)abs((int)iq[2 * k + 1]);
        uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
        uint16_t v = (uint16_t)((122u * hi + 51u * lo) >> 8);
        env[k] = v;
        acc += v;
    }
    return acc;
}

/* reference src/baseband.c:44 / :78 with AMP_TO_DB / MAG_TO_DB (include/baseband.h:36-37) */
float orc_level_db(uint32_t sum, uint32_t n, int is_magnitude)
{
    float x = 1.0f;
    if (n > 0 && sum >= n)
        x = (float)sum / n;
    float lg = x > 0 ? log10f(x) : 0;
    return is_magnitude ? 20.0f * lg - 84.2884f : 10.0f * lg - 42.1442f;
}

/* reference src/baseband.c:145-169: a1 = FIX(0.85408)>>1, b0 = FIX(0.07296)>>1, Q14 */
void orc_lowpass(orc_lpf_state *st, uint16_t const *x, int16_t *y, uint32_t n)
{
    int const a1 = ((int)(0.85408 * 32768)) >> 1;
    int const b0 = ((int)(0.07296 * 32768)) >> 1;
    if (n < 1)
        return;
    int yp = st->y_prev;
    int xp = st->x_prev; /* signed re-read of the stored u16, reference :161,:167 */
    for (uint32_t k = 0; k < n; ++k) {
        int xc = x[k];
        int16_t yc = (int16_t)((a1 * yp + b0 * (xc + xp)) >> 14);
        y[k] = yc;
        yp = yc;
        xp = xc;
    }
    st->y_prev = (int16_t)yp;
    st->x_prev = (int16_t)(uint16_t)x[n - 1];
}

/* reference src/baseband.c:181-202 */
static int16_t atan2_q15(int32_t y, int32_t x)
{
    int32_t const q = 32767 / 4, q3 = 3 * 32767 / 4;
    int32_t ay = abs(y);
    int32_t ang;
    if (x == 0 && y == 0)
        return 0;
    if (x >= 0) {
        int32_t den = ay + x;
        if (den == 0)
            den = 1;
        ang = q - q * (x - ay) / den;
    }
    else {
        int32_t den = ay - x;
        if (den == 0)
            den = 1;
        ang = q3 - q * (x + ay) / den;
    }
    return (int16_t)(y < 0 ? -ang : ang);
}

/* reference src/baseband.c:281-300 (arguments arrive already truncated to int32) */
static int32_t atan2_q31(int32_t y, int32_t x)
{
    int64_t const q = 2147483647 / 4, q3 = 3ll * 2147483647 / 4;
    int64_t ay = abs(y);
    int64_t ang;
    if (x >= 0) {
        int64_t den = ay + x;
        if (den == 0)
            den = 1;
        ang = q - q * (x - ay) / den;
    }
    else {
        int64_t den = ay - x;
        if (den == 0)
            den = 1;
        ang = q3 - q * (x + ay) / den;
    }
    if (y < 0)
        ang = -ang;
    return (int32_t)ang;
}

/* reference src/baseband.c:217-232 and :310-325 */
void orc_fm_coeffs(float low_pass, uint32_t rate, int32_t *a16, int32_t *b16, int64_t *a32, int64_t *b32)
{
    if (low_pass > 1e4f)
        low_pass = low_pass / rate;
    else if (low_pass >= 1.0f)
        low_pass = 1e6f / low_pass / rate;
    double ita = 1.0 / tan(M_PI_2 * low_pass);
    double g16 = 1.0 / (1.0 + ita) / 2;
    double g32 = 1.0 / (1.0 + ita);
    if (a16)
        *a16 = (int)((ita - 1.0) * g16 * 32768);
    if (b16)
        *b16 = (int)(g16 * 32768);
    if (a32)
        *a32 = (int)((ita - 1.0) * g32 * 1073741824);
    if (b32)
        *b32 = (int)(g32 * 1073741824);
}

/* reference src/baseband.c:210-272 */
void orc_fm_cu8(orc_fm_state *st, uint8_t const *iq, int16_t *out, uint32_t n, uint32_t rate, float low_pass)
{
    if (st->rate != rate) {
        orc_fm_coeffs(low_pass, rate, &st->a16, &st->b16, NULL, NULL);
        st->rate = rate;
    }
    int16_t re = (int16_t)st->xr, im = (int16_t)st->xi, fq = (int16_t)st->xf, lp = (int16_t)st->yf;
    for (uint32_t k = 0; k < n; ++k) {
        int16_t re1 = re, im1 = im, fq1 = fq, lp1 = lp;
        re = (int16_t)(iq[2 * k] - 128);
        im = (int16_t)(iq[2 * k + 1] - 128);
        int32_t dot = re * re1 + im * im1;
        int32_t crs = im * re1 - re * im1;
        fq = atan2_q15(crs, dot);
        lp = (int16_t)((st->a16 * lp1 + st->b16 * (fq + fq1)) >> 14);
        out[k] = lp;
    }
    st->xr = re;
    st->xi = im;
    st->xf = fq;
    st->yf = lp;
}

/* reference src/baseband.c:303-366 */
void orc_fm_cs16(orc_fm_state *st, int16_t const *iq, int16_t *out, uint32_t n, uint32_t rate, float low_pass)
{
    if (st->rate != rate) {
        orc_fm_coeffs(low_pass, rate, NULL, NULL, &st->a32, &st->b32);
        st->rate = rate;
    }
    int32_t re = st->xr, im = st->xi, fq = st->xf, lp = st->yf;
    for (uint32_t k = 0; k < n; ++k) {
        int32_t re1 = re, im1 = im, fq1 = fq, lp1 = lp;
        re = iq[2 * k];
        im = iq[2 * k + 1];
        int64_t dot = (int64_t)re * re1 + (int64_t)im * im1;
        int64_t crs = (int64_t)im * re1 - (int64_t)re * im1;
        fq = atan2_q31((int32_t)crs, (int32_t)dot); /* implicit 64->32 truncation at the call, :352 */
        lp = (int32_t)((st->a32 * lp1 + st->b32 * ((int64_t)fq + fq1)) >> 30);
        out[k] = (int16_t)(lp >> 16);
    }
    st->xr = re;
    st->xi = im;
    st->xf = fq;
    st->yf = lp;
}

/* ------------------------------------------------------------------ detector */

/* include/baseband.h:44-47 DB_TO_AMP/DB_TO_MAG/_F and src/pulse_detect.c:24,86-105 */
void orc_levels_from_db(orc_levels *lv, int use_mag, float fixed_db, float min_db, float ratio_db)
{
    lv->use_mag = use_mag;
    if (use_mag) {
        lv->fixed_high = fixed_db < 0.0 ? (int)powf(10, (fixed_db + 84.2884f) / 20.0f) : 0;
        lv->min_high = (int)powf(10, (min_db + 84.2884f) / 20.0f);
        lv->ratio = (int)(0.5 + powf(10, ratio_db / 20.0f));
    }
    else {
        lv->fixed_high = fixed_db < 0.0 ? (int)powf(10, (fixed_db + 42.1442f) / 10.0f) : 0;
        lv->min_high = (int)powf(10, (min_db + 42.1442f) / 10.0f);
        lv->ratio = (int)(0.5 + powf(10, ratio_db / 10.0f));
    }
    lv->max_high = (int)powf(10, (0 + 42.1442f) / 10.0f); /* OOK_MAX_HIGH_LEVEL is always the AMP form */
}

/* src/pulse_detect_fsk.c:26-32 */
static void fsk_reset(orc_fsk_state *f)
{
    memset(f, 0, sizeof(*f));
    f->vmax = INT16_MIN;
    f->vmin = INT16_MAX;
    f->skip = 40;
}

void orc_detector_reset(orc_detector *d)
{
    d->state = 0;
    d->run = 0;
    d->max_pulse = 0;
    d->pos = 0;
    d->lead_in = 0;
    d->low = 0;
    d->high = 0;
    fsk_reset(&d->fsk);
}

static void pulses_clear(orc_pulses *p)
{
    memset(p, 0, sizeof(*p));
}

/* src/pulse_data.c:27-34 */
static void pulses_drop_half(orc_pulses *p)
{
    int const h = R433_PD_MAX_PULSES / 2;
    memmove(p->pulse, p->pulse + h, (R433_PD_MAX_PULSES - h) * sizeof(int32_t));
    memmove(p->gap, p->gap + h, (R433_PD_MAX_PULSES - h) * sizeof(int32_t));
    p->num -= h;
    p->offset += h;
}

/* src/pulse_detect_fsk.c:34-141 */
static void fsk_classic(orc_fsk_state *f, int16_t v, orc_pulses *out)
{
    int d1 = abs(v - f->f1);
    int d2 = abs(v - f->f2);
    f->run += 1;
    switch (f->state) {
    case 0:
        if (f->run < 10) {
            f->f1 = f->f1 / 2 + v / 2;
        }
        else if (d1 > 6000 / 2) {
            if (v > f->f1) { /* started low: a leading gap */
                f->state = 1;
                f->f2 = f->f1;
                f->f1 = v;
                out->pulse[0] = 0;
                out->gap[0] = (int32_t)f->run;
                out->num += 1;
                f->run = 0;
            }
            else { /* started high */
                f->state = 2;
                f->f2 = v;
                out->pulse[0] = (int32_t)f->run;
                f->run = 0;
            }
        }
        else {
            f->f1 += v / 16 - f->f1 / 16;
        }
        break;
    case 1:
        if (d1 > d2) {
            f->state = 2;
            if (f->run >= 10) {
                out->pulse[out->num] = (int32_t)f->run;
                f->run = 0;
            }
            else {
                f->run += out->gap[out->num - 1];
                out->num -= 1;
                if (out->num == 0 && out->pulse[0] == 0) {
                    f->f1 = f->f2;
                    f->state = 0;
                }
            }
        }
        else if (v > f->f1) {
            f->f1 += v / 16 - f->f1 / 16;
        }
        else {
            f->f1 += v / 64 - f->f1 / 64;
        }
        break;
    case 2:
        if (d2 > d1) {
            f->state = 1;
            if (f->run >= 10) {
                out->gap[out->num] = (int32_t)f->run;
                out->num += 1;
                f->run = 0;
                if (out->num >= R433_PD_MAX_PULSES)
                    pulses_drop_half(out);
            }
            else {
                f->run += out->pulse[out->num];
                if (out->num == 0)
                    f->state = 0;
            }
        }
        else if (v < f->f2) {
            f->f2 += v / 16 - f->f2 / 16;
        }
        else {
            f->f2 += v / 64 - f->f2 / 64;
        }
        break;
    default:
        break;
    }
}

/* src/pulse_detect_fsk.c:143-156 */
static void fsk_wrap_up(orc_fsk_state *f, orc_pulses *out)
{
    if (out->num < R433_PD_MAX_PULSES) {
        f->run += 1;
        if (f->state == 1) {
            out->pulse[out->num] = (int32_t)f->run;
            out->gap[out->num] = 0;
        }
        else {
            out->gap[out->num] = (int32_t)f->run;
        }
        out->num += 1;
    }
}

/* src/pulse_detect_fsk.c:158-221 */
static void fsk_minmax(orc_fsk_state *f, int16_t v, orc_pulses *out)
{
    if (f->skip == 0) {
        f->vmax = (int16_t)IMAX(v, f->vmax);
        f->vmin = (int16_t)IMIN(v, f->vmin);
        int16_t mid = (int16_t)((f->vmax + f->vmin) / 2);
        if (v > mid)
            f->vmax = (int16_t)(f->vmax - 10);
        if (v < mid)
            f->vmin = (int16_t)(f->vmin + 10);
        f->run += 1;
        switch (f->state) {
        case 0:
            f->state = v > mid ? 1 : 2;
            break;
        case 1:
            if (v < mid) {
                f->state = 2;
                out->pulse[out->num] = (int32_t)f->run;
                f->run = 0;
            }
            f->f2 += v / 64 - f->f2 / 64; /* (sic) high state feeds f2, :192 */
            break;
        case 2:
            if (v > mid) {
                f->state = 1;
                out->gap[out->num] = (int32_t)f->run;
                out->num += 1;
                f->run = 0;
                if (out->num >= R433_PD_MAX_PULSES)
                    pulses_drop_half(out);
            }
            f->f1 += v / 64 - f->f1 / 64; /* (sic) :208 */
            break;
        default:
            break;
        }
    }
    if (f->skip > 0)
        f->skip -= 1;
}

static void fsk_feed(orc_detector *d, int16_t v, orc_pulses *fsk, unsigned fpdm)
{
    if (fpdm == 0)
        fsk_classic(&d->fsk, v, fsk);
    else
        fsk_minmax(&d->fsk, v, fsk);
}

/* common tail of an FSK package return, src/pulse_detect.c:239-253 and :387-410 */
static int emit_fsk(orc_detector *d, orc_pulses *ook, orc_pulses *fsk, unsigned fpdm, int len)
{
    if (fpdm == 0)
        fsk_wrap_up(&d->fsk, fsk);
    fsk->fsk_f1 = d->fsk.f1;
    fsk->fsk_f2 = d->fsk.f2;
    fsk->ook_low = d->low;
    fsk->ook_high = d->high;
    ook->end_ago = (uint32_t)(len - d->pos);
    fsk->end_ago = (uint32_t)(len - d->pos);
    d->state = 0;
    return R433_PKG_FSK;
}

/* common tail of an OOK package return, :264-272, :431-439, :451-468 */
static int emit_ook(orc_detector *d, orc_pulses *ook, int len)
{
    d->state = 0;
    ook->ook_low = d->low;
    ook->ook_high = d->high;
    ook->end_ago = (uint32_t)(len - d->pos);
    return R433_PKG_OOK;
}

/* src/pulse_detect.c:199-483 */
int orc_detect_package(orc_detector *d, int16_t const *am, int16_t const *fm, int len, uint32_t rate,
        uint64_t sample_offset, orc_pulses *ook, orc_pulses *fsk, unsigned fpdm)
{
    if (len == 0) { /* flush, :204-278 */
        int st = d->state;
        if (st == 1) {
            if (d->run < 10) {
                if (ook->num <= 1) {
                    d->state = 0;
                    st = 0;
                }
                else {
                    st = 2; /* falls into gap-start handling with state := gap */
                }
            }
            else {
                ook->pulse[ook->num] = d->run;
                d->max_pulse = IMAX(d->run, d->max_pulse);
                d->run = 0;
                st = 2;
            }
        }
        if (st == 2) {
            d->state = 3;
            if (fsk->num > 16)
                return emit_fsk(d, ook, fsk, fpdm, len);
            st = 3;
        }
        if (st == 3) {
            ook->gap[ook->num] = d->run;
            ook->num += 1;
            return emit_ook(d, ook, len);
        }
    }

    int const per_ms = (int)(rate / 1000);
    d->high = IMAX(d->high, d->lv.min_high);
    if (d->pos == 0) {
        ook->start_ago += (uint32_t)len;
        fsk->start_ago += (uint32_t)len;
    }
    int spurious_eop = 0;

    while (d->pos < len) {
        int16_t const a = am[d->pos];
        int16_t thr = (int16_t)((d->low + IMIN(d->high, d->lv.max_high)) / 2);
        if (d->lv.fixed_high != 0)
            thr = (int16_t)d->lv.fixed_high;
        int16_t const hys = (int16_t)(thr / 8);

        switch (d->state) {
        case 0:
            if (a > thr + hys && d->lead_in > 1024) {
                pulses_clear(ook);
                pulses_clear(fsk);
                ook->sample_rate = fsk->sample_rate = rate;
                ook->offset = fsk->offset = sample_offset + (uint64_t)d->pos;
                ook->start_ago = fsk->start_ago = (uint32_t)(len - d->pos);
                d->run = 0;
                d->max_pulse = 0;
                fsk_reset(&d->fsk);
                d->state = 1;
            }
            else {
                int dl = a - d->low;
                d->low += dl / 1024;
                d->low += dl > 0 ? 1 : -1;
                d->high = IMAX(d->lv.ratio * d->low, d->lv.min_high);
                if (d->lead_in <= 1024)
                    d->lead_in += 1;
            }
            break;
        case 1:
            d->run += 1;
            if (a < thr - hys) {
                if (d->run < 10) {
                    if (ook->num <= 1) {
                        d->state = 0;
                    }
                    else {
                        spurious_eop = 1;
                        d->state = 3;
                    }
                }
                else {
                    ook->pulse[ook->num] = d->run;
                    d->max_pulse = IMAX(d->run, d->max_pulse);
                    d->run = 0;
                    d->state = 2;
                }
            }
            else {
                d->high += a / 64 - d->high / 64;
                d->high = IMAX(d->high, d->lv.min_high);
                ook->fsk_f1 += fm[d->pos] / 64 - ook->fsk_f1 / 64;
            }
            if (ook->num == 0)
                fsk_feed(d, fm[d->pos], fsk, fpdm);
            break;
        case 2:
            d->run += 1;
            if (a > thr + hys) {
                d->run += ook->pulse[ook->num];
                d->state = 1;
            }
            else if (d->run >= 10) {
                d->state = 3;
                if (fsk->num > 16)
                    return emit_fsk(d, ook, fsk, fpdm, len);
            }
            if (ook->num == 0)
                fsk_feed(d, fm[d->pos], fsk, fpdm);
            break;
        case 3:
            d->run += 1;
            if (a > thr + hys) {
                ook->gap[ook->num] = d->run;
                ook->num += 1;
                if (ook->num >= R433_PD_MAX_PULSES)
                    return emit_ook(d, ook, len);
                d->run = 0;
                d->state = 1;
            }
            if (spurious_eop || (d->run > 10 * d->max_pulse && d->run > 10 * per_ms) || d->run > 100 * per_ms) {
                ook->gap[ook->num] = d->run;
                ook->num += 1;
                return emit_ook(d, ook, len);
            }
            break;
        default:
            d->state = 0;
        }
        d->pos += 1;
    }
    d->pos = 0;
    return 0;
}

/* ------------------------------------------------------------------ bitbuffer writer */

void orc_bb_clear(orc_bitbuf *b)
{
    memset(b, 0, sizeof(*b));
}

static void bb_touch(orc_bitbuf *b)
{
    if (b->num_rows == 0)
        b->num_rows = b->free_row = 1;
}

/* src/bitbuffer.c:22-58 */
void orc_bb_add_bit(orc_bitbuf *b, int bit)
{
    bb_touch(b);
    unsigned r = b->num_rows - 1u;
    unsigned len = b->bits[r];
    if (len == 65535u)
        return;
    if (len > 0 && len % (R433_BB_COLS * 8) == 0) {
        if (b->free_row < R433_BB_ROWS)
            b->free_row++;
        else
            return;
    }
    b->bb[r * R433_BB_COLS + len / 8] |= (uint8_t)(bit << (7 - len % 8));
    b->bits[r] = (uint16_t)(len + 1);
    if (b->bits[r] > b->extent[r])
        b->extent[r] = b->bits[r];
}

/* src/bitbuffer.c:104-122 */
void orc_bb_add_row(orc_bitbuf *b)
{
    bb_touch(b);
    if (b->free_row < R433_BB_ROWS) {
        b->free_row++;
        b->num_rows = b->free_row;
    }
    else {
        b->bits[b->num_rows - 1] = 0;
    }
}

/* src/bitbuffer.c:124-133 */
void orc_bb_add_sync(orc_bitbuf *b)
{
    bb_touch(b);
    if (b->bits[b->num_rows - 1])
        orc_bb_add_row(b);
    b->syncs[b->num_rows - 1]++;
}

/* ------------------------------------------------------------------ slicers */

typedef struct slice_ctx {
    orc_event_fn fn;
    void *ctx;
    unsigned dev;
    unsigned calls;
    int events;
    orc_bitbuf bits;
} slice_ctx;

/* src/pulse_slicer.c:26-66 minus statistics/logging */
static void fire(slice_ctx *s)
{
    int r = s->fn ? s->fn(s->ctx, s->dev, s->calls, &s->bits) : 0;
    s->calls += 1;
    if (r > 0)
        s->events += r;
    orc_bb_clear(&s->bits);
}

typedef struct timing {
    int sh, lo, rst, gap, syn, tol;
    float us; /* samples per microsecond */
} timing;

/* src/pulse_slicer.c:70-87 (same block opens every slicer) */
static int timing_make(timing *t, r433_dev_timing const *d, uint32_t rate, int full_check)
{
    t->us = rate / 1.0e6f;
    t->sh = (int)(d->short_width * t->us);
    t->lo = (int)(d->long_width * t->us);
    t->rst = (int)(d->reset_limit * t->us);
    t->gap = (int)(d->gap_limit * t->us);
    t->syn = (int)(d->sync_width * t->us);
    t->tol = (int)(d->tolerance * t->us);
    if ((d->short_width > 0 && t->sh <= 0) || (d->long_width > 0 && t->lo <= 0) || (d->reset_limit > 0 && t->rst <= 0))
        return 0;
    if (full_check && ((d->gap_limit > 0 && t->gap <= 0) || (d->sync_width > 0 && t->syn <= 0) || (d->tolerance > 0 && t->tol <= 0)))
        return 0;
    return 1;
}

static int within(int v, int centre, int tol)
{
    return v >= centre - tol && v <= centre + tol;
}

/* src/pulse_slicer.c:68-259 */
static void slice_pcm(orc_pulses const *p, r433_dev_timing const *d, timing const *t, slice_ctx *s)
{
    float f_sh = d->short_width > 0.0f ? 1.0f / (d->short_width * t->us) : 0;
    float f_lo = d->long_width > 0.0f ? 1.0f / (d->long_width * t->us) : 0;
    int const gap_limit = t->gap ? t->gap : t->rst;
    int const max_zeros = gap_limit / t->lo;
    int tol = t->tol > 0 ? t->tol : t->lo / 4;
    int const rz = t->sh != t->lo;
    unsigned const np = p->num;

    int need = rz ? 4 : 12;
    int preamble = 0;
    if (rz) { /* :105-132 longest run of in-tolerance RZ bits tunes both periods */
        for (unsigned n = 0; n < np; ++n) {
            int sw = 0, lw = 0, cnt = 0;
            while (n < np && within(p->pulse[n], t->sh, tol) && within(p->pulse[n] + p->gap[n], t->lo, tol)) {
                sw += p->pulse[n];
                lw += p->pulse[n] + p->gap[n];
                cnt++;
                n++;
            }
            if (cnt >= need) {
                f_lo = (float)cnt / lw;
                f_sh = (float)cnt / sw;
                need = cnt;
                preamble = cnt;
            }
        }
        if (preamble == 0) { /* :134-157 */
            int sw = 0, lw = 0, cnt = 0;
            for (unsigned n = 0; n < np; ++n) {
                if (within(p->pulse[n], t->sh, tol) && within(p->pulse[n] + p->gap[n], t->lo, tol)) {
                    sw += p->pulse[n];
                    lw += p->pulse[n] + p->gap[n];
                    cnt++;
                }
            }
            if (cnt > 8) {
                f_lo = (float)cnt / lw;
                f_sh = (float)cnt / sw;
            }
        }
    }
    else { /* NRZ :159-214 */
        for (unsigned n = 0; n < np; ++n) {
            int w = 0, cnt = 0;
            while (n < np && (int)(p->pulse[n] * f_sh + 0.5) == 1 && (int)(p->gap[n] * f_lo + 0.5) == 1) {
                w += p->pulse[n] + p->gap[n];
                cnt += 2;
                n++;
            }
            if (cnt >= need) {
                f_sh = f_lo = (float)cnt / w;
                need = cnt;
                preamble = cnt;
            }
        }
        if (preamble == 0) {
            int w = 0, cnt = 0;
            for (unsigned n = 0; n < np; ++n) {
                if (within(p->pulse[n], t->sh, tol)) {
                    w += p->pulse[n];
                    cnt += 1;
                }
                if (within(p->pulse[n], 2 * t->sh, tol)) {
                    w += p->pulse[n];
                    cnt += 2;
                }
                if (within(p->gap[n], t->lo, tol)) {
                    w += p->gap[n];
                    cnt += 1;
                }
                if (within(p->gap[n], 2 * t->lo, tol)) {
                    w += p->gap[n];
                    cnt += 2;
                }
            }
            if (cnt > 20)
                f_sh = f_lo = (float)cnt / w;
        }
    }

    for (unsigned n = 0; n < np; ++n) { /* :216-257 */
        int highs = (int)(p->pulse[n] * f_sh + 0.5f);
        int lows = (int)((p->gap[n] + t->sh - t->lo) * f_lo + 0.5f);
        for (int i = 0; i < highs; ++i)
            orc_bb_add_bit(&s->bits, 1);
        lows = IMIN(lows, max_zeros);
        for (int i = 0; i < lows; ++i)
            orc_bb_add_bit(&s->bits, 0);
        if (rz && abs(p->pulse[n] - t->sh) > tol)
            orc_bb_clear(&s->bits);
        else if (p->gap[n] > gap_limit && p->gap[n] <= t->rst)
            orc_bb_add_row(&s->bits);
        if ((n == np - 1 || p->gap[n] > t->rst) && (s->bits.bits[0] > 0 || s->bits.num_rows > 1))
            fire(s);
    }
}

/* src/pulse_slicer.c:261-337 */
static void slice_ppm(orc_pulses const *p, timing const *t, slice_ctx *s)
{
    int z_lo, z_hi, o_lo, o_hi, y_lo = 0, y_hi = 0;
    if (t->tol > 0) {
        z_lo = t->sh - t->tol;
        z_hi = t->sh + t->tol;
        o_lo = t->lo - t->tol;
        o_hi = t->lo + t->tol;
        if (t->syn > 0) {
            y_lo = t->syn - t->tol;
            y_hi = t->syn + t->tol;
        }
    }
    else {
        z_lo = 0;
        z_hi = (t->sh + t->lo) / 2 + 1;
        o_lo = z_hi - 1;
        o_hi = t->gap ? t->gap : t->rst;
    }
    for (unsigned n = 0; n < p->num; ++n) {
        int g = p->gap[n];
        if (g > z_lo && g < z_hi)
            orc_bb_add_bit(&s->bits, 0);
        else if (g > o_lo && g < o_hi)
            orc_bb_add_bit(&s->bits, 1);
        else if (g > y_lo && g < y_hi)
            orc_bb_add_sync(&s->bits);
        else if (g < t->rst)
            orc_bb_add_row(&s->bits);
        if ((n == p->num - 1 || g >= t->rst) && (s->bits.bits[0] > 0 || s->bits.num_rows > 1))
            fire(s);
    }
}

/* src/pulse_slicer.c:339-449 */
static void slice_pwm(orc_pulses const *p, timing const *t, slice_ctx *s)
{
    int const big = 2147483647;
    int o_lo, o_hi, z_lo, z_hi, y_lo = 0, y_hi = 0;
    if (t->tol > 0) {
        o_lo = t->sh - t->tol;
        o_hi = t->sh + t->tol;
        z_lo = t->lo - t->tol;
        z_hi = t->lo + t->tol;
        if (t->syn > 0) {
            y_lo = t->syn - t->tol;
            y_hi = t->syn + t->tol;
        }
    }
    else if (t->syn <= 0) {
        o_lo = 0;
        o_hi = (t->sh + t->lo) / 2 + 1;
        z_lo = o_hi - 1;
        z_hi = big;
    }
    else if (t->syn < t->sh) {
        y_lo = 0;
        y_hi = (t->syn + t->sh) / 2 + 1;
        o_lo = y_hi - 1;
        o_hi = (t->sh + t->lo) / 2 + 1;
        z_lo = o_hi - 1;
        z_hi = big;
    }
    else if (t->syn < t->lo) {
        o_lo = 0;
        o_hi = (t->sh + t->syn) / 2 + 1;
        y_lo = o_hi - 1;
        y_hi = (t->syn + t->lo) / 2 + 1;
        z_lo = y_hi - 1;
        z_hi = big;
    }
    else {
        o_lo = 0;
        o_hi = (t->sh + t->lo) / 2 + 1;
        z_lo = o_hi - 1;
        z_hi = (t->lo + t->syn) / 2 + 1;
        y_lo = z_hi - 1;
        y_hi = big;
    }
    for (unsigned n = 0; n < p->num; ++n) {
        int w = p->pulse[n];
        if (w > o_lo && w < o_hi)
            orc_bb_add_bit(&s->bits, 1);
        else if (w > z_lo && w < z_hi)
            orc_bb_add_bit(&s->bits, 0);
        else if (w > y_lo && w < y_hi)
            orc_bb_add_sync(&s->bits);
        else if (w <= o_lo)
            ; /* spurious short pulse ignored */
        else
            orc_bb_add_row(&s->bits);

        if ((n == p->num - 1 || p->gap[n] > t->rst) && s->bits.num_rows > 0)
            fire(s);
        else if (t->gap > 0 && p->gap[n] > t->gap && s->bits.num_rows > 0 && s->bits.bits[s->bits.num_rows - 1] > 0)
            orc_bb_add_row(&s->bits);
    }
}

/* src/pulse_slicer.c:451-527; "x > s_short * 1.5" in double == "2x > 3 s_short" in integers */
static void slice_mc(orc_pulses const *p, timing const *t, slice_ctx *s)
{
    int since = 0;
    orc_bb_add_bit(&s->bits, 0);
    for (unsigned n = 0; n < p->num; ++n) {
        int w = p->pulse[n], g = p->gap[n];
        if (t->tol > 0 && (w < t->sh - t->tol || w > t->sh * 2 + t->tol || g < t->sh - t->tol || g > t->sh * 2 + t->tol)) {
            if ((double)w > t->sh * 1.5 && w <= t->sh * 2 + t->tol)
                orc_bb_add_bit(&s->bits, 1);
            orc_bb_add_row(&s->bits);
            orc_bb_add_bit(&s->bits, 0);
            since = 0;
        }
        else if ((double)(w + since) > t->sh * 1.5) {
            orc_bb_add_bit(&s->bits, 1);
            since = 0;
        }
        else {
            since += w;
        }
        if ((n == p->num - 1 || g > t->rst) && s->bits.num_rows > 0) {
            fire(s);
            orc_bb_add_bit(&s->bits, 0);
            since = 0;
        }
        else if ((double)(g + since) > t->sh * 1.5) {
            orc_bb_add_bit(&s->bits, 0);
            since = 0;
        }
        else {
            since += g;
        }
    }
}

/* src/pulse_slicer.c:529-535 */
static int symbol_at(orc_pulses const *p, unsigned k)
{
    return (k & 1) ? p->gap[k / 2] : p->pulse[k / 2];
}

/* src/pulse_slicer.c:537-595 */
static void slice_dmc(orc_pulses const *p, timing const *t, slice_ctx *s)
{
    unsigned const ns = p->num * 2;
    for (unsigned k = 0; k < ns; ++k) {
        int sym = symbol_at(p, k);
        if (abs(sym - t->sh) < t->tol) {
            orc_bb_add_bit(&s->bits, 1);
            sym = k + 1 < ns ? symbol_at(p, ++k) : 0;
            if (abs(sym - t->sh) > t->tol) {
                if (sym >= t->rst - t->tol)
                    k--;
                else if (s->bits.num_rows > 0 && s->bits.bits[s->bits.num_rows - 1] > 0)
                    orc_bb_add_row(&s->bits);
            }
        }
        else if (abs(sym - t->lo) < t->tol) {
            orc_bb_add_bit(&s->bits, 0);
        }
        else if (sym >= t->rst - t->tol && s->bits.num_rows > 0) {
            fire(s);
        }
    }
}

/* src/pulse_slicer.c:597-657 */
static void slice_piwm_raw(orc_pulses const *p, r433_dev_timing const *d, timing const *t, slice_ctx *s)
{
    float f_sh = d->short_width > 0.0f ? 1.0f / (d->short_width * t->us) : 0;
    unsigned const ns = p->num * 2;
    for (unsigned k = 0; k < ns; ++k) {
        int sym = symbol_at(p, k);
        int w = (int)(sym * f_sh + 0.5);
        if (sym > t->lo) {
            orc_bb_add_row(&s->bits);
        }
        else if (abs(sym - w * t->sh) < t->tol) {
            for (; w > 0; --w)
                orc_bb_add_bit(&s->bits, 1 - (int)(k & 1));
        }
        else if (sym < t->rst && s->bits.num_rows > 0 && s->bits.bits[s->bits.num_rows - 1] > 0) {
            orc_bb_add_row(&s->bits);
        }
        if ((k == ns - 1 || sym > t->rst) && s->bits.num_rows > 0)
            fire(s);
    }
}

/* src/pulse_slicer.c:659-713 */
static void slice_piwm_dc(orc_pulses const *p, timing const *t, slice_ctx *s)
{
    unsigned const ns = p->num * 2;
    for (unsigned k = 0; k < ns; ++k) {
        int sym = symbol_at(p, k);
        if (abs(sym - t->sh) < t->tol)
            orc_bb_add_bit(&s->bits, 1);
        else if (abs(sym - t->lo) < t->tol)
            orc_bb_add_bit(&s->bits, 0);
        else if (sym < t->rst && s->bits.num_rows > 0 && s->bits.bits[s->bits.num_rows - 1] > 0)
            orc_bb_add_row(&s->bits);
        if ((k == ns - 1 || sym > t->rst) && s->bits.num_rows > 0)
            fire(s);
    }
}

/* src/pulse_slicer.c:715-759 */
static void slice_nrzs(orc_pulses const *p, timing const *t, slice_ctx *s)
{
    int const lim = t->sh;
    for (unsigned n = 0; n < p->num; ++n) {
        int w = p->pulse[n];
        if (w > lim) {
            if (lim <= 0)
                return; /* the reference would divide by zero here; no registered device has short_width == 0 */
            for (int i = 0; i < w / lim; ++i)
                orc_bb_add_bit(&s->bits, 1);
            orc_bb_add_bit(&s->bits, 0);
        }
        else if (w < lim) {
            orc_bb_add_bit(&s->bits, 0);
        }
        if (n == p->num - 1 || p->gap[n] >= t->rst)
            fire(s);
    }
}

static int pulse_or_zero(orc_pulses const *p, unsigned n, int want_gap)
{
    if (n >= R433_PD_MAX_PULSES)
        return 0;
    return want_gap ? p->gap[n] : p->pulse[n];
}

/* src/pulse_slicer.c:775-864 */
static void slice_osv1(orc_pulses const *p, timing const *t, slice_ctx *s)
{
    int const half_min = t->sh / 2;
    int const half_max = t->sh * 3 / 2;
    int const sync_min = 2 * half_max;
    unsigned n;
    int pre = 0, man = 0;
    for (n = 0; n < p->num; ++n) {
        if (p->pulse[n] > half_min && p->gap[n] > half_min) {
            pre++;
            if (p->gap[n] > half_max)
                break;
        }
        else
            return;
    }
    if (pre != 12)
        return;
    ++n;
    int sp = pulse_or_zero(p, n, 0), sg = pulse_or_zero(p, n, 1);
    if (sp < sync_min || sg < sync_min)
        return;
    if (sg > sp) {
        man ^= 1;
        if (man)
            orc_bb_add_bit(&s->bits, 0);
    }
    for (n++; n < p->num; ++n) {
        man ^= 1;
        if (man)
            orc_bb_add_bit(&s->bits, 1);
        if (p->pulse[n] > half_max) {
            man ^= 1;
            if (man)
                orc_bb_add_bit(&s->bits, 1);
        }
        if ((n == p->num - 1 || p->gap[n] > t->rst) && s->bits.num_rows > 0) {
            fire(s);
            return;
        }
        man ^= 1;
        if (man)
            orc_bb_add_bit(&s->bits, 0);
        if (p->gap[n] > half_max) {
            man ^= 1;
            if (man)
                orc_bb_add_bit(&s->bits, 0);
        }
    }
}

/* src/pulse_slicer.c:866-918 */
static void slice_rzi(orc_pulses const *p, timing const *t, slice_ctx *s)
{
    int const base = t->lo - t->sh;
    int fresh = 1;
    if (t->lo <= 0)
        return; /* the reference would divide by zero */
    for (unsigned n = 0; n < p->num; ++n) {
        int w = p->pulse[n];
        int ones = fresh ? (w + t->lo / 2) / t->lo : (w - base + t->lo / 2) / t->lo;
        fresh = 0;
        for (int k = 0; k < ones; ++k)
            orc_bb_add_bit(&s->bits, 1);
        if (p->gap[n] > t->rst || n == p->num - 1) {
            if (s->bits.bits[0] > 0)
                fire(s);
            orc_bb_clear(&s->bits);
            fresh = 1;
            continue;
        }
        orc_bb_add_bit(&s->bits, 0);
    }
}

/* one arm of the switch in src/r_api.c:456-497 / :520-547 */
int orc_slice(orc_pulses const *p, r433_dev_timing const *d, unsigned dev, int is_fsk_package, orc_event_fn fn,
        void *ctx, unsigned *n_calls)
{
    slice_ctx *s = calloc(1, sizeof(*s));
    s->fn = fn;
    s->ctx = ctx;
    s->dev = dev;
    unsigned m = d->modulation;
    int fsk_mod = m >= 16;
    timing t;
    if (fsk_mod == (is_fsk_package != 0) && timing_make(&t, d, p->sample_rate, m != 13)) {
        switch (m) {
        case 4:
        case 16:
            slice_pcm(p, d, &t, s);
            break;
        case 5:
            slice_ppm(p, &t, s);
            break;
        case 6:
        case 17:
            slice_pwm(p, &t, s);
            break;
        case 3:
        case 18:
            slice_mc(p, &t, s);
            break;
        case 8:
            slice_piwm_raw(p, d, &t, s);
            break;
        case 11:
            slice_piwm_dc(p, &t, s);
            break;
        case 9:
            slice_dmc(p, &t, s);
            break;
        case 10:
            slice_osv1(p, &t, s);
            break;
        case 12:
            slice_nrzs(p, &t, s);
            break;
        case 13:
            slice_rzi(p, &t, s);
            break;
        default:
            break;
        }
    }
    if (n_calls)
        *n_calls = s->calls;
    int ev = s->events;
    free(s);
    return ev;
}

/* ------------------------------------------------------------------ record serialisation */

static void blob_reserve(orc_blob *b, size_t extra)
{
    if (b->len + extra <= b->cap)
        return;
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->len + extra)
        nc *= 2;
    b->data = realloc(b->data, nc);
    b->cap = nc;
}

void orc_blob_free(orc_blob *b)
{
    free(b->data);
    memset(b, 0, sizeof(*b));
}

static void put_package(orc_blob *b, orc_pulses const *p, uint32_t stream, int type, uint32_t frame, uint32_t ret_pos)
{
    size_t sz = sizeof(r433_pkg_rec) + 8u * p->num;
    blob_reserve(b, sz);
    r433_pkg_rec h;
    memset(&h, 0, sizeof(h));
    h.total_bytes = (uint32_t)sz;
    h.stream = stream;
    h.type = (uint32_t)type;
    h.num_pulses = p->num;
    h.frame = frame;
    h.ret_pos = ret_pos;
    h.offset = p->offset;
    h.start_ago = p->start_ago;
    h.end_ago = p->end_ago;
    h.ook_low = p->ook_low;
    h.ook_high = p->ook_high;
    h.fsk_f1 = p->fsk_f1;
    h.fsk_f2 = p->fsk_f2;
    h.sample_rate = p->sample_rate;
    memcpy(b->data + b->len, &h, sizeof(h));
    int32_t *pairs = (int32_t *)(b->data + b->len + sizeof(h));
    for (uint32_t i = 0; i < p->num; ++i) {
        pairs[2 * i] = p->pulse[i];
        pairs[2 * i + 1] = p->gap[i];
    }
    b->len += sz;
    b->count += 1;
}

typedef struct evt_sink {
    orc_blob *blob;
    uint32_t pkg;
} evt_sink;

static int record_event(void *ctx, unsigned dev, unsigned ordinal, orc_bitbuf const *bits)
{
    evt_sink *k = ctx;
    orc_blob *b = k->blob;
    size_t sz = sizeof(r433_evt_rec);
    for (unsigned r = 0; r < bits->num_rows; ++r)
        sz += sizeof(r433_row_rec) + ((((size_t)bits->extent[r] + 7) / 8 + 3) & ~(size_t)3);
    blob_reserve(b, sz);
    uint8_t *w = b->data + b->len;
    memset(w, 0, sz);
    r433_evt_rec h = {(uint32_t)sz, k->pkg, (uint16_t)dev, (uint16_t)ordinal, bits->num_rows, bits->free_row};
    memcpy(w, &h, sizeof(h));
    w += sizeof(h);
    for (unsigned r = 0; r < bits->num_rows; ++r) {
        unsigned nb = ((unsigned)bits->extent[r] + 7) / 8;
        r433_row_rec rr = {bits->bits[r], bits->syncs[r], (uint16_t)nb, 0};
        memcpy(w, &rr, sizeof(rr));
        w += sizeof(rr);
        memcpy(w, bits->bb + r * R433_BB_COLS, nb);
        w += (nb + 3) & ~3u;
    }
    b->len += sz;
    b->count += 1;
    return 0;
}

static void record_to_pulses(uint8_t const *rec, orc_pulses *p)
{
    r433_pkg_rec h;
    memcpy(&h, rec, sizeof(h));
    memset(p, 0, sizeof(*p));
    p->offset = h.offset;
    p->sample_rate = h.sample_rate;
    p->start_ago = h.start_ago;
    p->end_ago = h.end_ago;
    p->num = h.num_pulses;
    p->ook_low = h.ook_low;
    p->ook_high = h.ook_high;
    p->fsk_f1 = h.fsk_f1;
    p->fsk_f2 = h.fsk_f2;
    for (uint32_t i = 0; i < h.num_pulses && i < R433_PD_MAX_PULSES; ++i) {
        memcpy(&p->pulse[i], rec + sizeof(h) + 8u * i, 4);
        memcpy(&p->gap[i], rec + sizeof(h) + 8u * i + 4, 4);
    }
}

int orc_slice_packages(uint8_t const *pkg_blob, size_t pkg_len, r433_dev_timing const *devs, unsigned n_devs,
        uint32_t pkg_base, orc_blob *events)
{
    orc_pulses *p = malloc(sizeof(*p));
    size_t at = 0;
    uint32_t idx = 0;
    while (at + sizeof(r433_pkg_rec) <= pkg_len) {
        r433_pkg_rec h;
        memcpy(&h, pkg_blob + at, sizeof(h));
        if (h.total_bytes < sizeof(h) || at + h.total_bytes > pkg_len)
            break;
        record_to_pulses(pkg_blob + at, p);
        evt_sink k = {events, pkg_base + idx};
        for (unsigned d = 0; d < n_devs; ++d)
            orc_slice(p, &devs[d], d, h.type == R433_PKG_FSK, record_event, &k, NULL);
        at += h.total_bytes;
        idx += 1;
    }
    free(p);
    return (int)idx;
}

/* ------------------------------------------------------------------ flow */

/* src/r_flow.c:104-340 restricted to file input: every frame is processed (:174), the FM buffer
 * aliases the raw envelope when FM demodulation is off (include/r_private.h:32-36). */
int orc_flow_run(orc_flow_cfg const *cfg, void const *iq, size_t n_bytes, r433_dev_timing const *devs,
        unsigned n_devs, uint32_t stream_index, uint32_t pkg_base, orc_flow_out *out)
{
    uint32_t const ss = cfg->sample_size;
    if (ss != 2 && ss != 4)
        return -1;
    size_t const total = n_bytes / ss;
    uint32_t const fs = cfg->frame_samples;
    uint16_t *env = malloc(sizeof(uint16_t) * (fs + 1));
    int16_t *am = malloc(sizeof(int16_t) * (fs + 1));
    int16_t *fmb = malloc(sizeof(int16_t) * (fs + 1));
    orc_pulses *ook = calloc(1, sizeof(*ook));
    orc_pulses *fsk = calloc(1, sizeof(*fsk));
    orc_lpf_state lpf = {0, 0};
    orc_fm_state fms;
    memset(&fms, 0, sizeof(fms));
    orc_detector det;
    memset(&det, 0, sizeof(det));
    orc_levels_from_db(&det.lv, (int)cfg->use_mag_est, cfg->level_limit_db, cfg->min_level_db, cfg->min_snr_db);
    orc_detector_reset(&det);
    float noise_level = 0.0f, min_level_auto = 0.0f;
    uint64_t input_pos = 0;
    size_t first_pkg = out->packages.len;

    uint32_t frame = 0;
    for (size_t done = 0;; ++frame) {
        /* the file loop (src/rtl_433.c:1826-1845) pushes full frames, a short tail, then a flush */
        uint32_t n = (uint32_t)((total - done) < fs ? (total - done) : fs);
        int flush = n == 0;
        if (!flush) {
            uint8_t const *src = (uint8_t const *)iq + done * ss;
            uint32_t sum;
            if (ss == 2)
                sum = cfg->use_mag_est ? orc_magest_cu8(src, env, n) : orc_envelope_cu8(src, env, n);
            else
                sum = orc_magest_cs16((int16_t const *)src, env, n);
            if (out->frame_sums)
                out->frame_sums[frame] = sum;
            float avg_db = orc_level_db(sum, n, ss == 4 || cfg->use_mag_est);
            /* squelch / auto level bookkeeping, src/r_flow.c:166-189 */
            if (min_level_auto == 0.0f)
                min_level_auto = cfg->min_level_db;
            if (noise_level == 0.0f)
                noise_level = min_level_auto - 3.0f;
            int noise_only = avg_db < noise_level + 3.0f;
            if (noise_only) {
                noise_level = (noise_level * 7 + avg_db) / 8;
                if (cfg->auto_level > 0 && noise_level < cfg->min_level_db - 3.0f
                        && fabsf(min_level_auto - noise_level - 3.0f) > 1.0f) {
                    min_level_auto = noise_level + 3.0f;
                    orc_levels_from_db(&det.lv, (int)cfg->use_mag_est, cfg->level_limit_db, min_level_auto, cfg->min_snr_db);
                }
            }
            else {
                noise_level = (noise_level * 31 + avg_db) / 32;
            }
            orc_lowpass(&lpf, env, am, n);
            if (cfg->enable_fm) {
                float lp = cfg->fm_low_pass != 0.0f ? cfg->fm_low_pass : cfg->fpdm ? 0.2f : 0.1f;
                if (ss == 2)
                    orc_fm_cu8(&fms, src, fmb, n, cfg->samp_rate, lp);
                else
                    orc_fm_cs16(&fms, (int16_t const *)src, fmb, n, cfg->samp_rate, lp);
            }
            else {
                memcpy(fmb, env, sizeof(uint16_t) * n); /* union aliasing */
            }
            /* "Handle special input formats" (src/r_flow.c:212-225): the file's words over the demodulated buffer */
            if (cfg->load_format == 1 && ss == 2)
                memcpy(am, src, sizeof(int16_t) * n);
            else if (cfg->load_format == 2 && ss == 2)
                memcpy(fmb, src, sizeof(int16_t) * n);
            if (out->env)
                memcpy(out->env + done, env, sizeof(uint16_t) * n);
            if (out->am)
                memcpy(out->am + done, am, sizeof(int16_t) * n);
            if (out->fm)
                memcpy(out->fm + done, fmb, sizeof(int16_t) * n);
        }
        for (;;) {
            int pos_before = det.pos;
            (void)pos_before;
            int type = orc_detect_package(&det, am, fmb, (int)n, cfg->samp_rate, input_pos, ook, fsk, cfg->fpdm);
            if (!type)
                break;
            uint32_t ret_pos = flush ? R433_RET_FLUSH : (uint32_t)det.pos;
            put_package(&out->packages, type == R433_PKG_OOK ? ook : fsk, stream_index, type, frame, ret_pos);
        }
        if (flush)
            break;
        input_pos += n;
        done += n;
    }

    int n_pkgs = 0;
    if (n_devs > 0)
        n_pkgs = orc_slice_packages(out->packages.data + first_pkg, out->packages.len - first_pkg, devs, n_devs,
                pkg_base, &out->events);
    else {
        size_t at = first_pkg;
        while (at < out->packages.len) {
            r433_pkg_rec h;
            memcpy(&h, out->packages.data + at, sizeof(h));
            at += h.total_bytes;
            n_pkgs++;
        }
    }
    free(env);
    free(am);
    free(fmb);
    free(ook);
    free(fsk);
    return n_pkgs;
}

/* ------------------------------------------------------------------ canonical forms */

size_t orc_events_normalize(uint8_t *blob, size_t len)
{
    size_t rd = 0, wr = 0;
    while (rd + sizeof(r433_evt_rec) <= len) {
        r433_evt_rec h;
        memcpy(&h, blob + rd, sizeof(h));
        if (h.total_bytes < sizeof(h) || rd + h.total_bytes > len)
            break;
        uint8_t const *src = blob + rd + sizeof(h);
        uint8_t *dst0 = blob + wr;
        uint8_t *dst = dst0 + sizeof(h);
        for (unsigned r = 0; r < h.num_rows; ++r) {
            r433_row_rec rr;
            memcpy(&rr, src, sizeof(rr));
            uint8_t const *bytes = src + sizeof(rr);
            unsigned padded = (rr.nbytes + 3u) & ~3u;
            unsigned nb = rr.nbytes;
            while (nb > 0 && bytes[nb - 1] == 0)
                nb--;
            unsigned npad = (nb + 3u) & ~3u;
            r433_row_rec nr = {rr.bits, rr.syncs, (uint16_t)nb, 0};
            uint8_t tmp[8];
            memcpy(tmp, &nr, sizeof(nr));
            memmove(dst + sizeof(nr), bytes, nb);
            memcpy(dst, tmp, sizeof(nr));
            memset(dst + sizeof(nr) + nb, 0, npad - nb);
            dst += sizeof(nr) + npad;
            src += sizeof(rr) + padded;
        }
        size_t old_total = h.total_bytes;
        h.total_bytes = (uint32_t)(dst - dst0);
        memcpy(dst0, &h, sizeof(h));
        wr += h.total_bytes;
        rd += old_total;
    }
    return wr;
}

uint64_t orc_events_digest(uint8_t const *blob, size_t len, uint32_t *n_events)
{
    uint64_t total = 0; /* sum of per-event FNV-1a hashes: independent of event order */
    uint32_t cnt = 0;
    size_t at = 0;
    static uint8_t img[4 + 200 + R433_BB_ROWS * R433_BB_COLS];
    while (at + sizeof(r433_evt_rec) <= len) {
        r433_evt_rec h;
        memcpy(&h, blob + at, sizeof(h));
        if (h.total_bytes < sizeof(h) || at + h.total_bytes > len)
            break;
        memset(img, 0, sizeof(img));
        memcpy(img, &h.num_rows, 2);
        memcpy(img + 2, &h.free_row, 2);
        uint8_t const *src = blob + at + sizeof(h);
        for (unsigned r = 0; r < h.num_rows && r < R433_BB_ROWS; ++r) {
            r433_row_rec rr;
            memcpy(&rr, src, sizeof(rr));
            memcpy(img + 4 + 2 * r, &rr.bits, 2);
            memcpy(img + 4 + 100 + 2 * r, &rr.syncs, 2);
            size_t room = (size_t)(R433_BB_ROWS - r) * R433_BB_COLS;
            memcpy(img + 204 + r * R433_BB_COLS, src + sizeof(rr), rr.nbytes < room ? rr.nbytes : room);
            src += sizeof(rr) + ((rr.nbytes + 3u) & ~3u);
        }
        uint8_t key[8];
        memcpy(key, &h.pkg, 4);
        memcpy(key + 4, &h.dev, 2);
        memcpy(key + 6, &h.ordinal, 2);
        uint64_t hsh = 1469598103934665603ull;
        for (unsigned i = 0; i < 8; ++i)
            hsh = (hsh ^ key[i]) * 1099511628211ull;
        for (size_t i = 0; i < sizeof(img); ++i)
            hsh = (hsh ^ img[i]) * 1099511628211ull;
        total += hsh;
        cnt++;
        at += h.total_bytes;
    }
    if (n_events)
        *n_events = cnt;
    return total;
}

/* Compact per-event checksum used by bench.py: FNV-1a 64 over {pkg, dev, ordinal, num_rows, free_row}
 * and, per row, {bits, syncs, the ceil(bits/8) payload bytes}; summed over events mod 2^64.  The same
 * function exists over a real bitbuffer_t in ref_harness.c and in the product's digest plugin. */
uint64_t orc_events_digest2(uint8_t const *blob, size_t len, uint32_t *n_events)
{
    uint64_t total = 0;
    uint32_t cnt = 0;
    size_t at = 0;
    while (at + sizeof(r433_evt_rec) <= len) {
        r433_evt_rec h;
        memcpy(&h, blob + at, sizeof(h));
        if (h.total_bytes < sizeof(h) || at + h.total_bytes > len)
            break;
        uint64_t x = 1469598103934665603ull;
        uint8_t const *k = blob + at + 4; /* pkg, dev, ordinal, num_rows, free_row = 12 bytes */
        for (unsigned i = 0; i < 12; ++i)
            x = (x ^ k[i]) * 1099511628211ull;
        uint8_t const *src = blob + at + sizeof(h);
        for (unsigned r = 0; r < h.num_rows; ++r) {
            r433_row_rec rr;
            memcpy(&rr, src, sizeof(rr));
            for (unsigned i = 0; i < 4; ++i)
                x = (x ^ src[i]) * 1099511628211ull;
            unsigned nb = ((unsigned)rr.bits + 7) / 8;
            for (unsigned i = 0; i < nb; ++i)
                x = (x ^ (i < rr.nbytes ? src[sizeof(rr) + i] : 0)) * 1099511628211ull;
            src += sizeof(rr) + ((rr.nbytes + 3u) & ~3u);
        }
        total += x;
        cnt++;
        at += h.total_bytes;
    }
    if (n_events)
        *n_events = cnt;
    return total;
}
