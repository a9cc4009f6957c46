/* TEST INFRASTRUCTURE: decode_fn plugins with the reference signature (include/r_device.h:59-92) whose first tests look
 * like the ones in the reference's src/devices/ -- for tests/test_prefilter.py.  Compiled by the test with gcc. */
#include <stdint.h>

typedef struct { uint16_t num_rows, free_row, bits_per_row[50], syncs_before_row[50]; uint8_t bb[50][128]; } bitbuffer_t;
struct r_device;

unsigned long pf_calls[8]; /* calls that reached each decoder */
unsigned pf_moody_n;        /* the moody decoder's memory */

static int payload_verdict(bitbuffer_t *b, int row)
{
    unsigned s = 0;
    for (int i = 0; i < (b->bits_per_row[row] + 7) / 8 && i < 128; ++i)
        s += b->bb[row][i];
    return (s % 5 == 0) ? 1 : (s % 5 == 1) ? -3 : (s % 5 == 2) ? -4 : 0; /* ok / MIC / sanity / legacy 0 */
}

/* like nice_flor_s.c:84: one exact length on row 0 */
int pf_dec_exact(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pf_calls[0]++;
    if (b->bits_per_row[0] != 24 && b->bits_per_row[0] != 25)
        return -1;
    return payload_verdict(b, 0);
}

/* like tpms_*.c: one row only, then a minimum length, then the data */
int pf_dec_onerow(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pf_calls[1]++;
    if (b->num_rows != 1)
        return -2;
    if (b->bits_per_row[0] < 16)
        return -1;
    return payload_verdict(b, 0);
}

/* walks the rows: looks at lengths the filter cannot vouch for */
int pf_dec_rows(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pf_calls[2]++;
    for (int r = 0; r < b->num_rows; ++r)
        if (b->bits_per_row[r] >= 20)
            return payload_verdict(b, r);
    return -1;
}

/* the data first (so nothing is learned from the head alone); a short row is refused whatever it holds -- unless sync pulses
 * came before it: what the exhaustive probe of tiny rows may and may not conclude */
int pf_dec_data(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pf_calls[3]++;
    if (b->syncs_before_row[0])
        return -3;
    if (b->bb[0][0] == 0xff)
        return -2;
    return b->bits_per_row[0] < 12 ? -1 : payload_verdict(b, 0);
}

/* keeps state that its length test looks at: the probe must refuse to learn anything from it */
int pf_dec_moody(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pf_calls[4]++;
    if (b->bits_per_row[0] < (++pf_moody_n % 3 == 0 ? 8 : 30))
        return -1;
    return payload_verdict(b, 0);
}

/* an empty bitbuffer and the "legacy" 0 code */
int pf_dec_zero(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pf_calls[5]++;
    if (b->num_rows == 0 || b->bits_per_row[0] == 0)
        return 0;
    if (b->num_rows > 3)
        return -4;
    return payload_verdict(b, 0);
}

/* ---- decoders made of a random list of first-line tests (tests/test_prefilter.py::test_prefilter_random_decoders) ----
 * decode_ctx (reference include/r_device.h: the last-but-one pointer of an r_device; offset 136 on LP64, include/r433_abi.h)
 * points at a program: steps of the kinds the reference's decoders open with -- tests of the head, of other rows' lengths, of
 * the sync count, of the content (after an inversion in place, through a search, by a sum) -- each returning its code. */
typedef struct pf_step { int op, a, b, code; } pf_step;
typedef struct pf_prog { int n; pf_step s[10]; unsigned long calls; } pf_prog;

int pf_dec_random(struct r_device *d, bitbuffer_t *b)
{
    pf_prog *p = *(pf_prog **)((char *)d + 136);
    p->calls++;
    for (int k = 0; k < p->n; ++k) {
        pf_step const *s = &p->s[k];
        unsigned const bits0 = b->bits_per_row[0], nbytes0 = (bits0 + 7) / 8 < 128 ? (bits0 + 7) / 8 : 128;
        switch (s->op) {
        case 0: if (b->num_rows != s->a) return s->code; break;
        case 1: if (b->num_rows < s->a || b->num_rows > s->b) return s->code; break;
        case 2: if ((int)bits0 < s->a) return s->code; break;
        case 3: if ((int)bits0 > s->b) return s->code; break;
        case 4: if ((int)bits0 != s->a && (int)bits0 != s->b) return s->code; break;
        case 5: /* bitbuffer_invert on row 0, then a look at its first byte */
            for (unsigned i = 0; i < nbytes0; ++i)
                b->bb[0][i] = (uint8_t)~b->bb[0][i];
            if (bits0 % 8)
                b->bb[0][nbytes0 - 1] &= (uint8_t)(0xff00 >> (bits0 % 8));
            if (nbytes0 && b->bb[0][0] == (uint8_t)s->a) return s->code;
            break;
        case 6: if (b->syncs_before_row[0] != 0) return s->code; break;
        case 7: { /* a search for a byte in row 0 */
            unsigned i = 0;
            while (i < nbytes0 && b->bb[0][i] != (uint8_t)s->a)
                ++i;
            if (i == nbytes0) return s->code;
            break;
        }
        case 8: { /* the first row long enough, or none */
            int r = 0;
            while (r < b->num_rows && b->bits_per_row[r] < s->a)
                ++r;
            if (r == b->num_rows) return s->code;
            break;
        }
        case 9: { /* a checksum over row 0 */
            unsigned sum = 0;
            for (unsigned i = 0; i < nbytes0; ++i)
                sum += b->bb[0][i];
            if (sum % (unsigned)s->a == (unsigned)s->b) return s->code;
            break;
        }
        case 10: if (b->num_rows > 1 && b->bits_per_row[1] != bits0) return s->code; break; /* another row's length */
        default: break;
        }
    }
    return payload_verdict(b, 0);
}

/* A decoder with a context (a create_fn's, src/decoder_util.c:19-45; the test hands its address in pf_ctx_offset, the offset of
 * r_device.decode_ctx): its state is a counter in that context.  Multi-row bitbuffers and long rows it refuses without a look at
 * its state; SHORT rows it counts in its state before it refuses them.  Asked with the context out of reach
 * (R433_KEEPS_CONTEXT) it gets verdicts for the first two and none for the short rows -- and no question moves the counter. */
unsigned pf_ctx_offset;
int pf_dec_context(struct r_device *d, bitbuffer_t *b)
{
    pf_calls[6]++;
    if (b->num_rows != 1)
        return -2;
    if (b->bits_per_row[0] > 150)
        return -1;
    unsigned *state = *(unsigned **)((char *)d + pf_ctx_offset);
    if (b->bits_per_row[0] < 64) {
        state[0] += 1; /* (the look at its state comes first) */
        return -1;
    }
    state[1] += 1;
    return payload_verdict(b, 0);
}
