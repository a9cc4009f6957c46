/* TEST INFRASTRUCTURE: the four bitbuffer helpers the decoders of tests/plugins/pf_helper_decoders.c call -- this test
 * host's own versions (what they compute is fixed by their callers' expectations, reference src/bitbuffer.c:135-149, :228-253,
 * :513-533).  A file of their own: ld --wrap redirects UNDEFINED references only, so the callers must live in another object,
 * as the reference's decoders (src/devices) and its src/bitbuffer.c do. */
#include <stdint.h>
#include <string.h>

typedef struct { uint16_t num_rows, free_row, bits_per_row[50], syncs_before_row[50]; uint8_t bb[50][128]; } bitbuffer_t;

static int bit(uint8_t const *b, unsigned i) { return (b[i >> 3] >> (7 - (i & 7))) & 1; }

void bitbuffer_invert(bitbuffer_t *bits)
{
    for (unsigned r = 0; r < bits->num_rows; ++r) {
        unsigned const n = bits->bits_per_row[r];
        if (!n)
            continue;
        unsigned const last = (n - 1) / 8;
        for (unsigned c = 0; c <= last; ++c)
            bits->bb[r][c] = (uint8_t)~bits->bb[r][c];
        bits->bb[r][last] ^= (uint8_t)(0xff >> (((n - 1) % 8) + 1));
    }
}

unsigned bitbuffer_search(bitbuffer_t *bits, unsigned row, unsigned start, const uint8_t *pattern, unsigned plen)
{
    unsigned const len = bits->bits_per_row[row];
    for (unsigned at = start; plen && at < len; ++at) { /* (looks at the row even where the pattern cannot fit any more, as the reference's does) */
        unsigned k = 0;
        while (k < plen && at + k < len && bit(bits->bb[row], at + k) == bit(pattern, k))
            ++k;
        if (k == plen)
            return at;
    }
    return len;
}

static int rows_equal(bitbuffer_t *bits, unsigned a, unsigned b, unsigned max_bits)
{
    if (max_bits == 0 || bits->bits_per_row[a] < max_bits || bits->bits_per_row[b] < max_bits)
        return bits->bits_per_row[a] == bits->bits_per_row[b] && !memcmp(bits->bb[a], bits->bb[b], (bits->bits_per_row[a] + 7) / 8);
    for (unsigned i = 0; i < max_bits; ++i)
        if (bit(bits->bb[a], i) != bit(bits->bb[b], i))
            return 0;
    return 1;
}

static int find_repeated(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits, unsigned max_bits)
{
    for (int i = 0; i < bits->num_rows; ++i) {
        unsigned cnt = 0;
        if (bits->bits_per_row[i] < min_bits)
            continue;
        for (int j = 0; j < bits->num_rows; ++j)
            cnt += (unsigned)rows_equal(bits, (unsigned)i, (unsigned)j, max_bits);
        if (cnt >= min_repeats)
            return i;
    }
    return -1;
}

int bitbuffer_find_repeated_row(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits) { return find_repeated(bits, min_repeats, min_bits, 0); }
int bitbuffer_find_repeated_prefix(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits) { return find_repeated(bits, min_repeats, min_bits, min_bits); }

