/* TEST INFRASTRUCTURE: decode_fn plugins that open the way most of the reference's decoders do -- with bitbuffer_invert,
 * bitbuffer_search or bitbuffer_find_repeated_row BEFORE their length test (src/devices/neptune_r900.c:88-101,
 * tpms_imars_t240.c:51-66, tfa_30_3221.c:46-52) -- for tests/test_prefilter.py::test_prefilter_helper_probe_plugins.  The
 * four helpers are in pf_helper_bitbuffer.c; the test links dropin/helper_wrap.c between the two with ld --wrap, as
 * dropin/Makefile does for the real decoders.  Compiled by the test with gcc. */
#include <stdint.h>
#include <string.h>

typedef struct { uint16_t num_rows, free_row, bits_per_row[50], syncs_before_row[50]; uint8_t bb[50][128]; } bitbuffer_t;
struct r_device;

unsigned long pfh_calls[12];

void bitbuffer_invert(bitbuffer_t *bits);
unsigned bitbuffer_search(bitbuffer_t *bits, unsigned row, unsigned start, const uint8_t *pattern, unsigned plen);
int bitbuffer_find_repeated_row(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits);
int bitbuffer_find_repeated_prefix(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits);

static int payload_verdict(bitbuffer_t *b, int row)
{
    unsigned s = 0;
    for (int i = 0; i < (b->bits_per_row[row] + 7) / 8 && i < 128; ++i)
        s += b->bb[row][i];
    return (s % 5 == 0) ? 1 : (s % 5 == 1) ? -3 : (s % 5 == 2) ? -4 : 0;
}

static uint8_t const kPreamble[3] = {0xaa, 0xa9, 0x66};

/* 0. like neptune_r900.c: one row, a search, then "too short" wherever the preamble sits: a matter of the length alone */
int pfh_dec_search_then_length(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[0]++;
    if (b->num_rows != 1)
        return -1;
    unsigned const at = bitbuffer_search(b, 0, 0, kPreamble, 20);
    if (at + 20 + 400 > b->bits_per_row[0])
        return -1;
    if (at == b->bits_per_row[0])
        return -2;
    return payload_verdict(b, 0);
}

/* 1. like tpms_imars_t240.c: "not found" and "found but too short" are different codes.  By the length alone only rows too
 * short to hold the preamble go; for the others the search itself is the test (the pre-filter's search rule: rows of fewer
 * than 64 bits are searched on the device) */
static uint8_t const kShortPreamble[1] = {0xcc}; /* 110011.. : in some rows of the test's captures, not in others */
int pfh_dec_search_two_codes(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[1]++;
    if (b->num_rows != 1)
        return -2;
    int const len = b->bits_per_row[0];
    int const at = (int)bitbuffer_search(b, 0, 1, kShortPreamble, 6);
    if (at >= len)
        return -2;
    if (len - at < 20)
        return -1;
    return payload_verdict(b, 0);
}

/* 2. like many PWM decoders: invert first, then an exact length */
int pfh_dec_invert_then_length(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[2]++;
    bitbuffer_invert(b);
    if (b->bits_per_row[0] != 36 && b->bits_per_row[0] != 37)
        return -1;
    return payload_verdict(b, 0);
}

/* 3. invert, then the payload decides for every length: nothing to learn */
int pfh_dec_invert_then_payload(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[3]++;
    bitbuffer_invert(b);
    if (b->bb[0][0] == 0x00)
        return -2;
    return b->bits_per_row[0] < 10 ? -1 : payload_verdict(b, 0);
}

/* 4. a search on a bitbuffer of its own decides first: the wrappers must leave that one to the real helper */
int pfh_dec_foreign_search(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[4]++;
    static bitbuffer_t own; /* (zeroed; one row of 64 zero bits: the preamble is not in it) */
    own.num_rows = own.free_row = 1;
    own.bits_per_row[0] = 64;
    unsigned const at = bitbuffer_search(&own, 0, 0, kPreamble, 20);
    if (at != 64)
        return 1; /* (never: a wrapper that answered for `own` would bring the probe here) */
    if (b->bits_per_row[0] < 30)
        return -1;
    return payload_verdict(b, 0);
}

/* 5. two searches: the second one's answer is not enumerated -- nothing may be concluded where it is reached */
int pfh_dec_two_searches(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[5]++;
    if (b->bits_per_row[0] < 8)
        return -1; /* (the head alone) */
    unsigned const a = bitbuffer_search(b, 0, 0, kPreamble, 8);
    unsigned const c = bitbuffer_search(b, 0, a, kPreamble + 1, 8);
    if (c + 24 > b->bits_per_row[0])
        return -1;
    return payload_verdict(b, 0);
}

/* 6. like tfa_30_3221.c: a repeated row of at least 40 bits, among at least two -- one row never is */
int pfh_dec_repeated_row(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[6]++;
    int const row = bitbuffer_find_repeated_row(b, b->num_rows > 4 ? 4 : 2, 40);
    if (row < 0)
        return -2;
    if (b->bits_per_row[row] > 41)
        return -1;
    return payload_verdict(b, row);
}

/* 7. one repeat is enough, so a one-row bitbuffer passes the helper (which compares the row with itself: a look at the
 * payload the fence stops); the length test behind it is a matter of the head */
int pfh_dec_repeated_once(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[7]++;
    int const row = bitbuffer_find_repeated_prefix(b, 1, 24);
    if (row < 0)
        return -1;
    if (b->bits_per_row[row] > 40)
        return -1;
    return payload_verdict(b, row);
}

/* 8. the code depends on WHERE the preamble was found, for some lengths only */
int pfh_dec_search_position(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[8]++;
    if (b->num_rows != 1)
        return -1;
    unsigned const len = b->bits_per_row[0];
    unsigned const at = bitbuffer_search(b, 0, 2, kPreamble, 10);
    if (len < 20)
        return -1; /* whatever the search said */
    if (at == len)
        return -2;
    if (at > 30)
        return -4;
    return at + 50 > len ? -1 : payload_verdict(b, 0);
}

/* 9. a repeated-row test, and where it fails the decoder walks the row lengths itself: the wrapper's "no row qualifies" rests on
 * a supposition about lengths the decoder then LOOKS at -- behind the fence -- so nothing may be concluded for several rows */
int pfh_dec_repeated_then_lengths(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[9]++;
    int const row = bitbuffer_find_repeated_row(b, 2, 30);
    if (row < 0) {
        for (int r = 0; r < b->num_rows; ++r)
            if (b->bits_per_row[r] > 2)
                return -1;
        return -2;
    }
    return payload_verdict(b, row);
}

/* 10. like tpms_eezrv.c:82-91: the row is inverted IN PLACE first, then searched; "not found" and "found, too short" are
 * different codes.  The search the decoder runs is over the inverted row: a search rule formed from its pattern and run
 * over the row as the slicer left it would look for the wrong bits -- no rule may be formed here (only the head-alone
 * verdicts, which hold whatever the row contains). */
static uint8_t const kEdgePreamble[1] = {0x0c}; /* 000011: not in the test's short rows as sliced (they hold 111100), in most of them once inverted */
int pfh_dec_invert_then_search(struct r_device *d, bitbuffer_t *b)
{
    (void)d;
    pfh_calls[10]++;
    if (b->num_rows != 1)
        return -2;
    bitbuffer_invert(b);
    int const len = b->bits_per_row[0];
    int const at = (int)bitbuffer_search(b, 0, 1, kEdgePreamble, 6);
    if (at >= len)
        return -2;
    if (len - at < 20)
        return -1;
    return payload_verdict(b, 0);
}
