/* helper_wrap.c -- the host's half of the pre-filter's second kind of question (include/r433_hip.h, r433_helper_probe).
 *
 * Linked into every program or library that holds the reference's decoders next to the GPU path (rtl_433_hip,
 * libr433plugins.so) with
 *     -Wl,--wrap=bitbuffer_invert,--wrap=bitbuffer_search,--wrap=bitbuffer_find_repeated_row,--wrap=bitbuffer_find_repeated_prefix
 * so that what the decoders (the sources under src/devices) call under those names lands here; the reference's own definitions
 * (src/bitbuffer.c:135-149, :228-253, :513-533) stay as they are and are what runs whenever nobody is asking.
 *
 * While the library asks a decoder about a HEAD (r433_batch_probe_prefilter: the payload lies on an inaccessible page) the
 * wrappers answer without the payload, from the block the library filled in -- see the header for what each may say and
 * why that is all the real helper could have said.
 *
 * Own code: only this file.  Compiled against the reference's headers; C99. */
#include <stdint.h>

#include "r433_hip.h"
#ifdef R433_WRAP_STANDALONE /* a host that is not built against the reference's headers (tests/plugins/pf_helper_decoders.c) */
typedef r433_bitbuffer bitbuffer_t;
#else
#include "bitbuffer.h"
#endif

void __real_bitbuffer_invert(bitbuffer_t *bits);
unsigned __real_bitbuffer_search(bitbuffer_t *bitbuffer, unsigned row, unsigned start, const uint8_t *pattern, unsigned pattern_bits_len);
int __real_bitbuffer_find_repeated_row(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits);
int __real_bitbuffer_find_repeated_prefix(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits);

static volatile int g_sessions;             /* threads that are asking right now */
static __thread r433_helper_probe t_block;

r433_helper_probe *r433_host_helper_probe(int session)
{
    if (session)
        __atomic_add_fetch(&g_sessions, session, __ATOMIC_SEQ_CST);
    return &t_block;
}

#define ASKED(bits) (__builtin_expect(g_sessions != 0, 0) && t_block.armed && (void const *)(bits) == t_block.subject)

void __wrap_bitbuffer_invert(bitbuffer_t *bits)
{
    if (ASKED(bits)) {
        t_block.inverts += 1; /* payload bytes only: nothing the question can see */
        return;
    }
    __real_bitbuffer_invert(bits);
}

unsigned __wrap_bitbuffer_search(bitbuffer_t *bitbuffer, unsigned row, unsigned start, const uint8_t *pattern, unsigned pattern_bits_len)
{
    if (ASKED(bitbuffer)) {
        unsigned const len = bitbuffer->bits_per_row[row]; /* (a row behind the first: behind the fence, the question ends here) */
        if (t_block.searches++ == 0) {
            t_block.row = row;
            t_block.start = start;
            t_block.pattern_bits = pattern_bits_len;
            for (unsigned i = 0; i < 8; ++i)
                t_block.pattern[i] = i < (pattern_bits_len + 7) / 8 ? pattern[i] : 0;
            return t_block.answer < 0 ? len : (unsigned)t_block.answer;
        }
        t_block.overflow = 1;
        return len;
    }
    return __real_bitbuffer_search(bitbuffer, row, start, pattern, pattern_bits_len);
}

/* one row: it is compared with itself only (bitbuffer_count_repeats -> bitbuffer_compare_rows(row, row): equal lengths, equal
 * bytes), so the count is 1 whatever the row holds */
int __wrap_bitbuffer_find_repeated_row(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits)
{
    if (ASKED(bits) && bits->num_rows == 1) {
        t_block.repeats += 1;
        return bits->bits_per_row[0] >= min_bits && 1u >= min_repeats ? 0 : -1;
    }
    return __real_bitbuffer_find_repeated_row(bits, min_repeats, min_bits);
}

int __wrap_bitbuffer_find_repeated_prefix(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits)
{
    if (ASKED(bits) && bits->num_rows == 1) {
        t_block.repeats += 1;
        return bits->bits_per_row[0] >= min_bits && 1u >= min_repeats ? 0 : -1;
    }
    return __real_bitbuffer_find_repeated_prefix(bits, min_repeats, min_bits);
}
