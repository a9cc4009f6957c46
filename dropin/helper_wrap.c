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

#ifdef R433_WRAP_PROFILE /* development: where the decoders' time goes (cycles and calls per helper, all threads) */
unsigned long long r433_wrap_cycles[4], r433_wrap_calls[4], r433_wrap_bits[4];
#define PROFILED(k, len, call) do { unsigned long long t0 = __builtin_ia32_rdtsc(); call; \
        __atomic_add_fetch(&r433_wrap_cycles[k], __builtin_ia32_rdtsc() - t0, __ATOMIC_RELAXED); \
        __atomic_add_fetch(&r433_wrap_calls[k], 1, __ATOMIC_RELAXED); __atomic_add_fetch(&r433_wrap_bits[k], (len), __ATOMIC_RELAXED); } while (0)
#else
#define PROFILED(k, len, call) call
#endif

/* bitbuffer_search for the replay: the same answer as src/bitbuffer.c:228-253 -- the first position from `start` on where all
 * pattern bits lie inside the row and match, else the row's length -- found eight positions per loaded word instead of a bit
 * per step.  The reference walks the row with a bit_at() per comparison, 13 cycles a bit; the decoders that still get records
 * behind the pre-filter are the ones that search long PCM rows for their preamble (one or two searches over 200-1600 bits per
 * call), and that loop was 57 % of the whole replay's CPU time (tools/pf_survivors.py, R433_WRAP_PROFILE).  Patterns of up to
 * 57 bits (a window of 64 holds the pattern at any of the 8 bit offsets of a byte); longer ones go to the reference's loop.
 * Where the pattern cannot fit behind `start` it answers without a look at the row (the reference reads a bit first): under
 * the pre-filter's fence that is one more head learned, and rightly -- the answer cannot depend on the payload there.
 * r433_host_wrap_selftest compares the two on random rows (tests/test_prefilter.py). */
static uint64_t load_be64(uint8_t const *p, unsigned have)
{
    uint64_t w = 0;
    if (have >= 8) {
        __builtin_memcpy(&w, p, 8);
        return __builtin_bswap64(w);
    }
    for (unsigned i = 0; i < have; ++i) /* the row's last bytes: nothing behind them is read */
        w |= (uint64_t)p[i] << (56 - 8 * i);
    return w;
}

static unsigned fast_search(uint8_t const *bits, unsigned len, unsigned start, uint8_t const *pattern, unsigned plen)
{
    if (plen == 0 || start >= len || plen > len - start)
        return len;
    /* the pattern and its mask at each of the eight bit offsets of a byte, left-aligned in the 64-bit window */
    uint64_t const top = ~0ull << (64 - plen), pat = load_be64(pattern, (plen + 7) / 8) & top;
    uint64_t pat_at[8], mask_at[8];
    for (unsigned s = 0; s < 8; ++s) {
        pat_at[s] = pat >> s;
        mask_at[s] = top >> s;
    }
    unsigned const nbytes = (len + 7) / 8;
    unsigned const last = len - plen; /* the last position a match can begin at */
    for (unsigned byte = start / 8; byte * 8 <= last; ++byte) {
        uint64_t const w = load_be64(bits + byte, nbytes - byte);
        unsigned const s0 = byte * 8 < start ? start - byte * 8 : 0;
        for (unsigned s = s0; s < 8; ++s) {
            if ((w & mask_at[s]) == pat_at[s]) {
                unsigned const at = byte * 8 + s;
                return at <= last ? at : len;
            }
        }
    }
    return len;
}

#define ASKED(bits) (__builtin_expect(g_sessions != 0, 0) && t_block.armed && (void const *)(bits) == t_block.subject)

void __wrap_bitbuffer_invert(bitbuffer_t *bits)
{
    if (ASKED(bits)) {
        t_block.inverts += 1; /* payload bytes only: nothing the question can see */
        return;
    }
    PROFILED(0, bits->bits_per_row[0], __real_bitbuffer_invert(bits));
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
    unsigned at;
    if (pattern_bits_len <= 57)
        PROFILED(1, bitbuffer->bits_per_row[row], at = fast_search(bitbuffer->bb[row], bitbuffer->bits_per_row[row], start, pattern, pattern_bits_len));
    else
        PROFILED(1, bitbuffer->bits_per_row[row], at = __real_bitbuffer_search(bitbuffer, row, start, pattern, pattern_bits_len));
    return at;
}

/* one row: it is compared with itself only (bitbuffer_count_repeats -> bitbuffer_compare_rows(row, row): equal lengths, equal
 * bytes), so the count is 1 whatever the row holds */
int __wrap_bitbuffer_find_repeated_row(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits)
{
    if (ASKED(bits) && bits->num_rows == 1) {
        t_block.repeats += 1;
        return bits->bits_per_row[0] >= min_bits && 1u >= min_repeats ? 0 : -1;
    }
    if (ASKED(bits) && t_block.rows_below && min_bits >= t_block.rows_below) {
        t_block.repeats += 1; /* every row is supposed shorter than min_bits: none qualifies (src/bitbuffer.c:513-533) */
        t_block.min_bits = min_bits;
        return -1;
    }
    int row;
    PROFILED(2, bits->num_rows, row = __real_bitbuffer_find_repeated_row(bits, min_repeats, min_bits));
    return row;
}

int __wrap_bitbuffer_find_repeated_prefix(bitbuffer_t *bits, unsigned min_repeats, unsigned min_bits)
{
    if (ASKED(bits) && bits->num_rows == 1) {
        t_block.repeats += 1;
        return bits->bits_per_row[0] >= min_bits && 1u >= min_repeats ? 0 : -1;
    }
    if (ASKED(bits) && t_block.rows_below && min_bits >= t_block.rows_below) {
        t_block.repeats += 1; /* every row is supposed shorter than min_bits: none qualifies (src/bitbuffer.c:513-533) */
        t_block.min_bits = min_bits;
        return -1;
    }
    int row;
    PROFILED(3, bits->num_rows, row = __real_bitbuffer_find_repeated_prefix(bits, min_repeats, min_bits));
    return row;
}

/* fast_search against the helper it stands in for, on `n` random rows and patterns (lengths, starts and pattern lengths
 * around every edge; patterns cut from the row itself half of the time, so that matches are common) -> mismatches */
unsigned r433_host_wrap_selftest(unsigned seed, unsigned n)
{
    static __thread bitbuffer_t bb;
    unsigned bad = 0;
    uint64_t x = 0x9e3779b97f4a7c15ull ^ seed;
    for (unsigned k = 0; k < n; ++k) {
#define RND() (x ^= x << 13, x ^= x >> 7, x ^= x << 17, (unsigned)(x >> 32))
        unsigned const row = RND() % 3;
        unsigned const len = RND() % 5 == 0 ? RND() % 1400 : RND() % 130;
        unsigned const sparse = RND() % 4; /* rows of few ones / few zeros make long partial matches */
        bb.num_rows = bb.free_row = (uint16_t)(row + 1);
        bb.bits_per_row[row] = (uint16_t)len;
        for (unsigned i = 0; i < (len + 7) / 8 + 9 && i + row * sizeof(bb.bb[0]) < sizeof(bb.bb); ++i) {
            unsigned v = RND();
            ((uint8_t *)bb.bb[row])[i] = (uint8_t)(sparse == 0 ? v : sparse == 1 ? v & (v >> 8) & (v >> 16) : sparse == 2 ? v | (v >> 8) | (v >> 16) : 0xaa);
        }
        uint8_t pat[8] = {0};
        unsigned const plen = RND() % 58;
        if (RND() & 1 && len > plen) { /* a piece of the row */
            unsigned const from = RND() % (len - plen + 1);
            for (unsigned i = 0; i < plen; ++i)
                if ((((uint8_t *)bb.bb[row])[(from + i) >> 3] >> (7 - ((from + i) & 7))) & 1)
                    pat[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
        }
        else
            for (unsigned i = 0; i < 8; ++i)
                pat[i] = (uint8_t)RND();
        unsigned const start = RND() % 3 == 0 ? RND() % (len + 3) : 0;
#undef RND
        bad += fast_search(bb.bb[row], len, start, pat, plen) != __real_bitbuffer_search(&bb, row, start, pat, plen);
    }
    return bad;
}
