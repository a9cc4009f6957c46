// prefilter.cpp -- the decoder pre-filter: which bitbuffers does a decoder refuse on a look at their
// head alone?  Learned by running the decoder itself under a memory fence, so every verdict is the decoder's own answer,
// given without having seen anything the verdict does not name.
//
// A reference bitbuffer_t (include/bitbuffer.h:34-40) starts { u16 num_rows, free_row, bits_per_row[50], ... }.  The probe
// bitbuffer is laid across a page boundary so that exactly num_rows, free_row and bits_per_row[0] sit on a readable page
// and bits_per_row[1] onwards (the other lengths, the sync counts, every data byte) on pages without access.  decode_fn
// either returns -- then it decided on those three values -- or faults, is caught and counts as "needs the record".
// Verdicts with a failure code become bytes of a (num_rows x bits_per_row[0]) table per decoder that the slicer kernel
// consults where it finishes a bitbuffer (BitSink::fire, slicer_device.hpp).
//
// Reference: account_event (src/pulse_slicer.c:26-66) is what a dropped record would have gone through; the dispatch
// functions add the kernel's per-decoder, per-code counts to the same statistics (dispatch.cpp, apply_prefilter_counts).
#include <climits>
#include <csetjmp>
#include <map>
#include <string>
#include <tuple>
#include <csignal>

#include <sys/mman.h>
#include <unistd.h>

#include "host_common.hpp"

using namespace r433;

namespace {

std::atomic<unsigned long> g_asks{0}, g_faults{0};
thread_local sigjmp_buf t_jump;
thread_local volatile sig_atomic_t t_armed = 0;
struct sigaction g_prev_segv, g_prev_bus;

void on_fault(int sig, siginfo_t *info, void *uctx)
{
    if (t_armed) {
        t_armed = 0;
        g_faults.fetch_add(1, std::memory_order_relaxed);
        siglongjmp(t_jump, 1);
    }
    // not ours: hand over to whoever was there before
    struct sigaction const &prev = sig == SIGSEGV ? g_prev_segv : g_prev_bus;
    if (prev.sa_flags & SA_SIGINFO) {
        if (prev.sa_sigaction) {
            prev.sa_sigaction(sig, info, uctx);
            return;
        }
    }
    else if (prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN) {
        prev.sa_handler(sig);
        return;
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

// the handlers are the process's (installed once per probe); every probing thread has fenced pages of its own
struct Handlers {
    bool ok = false;
    Handlers()
    {
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = on_fault;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        if (sigaction(SIGSEGV, &sa, &g_prev_segv) != 0)
            return;
        if (sigaction(SIGBUS, &sa, &g_prev_bus) != 0) {
            sigaction(SIGSEGV, &g_prev_segv, nullptr);
            return;
        }
        ok = true;
    }
    ~Handlers()
    {
        if (ok) {
            sigaction(SIGSEGV, &g_prev_segv, nullptr);
            sigaction(SIGBUS, &g_prev_bus, nullptr);
        }
    }
};

struct Fence {
    uint8_t *region = nullptr;
    size_t page = 0;
    r433_bitbuffer *bits = nullptr; // its first six (or eight) bytes end the readable page
    bool ok = false;

    explicit Fence(unsigned readable_words = 3) // num_rows, free_row, bits_per_row[0 .. readable_words - 3]
    {
        long const ps = sysconf(_SC_PAGESIZE);
        page = ps > 0 ? (size_t)ps : 4096;
        size_t const tail = (sizeof(r433_bitbuffer) + page - 1) / page * page + page;
        region = (uint8_t *)mmap(nullptr, page + tail, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (region == MAP_FAILED) {
            region = nullptr;
            return;
        }
        total = page + tail;
        if (mprotect(region + page, tail, PROT_NONE) != 0)
            return;
        bits = (r433_bitbuffer *)(region + page - 2 * readable_words);
        ok = true;
    }
    ~Fence()
    {
        if (region)
            munmap(region, total);
    }
    Fence(Fence const &) = delete;
    Fence &operator=(Fence const &) = delete;
    size_t total = 0;

    // the decoder's answer for this head: its return value, or INT_MIN when it reached for more
    int ask(r433_r_device *dev, unsigned rows, unsigned bits0)
    {
        uint16_t *head = (uint16_t *)bits;
        head[0] = (uint16_t)rows;
        head[1] = (uint16_t)rows;
        head[2] = (uint16_t)bits0;
        int ret = INT_MIN;
        if (sigsetjmp(t_jump, 0) == 0) { // (the handler runs with SA_NODEFER and an empty mask: no signal mask to put back)
            t_armed = 1;
            ret = dev->decode_fn(dev, bits);
            t_armed = 0;
        }
        return ret;
    }

    // a two-row head, on a fence that leaves bits_per_row[1] readable
    int ask2(r433_r_device *dev, unsigned bits0, unsigned bits1)
    {
        uint16_t *head = (uint16_t *)bits;
        head[0] = 2;
        head[1] = 2;
        head[2] = (uint16_t)bits0;
        head[3] = (uint16_t)bits1;
        int ret = INT_MIN;
        if (sigsetjmp(t_jump, 0) == 0) {
            t_armed = 1;
            ret = dev->decode_fn(dev, bits);
            t_armed = 0;
        }
        return ret;
    }

    // the same for the row-0 lengths from, from + step, ... below `to`: one jump target for the whole run (setting one
    // up costs more than a decoder that refuses on the spot)
    void ask_run(r433_r_device *dev, unsigned rows, unsigned from, unsigned to, unsigned step, int *out)
    {
        uint16_t *head = (uint16_t *)bits;
        volatile unsigned cur = from; // (lives across the jump)
        g_asks.fetch_add((to - from + step - 1) / step, std::memory_order_relaxed); // (once per run, whatever faults)
        if (sigsetjmp(t_jump, 0) != 0) {
            out[(cur - from) / step] = INT_MIN;
            cur = cur + step;
        }
        t_armed = 1;
        for (; cur < to; cur = cur + step) {
            unsigned const c = cur;
            // the whole head before every question: a decoder may have written to the readable words before it refused
            // (or faulted).  Decoders that allocate before their first look at the bitbuffer would leak on a fault: none
            // of the reference's does; document it for third-party ones.
            head[0] = (uint16_t)rows;
            head[1] = (uint16_t)rows;
            head[2] = (uint16_t)c;
            out[(c - from) / step] = dev->decode_fn(dev, bits);
        }
        t_armed = 0;
    }
};

std::mutex g_probe_lock; // signal dispositions are the process's
r433_helper_probe_fn g_helper = nullptr; // the host wraps its decoders' bitbuffer helpers (r433_prefilter_set_helper_probe)

// What a decoder answered is kept for the life of the process: a host that runs several engines over the same decoders
// (a pipeline of engines, the drop-in's engine per sample rate) asks once.  Keyed by the decoder object and everything of
// it the probe depends on; an empty table = nothing to filter.
struct ProbeKey {
    void const *dev, *fn, *ctx;
    int verbose;
    // ... and what tells one decoder from another that later lives at the same addresses (a decoder freed and another
    // registered in its place; flex decoders, which share one decode_fn): its number, line code, timings and name.  What sits
    // BEHIND decode_ctx cannot be seen from here: a host that changes it calls r433_prefilter_forget.
    unsigned protocol_num, modulation;
    uint32_t timing[6];
    std::string name;
    static ProbeKey of(r433_r_device const *d)
    {
        ProbeKey k{d, (void const *)d->decode_fn, d->decode_ctx, d->verbose, d->protocol_num, d->modulation, {0, 0, 0, 0, 0, 0},
                std::string(d->name ? d->name : "")};
        float const t[6] = {d->short_width, d->long_width, d->reset_limit, d->gap_limit, d->sync_width, d->tolerance};
        memcpy(k.timing, t, sizeof(t));
        return k;
    }
    bool operator<(ProbeKey const &o) const
    {
        return std::tie(dev, fn, ctx, verbose, protocol_num, modulation, timing[0], timing[1], timing[2], timing[3], timing[4], timing[5], name)
                < std::tie(o.dev, o.fn, o.ctx, o.verbose, o.protocol_num, o.modulation, o.timing[0], o.timing[1], o.timing[2], o.timing[3],
                        o.timing[4], o.timing[5], o.name);
    }
};
std::map<ProbeKey, std::vector<uint8_t>> g_known;

// While a decoder is being asked, whatever it hands out goes nowhere.  (A decoder that SUCCEEDS on a head alone -- a flex
// decoder without a minimum length does -- builds a message for it; the library cannot free a data_t, so the probe drops
// such a decoder at its first success and at most a handful of small messages are lost per engine.)
void swallow_output(r433_r_device *, struct data *) {}
void swallow_log(r433_r_device *, int, struct data *) {}

uint8_t verdict_of(int ret) // INT_MIN (the decoder wanted more), a success and an invalid code all mean: the host decides
{
    return ret <= 0 && ret >= R433_DECODE_FAIL_SANITY ? (uint8_t)(-ret) : (uint8_t)kPfKeep;
}

} // namespace

namespace r433 {

// what the last run's filter dropped goes into the decoders' statistics, once (the three dispatch functions call this)
void apply_prefilter_counts(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices)
{
    if (!b->pf_ran || b->pf_accounted)
        return;
    b->pf_accounted = true;
    for (uint32_t d = 0; d < n_devices && d < b->pf_index.size(); ++d) {
        if (!devices[d] || b->pf_index[d] < 0)
            continue;
        uint32_t const *c = b->h_pf_counts.p + (size_t)d * 5;
        for (int k = 0; k < 5; ++k) {
            devices[d]->decode_events += c[k];
            devices[d]->decode_fails[k] += c[k];
        }
    }
}

} // namespace r433

// One-row bitbuffers of a few bits are most of what a slicer makes of a signal that is not its decoder's (a PWM burst under a
// PCM slicer with a short reset limit: a bitbuffer per pulse; 70 % of the records that survive the head tables of the bench's
// decoders are one row of at most 16 bits), and the decoders they go to begin by inverting or searching the row -- a look
// past the head, so the fenced probe learns nothing.  But a row of n bits has 2^n contents: for n <= kTinyBits every one of
// them is ASKED, on an ordinary (readable, writable, otherwise cleared) bitbuffer_t.  If the decoder returns the same failure
// code for all of them, twice over, it returns that code for every real bitbuffer with that head whose row carries no sync
// count -- the verdict is stored with kPfTiny set, and the device applies it only to num_rows == free_row == 1,
// syncs_before_row[0] == 0, no byte ever written past the row's bits (BitSink::fire).
constexpr unsigned kTinyBits = 14;

// An unreadable page for r_device.decode_ctx to point at while a decoder that keeps its state behind that pointer is asked
// (R433_KEEPS_CONTEXT): one per process, never unmapped.
void *dead_context()
{
    static void *const page = [] {
        void *p = mmap(nullptr, 4096, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        return p == MAP_FAILED ? nullptr : p;
    }();
    return page;
}

void probe_tiny(r433_r_device *dev, std::vector<uint8_t> &tab, bool &useful, bool &accepts, bool guarded)
{
    r433_bitbuffer *bits = (r433_bitbuffer *)calloc(1, sizeof(r433_bitbuffer));
    if (!bits)
        return;
    unsigned tiny_bits = kTinyBits;
    if (char const *e = getenv("R433_PROBE_TINY_BITS")) // development: cost / yield of the exhaustive part
        tiny_bits = (unsigned)std::max(0, std::min(atoi(e), (int)kTinyBits)); // (a negative value asks nothing)
    (void)accepts; // (a tiny row it takes is a length left alone, nothing more: unlike a bare head it had the content to go on)
    for (unsigned n = 0; n <= tiny_bits; ++n) {
        if (tab[1 * kPfBits + n] != kPfKeep)
            continue; // refused on the head alone already
        int code = INT_MIN;
        bool same = true;
        for (int round = 0; round < 2 && same; ++round) {
            memset(bits, 0, sizeof *bits); // (all of it once a round, what a refusing decoder can have touched once a call)
            for (unsigned v = 0; v < (1u << n) && same; ++v) {
                memset(bits, 0, offsetof(r433_bitbuffer, bb) + R433_BITBUF_COLS); // head, row lengths, sync counts, row 0
                bits->num_rows = bits->free_row = 1;
                bits->bits_per_row[0] = (uint16_t)n;
                uint32_t const msb = n ? v << (16 - n) : 0u; // MSB first, like bitbuffer_add_bit
                bits->bb[0][0] = (uint8_t)(msb >> 8);
                bits->bb[0][1] = (uint8_t)msb;
                int ret = INT_MIN;
                if (!guarded) {
                    ret = dev->decode_fn(dev, bits);
                }
                else if (sigsetjmp(t_jump, 0) == 0) { // (its context is fenced: a content that makes it reach for its state faults)
                    t_armed = 1;
                    ret = dev->decode_fn(dev, bits);
                    t_armed = 0;
                }
                if (ret == INT_MIN) { // reached for its state: nothing is known about this length
                    same = false;
                    break;
                }
                // a decoder that added or extracted rows before it refused (bitbuffer_add_row, an in-place expansion) must not
                // leave them for the next question: every row its num_rows / free_row reach is cleared, as the replay's
                // call_one does between two real bitbuffers (dispatch.cpp); the head and row 0 are rewritten above anyway
                unsigned const reached = std::min<unsigned>(std::max<unsigned>(bits->num_rows, bits->free_row), R433_BITBUF_ROWS);
                if (reached > 1)
                    memset(bits->bb[1], 0, (size_t)(reached - 1) * R433_BITBUF_COLS);
                if (ret > 0 || ret < R433_DECODE_FAIL_SANITY)
                    same = false;
                else if (code == INT_MIN)
                    code = ret;
                else
                    same = ret == code;
            }
        }
        if (same && code != INT_MIN) {
            tab[1 * kPfBits + n] = (uint8_t)(kPfTiny | (unsigned)(-code));
            useful = true;
        }
    }
    free(bits);
}

bool probe_heads(Fence &fence, r433_r_device *dev, std::vector<uint8_t> &tab, bool &useful, bool &accepts);
bool probe_helpers(Fence &fence, r433_r_device *dev, r433_helper_probe *blk, std::vector<uint8_t> &tab, bool &useful);
bool probe_two_rows(Fence &fence2, r433_r_device *dev, r433_helper_probe *blk, std::vector<uint8_t> &tab, bool &useful);
bool probe_short_rows(Fence &fence, r433_r_device *dev, r433_helper_probe *blk, std::vector<uint8_t> &tab, bool &useful);

// Every question one decoder is asked.  True: `tab` holds its verdicts (something to filter, answers steady).
bool probe_one(Fence &fence, Fence &fence2, r433_r_device *dev, r433_helper_probe *blk, std::vector<uint8_t> &tab, bool fence_context)
{
    if (fence_context && !dead_context())
        return false;
    struct Quiet { // outputs off for the time of the questions
        r433_r_device *d;
        decltype(d->output_fn) out;
        decltype(d->log_fn) log;
        void *ctx;
        Quiet(r433_r_device *dev, bool fence_ctx) : d(dev), out(dev->output_fn), log(dev->log_fn), ctx(dev->decode_ctx)
        {
            d->output_fn = swallow_output;
            d->log_fn = swallow_log;
            if (fence_ctx) // the decoder's state is out of reach: what it answers without a fault it answers without it
                d->decode_ctx = dead_context();
        }
        ~Quiet()
        {
            d->output_fn = out;
            d->log_fn = log;
            d->decode_ctx = ctx;
        }
    } quiet(dev, fence_context);
    bool useful = false, accepts = false;
    tab.assign(kPfTable, (uint8_t)kPfKeep); // a head nobody asked about goes to the host: always safe
    bool const steady = probe_heads(fence, dev, tab, useful, accepts);
    if (!steady || accepts)
        return false;
    if (blk && !probe_helpers(fence, dev, blk, tab, useful))
        return false;
    if (!probe_two_rows(fence2, dev, blk, tab, useful))
        return false;
    if (blk && !probe_short_rows(fence, dev, blk, tab, useful))
        return false;
    probe_tiny(dev, tab, useful, accepts, fence_context);
    return useful && !accepts;
}

// Bitbuffers of several rows that are ALL short (a slicer with a short gap limit makes a row per pulse out of a burst that is
// not its decoder's: twenty rows of one to three bits): a decoder that opens with bitbuffer_find_repeated_row(min_repeats,
// min_bits) finds no row that qualifies and refuses -- but the helper walks the row lengths, which lie behind the fence.  With
// the host's wrappers the question SUPPOSES every row shorter than min_bits (r433_helper_probe.rows_below), the wrapper answers
// -1 on that supposition alone, and a refusal that follows without a look at anything else holds for every bitbuffer of that
// head whose longest row is that short: stored with kPfShort, the bound at kPfShortAt.  False: the answers moved.
bool probe_short_rows(Fence &fence, r433_r_device *dev, r433_helper_probe *blk, std::vector<uint8_t> &tab, bool &useful)
{
    auto ask = [&](unsigned rows, unsigned n, unsigned below) -> int {
        blk->answer = -1;
        blk->subject = fence.bits;
        blk->searches = blk->overflow = blk->row = blk->start = blk->pattern_bits = blk->inverts = blk->repeats = blk->min_bits = 0;
        blk->rows_below = below;
        blk->armed = 1;
        int const ret = fence.ask(dev, rows, n);
        blk->armed = 0;
        blk->rows_below = 0;
        return ret;
    };
    // the decoder's min_bits: what its helper asks for when every row is supposed to have at most one bit
    unsigned bound = 0;
    {
        int const r = ask(3, 1, 2);
        if (verdict_of(r) == kPfKeep || !blk->repeats || blk->searches || blk->min_bits < 2 || blk->min_bits > 0xffffu)
            return true; // (not this kind of decoder)
        bound = blk->min_bits;
    }
    std::vector<std::pair<unsigned, uint8_t>> got; // (table index, code)
    for (unsigned rows = 2; rows < kPfHeadRows; ++rows) {
        unsigned in_a_row = 0;
        for (unsigned n = 0; n < bound && n < kPfBits && in_a_row < 2; ++n) {
            if (tab[rows * kPfBits + n] != kPfKeep)
                continue; // (the head alone settles it)
            int const r = ask(rows, n, bound);
            uint8_t const v = verdict_of(r);
            bool const good = v != kPfKeep && blk->repeats && !blk->searches && blk->min_bits == bound;
            in_a_row = good ? 0 : in_a_row + 1;
            if (good)
                got.push_back({rows * kPfBits + n, v});
        }
    }
    for (auto const &g : got) // once more
        if (verdict_of(ask(g.first / kPfBits, g.first % kPfBits, bound)) != g.second)
            return false;
    if (got.empty())
        return true;
    for (auto const &g : got)
        tab[g.first] = (uint8_t)(kPfShort | g.second);
    uint16_t const b16 = (uint16_t)bound;
    memcpy(tab.data() + kPfShortAt, &b16, 2);
    useful = true;
    return true;
}

// Bitbuffers of exactly two rows, by both rows' lengths (kPfTwoAt): the fence leaves bits_per_row[1] readable as well, and a
// head is (2, 2, bits0, bits1).  With the host's wrappers the helpers answer as in probe_helpers (a search is told "not
// found" and every position it could have found; a repeated-row test on two rows goes to the real helper, which compares
// lengths first and payloads only if they tie).  False: the answers moved.
bool probe_two_rows(Fence &fence2, r433_r_device *dev, r433_helper_probe *blk, std::vector<uint8_t> &tab, bool &useful)
{
    auto ask = [&](unsigned n0, unsigned n1, int answer) -> int {
        if (blk) {
            blk->answer = answer;
            blk->subject = fence2.bits;
            blk->searches = blk->overflow = blk->row = blk->start = blk->pattern_bits = blk->inverts = blk->repeats = 0;
            blk->armed = 1;
        }
        int const ret = fence2.ask2(dev, n0, n1);
        if (blk)
            blk->armed = 0;
        return ret;
    };
    auto verdict = [&](unsigned n0, unsigned n1) -> uint8_t {
        int const r0 = ask(n0, n1, -1);
        uint8_t const v0 = verdict_of(r0);
        if (v0 == kPfKeep || (blk && blk->overflow))
            return (uint8_t)kPfKeep;
        if (blk && blk->searches) {
            unsigned const row = blk->row, start = blk->start, plen = blk->pattern_bits, n = row == 0 ? n0 : n1;
            if (row > 1)
                return (uint8_t)kPfKeep;
            if (plen >= 1 && plen <= n)
                for (unsigned pos = start; pos + plen <= n; ++pos)
                    if (ask(n0, n1, (int)pos) != r0 || blk->overflow || !blk->searches || blk->row != row || blk->start != start || blk->pattern_bits != plen)
                        return (uint8_t)kPfKeep;
        }
        return v0;
    };
    // a decoder that reaches for the payload whatever the two lengths say is not asked a thousand times
    static unsigned const sample[8][2] = {{1, 0}, {0, 0}, {1, 1}, {2, 0}, {40, 0}, {9, 3}, {64, 1}, {127, 7}};
    bool any = false;
    for (auto const &q : sample)
        any |= ask(q[0], q[1], -1) != INT_MIN;
    if (!any)
        return true;
    uint8_t *const two = tab.data() + kPfTwoAt;
    std::vector<uint8_t> got(kPfTwo0 * kPfTwo1, (uint8_t)kPfKeep);
    // (faults are what costs: two second-row lengths in a row that make the decoder reach for the payload end a first-row
    // length, twenty such first-row lengths in a row end the questions)
    unsigned faults = 0, barren = 0;
    for (unsigned n0 = 0; n0 < kPfTwo0 && faults < 300 && barren < 20; ++n0) {
        if (tab[2 * kPfBits + n0] != kPfKeep)
            continue; // (the head alone settles it)
        unsigned in_a_row = 0, learned = 0;
        for (unsigned n1 = 0; n1 < kPfTwo1 && in_a_row < 2; ++n1) {
            uint8_t const v = verdict(n0, n1);
            got[n0 * kPfTwo1 + n1] = v;
            in_a_row = v == kPfKeep ? in_a_row + 1 : 0;
            faults += v == kPfKeep;
            learned += v != kPfKeep;
        }
        barren = learned ? 0 : barren + 1;
    }
    for (unsigned i = 0; i < kPfTwo0 * kPfTwo1; ++i) // once more (only refusals: no faults unless the decoder moved)
        if (got[i] != kPfKeep && verdict_of(ask(i / kPfTwo1, i % kPfTwo1, -1)) != got[i])
            return false;
    for (unsigned i = 0; i < kPfTwo0 * kPfTwo1; ++i)
        if (got[i] != kPfKeep) {
            two[i] = got[i];
            useful = true;
        }
    return true;
}

// The questions a host with wrapped bitbuffer helpers makes possible (include/r433_hip.h, r433_helper_probe): one-row heads
// the decoder reached past under the bare fence are asked again with the helpers answering from `blk` -- once per answer
// bitbuffer_search could have given -- and a head becomes a verdict where every answer led to the same failure code without
// a look behind the head.  False: the answers moved between two askings.
constexpr unsigned kHelperBits = 512; // row lengths asked (a question per search position: n^2 / 2 calls up to here)

bool probe_helpers(Fence &fence, r433_r_device *dev, r433_helper_probe *blk, std::vector<uint8_t> &tab, bool &useful)
{
    auto ask = [&](unsigned n, int answer) -> int {
        blk->answer = answer;
        blk->subject = fence.bits;
        blk->searches = blk->overflow = blk->row = blk->start = blk->pattern_bits = blk->inverts = blk->repeats = 0;
        blk->armed = 1;
        int const ret = fence.ask(dev, 1, n);
        blk->armed = 0;
        return ret;
    };
    // a decoder whose first look at a one-row bitbuffer is not through a helper is not asked 500 times (each a fault)
    static unsigned const sample[6] = {1, 9, 24, 40, 130, 300};
    bool any = false;
    for (unsigned n : sample) {
        int const ret = ask(n, -1);
        any |= ret != INT_MIN || blk->searches || blk->inverts || blk->repeats;
        if (ret != INT_MIN && ret > 0)
            return true; // (takes a bare head under these answers: nothing to filter here, and nothing wrong)
    }
    if (!any)
        return true;
    // every answer the real helper could have given for this head; the verdict if they all agree on a refusal.  Where they
    // do not, but "not found" alone is a refusal, the head is a candidate for the decoder's search rule (kPfRule): the
    // search, run on the device, then says which of the answers applies.
    struct Cand {
        uint32_t start, plen, pat;
        uint8_t code;
        bool operator==(Cand const &o) const { return start == o.start && plen == o.plen && pat == o.pat && code == o.code; }
    };
    std::vector<std::pair<unsigned, Cand>> cands;
    bool reached = false; // the last question's first answer made the decoder reach for the payload (a fault)
    auto verdict = [&](unsigned n) -> uint8_t {
        int const r0 = ask(n, -1);
        reached = r0 == INT_MIN;
        // bitbuffer_invert before the search (tpms_eezrv.c:82-91): the decoder searches the INVERTED row, the device would search
        // the row as the slicer left it -- no search rule then; the head-alone verdicts below hold whatever the row contains
        bool const inverted = blk->inverts != 0;
        uint8_t const v0 = verdict_of(r0);
        if (v0 == kPfKeep || blk->overflow)
            return (uint8_t)kPfKeep;
        if (blk->searches) {
            if (blk->row != 0)
                return (uint8_t)kPfKeep; // (unreachable under the fence: the wrapper read that row's length)
            unsigned const start = blk->start, plen = blk->pattern_bits;
            uint32_t const pat = (uint32_t)blk->pattern[0] << 24 | (uint32_t)blk->pattern[1] << 16 | (uint32_t)blk->pattern[2] << 8 | blk->pattern[3];
            if (plen >= 1 && plen <= n) // a match ends inside the row: it begins at start .. n - plen
                for (unsigned pos = start; pos + plen <= n; ++pos) {
                    int const r = ask(n, (int)pos);
                    if (blk->overflow || !blk->searches || blk->start != start || blk->pattern_bits != plen)
                        return (uint8_t)kPfKeep;
                    if (r != r0) {
                        if (!inverted && !blk->inverts && n < kPfRuleMaxBits && plen <= 32 && start < 65536u)
                            cands.push_back({n, Cand{start, plen, plen == 32 ? pat : pat & ~(0xffffffffu >> plen), v0}});
                        return (uint8_t)kPfKeep;
                    }
                }
        }
        return v0;
    };
    // (the same economy as under the bare fence: a grid of every sixteenth length first, nothing asked between two grid
    // lengths that both made the decoder reach for the payload -- faults do not run side by side)
    constexpr unsigned kGrid = 16;
    bool grid_reached[kHelperBits / kGrid + 1];
    std::vector<uint8_t> got(kHelperBits, (uint8_t)kPfKeep);
    auto wanted = [&](unsigned n) { // heads the fence alone did not settle (a tiny-row verdict holds for plain rows only: ask)
        uint8_t const t = tab[1 * kPfBits + n];
        return t == kPfKeep || (t & kPfTiny);
    };
    for (unsigned g = 0; g * kGrid < kHelperBits; ++g) {
        reached = false;
        got[g * kGrid] = wanted(g * kGrid) ? verdict(g * kGrid) : tab[1 * kPfBits + g * kGrid];
        grid_reached[g] = reached;
    }
    for (unsigned g = 0; g * kGrid < kHelperBits; ++g) {
        bool const last = (g + 1) * kGrid >= kHelperBits;
        if (!last && grid_reached[g] && grid_reached[g + 1])
            continue;
        for (unsigned n = g * kGrid + 1; n < std::min(kHelperBits, (g + 1) * kGrid); ++n)
            if (wanted(n))
                got[n] = verdict(n);
    }
    // once more, the answer "not found" alone: verdicts of a decoder whose answers move are worth nothing
    for (unsigned n = 0; n < kHelperBits; ++n)
        if (wanted(n) && got[n] != kPfKeep && verdict_of(ask(n, -1)) != got[n])
            return false;
    for (unsigned n = 0; n < kHelperBits; ++n)
        if (wanted(n) && got[n] != kPfKeep) {
            tab[1 * kPfBits + n] = got[n];
            useful = true;
        }
    // the search rule: the (start, pattern, code) most candidate lengths share, and those lengths
    if (!cands.empty()) {
        size_t best = 0, best_n = 0;
        for (size_t i = 0; i < cands.size(); ++i) {
            size_t cnt = 0;
            for (auto const &c : cands)
                cnt += c.second == cands[i].second;
            if (cnt > best_n)
                best = i, best_n = cnt;
        }
        Cand const rule = cands[best].second;
        uint64_t lens = 0;
        for (auto const &c : cands)
            if (c.second == rule) {
                if (verdict_of(ask(c.first, -1)) != rule.code) // (asked again, as above)
                    return false;
                lens |= 1ull << c.first;
            }
        uint8_t *const r = tab.data() + kPfRule; // (in the unused part of the table's row for empty bitbuffers)
        r[0] = rule.code;
        r[1] = (uint8_t)rule.plen;
        uint16_t const st = (uint16_t)rule.start;
        memcpy(r + 2, &st, 2);
        memcpy(r + 4, &rule.pat, 4);
        memcpy(r + 8, &lens, 8);
        useful = true;
    }
    return true;
}

// the questions under the memory fence: heads a decoder refuses without looking further.  False: its answers moved.
bool probe_heads(Fence &fence, r433_r_device *dev, std::vector<uint8_t> &tab, bool &useful, bool &accepts)
{
    // a decoder that reaches past the head whatever the head says is not worth 50 000 faults, and one that accepts a
    // bare head is nothing to filter
    static unsigned const sample[8][2] = {{1, 0}, {1, 1}, {1, 7}, {2, 5}, {3, 200}, {1, 1000}, {5, 33}, {12, 12}};
    bool any = false;
    for (auto const &s : sample) {
        int const ret = fence.ask(dev, s[0], s[1]);
        any |= ret != INT_MIN;
        accepts |= ret != INT_MIN && ret > 0;
    }
    if (!any || accepts)
        return true; // (nothing learned under the fence; the tiny rows are still worth asking unless it accepted)
    unsigned faults = 0;
    unsigned blind_rows = 0; // row counts in a row for which the decoder reached past the head whatever the length
    // (bitbuffers of more than 24 rows are a few in a hundred: their heads are left unasked -- they go to the host)
    constexpr unsigned kAskedRows = 25;
    for (unsigned rows = 0; rows < kAskedRows && !accepts && faults < 6000 && blind_rows < 3; ++rows) {
        // a decoder that walks all its rows reaches past the head for every length of row 0: eight lengths tell (and three
        // such row counts in a row tell for the row counts behind them: those are left unasked)
        if (rows > 0) {
            static unsigned const spread[8] = {0, 1, 9, 40, 77, 200, 520, 1023};
            bool all_fault = true;
            for (unsigned l : spread)
                all_fault &= fence.ask(dev, rows, l) == INT_MIN;
            if (all_fault) {
                faults += 8;
                blind_rows += 1;
                continue;
            }
            blind_rows = 0;
        }
        // Row lengths a decoder wants to look at come in stretches (its minimum to its maximum): every question inside one
        // is a fault, microseconds each, and faults do not run side by side (the kernel hands a process its signals one at a
        // time).  So a grid of every sixteenth length first; between two grid lengths that both made the decoder reach for
        // more nothing is asked -- an unasked head goes to the host, which is always right.
        unsigned const n_len = rows ? kPfBits : 1u; // (an empty bitbuffer has no row)
        constexpr unsigned kGrid = 16;
        int grid_ret[kPfBits / kGrid + 2];
        auto record = [&](unsigned bits0, int ret) {
            faults += ret == INT_MIN;
            accepts |= ret != INT_MIN && ret > 0;
            uint8_t const v = verdict_of(ret);
            tab[rows * kPfBits + bits0] = v;
            useful |= v != kPfKeep;
        };
        // (and a grid of every 128th length before that one, for the same reason)
        unsigned const n_grid = (n_len + kGrid - 1) / kGrid;
        constexpr unsigned kCoarse = 8; // grid lengths per coarse step
        int coarse_ret[kPfBits / kGrid / kCoarse + 2];
        unsigned const n_coarse = (n_grid + kCoarse - 1) / kCoarse;
        fence.ask_run(dev, rows, 0, n_len, kGrid * kCoarse, coarse_ret);
        for (unsigned c = 0; c < n_coarse; ++c) {
            unsigned const g0 = c * kCoarse, g1 = std::min(n_grid, g0 + kCoarse);
            bool const last = c + 1 >= n_coarse;
            grid_ret[g0] = coarse_ret[c];
            if (!last && coarse_ret[c] == INT_MIN && coarse_ret[c + 1] == INT_MIN) {
                for (unsigned g = g0 + 1; g < g1; ++g)
                    grid_ret[g] = INT_MIN; // taken for a length the decoder looks at: unasked, kept
                continue;
            }
            if (g0 + 1 < g1)
                fence.ask_run(dev, rows, (g0 + 1) * kGrid, std::min(n_len, g1 * kGrid), kGrid, grid_ret + g0 + 1);
        }
        for (unsigned g = 0; g < n_grid; ++g)
            if (grid_ret[g] != INT_MIN || g % kCoarse == 0) // (the assumed ones are neither counted nor recorded)
                record(g * kGrid, grid_ret[g]);
        for (unsigned g = 0; g < n_grid && !accepts; ++g) {
            unsigned const from = g * kGrid + 1, to = std::min(n_len, (g + 1) * kGrid);
            bool const last = g + 1 >= n_grid; // (behind the last grid length: asked whatever that one said)
            if (from >= to || (!last && grid_ret[g] == INT_MIN && grid_ret[g + 1] == INT_MIN))
                continue;
            int between[kGrid];
            fence.ask_run(dev, rows, from, to, 1, between);
            for (unsigned bits0 = from; bits0 < to; ++bits0)
                record(bits0, between[bits0 - from]);
        }
    }
    if (!useful || accepts)
        return true;
    // the same questions again: a decoder whose answers move between calls keeps state that its length test looks at
    // (only refusals are asked again, in runs of neighbouring lengths: no faults here unless the decoder did move)
    std::vector<int> again(kPfBits);
    for (unsigned rows = 0; rows < kPfRows; ++rows) {
        unsigned const n_len = rows ? kPfBits : 1u;
        for (unsigned from = 0; from < n_len;) {
            if (tab[rows * kPfBits + from] == kPfKeep) {
                from += 1;
                continue;
            }
            unsigned to = from;
            while (to < n_len && tab[rows * kPfBits + to] != kPfKeep)
                to += 1;
            fence.ask_run(dev, rows, from, to, 1, again.data());
            for (unsigned bits0 = from; bits0 < to; ++bits0)
                if (tab[rows * kPfBits + bits0] != verdict_of(again[bits0 - from]))
                    return false;
            from = to;
        }
    }
    return true;
}

extern "C" {

int r433_batch_probe_prefilter(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices)
{
    if (!b || (!devices && n_devices))
        return fail(R433_EINVAL, "null argument");
    DeviceScope on_device(b->device);
    if (n_devices != b->timing.size())
        return fail(R433_EINVAL, "the probe needs the %zu devices the engine was created with", b->timing.size());
    std::lock_guard<std::mutex> guard(g_probe_lock);
    Handlers handlers;
    if (!handlers.ok)
        return fail(R433_ENOMEM, "pre-filter probe: sigaction failed");
    uint32_t const lowest = b->prio_levels.empty() ? 0u : b->prio_levels.front();
    b->pf_tables.clear();
    b->pf_index.assign(n_devices, -1);
    // who has to be asked (the others are known from an earlier engine of this process, or are not filtered at all)
    std::vector<uint32_t> ask_list;
    std::vector<ProbeKey> keys(n_devices);
    std::vector<uint8_t> eligible(n_devices, 0);
    for (uint32_t d = 0; d < n_devices; ++d) {
        r433_r_device *dev = devices[d];
        // only quiet decoders (account_event prints refused bitbuffers at -vv).  Those of later priority levels are not called
        // for every package: what the filter proves refused goes along as a stub, the replay books it if it gets there (kPfStub)
        if (!dev || !dev->decode_fn || dev->verbose)
            continue;
        // A decoder the host says keeps state between calls (r433_batch_set_stateless, made before the probe: statics like
        // src/devices/secplus_v1.c:142-143, a create_fn's context) is not asked: what it answers may depend on what it was asked
        // before -- a refusal on the head alone is then no function of the head --, and the questions themselves would leave
        // made-up half-messages in its state for the replay to pair real ones with.  Its records all cross.
        // (2 = all of its state sits behind decode_ctx: asked with that pointer on an unreadable page, probe_one)
        if (!b->stateless.empty() && d < b->stateless.size() && !b->stateless[d])
            continue;
        if (!b->stateless.empty() && d < b->stateless.size() && b->stateless[d] == R433_KEEPS_CONTEXT && !dev->decode_ctx)
            continue; // (says its state is behind a context and has none: not to be believed)
        eligible[d] = 1;
        keys[d] = ProbeKey::of(dev);
        if (g_known.find(keys[d]) == g_known.end()) {
            bool twice = false; // (one decoder object registered twice is asked once)
            for (uint32_t o : ask_list)
                twice |= devices[o] == dev;
            if (!twice)
                ask_list.push_back(d);
        }
    }
    // Decoders are asked side by side (each thread its own fenced pages, its own decoder objects).
    std::vector<std::vector<uint8_t>> answers(ask_list.size());
    std::vector<uint8_t> answered(ask_list.size(), 0);
    if (!ask_list.empty()) {
        unsigned const hw = std::thread::hardware_concurrency();
        // (eight: the refusals and the seven million tiny rows run side by side, the faults do not -- a process takes its
        // signals one at a time.  Measured for the reference's 335 decoders: 0.95 s on one thread, 0.49 on two, 0.29 on
        // four, 0.21 on eight; the fenced questions alone were 0.21 s on one thread and 0.11 s on four or eight.)
        unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(hw ? hw : 1u, 8u), ask_list.size() / 4));
        if (char const *e = getenv("R433_PROBE_THREADS")) // development: A/B timing
            nt = (unsigned)std::max(1, atoi(e));
        std::atomic<uint32_t> cursor{0};
        std::atomic<int> no_pages{0};
        r433_helper_probe_fn const helper = g_helper;
        b->pool.run(nt, [&](unsigned) {
            Fence fence, fence2(4);
            if (!fence.ok || !fence2.ok) {
                no_pages.store(1);
                return;
            }
            r433_helper_probe *const blk = helper ? helper(+1) : nullptr; // this thread's block of the host's wrappers
            for (;;) {
                uint32_t const k = cursor.fetch_add(1, std::memory_order_relaxed);
                if (k >= ask_list.size())
                    break;
                uint32_t const d = ask_list[k];
                bool const fence_context = d < b->stateless.size() && b->stateless[d] == R433_KEEPS_CONTEXT;
                answered[k] = probe_one(fence, fence2, devices[d], blk, answers[k], fence_context) ? 1 : 0;
            }
            if (helper)
                helper(-1);
        });
        if (no_pages.load() && cursor.load() < ask_list.size())
            return fail(R433_ENOMEM, "pre-filter probe: no fenced page (mmap / mprotect)");
        for (size_t k = 0; k < ask_list.size(); ++k) { // (a thread without pages leaves its share to the others; unasked = unknown)
            std::vector<uint8_t> &verdicts = g_known[keys[ask_list[k]]]; // stays empty unless the questions ended well
            if (answered[k])
                verdicts = answers[k];
        }
    }
    if (char const *trace = getenv("R433_PROBE_TRACE")) {
        fprintf(stderr, "r.probe: %zu decoders asked, %lu questions in runs, %lu faults\n", ask_list.size(), g_asks.load(), g_faults.load());
        if (atoi(trace) >= 2) // development: what was learned per decoder beyond the head tables
            for (uint32_t d = 0; d < n_devices; ++d) {
                auto const known = eligible[d] ? g_known.find(keys[d]) : g_known.end();
                if (known == g_known.end() || known->second.empty())
                    continue;
                uint8_t const *t = known->second.data();
                unsigned heads = 0, one_row = 0, two = 0, shorts = 0;
                for (unsigned i = 0; i < kPfHeadRows * kPfBits; ++i) {
                    bool const v = t[i] != kPfKeep && !(i < kPfBits && i > 0);
                    heads += v;
                    one_row += v && i / kPfBits == 1;
                    shorts += v && (t[i] & kPfShort);
                }
                for (unsigned i = 0; i < kPfTwo0 * kPfTwo1; ++i)
                    two += t[kPfTwoAt + i] != kPfKeep;
                uint64_t lens;
                uint32_t pat;
                uint16_t st, bound;
                memcpy(&lens, t + kPfRule + 8, 8);
                memcpy(&pat, t + kPfRule + 4, 4);
                memcpy(&st, t + kPfRule + 2, 2);
                memcpy(&bound, t + kPfShortAt, 2);
                fprintf(stderr, "r.probe: [%3u] %-40.40s heads %5u (one row %4u, short rows %4u below %u bits) two rows %4u", d, devices[d]->name ? devices[d]->name : "?",
                        heads, one_row, shorts, shorts ? bound : 0u, two);
                if (t[kPfRule] != kPfKeep)
                    fprintf(stderr, "  search rule: code %u, %u bits %08x from %u, lengths %016llx", t[kPfRule], t[kPfRule + 1], pat, st, (unsigned long long)lens);
                fprintf(stderr, "\n");
            }
    }
    int filtered = 0;
    for (uint32_t d = 0; d < n_devices; ++d) {
        if (!eligible[d])
            continue;
        auto const known = g_known.find(keys[d]);
        if (known != g_known.end() && !known->second.empty()) {
            b->pf_index[d] = (int)(b->pf_tables.size() / kPfTable);
            b->pf_tables.insert(b->pf_tables.end(), known->second.begin(), known->second.end());
            b->pf_tables[b->pf_tables.size() - kPfTable + kPfStub] = b->timing[d].priority != lowest ? 1 : 0; // (this engine's copy)
            filtered += 1;
        }
    }
    for (DevRow &r : b->rows)
        r.pf = r.orig >= 0 ? b->pf_index[(size_t)r.orig] : -1;
    int rc;
    if ((rc = b->d_pf_tables.ensure(std::max<size_t>(b->pf_tables.size(), 16))) || (rc = b->d_pf_counts.ensure((size_t)n_devices * 5 + 16))
            || (rc = b->h_pf_counts.ensure((size_t)n_devices * 5 + 16)))
        return rc;
    if (!b->pf_tables.empty())
        HIP_TRY(hipMemcpy(b->d_pf_tables.p, b->pf_tables.data(), b->pf_tables.size(), hipMemcpyHostToDevice));
    if (!b->rows.empty())
        HIP_TRY(hipMemcpy(b->d_rows.p, b->rows.data(), b->rows.size() * sizeof(DevRow), hipMemcpyHostToDevice));
    b->pf_on = filtered > 0;
    return filtered;
}

void r433_prefilter_set_helper_probe(r433_helper_probe_fn host_block)
{
    std::lock_guard<std::mutex> guard(g_probe_lock);
    g_helper = host_block;
}

void r433_prefilter_forget(void)
{
    std::lock_guard<std::mutex> guard(g_probe_lock);
    g_known.clear();
}

int r433_batch_set_prefilter(r433_batch *b, int on)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (on && b->pf_tables.empty())
        return fail(R433_EINVAL, "no pre-filter tables: call r433_batch_probe_prefilter first");
    b->pf_on = on != 0;
    return 0;
}

int r433_batch_prefilter_counts(r433_batch *b, uint32_t const **counts, uint32_t *n_devices)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    if (counts)
        *counts = b->pf_ran ? b->h_pf_counts.p : nullptr;
    if (n_devices)
        *n_devices = b->pf_ran ? (uint32_t)b->pf_index.size() : 0u;
    return 0;
}

} // extern "C"
