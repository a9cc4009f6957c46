// slicer_device.hpp -- line-code slicers and the bitbuffer writer, one (package, r_device) per lane.
//
// Each lane replays one of the reference's pulse_slicer_* functions (src/pulse_slicer.c:68-918) over a
// pulse package staged in LDS and serialises every bitbuffer it would hand to decode_fn
// (account_event, src/pulse_slicer.c:26-66) as an r433_evt_rec.  The writer reproduces
// bitbuffer_add_bit/add_row/add_sync (src/bitbuffer.c:22-133) including row spill past 1024 bits,
// the 50-row overflow rule (the last row's length is zeroed but its bytes stay and later bits are
// OR-ed on top) and the rows a spill skips.  The same code runs twice: COUNT sizes the records,
// WRITE emits them at the scanned offsets, so the event stream is dense, ordered and atomics-free.
//
// Float spots of the reference are evaluated with explicit round-to-nearest single operations
// (__fmul_rn/__fadd_rn/__fdiv_rn) so no FMA contraction can change a truncation.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "r433_internal.hpp"

namespace r433 {

struct PulseView {
    int2 const *pairs; // LDS: {pulse, gap}
    uint32_t num;
    __device__ __forceinline__ int pulse(uint32_t n) const { return pairs[n].x; }
    __device__ __forceinline__ int gap(uint32_t n) const { return pairs[n].y; }
    __device__ __forceinline__ int symbol(uint32_t k) const { return (k & 1) ? pairs[k >> 1].y : pairs[k >> 1].x; }
};

template <bool WRITE> struct BitSink {
    uint8_t *out;       // lane's record region (WRITE only)
    uint32_t limit;     // size of that region: the bytes COUNT attributed to fired events.  A bitbuffer that is
                        // cleared or never fired is built past `off` too, but must not leak into the next lane's
                        // region, so every store is bounded by this.
    uint32_t off;       // bytes of finished events
    uint32_t pkg;
    uint32_t dev_ord;   // device | ordinal << 16 (the third word of a record, as it is written)
    // bitbuffer under construction
    uint32_t num_rows, free_row;
    uint32_t row0_bits; // bits_per_row[0]
    uint32_t cur_bits;  // bits_per_row[num_rows-1]
    uint32_t cur_syncs;
    uint32_t extent;    // most bits the current row ever held
    uint32_t written;   // dwords of the current row already stored
    uint32_t acc;       // bits of the dword being filled, MSB first
    uint32_t max_bits;  // the longest CLOSED row of the bitbuffer under construction (the pre-filter's short-rows verdicts)
    uint32_t w0;        // the first dword of the current row once it is full (the pre-filter's search rule looks at rows of < 64 bits)
    uint32_t row_hdr;   // offset of the current row's header (relative to out)
    uint32_t wr;        // write cursor (relative to out)
    // decoder pre-filter (r433_batch_probe_prefilter): the decoder's table, and what it refused so far
    uint8_t const *pf;  // kPfTable verdicts, or nullptr
    uint64_t pf_dropped; // ... per code 0..4 in fields of 12 bits (a package has at most 1200 pulses, so at most 1200 bitbuffers per
                         // decoder; one value, not an array: a dynamically indexed array would live in scratch memory)

    __device__ __forceinline__ void put32(uint32_t at, uint32_t v)
    {
        if (WRITE && at + 4u <= limit)
            *(uint32_t *)(out + at) = v;
    }

    __device__ __forceinline__ void begin(uint8_t *o, uint32_t limit_, uint32_t pkg_, uint32_t dev_)
    {
        out = o;
        limit = limit_;
        off = 0;
        pkg = pkg_;
        dev_ord = dev_ & 0xffffu;
        pf = nullptr;
        pf_dropped = 0;
        clear();
    }

    // bitbuffer_clear, src/bitbuffer.c:17-20
    __device__ __forceinline__ void clear()
    {
        num_rows = free_row = 0;
        row0_bits = cur_bits = cur_syncs = max_bits = 0;
        extent = written = acc = w0 = 0;
        wr = off + (uint32_t)sizeof(r433_evt_rec);
        row_hdr = wr;
    }

    __device__ __forceinline__ void open_row()
    {
        row_hdr = wr;
        wr += (uint32_t)sizeof(r433_row_rec);
        cur_bits = cur_syncs = 0;
        extent = written = acc = w0 = 0;
    }

    __device__ __forceinline__ void touch()
    {
        if (num_rows == 0) {
            num_rows = free_row = 1;
            open_row();
        }
    }

    __device__ __forceinline__ void store_word(uint32_t k, uint32_t bits_be)
    {
        if (WRITE) {
            uint32_t at = row_hdr + (uint32_t)sizeof(r433_row_rec) + 4u * k;
            if (at + 4u <= limit) {
                uint32_t *p = (uint32_t *)(out + at);
                uint32_t v = __builtin_bswap32(bits_be);
                if (k < written)
                    v |= *p; // only after the 50-row overflow reset: OR onto what the row already holds
                *p = v;
            }
        }
    }

    // finish the current row: flush the partial word, write its header, advance past its bytes
    __device__ __forceinline__ void close_row()
    {
        uint32_t k = cur_bits >> 5;
        if (cur_bits & 31u)
            store_word(k, acc);
        max_bits = max(max_bits, cur_bits);
        uint32_t nbytes = (extent + 7u) >> 3;
        put32(row_hdr, (cur_bits & 0xffffu) | (cur_syncs << 16));
        put32(row_hdr + 4, nbytes & 0xffffu);
        wr = row_hdr + (uint32_t)sizeof(r433_row_rec) + ((nbytes + 3u) & ~3u);
    }

    __device__ __forceinline__ void add_bit(int bit)
    {
        touch();
        if (cur_bits == 65535u)
            return;
        if (cur_bits > 0 && (cur_bits & 1023u) == 0) {
            if (free_row < R433_BB_ROWS)
                free_row++;
            else
                return;
        }
        acc |= (uint32_t)bit << (31u - (cur_bits & 31u));
        cur_bits += 1;
        if (num_rows == 1)
            row0_bits = cur_bits;
        if (cur_bits > extent)
            extent = cur_bits;
        if ((cur_bits & 31u) == 0) {
            uint32_t k = (cur_bits >> 5) - 1;
            store_word(k, acc);
            if (k >= written)
                written = k + 1;
            if (k == 0)
                w0 = acc;
            acc = 0;
        }
    }

    // n calls of add_bit(bit), a word at a time: the long constant runs of the PCM-type slicers
    // (a 500 us pulse against a 5 us bit period is 100 bits) otherwise serialise a whole wavefront
    // behind one lane.  Stops exactly where add_bit would start refusing bits.
    __device__ __forceinline__ void add_run(int bit, int n)
    {
        if (n <= 0)
            return;
        touch();
        while (n > 0) {
            if (cur_bits == 65535u)
                return;
            if (cur_bits > 0 && (cur_bits & 1023u) == 0) {
                if (free_row < R433_BB_ROWS)
                    free_row++;
                else
                    return;
            }
            uint32_t const room = 32u - (cur_bits & 31u);
            uint32_t take = min(min((uint32_t)n, room), 65535u - cur_bits);
            if (bit) {
                uint32_t const ones = take == 32u ? 0xffffffffu : ((1u << take) - 1u);
                acc |= ones << (room - take);
            }
            cur_bits += take;
            n -= (int)take;
            if (num_rows == 1)
                row0_bits = cur_bits;
            if (cur_bits > extent)
                extent = cur_bits;
            if ((cur_bits & 31u) == 0) {
                uint32_t k = (cur_bits >> 5) - 1;
                store_word(k, acc);
                if (k >= written)
                    written = k + 1;
                if (k == 0)
                    w0 = acc;
                acc = 0;
            }
        }
    }

    __device__ __forceinline__ void add_row()
    {
        touch();
        if (free_row < R433_BB_ROWS) {
            close_row();
            uint32_t skipped = free_row - num_rows; // rows the spill ran through: logical rows of length 0
            for (uint32_t i = 0; i < skipped; ++i) {
                put32(wr, 0);
                put32(wr + 4, 0);
                wr += (uint32_t)sizeof(r433_row_rec);
            }
            free_row++;
            num_rows = free_row;
            open_row();
        }
        else {
            // no room: the reference zeroes the length of the last row and keeps its bytes
            uint32_t k = cur_bits >> 5;
            if (cur_bits & 31u) {
                store_word(k, acc);
                if (k >= written)
                    written = k + 1;
            }
            cur_bits = 0;
            if (num_rows == 1)
                row0_bits = 0;
            acc = 0;
        }
    }

    __device__ __forceinline__ void add_sync()
    {
        touch();
        if (cur_bits)
            add_row();
        cur_syncs++;
    }

    __device__ __forceinline__ uint32_t last_row_bits() const { return cur_bits; }

    // account_event + bitbuffer_clear, src/pulse_slicer.c:26-66
    __device__ __forceinline__ void fire()
    {
        // A bitbuffer its decoder provably refuses on num_rows / free_row / bits_per_row[0] alone never becomes a record:
        // the next one is built over it (same offset, next ordinal), the refusal is counted under the code the decoder
        // would have returned.  Every pass of the slicer takes the same decisions, so sizes and offsets agree.
        if (pf && num_rows < kPfHeadRows && free_row == num_rows && row0_bits < kPfBits) {
            uint32_t verdict = pf[num_rows * kPfBits + row0_bits];
            if (verdict != kPfKeep && (verdict & kPfTiny)) // asked content by content for plain one-row bitbuffers only
                verdict = (num_rows == 1 && cur_syncs == 0 && extent == cur_bits) ? (verdict & ~kPfTiny) : kPfKeep;
            if (verdict != kPfKeep && (verdict & kPfShort)) // holds where no row reaches the decoder's min_bits (kPfShortAt)
                verdict = max(max_bits, cur_bits) < (uint32_t) * (uint16_t const *)(pf + kPfShortAt) ? (verdict & ~kPfShort) : kPfKeep;
            if (verdict == kPfKeep && num_rows == 2 && row0_bits < kPfTwo0 && cur_bits < kPfTwo1) // both rows' lengths (kPfTwoAt)
                verdict = pf[kPfTwoAt + row0_bits * kPfTwo1 + cur_bits];
            // The decoder's search rule (kPfRule): its first act on a one-row bitbuffer of this length is bitbuffer_search(row 0,
            // start, pattern) and "not found" makes it refuse without a look at anything else.  A row of fewer than 64 bits that
            // never shrank is still in w0 / acc: look for the pattern (src/bitbuffer.c:228-253 finds the first place from `start`
            // on where all pattern bits lie inside the row and match).
            if (verdict == kPfKeep && num_rows == 1 && row0_bits < kPfRuleMaxBits && extent == cur_bits && pf[kPfRule] != kPfKeep
                    && ((*(uint64_t const *)(pf + kPfRule + 8) >> row0_bits) & 1ull)) {
                uint32_t const plen = pf[kPfRule + 1], start = *(uint16_t const *)(pf + kPfRule + 2);
                uint32_t const pat = *(uint32_t const *)(pf + kPfRule + 4) >> (32u - plen);
                uint64_t const row = cur_bits < 32u ? (uint64_t)acc << 32 : ((uint64_t)w0 << 32) | acc; // bit i of the row = bit 63 - i
                bool found = false;
                for (uint32_t at = start; at + plen <= cur_bits && !found; ++at)
                    found = (uint32_t)((row << at) >> (64u - plen)) == pat;
                if (!found)
                    verdict = pf[kPfRule];
            }
            if (verdict != kPfKeep && pf[kPfStub] == 1u) { // a later priority level: the refusal goes along as a stub
                put32(off, (uint32_t)sizeof(r433_evt_rec));
                put32(off + 4, pkg);
                put32(off + 8, dev_ord);
                put32(off + 12, kPfStubRows | ((verdict <= 4u ? verdict : 0u) << 16)); // (the same mapping as the drop below: codes are 0..4)
                off += (uint32_t)sizeof(r433_evt_rec);
                dev_ord += 0x10000u;
                clear();
                return;
            }
            if (verdict != kPfKeep) {
                pf_dropped += 1ull << (12u * (verdict <= 4u ? verdict : 0u));
                dev_ord += 0x10000u;
                clear();
                return;
            }
        }
        if (num_rows > 0)
            close_row();
        put32(off, wr - off);
        put32(off + 4, pkg);
        put32(off + 8, dev_ord);
        put32(off + 12, (num_rows & 0xffffu) | (free_row << 16));
        off = wr;
        dev_ord += 0x10000u;
        clear();
    }
};

__device__ __forceinline__ bool within(int v, int centre, int tol)
{
    return v >= centre - tol && v <= centre + tol;
}

// (int)(x * f + 0.5f) with float product and float sum, src/pulse_slicer.c:218,221
__device__ __forceinline__ int round_f(int x, float f)
{
    return (int)__fadd_rn(__fmul_rn((float)x, f), 0.5f);
}

// (int)(x * f + 0.5): float product widened to double for the sum, src/pulse_slicer.c:163-164,629
__device__ __forceinline__ int round_d(int x, float f)
{
    return (int)((double)__fmul_rn((float)x, f) + 0.5);
}

// src/pulse_slicer.c:68-259
template <bool W> __device__ __forceinline__ void slice_pcm(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    if (t.s_long <= 0)
        return; // the reference divides by s_long below
    float f_sh = t.f_short, f_lo = t.f_long;
    int const gap_limit = t.s_gap ? t.s_gap : t.s_reset;
    int const max_zeros = gap_limit / t.s_long;
    int const tol = t.s_tol > 0 ? t.s_tol : t.s_long / 4;
    bool const rz = t.s_short != t.s_long;
    uint32_t const np = p.num;

    int need = rz ? 4 : 12;
    int preamble = 0;
    if (rz) {
        for (uint32_t n = 0; n < np; ++n) {
            int sw = 0, lw = 0, cnt = 0;
            while (n < np && within(p.pulse(n), t.s_short, tol) && within(p.pulse(n) + p.gap(n), t.s_long, tol)) {
                sw += p.pulse(n);
                lw += p.pulse(n) + p.gap(n);
                cnt++;
                n++;
            }
            if (cnt >= need) {
                f_lo = __fdiv_rn((float)cnt, (float)lw);
                f_sh = __fdiv_rn((float)cnt, (float)sw);
                need = cnt;
                preamble = cnt;
            }
        }
        if (preamble == 0) {
            int sw = 0, lw = 0, cnt = 0;
            for (uint32_t n = 0; n < np; ++n) {
                if (within(p.pulse(n), t.s_short, tol) && within(p.pulse(n) + p.gap(n), t.s_long, tol)) {
                    sw += p.pulse(n);
                    lw += p.pulse(n) + p.gap(n);
                    cnt++;
                }
            }
            if (cnt > 8) {
                f_lo = __fdiv_rn((float)cnt, (float)lw);
                f_sh = __fdiv_rn((float)cnt, (float)sw);
            }
        }
    }
    else {
        for (uint32_t n = 0; n < np; ++n) {
            int w = 0, cnt = 0;
            while (n < np && round_d(p.pulse(n), f_sh) == 1 && round_d(p.gap(n), f_lo) == 1) {
                w += p.pulse(n) + p.gap(n);
                cnt += 2;
                n++;
            }
            if (cnt >= need) {
                f_sh = f_lo = __fdiv_rn((float)cnt, (float)w);
                need = cnt;
                preamble = cnt;
            }
        }
        if (preamble == 0) {
            int w = 0, cnt = 0;
            for (uint32_t n = 0; n < np; ++n) {
                int pu = p.pulse(n), ga = p.gap(n);
                if (within(pu, t.s_short, tol)) {
                    w += pu;
                    cnt += 1;
                }
                if (within(pu, 2 * t.s_short, tol)) {
                    w += pu;
                    cnt += 2;
                }
                if (within(ga, t.s_long, tol)) {
                    w += ga;
                    cnt += 1;
                }
                if (within(ga, 2 * t.s_long, tol)) {
                    w += ga;
                    cnt += 2;
                }
            }
            if (cnt > 20)
                f_sh = f_lo = __fdiv_rn((float)cnt, (float)w);
        }
    }

    for (uint32_t n = 0; n < np; ++n) {
        int pu = p.pulse(n), ga = p.gap(n);
        int highs = round_f(pu, f_sh);
        int lows = round_f(ga + t.s_short - t.s_long, f_lo);
        s.add_run(1, highs);
        lows = min(lows, max_zeros);
        s.add_run(0, lows);
        if (rz && abs(pu - t.s_short) > tol)
            s.clear();
        else if (ga > gap_limit && ga <= t.s_reset)
            s.add_row();
        if ((n == np - 1 || ga > t.s_reset) && (s.row0_bits > 0 || s.num_rows > 1))
            s.fire();
    }
}

// src/pulse_slicer.c:261-337
template <bool W> __device__ __forceinline__ void slice_ppm(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    int z_lo, z_hi, o_lo, o_hi, y_lo = 0, y_hi = 0;
    if (t.s_tol > 0) {
        z_lo = t.s_short - t.s_tol;
        z_hi = t.s_short + t.s_tol;
        o_lo = t.s_long - t.s_tol;
        o_hi = t.s_long + t.s_tol;
        if (t.s_sync > 0) {
            y_lo = t.s_sync - t.s_tol;
            y_hi = t.s_sync + t.s_tol;
        }
    }
    else {
        z_lo = 0;
        z_hi = (t.s_short + t.s_long) / 2 + 1;
        o_lo = z_hi - 1;
        o_hi = t.s_gap ? t.s_gap : t.s_reset;
    }
    // (One add_bit and one add_row per symbol, whatever the symbol is: the lanes of a wavefront are 64 decoders with different
    // timings, most symbols are a one to some of them and a zero to others, and a wavefront executes every inlined copy that any
    // lane takes -- with a copy per branch of the reference's if-chain, the bit writer ran twice and the row writer three times
    // per symbol.)
    for (uint32_t n = 0; n < p.num; ++n) {
        int g = p.gap(n);
        int const bit = (g > z_lo && g < z_hi) ? 0 : (g > o_lo && g < o_hi) ? 1 : -1;
        bool const sync = bit < 0 && g > y_lo && g < y_hi;
        bool const row = bit < 0 && !sync && g < t.s_reset;
        if (bit >= 0)
            s.add_bit(bit);
        if (sync)
            s.touch();
        if (row || (sync && s.cur_bits)) // add_sync (src/bitbuffer.c:122-133): a row first if the current one holds bits
            s.add_row();
        if (sync)
            s.cur_syncs++;
        if ((n == p.num - 1 || g >= t.s_reset) && (s.row0_bits > 0 || s.num_rows > 1))
            s.fire();
    }
}

// src/pulse_slicer.c:339-449
template <bool W> __device__ __forceinline__ void slice_pwm(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    int const big = 2147483647;
    int o_lo, o_hi, z_lo, z_hi, y_lo = 0, y_hi = 0;
    if (t.s_tol > 0) {
        o_lo = t.s_short - t.s_tol;
        o_hi = t.s_short + t.s_tol;
        z_lo = t.s_long - t.s_tol;
        z_hi = t.s_long + t.s_tol;
        if (t.s_sync > 0) {
            y_lo = t.s_sync - t.s_tol;
            y_hi = t.s_sync + t.s_tol;
        }
    }
    else if (t.s_sync <= 0) {
        o_lo = 0;
        o_hi = (t.s_short + t.s_long) / 2 + 1;
        z_lo = o_hi - 1;
        z_hi = big;
    }
    else if (t.s_sync < t.s_short) {
        y_lo = 0;
        y_hi = (t.s_sync + t.s_short) / 2 + 1;
        o_lo = y_hi - 1;
        o_hi = (t.s_short + t.s_long) / 2 + 1;
        z_lo = o_hi - 1;
        z_hi = big;
    }
    else if (t.s_sync < t.s_long) {
        o_lo = 0;
        o_hi = (t.s_short + t.s_sync) / 2 + 1;
        y_lo = o_hi - 1;
        y_hi = (t.s_sync + t.s_long) / 2 + 1;
        z_lo = y_hi - 1;
        z_hi = big;
    }
    else {
        o_lo = 0;
        o_hi = (t.s_short + t.s_long) / 2 + 1;
        z_lo = o_hi - 1;
        z_hi = (t.s_long + t.s_sync) / 2 + 1;
        y_lo = z_hi - 1;
        y_hi = big;
    }
    for (uint32_t n = 0; n < p.num; ++n) {
        int w = p.pulse(n), g = p.gap(n);
        // (one add_bit and one add_row for the symbol: see slice_ppm)
        int const bit = (w > o_lo && w < o_hi) ? 1 : (w > z_lo && w < z_hi) ? 0 : -1;
        bool const sync = bit < 0 && w > y_lo && w < y_hi;
        bool const row = bit < 0 && !sync && w > o_lo;
        if (bit >= 0)
            s.add_bit(bit);
        if (sync)
            s.touch();
        if (row || (sync && s.cur_bits))
            s.add_row();
        if (sync)
            s.cur_syncs++;
        if ((n == p.num - 1 || g > t.s_reset) && s.num_rows > 0)
            s.fire();
        else if (t.s_gap > 0 && g > t.s_gap && s.num_rows > 0 && s.last_row_bits() > 0)
            s.add_row();
    }
}

// src/pulse_slicer.c:451-527; "x > s_short * 1.5" (double) is "2x > 3 s_short" in integers
template <bool W> __device__ __forceinline__ void slice_mc(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    int since = 0;
    int const sh3 = 3 * t.s_short;
    s.add_bit(0);
    for (uint32_t n = 0; n < p.num; ++n) {
        int w = p.pulse(n), g = p.gap(n);
        if (t.s_tol > 0
                && (w < t.s_short - t.s_tol || w > t.s_short * 2 + t.s_tol || g < t.s_short - t.s_tol
                        || g > t.s_short * 2 + t.s_tol)) {
            if (2 * w > sh3 && w <= t.s_short * 2 + t.s_tol)
                s.add_bit(1);
            s.add_row();
            s.add_bit(0);
            since = 0;
        }
        else if (2 * (w + since) > sh3) {
            s.add_bit(1);
            since = 0;
        }
        else {
            since += w;
        }
        if ((n == p.num - 1 || g > t.s_reset) && s.num_rows > 0) {
            s.fire();
            s.add_bit(0);
            since = 0;
        }
        else if (2 * (g + since) > sh3) {
            s.add_bit(0);
            since = 0;
        }
        else {
            since += g;
        }
    }
}

// src/pulse_slicer.c:537-595
template <bool W> __device__ __forceinline__ void slice_dmc(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    uint32_t const ns = p.num * 2;
    for (uint32_t k = 0; k < ns; ++k) {
        int sym = p.symbol(k);
        if (abs(sym - t.s_short) < t.s_tol) {
            s.add_bit(1);
            sym = k + 1 < ns ? p.symbol(++k) : 0;
            if (abs(sym - t.s_short) > t.s_tol) {
                if (sym >= t.s_reset - t.s_tol)
                    k--;
                else if (s.num_rows > 0 && s.last_row_bits() > 0)
                    s.add_row();
            }
        }
        else if (abs(sym - t.s_long) < t.s_tol) {
            s.add_bit(0);
        }
        else if (sym >= t.s_reset - t.s_tol && s.num_rows > 0) {
            s.fire();
        }
    }
}

// src/pulse_slicer.c:597-657
template <bool W> __device__ __forceinline__ void slice_piwm_raw(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    uint32_t const ns = p.num * 2;
    for (uint32_t k = 0; k < ns; ++k) {
        int sym = p.symbol(k);
        int w = round_d(sym, t.f_short);
        if (sym > t.s_long) {
            s.add_row();
        }
        else if (abs(sym - w * t.s_short) < t.s_tol) {
            s.add_run(1 - (int)(k & 1), w);
        }
        else if (sym < t.s_reset && s.num_rows > 0 && s.last_row_bits() > 0) {
            s.add_row();
        }
        if ((k == ns - 1 || sym > t.s_reset) && s.num_rows > 0)
            s.fire();
    }
}

// src/pulse_slicer.c:659-713
template <bool W> __device__ __forceinline__ void slice_piwm_dc(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    uint32_t const ns = p.num * 2;
    for (uint32_t k = 0; k < ns; ++k) {
        int sym = p.symbol(k);
        if (abs(sym - t.s_short) < t.s_tol)
            s.add_bit(1);
        else if (abs(sym - t.s_long) < t.s_tol)
            s.add_bit(0);
        else if (sym < t.s_reset && s.num_rows > 0 && s.last_row_bits() > 0)
            s.add_row();
        if ((k == ns - 1 || sym > t.s_reset) && s.num_rows > 0)
            s.fire();
    }
}

// src/pulse_slicer.c:715-759
template <bool W> __device__ __forceinline__ void slice_nrzs(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    int const lim = t.s_short;
    for (uint32_t n = 0; n < p.num; ++n) {
        int w = p.pulse(n);
        if (w > lim) {
            if (lim <= 0)
                return; // the reference would divide by zero; no registered device has short_width == 0
            int ones = w / lim;
            s.add_run(1, ones);
            s.add_bit(0);
        }
        else if (w < lim) {
            s.add_bit(0);
        }
        if (n == p.num - 1 || p.gap(n) >= t.s_reset)
            s.fire();
    }
}

// src/pulse_slicer.c:775-864
template <bool W> __device__ __forceinline__ void slice_osv1(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    int const half_min = t.s_short / 2;
    int const half_max = t.s_short * 3 / 2;
    int const sync_min = 2 * half_max;
    uint32_t n;
    int pre = 0, man = 0;
    for (n = 0; n < p.num; ++n) {
        if (p.pulse(n) > half_min && p.gap(n) > half_min) {
            pre++;
            if (p.gap(n) > half_max)
                break;
        }
        else
            return;
    }
    if (pre != 12)
        return;
    ++n;
    int sp = n < p.num ? p.pulse(n) : 0; // past the end the reference reads the cleared tail of the arrays
    int sg = n < p.num ? p.gap(n) : 0;
    if (sp < sync_min || sg < sync_min)
        return;
    if (sg > sp) {
        man ^= 1;
        if (man)
            s.add_bit(0);
    }
    for (n++; n < p.num; ++n) {
        man ^= 1;
        if (man)
            s.add_bit(1);
        if (p.pulse(n) > half_max) {
            man ^= 1;
            if (man)
                s.add_bit(1);
        }
        if ((n == p.num - 1 || p.gap(n) > t.s_reset) && s.num_rows > 0) {
            s.fire();
            return;
        }
        man ^= 1;
        if (man)
            s.add_bit(0);
        if (p.gap(n) > half_max) {
            man ^= 1;
            if (man)
                s.add_bit(0);
        }
    }
}

// src/pulse_slicer.c:866-918
template <bool W> __device__ __forceinline__ void slice_rzi(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    if (t.s_long <= 0)
        return; // the reference would divide by zero
    int const base = t.s_long - t.s_short;
    bool fresh = true;
    for (uint32_t n = 0; n < p.num; ++n) {
        int w = p.pulse(n);
        int ones = fresh ? (w + t.s_long / 2) / t.s_long : (w - base + t.s_long / 2) / t.s_long;
        fresh = false;
        s.add_run(1, ones);
        if (p.gap(n) > t.s_reset || n == p.num - 1) {
            if (s.row0_bits > 0)
                s.fire();
            s.clear();
            fresh = true;
            continue;
        }
        s.add_bit(0);
    }
}

// one arm of the switch in src/r_api.c:456-497 / :520-547
template <bool W> __device__ __forceinline__ void slice_dispatch(PulseView const &p, DevRow const &t, BitSink<W> &s)
{
    switch (t.modulation) {
    case 4:
    case 16:
        slice_pcm(p, t, s);
        break;
    case 5:
        slice_ppm(p, t, s);
        break;
    case 6:
    case 17:
        slice_pwm(p, t, s);
        break;
    case 3:
    case 18:
        slice_mc(p, t, s);
        break;
    case 8:
        slice_piwm_raw(p, t, s);
        break;
    case 11:
        slice_piwm_dc(p, t, s);
        break;
    case 9:
        slice_dmc(p, t, s);
        break;
    case 10:
        slice_osv1(p, t, s);
        break;
    case 12:
        slice_nrzs(p, t, s);
        break;
    case 13:
        slice_rzi(p, t, s);
        break;
    default:
        break;
    }
}

} // namespace r433
