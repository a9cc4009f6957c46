// detect_device.hpp -- OOK level-tracking pulse detector with the embedded FSK sub-detectors:
// the exact, general per-sample step.  One capture per wavefront, executed WAVE-UNIFORMLY: every
// lane carries the same detector state and takes the same (scalar) branches; only the lane flagged
// `writer` touches memory (package arena, FSK candidate ring) and what it reads back from the ring
// is broadcast.  The kernel (stream_kernels.hip) uses the whole wavefront to skip over samples that
// provably cannot change the state machine and comes here for everything else.
//
// Behaviour follows the reference's pulse_detect_package() (src/pulse_detect.c:199-483),
// pulse_detect_fsk_classic/minmax/wrap_up (src/pulse_detect_fsk.c:34-221) and pulse_data_shift
// (src/pulse_data.c:27-34) exactly, including the per-call quirks: the high estimate is re-clamped
// and the start age re-applied at every call entry, a sample that ends a package is examined again
// in the idle state, "eop on spurious pulse" is forgotten at call boundaries.
//
// Output: r433_pkg_rec records (include/r433_records.h) appended to a per-capture arena in HBM.
// The open OOK package grows in place at the arena cursor; the FSK candidate lives in a
// per-capture scratch ring because the reference keeps it alive next to the OOK package.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "r433_records.h"

namespace r433 {

struct DetCfg {
    int fixed_high; // 0 = adaptive threshold
    int min_high;
    int ratio;
    int max_high; // OOK_MAX_HIGH_LEVEL
    int per_ms;   // samples per millisecond (integer division, src/pulse_detect.c:281)
    uint32_t rate;
    int fpdm;     // 0 classic, 1 minmax
};

enum { ST_IDLE = 0, ST_PULSE = 1, ST_GAP_START = 2, ST_GAP = 3 };

struct DetLane {
    // level tracker / state machine (struct pulse_detect, src/pulse_detect.c:30-54)
    int state, run, max_pulse, lead_in, low, high;
    // FSK sub-detector (pulse_detect_fsk_t)
    uint32_t f_run;
    int f_state, f_f1, f_f2, f_vmax, f_vmin, f_skip;
    // open packages
    uint32_t ook_num;   // completed (pulse, gap) pairs of the OOK package
    int cur_pulse;      // pulses->pulse[num_pulses], the slot being filled
    int ook_f1;         // pulses->fsk_f1_est (carrier estimate during pulses)
    uint32_t fsk_num;   // fsk_pulses->num_pulses
    uint64_t offset;    // pulses->offset
    uint64_t fsk_offset;
    uint32_t start_ago; // identical for both packages at all times
    int eop_spurious;
    // arena
    uint8_t *arena;
    int2 *fsk_ring;     // per-capture scratch, R433_PD_MAX_PULSES entries
    bool writer;        // the one lane that performs the memory accesses
    uint32_t arena_cap;
    uint32_t cursor;    // bytes of finished records
    uint32_t ook_base;  // cursor when the current (or last) package began: its OOK pairs sit at arena + ook_base + 64
    uint32_t n_pkgs;
    uint32_t overflow;
    uint32_t stream;
};

__device__ __forceinline__ void fsk_reset(DetLane &d)
{
    d.f_run = 0;
    d.f_state = 0;
    d.f_f1 = 0;
    d.f_f2 = 0;
    d.f_vmax = -32768;
    d.f_vmin = 32767;
    d.f_skip = 40;
}

__device__ __forceinline__ void det_reset(DetLane &d)
{
    d.state = ST_IDLE;
    d.run = 0;
    d.max_pulse = 0;
    d.lead_in = 0;
    d.low = 0;
    d.high = 0;
    fsk_reset(d);
    d.ook_num = 0;
    d.cur_pulse = 0;
    d.ook_f1 = 0;
    d.fsk_num = 0;
    d.offset = 0;
    d.fsk_offset = 0;
    d.start_ago = 0;
    d.eop_spurious = 0;
}

// ---- FSK candidate ring: element k is pulse (k even) / gap (k odd) of pair k/2 ----

__device__ __forceinline__ int ring_get(DetLane const &d, uint32_t k)
{
    int v = 0;
    if (d.writer)
        v = ((int const *)d.fsk_ring)[k];
    return __builtin_amdgcn_readfirstlane(v); // the writer is lane 0
}

__device__ __forceinline__ void ring_set(DetLane const &d, uint32_t k, int v)
{
    if (d.writer)
        ((int *)d.fsk_ring)[k] = v;
}

// ---- arena helpers ----

__device__ __forceinline__ bool arena_room(DetLane &d, uint32_t pairs_after)
{
    if (d.cursor + (uint32_t)sizeof(r433_pkg_rec) + 8u * pairs_after > d.arena_cap) {
        d.overflow = 1;
        return false;
    }
    return true;
}

__device__ __forceinline__ void ook_push_pair(DetLane &d, int gap)
{
    if (arena_room(d, d.ook_num + 1) && d.writer) {
        int2 *pairs = (int2 *)(d.arena + d.cursor + sizeof(r433_pkg_rec));
        pairs[d.ook_num] = make_int2(d.cur_pulse, gap);
    }
    d.ook_num += 1;
    d.cur_pulse = 0;
}

__device__ __forceinline__ void write_header(DetLane &d, uint32_t type, uint32_t num, uint64_t offset, uint32_t end_ago,
        int f1, int f2, DetCfg const &c, uint32_t frame, uint32_t ret_pos)
{
    if (!arena_room(d, num))
        return;
    uint32_t *h = (uint32_t *)(d.arena + d.cursor);
    uint32_t total = (uint32_t)sizeof(r433_pkg_rec) + 8u * num;
    if (d.writer) {
        h[0] = total;
        h[1] = d.stream;
        h[2] = type;
        h[3] = num;
        h[4] = frame;
        h[5] = ret_pos;
        h[6] = (uint32_t)offset;
        h[7] = (uint32_t)(offset >> 32);
        h[8] = d.start_ago;
        h[9] = end_ago;
        h[10] = (uint32_t)d.low;
        h[11] = (uint32_t)d.high;
        h[12] = (uint32_t)f1;
        h[13] = (uint32_t)f2;
        h[14] = c.rate;
        h[15] = 0;
    }
    d.cursor += total;
    d.n_pkgs += 1;
}

// src/pulse_detect.c:264-272 / 431-439 / 451-468
__device__ __forceinline__ int emit_ook(DetLane &d, DetCfg const &c, int len, int pos, uint32_t frame, uint32_t ret_pos)
{
    d.state = ST_IDLE;
    write_header(d, R433_PKG_OOK, d.ook_num, d.offset, (uint32_t)(len - pos), d.ook_f1, 0, c, frame, ret_pos);
    return R433_PKG_OOK;
}

__device__ __forceinline__ void move_pairs(int2 *dst, int2 const *src, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i)
        dst[i] = src[i];
}

// src/pulse_data.c:27-34
__device__ __forceinline__ void fsk_drop_half(DetLane &d)
{
    if (d.writer)
        move_pairs(d.fsk_ring, d.fsk_ring + R433_PD_MAX_PULSES / 2, R433_PD_MAX_PULSES / 2);
    d.fsk_num -= R433_PD_MAX_PULSES / 2;
    d.fsk_offset += R433_PD_MAX_PULSES / 2;
}

// src/pulse_detect.c:239-253 / 387-410 with pulse_detect_fsk_wrap_up (src/pulse_detect_fsk.c:143-156)
__device__ __forceinline__ int emit_fsk(DetLane &d, DetCfg const &c, int len, int pos, uint32_t frame, uint32_t ret_pos)
{
    if (c.fpdm == 0 && d.fsk_num < R433_PD_MAX_PULSES) {
        d.f_run += 1;
        if (d.f_state == 1) {
            ring_set(d, 2 * d.fsk_num, (int)d.f_run);
            ring_set(d, 2 * d.fsk_num + 1, 0);
        }
        else {
            ring_set(d, 2 * d.fsk_num + 1, (int)d.f_run);
        }
        d.fsk_num += 1;
    }
    d.state = ST_IDLE;
    if (arena_room(d, d.fsk_num) && d.writer)
        move_pairs((int2 *)(d.arena + d.cursor + sizeof(r433_pkg_rec)), d.fsk_ring, d.fsk_num);
    write_header(d, R433_PKG_FSK, d.fsk_num, d.fsk_offset, (uint32_t)(len - pos), d.f_f1, d.f_f2, c, frame, ret_pos);
    return R433_PKG_FSK;
}

// ---- FSK sub-detectors ----

// src/pulse_detect_fsk.c:34-141
__device__ __forceinline__ void fsk_classic(DetLane &d, int v)
{
    int d1 = abs(v - d.f_f1);
    int d2 = abs(v - d.f_f2);
    d.f_run += 1;
    if (d.f_state == 0) {
        if (d.f_run < 10u) {
            d.f_f1 = d.f_f1 / 2 + v / 2;
        }
        else if (d1 > 3000) {
            if (v > d.f_f1) {
                d.f_state = 1;
                d.f_f2 = d.f_f1;
                d.f_f1 = v;
                ring_set(d, 0, 0);
                ring_set(d, 1, (int)d.f_run);
                d.fsk_num += 1;
                d.f_run = 0;
            }
            else {
                d.f_state = 2;
                d.f_f2 = v;
                ring_set(d, 0, (int)d.f_run);
                d.f_run = 0;
            }
        }
        else {
            d.f_f1 += v / 16 - d.f_f1 / 16;
        }
    }
    else if (d.f_state == 1) {
        if (d1 > d2) {
            d.f_state = 2;
            if (d.f_run >= 10u) {
                ring_set(d, 2 * d.fsk_num, (int)d.f_run);
                d.f_run = 0;
            }
            else {
                d.f_run += (uint32_t)ring_get(d, 2 * (d.fsk_num - 1) + 1);
                d.fsk_num -= 1;
                if (d.fsk_num == 0 && ring_get(d, 0) == 0) {
                    d.f_f1 = d.f_f2;
                    d.f_state = 0;
                }
            }
        }
        else if (v > d.f_f1) {
            d.f_f1 += v / 16 - d.f_f1 / 16;
        }
        else {
            d.f_f1 += v / 64 - d.f_f1 / 64;
        }
    }
    else if (d.f_state == 2) {
        if (d2 > d1) {
            d.f_state = 1;
            if (d.f_run >= 10u) {
                ring_set(d, 2 * d.fsk_num + 1, (int)d.f_run);
                d.fsk_num += 1;
                d.f_run = 0;
                if (d.fsk_num >= R433_PD_MAX_PULSES)
                    fsk_drop_half(d);
            }
            else {
                d.f_run += (uint32_t)ring_get(d, 2 * d.fsk_num);
                if (d.fsk_num == 0)
                    d.f_state = 0;
            }
        }
        else if (v < d.f_f2) {
            d.f_f2 += v / 16 - d.f_f2 / 16;
        }
        else {
            d.f_f2 += v / 64 - d.f_f2 / 64;
        }
    }
}

// src/pulse_detect_fsk.c:158-221
__device__ __forceinline__ void fsk_minmax(DetLane &d, int v)
{
    if (d.f_skip == 0) {
        d.f_vmax = max(v, d.f_vmax);
        d.f_vmin = min(v, d.f_vmin);
        int mid = (int)(int16_t)((d.f_vmax + d.f_vmin) / 2);
        if (v > mid)
            d.f_vmax = (int)(int16_t)(d.f_vmax - 10);
        if (v < mid)
            d.f_vmin = (int)(int16_t)(d.f_vmin + 10);
        d.f_run += 1;
        if (d.f_state == 0) {
            d.f_state = v > mid ? 1 : 2;
        }
        else if (d.f_state == 1) {
            if (v < mid) {
                d.f_state = 2;
                ring_set(d, 2 * d.fsk_num, (int)d.f_run);
                d.f_run = 0;
            }
            d.f_f2 += v / 64 - d.f_f2 / 64; // (sic) the reference updates f2 while high
        }
        else if (d.f_state == 2) {
            if (v > mid) {
                d.f_state = 1;
                ring_set(d, 2 * d.fsk_num + 1, (int)d.f_run);
                d.fsk_num += 1;
                d.f_run = 0;
                if (d.fsk_num >= R433_PD_MAX_PULSES)
                    fsk_drop_half(d);
            }
            d.f_f1 += v / 64 - d.f_f1 / 64;
        }
    }
    if (d.f_skip > 0)
        d.f_skip -= 1;
}

__device__ __forceinline__ void fsk_feed(DetLane &d, DetCfg const &c, int v)
{
    if (c.fpdm == 0)
        fsk_classic(d, v);
    else
        fsk_minmax(d, v);
}

// ---- call entry / per-sample step / flush ----

// what every pulse_detect_package() call does before its sample loop, src/pulse_detect.c:283-291
__device__ __forceinline__ void det_call_entry(DetLane &d, DetCfg const &c, int len, int pos)
{
    d.high = max(d.high, c.min_high);
    if (pos == 0)
        d.start_ago += (uint32_t)len;
    d.eop_spurious = 0;
}

// The idle arm of the sample loop (src/pulse_detect.c:308-335): either a pulse starts or the
// noise-floor estimate takes one step.  Also what re-examines a sample after a package return.
__device__ __forceinline__ void det_idle(DetLane &d, DetCfg const &c, int am, int len, int pos, uint64_t input_pos)
{
    int thr = (int)(int16_t)((d.low + min(d.high, c.max_high)) / 2);
    if (c.fixed_high != 0)
        thr = (int)(int16_t)c.fixed_high;
    int const hys = (int)(int16_t)(thr / 8);
    if (am > thr + hys && d.lead_in > 1024) {
        d.ook_num = 0;
        d.ook_base = d.cursor;
        d.cur_pulse = 0;
        d.ook_f1 = 0;
        d.fsk_num = 0;
        d.offset = d.fsk_offset = input_pos + (uint64_t)pos;
        d.start_ago = (uint32_t)(len - pos);
        d.run = 0;
        d.max_pulse = 0;
        fsk_reset(d);
        ring_set(d, 0, 0);
        ring_set(d, 1, 0);
        d.state = ST_PULSE;
    }
    else {
        int dl = am - d.low;
        d.low += dl / 1024;
        d.low += dl > 0 ? 1 : -1;
        d.high = max(c.ratio * d.low, c.min_high);
        if (d.lead_in <= 1024)
            d.lead_in += 1;
    }
}

// One iteration of the sample loop (src/pulse_detect.c:293-476).  Returns 0 if the sample was
// consumed, else the package type; in that case the caller re-enters (det_call_entry) and presents
// the same sample again to det_idle, as the reference does by returning without advancing
// data_counter (the state is always idle after a return).
__device__ __forceinline__ int det_step(DetLane &d, DetCfg const &c, int am, int fm, int len, int pos,
        uint64_t input_pos, uint32_t frame)
{
    if (d.state == ST_IDLE) {
        det_idle(d, c, am, len, pos, input_pos);
        return 0;
    }
    int thr = (int)(int16_t)((d.low + min(d.high, c.max_high)) / 2);
    if (c.fixed_high != 0)
        thr = (int)(int16_t)c.fixed_high;
    int const hys = (int)(int16_t)(thr / 8);
    bool const above = am > thr + hys;
    int ret = 0;
    bool feed = false;

    d.run += 1;
    if (d.state == ST_PULSE) {
        if (am < thr - hys) {
            if (d.run < 10) {
                if (d.ook_num <= 1) {
                    d.state = ST_IDLE;
                }
                else {
                    d.eop_spurious = 1;
                    d.state = ST_GAP;
                }
            }
            else {
                d.cur_pulse = d.run;
                d.max_pulse = max(d.run, d.max_pulse);
                d.run = 0;
                d.state = ST_GAP_START;
            }
        }
        else {
            d.high += am / 64 - d.high / 64;
            d.high = max(d.high, c.min_high);
            d.ook_f1 += fm / 64 - d.ook_f1 / 64;
        }
        feed = d.ook_num == 0;
    }
    else if (d.state == ST_GAP_START) {
        if (above) {
            d.run += d.cur_pulse;
            d.state = ST_PULSE;
        }
        else if (d.run >= 10) {
            d.state = ST_GAP;
            if (d.fsk_num > 16)
                ret = R433_PKG_FSK;
        }
        feed = ret == 0 && d.ook_num == 0;
    }
    else { // ST_GAP
        if (above) {
            ook_push_pair(d, d.run);
            if (d.ook_num >= R433_PD_MAX_PULSES) {
                ret = R433_PKG_OOK;
            }
            else {
                d.run = 0;
                d.state = ST_PULSE;
            }
        }
        if (ret == 0
                && (d.eop_spurious || (d.run > 10 * d.max_pulse && d.run > 10 * c.per_ms) || d.run > 100 * c.per_ms)) {
            ook_push_pair(d, d.run);
            ret = R433_PKG_OOK;
        }
    }
    if (feed)
        fsk_feed(d, c, fm);
    if (ret == R433_PKG_FSK)
        return emit_fsk(d, c, len, pos, frame, (uint32_t)pos);
    if (ret == R433_PKG_OOK)
        return emit_ook(d, c, len, pos, frame, (uint32_t)pos);
    return 0;
}

// the len == 0 call, src/pulse_detect.c:204-278.  Returns the package type or 0.
__device__ __forceinline__ int det_flush(DetLane &d, DetCfg const &c, uint32_t frame)
{
    int st = d.state;
    if (st == ST_PULSE) {
        if (d.run < 10) {
            if (d.ook_num <= 1) {
                d.state = ST_IDLE;
                st = ST_IDLE;
            }
            else {
                st = ST_GAP_START;
            }
        }
        else {
            d.cur_pulse = d.run;
            d.max_pulse = max(d.run, d.max_pulse);
            d.run = 0;
            st = ST_GAP_START;
        }
    }
    if (st == ST_GAP_START) {
        d.state = ST_GAP;
        if (d.fsk_num > 16)
            return emit_fsk(d, c, 0, 0, frame, R433_RET_FLUSH);
        st = ST_GAP;
    }
    if (st == ST_GAP) {
        ook_push_pair(d, d.run);
        return emit_ook(d, c, 0, 0, frame, R433_RET_FLUSH);
    }
    return 0;
}

} // namespace r433
