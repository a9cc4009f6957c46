// r433_internal.hpp -- kernel parameter blocks and launch entry points shared by the
// translation units of librtl433hip.so.  Not part of the public C ABI (include/r433_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "detect_device.hpp"
#include "r433_records.h"

namespace r433 {

// Per-capture carry between launches: everything the reference keeps in filter_state_t,
// demodfm_state_t, struct pulse_detect and the two open pulse_data_t.
struct StreamState {
    int lpf_y, lpf_x;
    int fm_xr, fm_xi, fm_xf, fm_yf;
    int state, run, max_pulse, lead_in, low, high;
    uint32_t f_run;
    int f_state, f_f1, f_f2, f_vmax, f_vmin, f_skip;
    uint32_t ook_num;
    int cur_pulse, ook_f1;
    uint32_t fsk_num;
    uint32_t start_ago;
    uint64_t offset, fsk_offset;
    uint64_t input_pos;
    uint32_t frame;
    uint32_t cursor;
    uint32_t n_pkgs;
    uint32_t overflow;
    // what a segment of a split capture reports for the stitch (SegDesc below)
    int seg_init_low;  // noise floor it assumed at its first sample (its parity variant)
    int seg_init_high; // and the level estimate that goes with an idle detector at that floor
    int seg_fail;      // its start could not be established (filter carry or floor not provable)
    int seg_end_state; // detector state, lead-in counter and noise floor after its last sample
    int seg_end_lead;
    int seg_end_low;
    int seg_end_high;
};

// A capture processed by several wavefronts.  Cuts are speculative: a segment that does not start the
// capture assumes the detector idle at its first sample (lead-in saturated, nothing open), takes the
// filter carries from an all-extremes pass over the tile before it and the noise floor from the last
// 128 samples of that tile walked from both extremes of ONE assumed parity.  Both parities are run;
// the host keeps the variant whose assumed floor equals the floor the previous segment really ended
// with (and requires that segment to have ended idle) -- otherwise the capture is redone in one piece.
struct SegDesc {
    uint32_t capture;  // index of the capture (arena records carry it)
    uint32_t start;    // first sample, a multiple of the tile; > 0 unless SEG_FIRST
    uint32_t end;      // one past the last sample: a multiple of the tile or the capture length
    uint32_t flags;
};
enum : uint32_t {
    SEG_FIRST = 1u,   // starts the capture: reset state, no assumptions
    SEG_LAST = 2u,    // ends the capture: end-of-input flush
    SEG_ODD = 4u,     // assumed parity of the noise floor at `start`
    SEG_PRIMARY = 8u, // the variant that also delivers frame sums and taps
};

enum : uint32_t {
    RUN_NOFLUSH = 2u,  // do not issue the end-of-input flush call
    RUN_ENV_RAW16 = 4u, // filters-only launches: the input is a u16 envelope (2 B/sample), not IQ (baseband_low_pass_filter's x_buf)
    // profiling aids (env R433_DEBUG_FLAGS, never set by the product path): stop after a phase
    RUN_AM_IS_INPUT = 65536u,  // cu8 launches: the capture's 16-bit words ARE the AM samples (am.s16 files, src/r_flow.c:213-217)
    RUN_FM_IS_INPUT = 131072u, // ... the FM samples (fm.s16 files, src/r_flow.c:218-224)
    RUN_DBG_SKIP_DETECT = 256u,
    RUN_DBG_SKIP_FILTERS = 512u,
    RUN_DBG_TIMING = 1024u, // per-phase shader-clock ticks into unused StreamState slots (r433_batch_debug_state)
    RUN_NO_ROLE_SWAP = 8192u, // development: wavefront 0 always produces
    RUN_NO_PRIO = 16384u,     // development: the consumer does not raise its issue priority
    RUN_PAIR = 32768u,        // development: a producer / consumer pair per capture whatever the launch size
    RUN_ONE_WAVE = 4096u, // development: one wavefront per capture does phases A+B and C in turn (A/B timing, same results)
    RUN_NO_TRAIN_ENGINE = 2048u, // development: in-package legs through the older per-leg code (A/B timing, same results)
    RUN_NO_LAZY = 262144u,       // development: every tile filtered, also the ones that provably cannot move the detector (A/B timing, same results)
    RUN_SPLIT_ROLES = 524288u,   // launch_stream: producers and consumers as two launches over StreamParams::tile_store (set by the host where it pays; development: forced / forbidden)
    RUN_RETRY_PASS = 1048576u,   // (set by launch_stream) the run-again launch behind such a pass: pair kernel, every tile filtered, over retry_list
};

// what a filtered tile hands from the producers' launch to the consumers' (stream_kernels.hip): 2048 filtered envelope samples,
// 2048 filtered discriminator samples (int16 each), the extrema of the 64 chunks of the envelope (int16 max, int16 min)
constexpr uint32_t kTileSamples = 2048;
constexpr uint32_t kTileRecBytes = 2 * kTileSamples * 2 + 2 * 64 * 2;

struct StreamParams {
    uint8_t const *iq;            // n_streams captures, stride_bytes apart (16-byte aligned)
    uint64_t stride_bytes;
    uint32_t const *stream_bytes; // per capture, or nullptr: all uniform_bytes
    uint32_t uniform_bytes;
    uint32_t n_streams;
    uint32_t frame_samples;
    uint32_t flags;
    DetCfg det;
    int use_mag;
    int enable_fm;
    int a16, b16;
    long long a32, b32;
    uint8_t *arena;               // n_streams * arena_stride bytes
    uint32_t arena_stride;
    int2 *fsk_ring;               // n_streams * 1200 pairs
    SegDesc const *segs;          // n_streams segments (nullptr: one whole capture per wavefront)
    // Split captures: the workgroups of the launch, heaviest first.  Entry = first slot of the workgroup; bit 31 set = the
    // slot after it is the same piece's odd-parity variant: one producer wavefront then feeds two consumer wavefronts
    // (the filters of a piece are computed once, its samples read once).  nullptr: slot i is workgroup i.
    uint32_t const *wg_slot;
    uint32_t n_wgs;
    StreamState *state;           // n_streams
    uint32_t *frame_sums;         // n_streams * frames_cap (may be null)
    uint32_t frames_cap;
    int const *frame_min_high;    // optional per (capture, frame) override of det.min_high
    // optional taps (parity tests): n_streams * tap_stride samples each, may be null
    uint16_t *tap_env;
    int16_t *tap_am;
    int16_t *tap_fm;
    uint64_t tap_stride;
    // filters-only launches (launch_filters): the carries a frame starts from -- AM y[-1], x[-1]; FM y[-1], discriminator[-1];
    // the last IQ sample, centred -- i.e. filter_state_t / demodfm_state_t of the reference (include/baseband.h:91-107).
    // The carries after the frame's last sample come back in StreamState::lpf_y, lpf_x, fm_yf, fm_xf.
    int const *seam_init;
    // optional `u8` logic dump (-w file.u8, reference src/r_flow.c:236-237,271-272,314-315,364-371, src/pulse_data.c:58-67):
    // one byte per sample, n_streams * logic_stride bytes, zeroed by the host before the launch
    uint8_t *logic;
    uint64_t logic_stride;
    // Producers and consumers as two launches (RUN_SPLIT_ROLES; nullptr: not available, launch_stream keeps them in one
    // workgroup): per (slot, tile) a record of kTileRecBytes and a descriptor word; per slot what the producer has to tell
    // the consumer beside them (a refused filter carry; attempts | reasons << 8), and the list of the slots whose consumer
    // found that the capture cannot be carried across its unfiltered tiles (run again by a third launch, every tile filtered).
    uint8_t *tile_store;          // n_streams * tiles_cap * kTileRecBytes
    int *tile_desc;               // n_streams * tiles_cap
    uint32_t tiles_cap;
    int *tile_over;               // n_streams
    uint32_t *tile_info;          // n_streams
    uint32_t *retry_count;        // device scalar, zeroed by launch_stream
    uint32_t *retry_list;         // n_streams
    uint32_t *cons_weight;        // n_streams: the consumers' launch takes its captures by the work the producers left them (launch_stream)
    uint32_t *cons_order;         // n_streams
    uint32_t *retry_why;          // n_streams
    uint32_t const *wg_count;     // (the run-again launch) workgroups at and beyond *wg_count leave at once
};

// -> true: the pass went out as a launch of producers, a launch of consumers and the run-again launch (RUN_SPLIT_ROLES asked for
// it and nothing -- taps, logic dump, pieces, sample files that are not IQ -- stood against it)
bool launch_stream(StreamParams const &p, uint32_t sample_size, hipStream_t st);
// phases A + B only, one frame, state in and out: the function-level seam of include/baseband.h
void launch_filters(StreamParams const &p, uint32_t sample_size, hipStream_t st);

// ---- slicer fan-out ----

// integer timing of one r_device at one sample rate (host-resolved, reference src/pulse_slicer.c:70-99)
struct DevRow {
    int modulation;
    int s_short, s_long, s_reset, s_gap, s_sync, s_tol;
    float f_short, f_long;
    int valid;    // 0: "sample rate too low" early return, or degenerate timing
    int orig;     // index in registration order; -1 = padding row
    int is_fsk;
    int pf;       // index of the decoder's pre-filter table (SliceParams.pf_tables), -1 = none: every record goes to the host
};

// pre-filter tables: one byte per (num_rows 0..50, bits_per_row[0] 0..kPfBits-1); kPfKeep, or the decode_fn failure code negated
constexpr uint32_t kPfRows = 51, kPfBits = 1024, kPfKeep = 0xff;
// a verdict with this bit holds for one-row bitbuffers without a sync count and nothing written past the row's bits only: it was
// found by asking the decoder every content such a row can have (prefilter.cpp probe_tiny), not under the memory fence
constexpr uint32_t kPfTiny = 0x40;
// a verdict with this bit holds for bitbuffers whose LONGEST row has fewer bits than the decoder's bound ([44..45] of the table's
// first row, kPfShortAt): the decoder's first act is bitbuffer_find_repeated_row / _prefix with that min_bits, and where no row
// can qualify it refuses without a look at anything else (r433_helper_probe.rows_below)
constexpr uint32_t kPfShort = 0x20, kPfShortAt = 44;
constexpr uint32_t kPfTable = kPfRows * kPfBits;
// The table's row for num_rows == 0 only ever uses its first byte (an empty bitbuffer has no row 0): bytes 16..31 of it hold the
// decoder's SEARCH RULE (r433_helper_probe: the decoder's first act on a one-row bitbuffer is bitbuffer_search(row 0, start,
// pattern), and with the answer "not found" it refuses under `code` without a look at anything else) --
//   [16] code (kPfKeep: no rule)  [17] pattern bits (1..32)  [18..19] start  [20..23] the pattern, left-aligned in a u32
//   [24..31] u64: bit n set = the rule holds for one-row bitbuffers of n bits (n < 64)
constexpr uint32_t kPfRule = 16, kPfRuleMaxBits = 64;
// [40] of the same row: the decoder sits on a LATER priority level (src/r_api.c:442-451: called only for packages the levels
// before it decoded nothing of), so whether a refused bitbuffer counts in its statistics is the replay's to know: the slicer
// then leaves a 16-byte STUB in the record's place -- an r433_evt_rec alone with num_rows = kPfStubRows and the refusal's code
// in free_row -- which the replay books like the refusal it stands for, when and if it reaches that level for the package
constexpr uint32_t kPfStub = 40, kPfStubRows = 0xffffu;
// Heads are asked for up to kPfHeadRows - 1 rows; the table's rows behind them are free.  The last one holds the TWO-ROW table:
// verdicts by (bits_per_row[0] < kPfTwo0, bits_per_row[1] < kPfTwo1) for bitbuffers of exactly two rows, learned like the heads
// but with the second row's length on the readable side of the fence (a PWM slicer's "one bit, then a sync pulse opens an empty
// second row" is most of what two-row bitbuffers are)
constexpr uint32_t kPfHeadRows = 25, kPfTwo0 = 128, kPfTwo1 = 8, kPfTwoAt = (kPfRows - 1) * kPfBits;
static_assert(kPfTwo0 * kPfTwo1 <= kPfBits, "the two-row table fits one row of the table");

struct SliceParams {
    uint8_t const *arena;
    uint32_t arena_stride;
    uint32_t const *dir_stream; // per package: capture index
    uint32_t const *dir_off;    // per package: byte offset of its record in that capture's arena
    uint32_t const *n_pkgs;     // device scalar: total packages
    DevRow const *devs;         // n_rows slicer rows in chunks of 64, padded: a chunk holds one kind (OOK / FSK) and one or a few line codes (host_api.cpp)
    uint32_t n_rows;
    uint32_t n_devs;            // registered devices (width of `sizes`)
    uint32_t *sizes;            // [pkg][orig dev] bytes of event records
    uint32_t *dev_off;          // [pkg][orig dev] exclusive prefix of sizes[pkg][.] (k_dev_prefix, before the placing pass)
    uint32_t *pkg_bytes;        // [pkg] total
    uint32_t const *pkg_off;    // [pkg] exclusive scan of pkg_bytes
    uint8_t *events;
    uint32_t events_cap;
    uint8_t *stage;             // [pkg][row] slots of stage_cap bytes, or nullptr: count + write passes
    uint32_t stage_cap;
    uint32_t max_pkgs;
    uint32_t pkg_begin, pkg_end; // the packages of this launch: [pkg_begin, min(pkg_end, *n_pkgs)); staging slots count from pkg_begin
    uint8_t const *pf_tables;   // pre-filter (r433_batch_probe_prefilter), or nullptr
    uint32_t *pf_counts;        // [orig dev][5]: records the filter dropped, by failure code
    // the sizing pass draws its items from a cursor, the heavy packages first (k_slice; launch_slice_count fills both arrays)
    uint32_t draw;              // 0: fixed strides; 1: from one cursor per chunk of devices; 2: large and small packages apart (two launches)
    uint32_t *pkg_order;        // the packages of this launch: OOK then FSK, each by pulse count, descending
    uint32_t *cursor;           // [2 * n_rows / 64 + 4] next entry of pkg_order per chunk of devices: its kind's list / large part, small part; the four ends (k_pkg_order)
    // Which chunk a workgroup of a drawn sizing launch serves: chunk_deal[blockIdx] when deal_grid is the launch's grid (k_deal
    // made it from the shares the host worked out from what the engine's last runs measured, chunk_work), else blockIdx % chunks
    // as ever.  The PCM chunk of the default decoders has twice the work of the average chunk, and with an equal share of the
    // workgroups the launch waited for it (batch_run.cpp).
    uint8_t *chunk_deal;        // [2][16384]: large / small launch
    uint32_t deal_grid;
    // [2][16][2] (large / small launch, chunk): when the chunk's last wavefront left (100 MHz wall clock) and how many wavefronts it
    // had; [64]: when the launches could begin (k_pkg_order) -- next run's shares
    unsigned long long *chunk_work;
};

// `order` (may be null = identity) lists the wavefront slots (whole captures or the chosen segments of split
// captures) in canonical order; n = its length.
// scal[0] = total packages, scal[1] = any arena overflow
void launch_pkg_scan(StreamState const *state, uint32_t const *order, uint32_t n, uint32_t *pkg_base, uint32_t *scal,
        hipStream_t st);
// fills dir_stream (arena slot) / dir_off / rec_bytes for every package (canonical order)
void launch_directory(uint8_t const *arena, uint32_t arena_stride, StreamState const *state, uint32_t const *order,
        uint32_t n, uint32_t const *pkg_base, uint32_t *dir_stream, uint32_t *dir_off, uint32_t *rec_bytes,
        uint32_t max_pkgs, hipStream_t st);
// out[i] = sum(in[0..i)), *total = sum(in[0..n)), n = min(*n_ptr, n_cap); single block
// (carry_in, may be null or equal to total: the scan starts from *carry_in, e.g. the total of the stretch before this one)
void launch_scan_u32(uint32_t const *in, uint32_t *out, uint32_t const *n_ptr, uint32_t n_cap, uint32_t *total,
        hipStream_t st, uint32_t const *carry_in = nullptr, uint32_t n_skip = 0);
// dense copy of all package records in canonical order
void launch_gather_packages(uint8_t const *arena, uint32_t arena_stride, uint32_t const *dir_stream,
        uint32_t const *dir_off, uint32_t const *rec_off, uint32_t const *n_pkgs, uint32_t max_pkgs, uint8_t *dst,
        uint32_t dst_cap, uint32_t grid_pkgs, hipStream_t st);
// p.draw == 2: the sizing pass as two launches, the large packages and the small ones (less LDS, more wavefronts); with a fork
// the small ones run on its second stream beside the large ones and the caller's stream waits for both
struct SliceFork {
    hipStream_t st2;
    hipEvent_t forked, joined;
};
void launch_slice_count(SliceParams const &p, uint32_t grid_pkgs, hipStream_t st, SliceFork const *fork = nullptr, double const *shares = nullptr,
        uint32_t least = 8);
void launch_slice_write(SliceParams const &p, uint32_t grid_pkgs, hipStream_t st);
// the slice index (slicer_kernels.hip): per decoder the (offset, bytes) of its non-empty slices of the event stream, package order.
// count: cnt[blocks][n_devs] (scratch, then every block's first entry), start[n_devs + 1], *total; fill: after dev_off / pkg_off are final
uint32_t slice_index_blocks(uint32_t n_pkgs);
void launch_slice_index_count(uint32_t const *sizes, uint32_t const *n_pkgs_ptr, uint32_t max_pkgs, uint32_t n_pkgs, uint32_t n_devs,
        uint32_t *cnt, uint32_t *start, uint32_t *total, hipStream_t st);
void launch_slice_index_fill(uint32_t const *sizes, uint32_t const *dev_off, uint32_t const *pkg_off, uint32_t const *n_pkgs_ptr,
        uint32_t max_pkgs, uint32_t n_pkgs, uint32_t n_devs, uint32_t const *base, uint2 *slices, uint32_t cap, hipStream_t st);

// ---- function-level baseband kernels (device pointers) ----
enum { ENV_AMP_CU8 = 0, ENV_MAG_CU8 = 1, ENV_MAG_CS16 = 2, ENV_TRUE_CU8 = 3, ENV_TRUE_CS16 = 4 }; // == R433_ENV_* of r433_hip.h
// input_format 1: cs8 -> cu8, 2: cf32 -> cs16; n_rows captures, row_in_bytes input bytes each
void launch_analyze(uint8_t const *arena, uint32_t arena_stride, uint32_t const *dir_stream, uint32_t const *dir_off, uint32_t n_pkgs,
        r433_analysis *out, hipStream_t st);
// -w dump formats (R433_DUMP_*): n_out output values; returns -1 for a format that is not a conversion
int launch_dump(int format, uint32_t sample_size, void const *d_in, void *d_out, uint64_t n_out, hipStream_t st);
void launch_convert(int input_format, void const *d_in, uint64_t in_stride_bytes, void *d_out, uint64_t out_stride_bytes,
        uint64_t row_in_bytes, uint32_t n_rows, hipStream_t st);
// heaviest captures first: weight[n_streams] (scratch), order[n_streams] = capture indices by guessed weight, descending
void launch_capture_order(int kind, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes, uint32_t uniform_bytes,
        uint32_t n_streams, uint32_t *weight, uint32_t *order, hipStream_t st);
void launch_tile_max(int kind, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes, uint32_t uniform_bytes,
        uint32_t n_streams, uint32_t tiles_cap, uint32_t *tile_max, hipStream_t st);
void launch_frame_sums(int kind, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes, uint32_t uniform_bytes,
        uint32_t n_streams, uint32_t frame_samples, uint32_t frames_cap, uint32_t *sums, hipStream_t st);
void launch_envelope(int kind, void const *d_iq, uint16_t *d_env, uint32_t n, uint32_t *d_sum, hipStream_t st);

} // namespace r433
