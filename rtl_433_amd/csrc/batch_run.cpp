// batch_run.cpp -- r433_batch_run / r433_batch_run_pulses: the kernel sequence of one pass of the hot path, stage by
// stage (input conversion, autolevel, segment planning, detection + stitch, slicer fan-out, host mirrors).
#include <chrono>

#include "host_common.hpp"

using namespace r433;

std::mutex g_detect_turn[kTurnDevices]; // one per GPU: engines on different GPUs have nothing to take turns on

// ---- r433_batch_run, stage by stage --------------------------------------------------------------
namespace {

// R433_TRACE_LEGS=1: one stderr line per stage of a pass with the monotonic clock in milliseconds (the clock Python's
// time.monotonic() reads), so that a host which keeps several engines in flight can lay its own stamps beside them
// (tools/leg_timeline.py).  Off: one read of a static flag per stamp.
void leg_stamp(r433_batch const *b, char const *what)
{
    static int const on = getenv("R433_TRACE_LEGS") != nullptr;
    if (on) {
        double const ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
        fprintf(stderr, "r433-leg %p %s %.3f\n", (void const *)b, what, ms);
    }
}

// What one r433_batch_run call carries from stage to stage.
struct RunCtx {
    std::unique_lock<std::mutex> turn; // exclusive_detect: this pass's turn on the detection kernel of the engine's GPU (bound in run_detect)
    r433_batch *b;
    hipStream_t st;
    uint32_t ss;                  // bytes per sample: 2 = cu8, 4 = cs16
    void const *d_iq;             // the captures as the detector sees them (after input conversion)
    uint64_t stride_bytes;
    uint32_t const *stream_bytes; // host, per capture; null = every capture fills the stride
    uint32_t n_streams;
    uint32_t max_samples = 0, frames_cap = 0, want_stride = 0;
    int const *d_min_high = nullptr; // per-frame detection level (-Y autolevel), device
    // plan: one wavefront per capture, or several per long capture (speculative cuts)
    bool split = false;
    std::vector<SegDesc> segs;
    std::vector<uint32_t> seg_first_of; // segs of capture c: [seg_first_of[c], seg_first_of[c+1])
    std::vector<uint32_t> cap_n;        // samples per capture
    uint32_t max_seg_samples = 0, n_planned = 0, n_slots = 0;
    uint32_t tiles_cap = 0;             // split: tile sums per capture in b->h_tile_max
    std::vector<uint64_t> quiet_below;  // split: per capture, the tile sum below which a tile counts as quiet
    // detection result
    uint32_t n_order = 0;              // slots that make up the result, in capture order
    uint32_t const *d_order = nullptr; // null = slot i is capture i
    std::vector<uint32_t> order;
    uint32_t total_pkgs = 0;

    uint32_t const *d_lens() const { return stream_bytes ? b->d_stream_bytes.p : nullptr; }
    int env_kind() const { return ss == 4 ? ENV_MAG_CS16 : b->cfg.use_mag_est ? ENV_MAG_CU8 : ENV_AMP_CU8; }
};


// cs8 / cf32 input -> cu8 / cs16 in an internal buffer
int run_convert_input(RunCtx &r)
{
    r433_batch *const b = r.b;
    if (b->cfg.input_format == R433_IN_NATIVE || b->cfg.input_format == R433_IN_S16_AM || b->cfg.input_format == R433_IN_S16_FM)
        return 0; // (am.s16 / fm.s16 words go through the flow as if they were cu8 pairs, src/rtl_433.c:1735-1739)
    // The reference converts these formats while it loads a file (src/rtl_433.c:1811-1834): one HBM-bound
    // map into an internal buffer, then everything below sees cu8 / cs16 like the reference's flow does.
    uint32_t const shrink = b->cfg.input_format == R433_IN_CF32 ? 2 : 1; // 8 B -> 4 B per sample
    uint64_t in_max = 0;
    b->conv_bytes.resize(r.n_streams);
    for (uint32_t i = 0; i < r.n_streams; ++i) {
        uint64_t const nb = r.stream_bytes ? r.stream_bytes[i] : r.stride_bytes;
        if (nb > r.stride_bytes)
            return fail(R433_EINVAL, "capture %u is longer than the stride", i);
        in_max = std::max(in_max, nb);
        b->conv_bytes[i] = (uint32_t)(nb / (shrink * r.ss) * r.ss); // whole samples
    }
    uint64_t const out_stride = ((in_max / shrink) + 15) & ~15ull;
    if (int rc = b->d_converted.ensure((size_t)r.n_streams * out_stride + 16))
        return rc;
    launch_convert((int)b->cfg.input_format, r.d_iq, r.stride_bytes, b->d_converted.p, out_stride, in_max, r.n_streams, r.st);
    HIP_TRY(hipGetLastError());
    r.d_iq = b->d_converted.p;
    r.stride_bytes = out_stride;
    r.stream_bytes = b->conv_bytes.data();
    return 0;
}

// capture lengths to the device, per-capture scratch, first guess of the package arena
int run_size_buffers(RunCtx &r)
{
    r433_batch *const b = r.b;
    uint32_t max_bytes = 0;
    if (r.stream_bytes) {
        for (uint32_t i = 0; i < r.n_streams; ++i) {
            if (r.stream_bytes[i] > r.stride_bytes)
                return fail(R433_EINVAL, "capture %u is longer than the stride", i);
            max_bytes = std::max(max_bytes, r.stream_bytes[i]);
        }
    }
    else {
        max_bytes = (uint32_t)r.stride_bytes;
    }
    r.max_samples = max_bytes / r.ss;
    r.frames_cap = r.max_samples / b->cfg.frame_samples + 2;

    int rc;
    if ((rc = b->d_ring.ensure((size_t)r.n_streams * R433_PD_MAX_PULSES)) || (rc = b->d_state.ensure(r.n_streams))
            || (rc = b->d_frame_sums.ensure((size_t)r.n_streams * r.frames_cap)) || (rc = b->d_pkg_base.ensure(r.n_streams)))
        return rc;
    if (r.stream_bytes) {
        if ((rc = b->d_stream_bytes.ensure(r.n_streams)))
            return rc;
        HIP_TRY(hipMemcpyAsync(b->d_stream_bytes.p, r.stream_bytes, r.n_streams * sizeof(uint32_t), hipMemcpyHostToDevice, r.st));
    }
    b->frames_cap = r.frames_cap;
    b->n_streams = r.n_streams;
    if (b->logic_on) {
        b->logic_stride = ((uint64_t)r.max_samples + 15u) & ~15ull;
        size_t const bytes = (size_t)r.n_streams * b->logic_stride + 16;
        if ((rc = b->d_logic.ensure(bytes)) || (rc = b->h_logic.ensure(bytes)))
            return rc;
        HIP_TRY(hipMemsetAsync(b->d_logic.p, 0, bytes, r.st));
    }

    // arena: worst case is one (pulse, gap) pair per 20 samples plus headers; start at ~1 B/sample
    r.want_stride = std::max<uint32_t>(16384u, ((r.max_samples + 4096u) + 15u) & ~15u);
    // sized from THIS run's longest piece (a capture here, a segment once the plan is known) times what earlier overflows
    // taught; never inherited from a run with longer captures
    b->arena_stride = (uint32_t)std::min<uint64_t>((uint64_t)r.want_stride * b->arena_growth, 1u << 30);

    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[0], r.st));
    return 0;
}

// -Y autolevel (reference src/r_flow.c:166-186): the detection level of a frame follows the noise
// estimate, which follows the mean envelope of the frames so far -- a short recurrence in host
// floats (same libm as the reference) over per-frame sums that one HBM-bound pass provides.
int run_autolevel(RunCtx &r)
{
    r433_batch *const b = r.b;
    if (!(b->cfg.auto_level > 0))
        return 0;
    int rc;
    if ((rc = b->d_frame_min_high.ensure((size_t)r.n_streams * r.frames_cap)) || (rc = b->h_frame_sums.ensure((size_t)r.n_streams * r.frames_cap)))
        return rc;
    launch_frame_sums(r.env_kind(), r.d_iq, r.stride_bytes, r.d_lens(), (uint32_t)r.stride_bytes, r.n_streams,
            b->cfg.frame_samples, r.frames_cap, b->d_frame_sums.p, r.st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_frame_sums.p, b->d_frame_sums.p, (size_t)r.n_streams * r.frames_cap * sizeof(uint32_t),
            hipMemcpyDeviceToHost, r.st));
    if (b->logic_on && r.d_iq)
        HIP_TRY(hipMemcpyAsync(b->h_logic.p, b->d_logic.p, (size_t)r.n_streams * b->logic_stride, hipMemcpyDeviceToHost, r.st));
    HIP_TRY(stream_wait(b, r.st));
    b->h_frame_min_high.assign((size_t)r.n_streams * r.frames_cap, b->det.min_high);
    int const is_mag = r.ss == 4 || b->cfg.use_mag_est;
    for (uint32_t s = 0; s < r.n_streams; ++s) {
        uint32_t const n = (r.stream_bytes ? r.stream_bytes[s] : (uint32_t)r.stride_bytes) / r.ss;
        float noise_level = 0.0f, min_level_auto = 0.0f;
        DetCfg lv = b->det;
        for (uint32_t f = 0; f < r.frames_cap; ++f) {
            uint64_t const start = (uint64_t)f * b->cfg.frame_samples;
            if (start < n) {
                uint32_t const cnt = (uint32_t)std::min<uint64_t>(b->cfg.frame_samples, n - start);
                float const avg_db = r433_level_db(b->h_frame_sums.p[(size_t)s * r.frames_cap + f], cnt, is_mag);
                if (min_level_auto == 0.0f)
                    min_level_auto = b->cfg.min_level_db;
                if (noise_level == 0.0f)
                    noise_level = min_level_auto - 3.0f;
                if (avg_db < noise_level + 3.0f) {
                    noise_level = (noise_level * 7 + avg_db) / 8;
                    if (noise_level < b->cfg.min_level_db - 3.0f && fabsf(min_level_auto - noise_level - 3.0f) > 1.0f) {
                        min_level_auto = noise_level + 3.0f;
                        levels_from_db(lv, (int)b->cfg.use_mag_est, b->cfg.level_limit_db, min_level_auto, b->cfg.min_snr_db);
                    }
                }
                else {
                    noise_level = (noise_level * 31 + avg_db) / 32;
                }
            }
            b->h_frame_min_high[(size_t)s * r.frames_cap + f] = lv.min_high;
        }
    }
    HIP_TRY(hipMemcpyAsync(b->d_frame_min_high.p, b->h_frame_min_high.data(), b->h_frame_min_high.size() * sizeof(int),
            hipMemcpyHostToDevice, r.st));
    r.d_min_high = b->d_frame_min_high.p;
    return 0;
}

// Where long captures may be cut: at the end of 12.5 ms of quiet, about every split_samples samples.
int run_plan_split(RunCtx &r, uint32_t split_samples)
{
    r433_batch *const b = r.b;
    int rc;
    constexpr uint32_t kTileS = 2048;
    uint32_t const tiles_cap = r.max_samples / kTileS + 1;
    r.tiles_cap = tiles_cap;
    r.quiet_below.assign(r.n_streams, 0);
    if ((rc = b->d_tile_max.ensure((size_t)r.n_streams * tiles_cap)) || (rc = b->h_tile_max.ensure((size_t)r.n_streams * tiles_cap)))
        return rc;
    launch_tile_max(r.env_kind(), r.d_iq, r.stride_bytes, r.d_lens(), (uint32_t)r.stride_bytes, r.n_streams,
            tiles_cap, b->d_tile_max.p, r.st);
    HIP_TRY(hipGetLastError());
    auto const t_plan = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(b->h_tile_max.p, b->d_tile_max.p, (size_t)r.n_streams * tiles_cap * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
    HIP_TRY(stream_wait(b, r.st));
    if (b->debug_flags & R433_DEBUG_SPLIT_TRACE)
        fprintf(stderr, "r.split: tile estimate %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan).count());
    // A tile is quiet when it carries no more energy than the noise floor: mean envelope at most 1.5x
    // the capture's median tile, or -- for captures that are mostly signal -- below half the
    // falling-edge level of the lowest threshold the detector can have (pulse_detect.c:300-304).
    int thr = (-1 + std::min(b->det.min_high, b->det.max_high)) / 2;
    if (b->det.fixed_high)
        thr = b->det.fixed_high;
    uint64_t const abs_quiet = (uint64_t)std::max(1, (thr - thr / 8) / 2) * 2048u;
    bool const blind = (b->debug_flags & R433_DEBUG_SPLIT_BLIND) != 0; // tests: cut anywhere, let the verification sort it out
    uint32_t const seg_len = (split_samples + kTileS - 1) / kTileS * kTileS;
    // a package stays open until its last gap exceeds 10 pulse widths and 10 ms (pulse_detect.c:446-450):
    // ask for 12.5 ms of quiet before a cut (pulses up to 1.25 ms; the stitch catches the rest)
    uint32_t const quiet_tiles = std::max<uint32_t>(2u, (b->cfg.samp_rate / 80u + kTileS - 1) / kTileS);
    r.max_seg_samples = 0;
    for (uint32_t c = 0; c < r.n_streams; ++c) {
        r.seg_first_of[c] = (uint32_t)r.segs.size();
        uint32_t const n = r.cap_n[c];
        uint32_t const *tm = b->h_tile_max.p + (size_t)c * tiles_cap;
        uint64_t quiet_below = abs_quiet; // on tile sums
        if (n >= 2 * kTileS) {
            // (only a workgroup's estimated cost depends on it: for long captures the median of every 16th tile -- the full
            // selection over 131072 tiles was a third of a millisecond between k_tile_max and k_wave, device idle)
            uint32_t const nt = n / kTileS, step = nt >= 4096 ? 16u : 1u;
            std::vector<uint32_t> med;
            med.reserve(nt / step + 1);
            for (uint32_t t = 0; t < nt; t += step)
                med.push_back(tm[t]);
            std::nth_element(med.begin(), med.begin() + med.size() / 2, med.end());
            quiet_below = std::max<uint64_t>(abs_quiet, (uint64_t)med[med.size() / 2] * 3 / 2);
        }
        r.quiet_below[c] = quiet_below; // (what a workgroup's cost is estimated with, plan_workgroups)
        // For the cuts the floor is taken locally: 1.5x the quietest tile among the ~190 around (blocks of 64 tiles, the
        // tile's own and its two neighbours).  The median of the whole capture sits on a half-loud tile when about half of
        // the capture is signal (config 3: bursts 47 % of the time), and calls everything quiet then; a floor that steps
        // (config 5) makes any capture-wide quantile wrong for part of the capture.
        uint32_t const n_whole = n / kTileS;
        std::vector<uint32_t> block_min((n_whole + 63) / 64 + 1, 0xffffffffu);
        for (uint32_t t = 0; t < n_whole; ++t)
            block_min[t / 64] = std::min(block_min[t / 64], tm[t]);
        auto quiet_tile = [&](uint32_t t) {
            uint32_t const bk = t / 64;
            uint32_t lo = std::min(block_min[bk], block_min[bk + 1]);
            if (bk > 0)
                lo = std::min(lo, block_min[bk - 1]);
            return tm[t] < std::max<uint64_t>(abs_quiet, (uint64_t)lo * 3 / 2);
        };
        // every tile judged once, and the loud ones counted up to each tile: "the quiet_tiles tiles before P are all quiet" is
        // one subtraction (a candidate used to walk its 12.5 ms again: 0.8 M tile tests per pass over a 256 Mi-sample stream,
        // a millisecond of host time with the device idle -- profiles/r06_stream_phases.txt)
        std::vector<uint32_t> loud_before(n_whole + 2, 0);
        for (uint32_t t = 0; t < n_whole; ++t)
            loud_before[t + 1] = loud_before[t] + (quiet_tile(t) ? 0u : 1u);
        loud_before[n_whole + 1] = loud_before[n_whole] + 1u; // (a tile that is not whole is never quiet)
        auto quiet_at = [&](uint32_t t) { return t < n_whole && loud_before[t + 1] == loud_before[t]; };
        auto quiet_run = [&](uint32_t first, uint32_t count) { // tiles first .. first + count - 1, all whole and quiet
            return first + count <= n_whole && loud_before[first + count] == loud_before[first];
        };
        std::vector<uint32_t> cuts;
        uint32_t pos = seg_len;
        while (n > seg_len && pos + seg_len / 2 < n) {
            // The tile sums are taken from a subsample (four 128-byte lines of a tile, k_tile_max): a burst that starts behind the
            // last line of the tile in front of the cut goes unseen there, and that tile is the one the new piece
            // establishes its filter carries and its floor on.  A cut whose own first tile is quiet too cannot have that
            // (anything longer than the space between two lines shows in the next tile's first line): such cuts first.
            uint32_t cut = 0, second_best = 0;
            for (uint32_t P = pos; P < std::min(n, pos + seg_len) && P + kTileS <= n; P += kTileS) {
                bool const quiet = P / kTileS >= quiet_tiles && quiet_run(P / kTileS - quiet_tiles, quiet_tiles);
                if (blind || (quiet && quiet_at(P / kTileS))) {
                    cut = P;
                    break;
                }
                if (quiet && !second_best)
                    second_best = P;
            }
            if (!cut)
                cut = second_best;
            if (cut) {
                cuts.push_back(cut);
                pos = cut + seg_len;
            }
            else {
                pos += seg_len;
            }
        }
        uint32_t from = 0;
        for (size_t k = 0; k <= cuts.size(); ++k) {
            uint32_t const to = k < cuts.size() ? cuts[k] : n;
            uint32_t const last = k == cuts.size() ? SEG_LAST : 0u;
            if (k == 0) {
                r.segs.push_back(SegDesc{c, 0u, to, SEG_FIRST | SEG_PRIMARY | last});
            }
            else { // both parities of the noise floor
                r.segs.push_back(SegDesc{c, from, to, SEG_PRIMARY | last});
                r.segs.push_back(SegDesc{c, from, to, SEG_ODD | last});
            }
            r.max_seg_samples = std::max(r.max_seg_samples, to - from + kTileS);
            from = to;
        }
    }
    r.seg_first_of[r.n_streams] = (uint32_t)r.segs.size();
    return 0;
}

// one wavefront per capture, or several per long capture
int run_plan(RunCtx &r)
{
    r433_batch *const b = r.b;
    int rc;
    r.seg_first_of.assign(r.n_streams + 1, 0);
    r.cap_n.resize(r.n_streams);
    for (uint32_t c = 0; c < r.n_streams; ++c)
        r.cap_n[c] = (r.stream_bytes ? r.stream_bytes[c] : (uint32_t)r.stride_bytes) / r.ss;
    b->stream_samples = r.cap_n;
    // automatic: only where one wavefront per capture would leave the chip empty -- few, long captures.
    // Aim at ~8192 segments, at least 24 Ki samples each (tools/splitsweep.py, tools/longbench.py: a piece that has to be run
    // again across a cut that did not verify is one wavefront's serial time, so short pieces keep the later rounds short;
    // 64 Mi-sample cs16 FSK stream 20.0 -> 15.5 ms, 256 Mi-sample 2 MS/s stream 27.0 -> 22.5 ms against 32 Ki / 4096).
    uint32_t split_samples = b->logic_on ? 0u : b->split_samples; // the logic dump is painted by whole-capture wavefronts
    if (b->cfg.input_format == R433_IN_S16_AM || b->cfg.input_format == R433_IN_S16_FM)
        split_samples = 0; // where a cut may go is read off the IQ envelope, which these files do not have
    if (split_samples == R433_SPLIT_AUTO) {
        uint64_t total = 0;
        for (uint32_t c = 0; c < r.n_streams; ++c)
            total += r.cap_n[c];
        split_samples = (r.n_streams <= 64 && r.max_samples >= (1u << 20)) ? (uint32_t)std::max<uint64_t>(24576, total / 8192) : 0u;
    }
    r.split = split_samples > 0;
    r.max_seg_samples = r.max_samples;
    r.n_order = r.n_streams;
    if (r.split && (rc = run_plan_split(r, split_samples)))
        return rc;
    r.n_planned = r.split ? (uint32_t)r.segs.size() : r.n_streams;
    r.n_slots = r.split ? 3 * r.n_planned : r.n_streams; // + re-run slots for cuts that have to be dropped
    if (r.split) // sized for whole captures above: segments need less (merged pieces that outgrow it take the overflow retry)
        b->arena_stride = (uint32_t)std::min<uint64_t>((uint64_t)std::max<uint32_t>(16384u, ((r.max_seg_samples + 4096u) + 15u) & ~15u) * b->arena_growth, 1u << 30);
    if ((rc = b->d_ring.ensure((size_t)r.n_slots * R433_PD_MAX_PULSES)) || (rc = b->d_state.ensure(r.n_slots))
            || (rc = b->d_pkg_base.ensure(r.n_slots)) || (rc = b->h_state.ensure(r.n_slots)) || (rc = b->d_order.ensure(r.n_slots))
            || (rc = b->d_segs.ensure(r.n_slots)) || (rc = b->d_wg.ensure(r.n_slots)))
        return rc;
    if (r.split) // a slot no workgroup wrote reads as one that failed, whatever the memory held before (seg_fail = -1)
        HIP_TRY(hipMemsetAsync(b->d_state.p, 0xff, (size_t)r.n_slots * sizeof(StreamState), r.st));
    return 0;
}

// what every launch of the detection kernel in this run shares
StreamParams stream_params(RunCtx const &r)
{
    r433_batch *const b = r.b;
    StreamParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.iq = (uint8_t const *)r.d_iq;
    sp.stride_bytes = r.stride_bytes;
    sp.stream_bytes = r.d_lens();
    sp.uniform_bytes = (uint32_t)r.stride_bytes;
    sp.n_streams = r.n_planned;
    sp.frame_samples = b->cfg.frame_samples;
    sp.flags = b->cfg.input_format == R433_IN_S16_AM ? RUN_AM_IS_INPUT : b->cfg.input_format == R433_IN_S16_FM ? RUN_FM_IS_INPUT : 0u;
    sp.flags |= b->debug_flags & (RUN_DBG_SKIP_DETECT | RUN_DBG_SKIP_FILTERS | RUN_DBG_TIMING | RUN_NO_TRAIN_ENGINE | RUN_ONE_WAVE | RUN_NO_ROLE_SWAP | RUN_NO_PRIO | RUN_PAIR | RUN_NO_LAZY); // r433_batch_set_debug
    sp.det = b->det;
    sp.use_mag = (int)b->cfg.use_mag_est;
    sp.enable_fm = (int)b->cfg.enable_fm;
    sp.a16 = b->a16;
    sp.b16 = b->b16;
    sp.a32 = b->a32;
    sp.b32 = b->b32;
    sp.arena = b->d_arena.p;
    sp.arena_stride = b->arena_stride;
    sp.fsk_ring = b->d_ring.p;
    sp.state = b->d_state.p;
    sp.frame_sums = b->d_frame_sums.p;
    sp.frames_cap = r.frames_cap;
    sp.frame_min_high = r.d_min_high;
    if (b->logic_on) {
        sp.logic = b->d_logic.p;
        sp.logic_stride = b->logic_stride;
    }
    sp.tap_env = (uint16_t *)b->tap_env;
    sp.tap_am = (int16_t *)b->tap_am;
    sp.tap_fm = (int16_t *)b->tap_fm;
    sp.tap_stride = b->tap_stride;
    return sp;
}

// The workgroups of a launch over `n` slots of split captures: a piece whose two parity variants sit in neighbouring slots
// becomes ONE workgroup (a producer wavefront feeding two consumers, bit 31 of the entry), and the workgroups go out
// heaviest first -- a piece costs its tiles, a loud tile (a burst the detector has to walk) several times a quiet one --
// so that the long pieces do not start last and finish alone.
void plan_workgroups(RunCtx const &r, SegDesc const *segs, uint32_t n, std::vector<uint32_t> &wgs)
{
    r433_batch *const b = r.b;
    std::vector<std::pair<uint64_t, uint32_t>> order; // (cost, entry)
    for (uint32_t i = 0; i < n;) {
        SegDesc const &d = segs[i];
        // (R433_DEBUG_ONE_WAVE: workgroups of one wavefront run one slot each -- no twins then)
        bool const twin = !(b->debug_flags & R433_DEBUG_ONE_WAVE) && i + 1 < n && !(d.flags & (SEG_FIRST | SEG_ODD)) && (segs[i + 1].flags & SEG_ODD)
                && segs[i + 1].capture == d.capture && segs[i + 1].start == d.start && segs[i + 1].end == d.end;
        uint64_t cost = 0;
        if (r.tiles_cap && d.capture < r.quiet_below.size()) {
            uint32_t const *tm = b->h_tile_max.p + (size_t)d.capture * r.tiles_cap;
            for (uint32_t t = d.start / 2048u; t < (d.end + 2047u) / 2048u && t < r.tiles_cap; ++t)
                cost += tm[t] >= r.quiet_below[d.capture] ? 6u : 1u;
        }
        order.emplace_back(cost, i | (twin ? 0x80000000u : 0u));
        i += twin ? 2 : 1;
    }
    std::stable_sort(order.begin(), order.end(), [](auto const &x, auto const &y) { return x.first > y.first; });
    wgs.clear();
    for (auto const &o : order)
        wgs.push_back(o.second);
}

// Stitch.  Per capture an ordered list of pieces; every piece but the first exists in two
// parity variants (two slots).  Walk the pieces in order; at each cut keep the variant that
// assumed exactly the floor (and the level estimate that an idle step leaves behind: a spurious
// short pulse returns to idle without one) the piece before it really ended with, and require
// that piece to have ended idle with the lead-in saturated -- that is the detector's whole
// state between packages (everything else is reset when a pulse starts).  A cut that does not verify is
// dropped: the piece before it is run again through to the end of the next piece, and the walk resumes from there.  Every
// round removes at least one cut per capture that still has a problem, so this terminates.
int run_stitch(RunCtx &r, StreamParams const &sp)
{
    r433_batch *const b = r.b;
    int rc;
    struct Piece {
        SegDesc seg;      // flags without SEG_ODD / SEG_PRIMARY
        uint32_t slot[2]; // even / odd parity variant (the first piece of a capture: slot[0] only)
    };
    std::vector<std::vector<Piece>> pieces(r.n_streams);
    for (uint32_t c = 0; c < r.n_streams; ++c)
        for (uint32_t k = r.seg_first_of[c]; k < r.seg_first_of[c + 1]; k += (k == r.seg_first_of[c] ? 1 : 2)) {
            Piece pc;
            pc.seg = r.segs[k];
            pc.seg.flags &= ~(uint32_t)(SEG_ODD | SEG_PRIMARY);
            pc.slot[0] = k;
            pc.slot[1] = k == r.seg_first_of[c] ? k : k + 1;
            pieces[c].push_back(pc);
        }
    uint32_t n_have = r.n_planned; // slots whose state is on the host
    std::vector<SegDesc> slot_seg(r.segs), launch_list;
    auto new_slot = [&](SegDesc const &d) {
        launch_list.push_back(d);
        slot_seg.push_back(d);
        return n_have + (uint32_t)launch_list.size() - 1;
    };
    auto run_launch_list = [&]() -> int {
        if (launch_list.empty())
            return 0;
        if (n_have + launch_list.size() > r.n_slots)
            return fail(R433_EHIP, "r.split bookkeeping ran out of slots");
        b->last_redone += (uint32_t)launch_list.size();
        auto const t_round = std::chrono::steady_clock::now();
        HIP_TRY(hipMemcpyAsync(b->d_segs.p + n_have, launch_list.data(), launch_list.size() * sizeof(SegDesc), hipMemcpyHostToDevice, r.st));
        StreamParams sr = sp;
        sr.n_streams = (uint32_t)launch_list.size();
        sr.frame_sums = nullptr; // counted by the first launch
        std::vector<uint32_t> wgs;
        plan_workgroups(r, launch_list.data(), (uint32_t)launch_list.size(), wgs);
        HIP_TRY(hipMemcpyAsync(b->d_wg.p + n_have, wgs.data(), wgs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, r.st));
        sr.wg_slot = b->d_wg.p + n_have;
        sr.n_wgs = (uint32_t)wgs.size();
        sr.segs = b->d_segs.p + n_have;
        sr.arena = b->d_arena.p + (size_t)n_have * b->arena_stride;
        sr.fsk_ring = b->d_ring.p + (size_t)n_have * R433_PD_MAX_PULSES;
        sr.state = b->d_state.p + n_have;
        launch_stream(sr, r.ss, r.st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(b->h_state.p + n_have, b->d_state.p + n_have, launch_list.size() * sizeof(StreamState), hipMemcpyDeviceToHost, r.st));
        HIP_TRY(stream_wait(b, r.st));
        if (b->debug_flags & R433_DEBUG_SPLIT_TRACE)
            fprintf(stderr, "r.split: round over %zu slots %.3f ms\n", launch_list.size(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count());
        n_have += (uint32_t)launch_list.size();
        launch_list.clear();
        return 0;
    };
    auto const t_first = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(b->h_state.p, b->d_state.p, (size_t)r.n_planned * sizeof(StreamState), hipMemcpyDeviceToHost, r.st));
    HIP_TRY(stream_wait(b, r.st));
    if (b->debug_flags & R433_DEBUG_SPLIT_TRACE)
        fprintf(stderr, "r.split: waited %.3f ms for the first launch (%u slots)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_first).count(), r.n_planned);
    // (1) cuts neither variant could start from (no provable filter carry or floor: digital
    // silence does that) are known now, all at once: merge across them in one extra launch
    for (uint32_t c = 0; c < r.n_streams; ++c) {
        std::vector<Piece> merged;
        for (Piece const &pc : pieces[c]) {
            bool const unstartable = !merged.empty() && b->h_state.p[pc.slot[0]].seg_fail && b->h_state.p[pc.slot[1]].seg_fail;
            if (!unstartable) {
                merged.push_back(pc);
                continue;
            }
            if (b->debug_flags & R433_DEBUG_SPLIT_TRACE)
                fprintf(stderr, "r.split: capture %u cut at %u cannot be started from (reasons %d / %d)\n", c, pc.seg.start,
                        b->h_state.p[pc.slot[0]].seg_fail, b->h_state.p[pc.slot[1]].seg_fail);
            Piece &m = merged.back();
            m.seg.end = pc.seg.end;
            m.seg.flags |= pc.seg.flags & SEG_LAST;
            m.slot[0] = m.slot[1] = UINT32_MAX; // to be run again
        }
        for (Piece &m : merged)
            if (m.slot[0] == UINT32_MAX) {
                SegDesc d = m.seg;
                d.flags |= SEG_PRIMARY;
                m.slot[0] = new_slot(d);
                m.slot[1] = m.slot[0];
                if (!(d.flags & SEG_FIRST)) {
                    d.flags = (d.flags & ~(uint32_t)SEG_PRIMARY) | SEG_ODD;
                    m.slot[1] = new_slot(d);
                }
            }
        pieces[c].swap(merged);
    }
    if ((rc = run_launch_list()))
        return rc;
    // (2) the walk proper
    std::vector<std::vector<uint32_t>> chosen(r.n_streams);
    std::vector<size_t> at(r.n_streams, 1);
    std::vector<uint32_t> dropped(r.n_streams, 0);
    for (uint32_t c = 0; c < r.n_streams; ++c)
        chosen[c].push_back(pieces[c][0].slot[0]);
    for (;;) {
        for (uint32_t c = 0; c < r.n_streams; ++c) {
            std::vector<Piece> &pcs = pieces[c];
            while (at[c] < pcs.size()) {
                uint32_t const cur = chosen[c].back();
                StreamState const &P = b->h_state.p[cur];
                Piece const &nx = pcs[at[c]];
                uint32_t pick = UINT32_MAX;
                if (P.seg_end_state == ST_IDLE && P.seg_end_lead == 1025)
                    for (int v = 0; v < 2; ++v)
                        if (!b->h_state.p[nx.slot[v]].seg_fail && b->h_state.p[nx.slot[v]].seg_init_low == P.seg_end_low
                                && b->h_state.p[nx.slot[v]].seg_init_high == P.seg_end_high)
                            pick = nx.slot[v];
                if (pick != UINT32_MAX) {
                    chosen[c].push_back(pick);
                    at[c] += 1;
                    dropped[c] = 0;
                    continue;
                }
                // drop this cut: the standing piece continues through the next one.  (Three cuts in a
                // row that fail are not worth a fourth try: the piece then runs to the capture's end.)
                bool const give_up = ++dropped[c] >= 3;
                if (b->debug_flags & R433_DEBUG_SPLIT_TRACE)
                    fprintf(stderr, "r.split: capture %u cut at %u dropped (end state %d, floor %d vs %d/%d, fail %d/%d)\n", c, nx.seg.start,
                            P.seg_end_state, P.seg_end_low, b->h_state.p[nx.slot[0]].seg_init_low, b->h_state.p[nx.slot[1]].seg_init_low,
                            b->h_state.p[nx.slot[0]].seg_fail, b->h_state.p[nx.slot[1]].seg_fail);
                SegDesc d = slot_seg[cur];
                d.end = give_up ? r.cap_n[c] : nx.seg.end;
                d.flags = (d.flags & ~(uint32_t)SEG_LAST) | (give_up ? (uint32_t)SEG_LAST : (nx.seg.flags & SEG_LAST));
                chosen[c].back() = new_slot(d);
                at[c] = give_up ? pcs.size() : at[c] + 1;
                break; // its end state is not known yet: resume in the next round
            }
        }
        if (launch_list.empty())
            break;
        if ((rc = run_launch_list()))
            return rc;
    }
    r.order.clear();
    for (uint32_t c = 0; c < r.n_streams; ++c)
        r.order.insert(r.order.end(), chosen[c].begin(), chosen[c].end());
    r.n_order = (uint32_t)r.order.size();
    HIP_TRY(hipMemcpyAsync(b->d_order.p, r.order.data(), r.order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, r.st));
    r.d_order = b->d_order.p;
    return 0;
}

// envelope, filters, pulse detection: packages per slot in the arena (grown and repeated if it overflows)
constexpr uint32_t kOrderFrom = 2048; // captures in a grid from which on their order is worth a look (1536 pairs fit the chip at once)
// Captures in a grid from which on the two roles of a capture run as two launches (stream_kernels.hip FORM 4 / FORM 5; measured:
// 4096 captures 2.02 ms as pairs / 2.39 ms as two launches, 8192 captures 3.66 / 3.41 -- 3.05 with the consumers in their own
// order --, profiles/r05_kbench.txt, r05_h_kbench_consumer_order.txt): below,
// a launch lasts as long as its slowest capture and the pair's overlap of filters and detector is what shortens that; above,
// the grid is several rounds deep and what counts is how many CONSUMERS the chip holds at once (three to a SIMD instead of
// one and a half).  The tile records are 8.4 KB per tile of 2048 samples: at most kRolesStoreMax bytes of HBM per engine.
constexpr uint32_t kRolesFrom = 6144;
constexpr uint64_t kRolesStoreMax = 12ull << 30;

int run_detect(RunCtx &r)
{
    r433_batch *const b = r.b;
    int rc;
    // the calling thread waits for the detection kernel below anyway (it needs the package count): holding the turn until
    // then keeps two engines' detection kernels from running side by side
    std::unique_lock<std::mutex> &turn = r.turn;
    leg_stamp(b, "enter");
    if (b->exclusive_detect) {
        turn = std::unique_lock<std::mutex>(g_detect_turn[(unsigned)(b->device < 0 ? 0 : b->device) % kTurnDevices], std::defer_lock);
        turn.lock();
        leg_stamp(b, "turn");
        if (b->profiling) // the time spent waiting for the turn is not the kernel's
            HIP_TRY(hipEventRecord(b->ev[0], r.st));
    }
    for (int attempt = 0;; ++attempt) {
        if ((rc = b->d_arena.ensure((size_t)r.n_slots * b->arena_stride)))
            return rc;
        StreamParams sp = stream_params(r);
        // Per-frame envelope sums come out of the producers (atomic adds: a frame may be shared by several pieces).  The
        // pieces of the FIRST launch partition every capture and only their primary variant adds, so every sample is
        // counted exactly once there; the launches that run merged pieces again (run_stitch) add nothing.
        HIP_TRY(hipMemsetAsync(b->d_frame_sums.p, 0, (size_t)r.n_streams * r.frames_cap * sizeof(uint32_t), r.st));
        r.d_order = nullptr;
        b->last_segments = r.n_planned;
        b->last_redone = 0;
        std::vector<uint32_t> wgs; // (outlives the copy below: the stream is waited for before this scope ends)
        if (!r.split && (r.n_streams >= kOrderFrom || (b->debug_flags & R433_DEBUG_FORCE_ORDER)) && !(b->debug_flags & R433_DEBUG_NO_ORDER)) {
            // several rounds of workgroups: the heavy captures first (k_capture_weight; the order is made on the device)
            if ((rc = b->d_wg.ensure(2 * (size_t)r.n_streams)))
                return rc;
            launch_capture_order(r.env_kind(), r.d_iq, r.stride_bytes, r.d_lens(), (uint32_t)r.stride_bytes, r.n_streams,
                    b->d_wg.p + r.n_streams, b->d_wg.p, r.st);
            HIP_TRY(hipGetLastError());
            sp.wg_slot = b->d_wg.p;
            sp.n_wgs = r.n_streams;
        }
        if (!r.split && !(b->debug_flags & R433_DEBUG_NO_SPLIT_ROLES) && (r.n_streams >= kRolesFrom || (b->debug_flags & R433_DEBUG_SPLIT_ROLES))) {
            uint32_t const tiles_cap = (uint32_t)((r.stride_bytes / r.ss + kTileSamples - 1) / kTileSamples);
            uint64_t const store = (uint64_t)r.n_streams * tiles_cap * kTileRecBytes;
            size_t const n = r.n_streams;
            // (the two-launch form is an optimisation: where its store -- twice the cu8 input -- does not fit beside the other
            // engines of the device, the pass goes out as pairs, which need none)
            bool have_store = tiles_cap && store <= kRolesStoreMax;
            if (have_store && (b->d_tile_store.ensure(store) || b->d_tile_desc.ensure(n * tiles_cap) || b->d_tile_words.ensure(6 * n + 4))) {
                (void)hipGetLastError();
                b->d_tile_store.release();
                have_store = false;
            }
            if (have_store) {
                sp.flags |= RUN_SPLIT_ROLES;
                sp.tile_store = b->d_tile_store.p;
                sp.tile_desc = b->d_tile_desc.p;
                sp.tiles_cap = tiles_cap;
                sp.tile_over = (int *)b->d_tile_words.p;
                sp.tile_info = b->d_tile_words.p + n;
                sp.retry_count = b->d_tile_words.p + 2 * n;
                sp.retry_list = b->d_tile_words.p + 2 * n + 4;
                sp.retry_why = b->d_tile_words.p + 3 * n + 4;
                sp.cons_weight = sp.wg_slot ? b->d_tile_words.p + 4 * n + 4 : nullptr; // (no order asked for: none made)
                sp.cons_order = b->d_tile_words.p + 5 * n + 4;
            }
        }
        if (r.split) {
            HIP_TRY(hipMemcpyAsync(b->d_segs.p, r.segs.data(), r.segs.size() * sizeof(SegDesc), hipMemcpyHostToDevice, r.st));
            sp.segs = b->d_segs.p;
            plan_workgroups(r, r.segs.data(), (uint32_t)r.segs.size(), wgs);
            HIP_TRY(hipMemcpyAsync(b->d_wg.p, wgs.data(), wgs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, r.st));
            sp.wg_slot = b->d_wg.p;
            sp.n_wgs = (uint32_t)wgs.size();
        }
        b->last_roles = launch_stream(sp, r.ss, r.st);
        HIP_TRY(hipGetLastError());
        b->h_scal.p[8] = 0;
        if (b->last_roles) // how many captures the run-again launch took (r433_batch_split_stats)
            HIP_TRY(hipMemcpyAsync(b->h_scal.p + 8, sp.retry_count, sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
        if (r.split && (rc = run_stitch(r, sp)))
            return rc;
        if (b->profiling && attempt == 0)
            HIP_TRY(hipEventRecord(b->ev[1], r.st));
        launch_pkg_scan(b->d_state.p, r.d_order, r.n_order, b->d_pkg_base.p, b->d_scal.p, r.st);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(b->h_scal.p, b->d_scal.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
        HIP_TRY(stream_wait(b, r.st));
        r.total_pkgs = b->h_scal.p[0];
        if (b->last_roles)
            b->last_redone = b->h_scal.p[8];
        leg_stamp(b, "detected");
        if (!b->h_scal.p[1])
            break;
        if (attempt >= 6 || b->arena_stride > (1u << 28))
            return fail(R433_EOVERFLOW, "package arena overflow (stride %u)", b->arena_stride);
        b->arena_growth *= 4;
        b->arena_stride = (uint32_t)std::min<uint64_t>((uint64_t)b->arena_stride * 4, 1u << 30);
    }
    if (turn.owns_lock() && b->exclusive_detect < 2)
        turn.unlock(); // level 2 keeps the turn through the slicer kernels (run_slice_and_mirror)
    return 0;
}

// package directory, slicer fan-out, results to the host mirrors
int run_slice_and_mirror(RunCtx &r)
{
    r433_batch *const b = r.b;
    int rc;
    b->n_pkgs = r.total_pkgs;
    b->n_events = 0;
    b->pkg_bytes = b->evt_bytes = 0;
    b->slices_valid = false;
    uint32_t const n_devs = (uint32_t)b->timing.size();
    uint32_t const max_pkgs = std::max<uint32_t>(r.total_pkgs, 1);

    if ((rc = b->d_dir_stream.ensure(max_pkgs)) || (rc = b->d_dir_off.ensure(max_pkgs))
            || (rc = b->d_rec_bytes.ensure(max_pkgs)) || (rc = b->d_rec_off.ensure(max_pkgs))
            || (rc = b->d_pkg_bytes.ensure(max_pkgs)) || (rc = b->d_pkg_off.ensure(max_pkgs))
            || (rc = b->d_sizes.ensure((size_t)max_pkgs * std::max<uint32_t>(n_devs, 1)))
            || (rc = b->d_dev_off.ensure((size_t)max_pkgs * std::max<uint32_t>(n_devs, 1)))
            || (rc = b->d_pkg_order.ensure(max_pkgs)) || (rc = b->d_slice_cursor.ensure(2 * std::max<size_t>(b->rows.size() / 64, 1) + 4)))
        return rc;

    launch_directory(b->d_arena.p, b->arena_stride, b->d_state.p, r.split ? b->d_order.p : nullptr, r.n_order, b->d_pkg_base.p,
            b->d_dir_stream.p, b->d_dir_off.p, b->d_rec_bytes.p, max_pkgs, r.st);
    launch_scan_u32(b->d_rec_bytes.p, b->d_rec_off.p, b->d_scal.p, max_pkgs, b->d_scal.p + 2, r.st);
    HIP_TRY(hipGetLastError());
    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[2], r.st));

    SliceParams lp;
    memset(&lp, 0, sizeof(lp));
    lp.arena = b->d_arena.p;
    lp.arena_stride = b->arena_stride;
    lp.dir_stream = b->d_dir_stream.p;
    lp.dir_off = b->d_dir_off.p;
    lp.n_pkgs = b->d_scal.p;
    lp.devs = b->d_rows.p;
    lp.n_rows = (uint32_t)b->rows.size();
    lp.n_devs = n_devs;
    lp.sizes = b->d_sizes.p;
    lp.dev_off = b->d_dev_off.p;
    lp.pkg_bytes = b->d_pkg_bytes.p;
    lp.pkg_off = b->d_pkg_off.p;
    lp.max_pkgs = max_pkgs;
    SliceFork fork{nullptr, nullptr, nullptr};
    if (!(b->debug_flags & R433_DEBUG_STATIC_SLICE)) {
        lp.draw = (b->debug_flags & R433_DEBUG_ONE_SLICE_LAUNCH) ? 1 : 2;
        lp.pkg_order = b->d_pkg_order.p;
        lp.cursor = b->d_slice_cursor.p;
        if (lp.draw == 2) { // the second stream of the sizing pass (made once per engine)
            if (!b->slice_stream) {
                HIP_TRY(hipStreamCreateWithFlags(&b->slice_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&b->slice_forked, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&b->slice_joined, hipEventDisableTiming));
            }
            fork = SliceFork{b->slice_stream, b->slice_forked, b->slice_joined};
        }
    }
    // The sizing pass shares its workgroups out by the work each chunk of devices had in this engine's runs so far (wavefront
    // clocks, SliceParams::chunk_work -> slice_w, smoothed): with even shares the launch waited for the chunk with
    // the most work -- the PCM slicers of the default decoders, twice the average -- at a seventh of the chip.
    double const *shares = nullptr;
    uint32_t least_share = 8;
    double skewed[2][16];
    if (lp.draw && n_devs && r.total_pkgs && !(b->debug_flags & R433_DEBUG_EVEN_SLICE)) {
        if ((rc = b->d_chunk_work.ensure(2 * 16 * 2 + 1)) || (rc = b->h_chunk_work.ensure(2 * 16 * 2 + 1)) || (rc = b->d_chunk_deal.ensure(2 * 16384)))
            return rc;
        lp.chunk_deal = b->d_chunk_deal.p;
        HIP_TRY(hipMemsetAsync(b->d_chunk_work.p, 0, (2 * 16 * 2 + 1) * sizeof(unsigned long long), r.st));
        lp.chunk_work = b->d_chunk_work.p;
        if (b->debug_flags & R433_DEBUG_SKEW_SLICE) { // tests: unequal shares however small the launch
            for (int l = 0; l < 2; ++l)
                for (int c = 0; c < 16; ++c)
                    skewed[l][c] = 1.0 + 5.0 * ((c + l) % 3);
            shares = &skewed[0][0];
            least_share = 1;
        }
        else if (b->slice_w_valid)
            shares = &b->slice_w[0][0];
    }
    bool placed = false; // the event records are in d_events already (large batches: stretch by stretch)
    bool want_index = false; // the slice index of this run is being made
    lp.pkg_begin = 0;
    lp.pkg_end = max_pkgs;
    b->pf_ran = b->pf_accounted = false;
    if (b->pf_on && n_devs && r.total_pkgs) { // decoder pre-filter (prefilter.cpp): records their decoder provably refuses stay here
        lp.pf_tables = b->d_pf_tables.p;
        lp.pf_counts = b->d_pf_counts.p;
        HIP_TRY(hipMemsetAsync(b->d_pf_counts.p, 0, (size_t)n_devs * 5 * sizeof(uint32_t), r.st));
    }
    if (n_devs && r.total_pkgs) {
        // (a chunk of devices only visits the packages of its own kind: what it never visits has no records)
        if (lp.draw)
            HIP_TRY(hipMemsetAsync(b->d_sizes.p, 0, (size_t)std::min(r.total_pkgs, max_pkgs) * n_devs * sizeof(uint32_t), r.st));
        // One slicing pass into staging slots when they fit.  A default device set yields ~135 B per
        // (package, device) on average, but the heavy PCM rows fill whole bitbuffers -- 50 rows x (4 + 128) B -- and
        // those are exactly the slow lanes: a record that outgrows its slot is sliced a second time by the placing
        // pass (with 4 KB slots that second slicing WAS the placing pass: 0.35 ms of 0.37).  8 KB holds every
        // record there can be; the slots are written sparsely, so their size costs address space, not bandwidth
        // (3.2 GB for the 1024 x 384 rows of the bench batch, of 288 GB).  Below 512 B the classic count + write
        // pair runs instead.
        constexpr size_t kStageMax = (size_t)32 << 30;
        // A batch whose staging slots would not fit the arena's limit goes through the slicers a stretch of packages at a
        // time, every stretch into the SAME slots (slice, scan its sizes on from the total so far, place): full-size slots
        // for any number of packages.  (Not a matter of speed: 8192 packages in one go take 3.9 ms, in eight stretches 5.0.)
        uint32_t const fit = (uint32_t)std::max<size_t>(1, kStageMax / ((size_t)n_devs * 8192u)); // (a slot per device, not per padded row)
        uint32_t const stretch = (b->debug_flags & (R433_DEBUG_TWO_PASS_SLICER | R433_DEBUG_ONE_STRETCH)) ? r.total_pkgs
                : (b->debug_flags & R433_DEBUG_SMALL_STRETCH)                                            ? std::min<uint32_t>(3u, r.total_pkgs)
                                                                                                         : std::min(r.total_pkgs, fit);
        uint32_t stage_cap = b->stage_slot ? b->stage_slot : 8192u;
        if (char const *e = getenv("R433_STAGE_CAP")) // development: A/B timing of smaller slots (records over the slot are sliced again by the placing pass)
            stage_cap = std::max(512, std::min(8192, atoi(e))) & ~511u;
        while (stage_cap >= 512 && (size_t)stretch * n_devs * stage_cap > kStageMax)
            stage_cap >>= 1;
        if (!(b->debug_flags & R433_DEBUG_TWO_PASS_SLICER)) {
            // (a device with less free memory than that: smaller slots -- a record that outgrows its slot is sliced a second
            // time -- and in the end the count + write pair, never a failed run)
            // do not even ask for more than the device has free: ensure() rounds up by a quarter
            size_t mem_free = 0, mem_total = 0;
            // (only when the arena has to grow: the call is a millisecond of driver time with the stream idle)
            if (b->d_stage.cap < (size_t)stretch * n_devs * stage_cap && hipMemGetInfo(&mem_free, &mem_total) == hipSuccess)
                while (stage_cap >= 512 && b->d_stage.cap < (size_t)stretch * n_devs * stage_cap
                        && (size_t)stretch * n_devs * stage_cap / 4 * 5 > mem_free + b->d_stage.cap)
                    stage_cap >>= 1;
            for (; stage_cap >= 512; stage_cap >>= 1) {
                if (b->d_stage.ensure((size_t)stretch * n_devs * stage_cap) == 0) {
                    lp.stage = b->d_stage.p;
                    lp.stage_cap = stage_cap;
                    break;
                }
                (void)hipGetLastError(); // a refused allocation must not fail the launch checks below (sticky on some ROCm releases)
            }
        }
        if (lp.stage && stretch < r.total_pkgs) {
            // The event stream has to be there before the first stretch is placed, and its size is only known after the
            // last: what the engine's earlier runs needed (or ~40 KB a package) is taken, records beyond it are left out
            // by the placing pass (bounds-checked), and a total that did not fit means one more round with the exact size.
            size_t want = std::max<size_t>(b->d_events.cap, (size_t)r.total_pkgs * 40960u);
            for (int round = 0;; ++round) {
                if ((rc = b->d_events.ensure(std::min<size_t>(want, 0xf0000010ull))))
                    return rc;
                lp.events = b->d_events.p;
                lp.events_cap = (uint32_t)std::min<size_t>(b->d_events.cap, 0xffffffffu);
                HIP_TRY(hipMemsetAsync(b->d_pkg_bytes.p, 0, (size_t)max_pkgs * sizeof(uint32_t), r.st));
                HIP_TRY(hipMemsetAsync(b->d_scal.p + 3, 0, sizeof(uint32_t), r.st));
                if (lp.pf_counts)
                    HIP_TRY(hipMemsetAsync(b->d_pf_counts.p, 0, (size_t)n_devs * 5 * sizeof(uint32_t), r.st));
                for (uint32_t p0 = 0; p0 < r.total_pkgs; p0 += stretch) {
                    lp.pkg_begin = p0;
                    lp.pkg_end = std::min(r.total_pkgs, p0 + stretch);
                    launch_slice_count(lp, lp.pkg_end - p0, r.st, fork.st2 ? &fork : nullptr, shares, least_share);
                    launch_scan_u32(b->d_pkg_bytes.p, b->d_pkg_off.p, b->d_scal.p, lp.pkg_end, b->d_scal.p + 3, r.st, b->d_scal.p + 3, p0);
                    launch_slice_write(lp, lp.pkg_end - p0, r.st);
                }
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(b->h_scal.p, b->d_scal.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
                HIP_TRY(stream_wait(b, r.st));
                size_t const total = b->h_scal.p[3];
                if (total + 16 <= b->d_events.cap || total > 0xf0000000ull)
                    break;
                if (round >= 2) // sizes are a function of the packages: a third round cannot differ from the second
                    return fail(R433_EOVERFLOW, "event records (%zu bytes) do not fit the buffer sized for them (%zu)", total, b->d_events.cap);
                want = total + total / 8 + 16;
            }
            placed = true;
            if (b->profiling)
                HIP_TRY(hipEventRecord(b->ev[3], r.st));
        }
        else {
            HIP_TRY(hipMemsetAsync(b->d_pkg_bytes.p, 0, (size_t)max_pkgs * sizeof(uint32_t), r.st));
            launch_slice_count(lp, r.total_pkgs, r.st, fork.st2 ? &fork : nullptr, shares, least_share);
            HIP_TRY(hipGetLastError());
            if (b->profiling)
                HIP_TRY(hipEventRecord(b->ev[3], r.st));
            launch_scan_u32(b->d_pkg_bytes.p, b->d_pkg_off.p, b->d_scal.p, max_pkgs, b->d_scal.p + 3, r.st);
        }
        // the slice index: how many non-empty (package, decoder) slices each decoder has (the sizes are final here)
        uint32_t const idx_blocks = slice_index_blocks(r.total_pkgs);
        if ((rc = b->d_idx_cnt.ensure((size_t)idx_blocks * n_devs + 16)) || (rc = b->d_slice_start.ensure(n_devs + 16)))
            return rc;
        launch_slice_index_count(b->d_sizes.p, b->d_scal.p, max_pkgs, r.total_pkgs, n_devs, b->d_idx_cnt.p, b->d_slice_start.p, b->d_scal.p + 4, r.st);
        HIP_TRY(hipGetLastError());
        want_index = true;
    }
    else {
        HIP_TRY(hipMemsetAsync(b->d_scal.p + 3, 0, sizeof(uint32_t), r.st));
        if (b->profiling)
            HIP_TRY(hipEventRecord(b->ev[3], r.st));
    }
    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[4], r.st));
    HIP_TRY(hipMemcpyAsync(b->h_scal.p, b->d_scal.p, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
    if (lp.chunk_work)
        HIP_TRY(hipMemcpyAsync(b->h_chunk_work.p, b->d_chunk_work.p, (2 * 16 * 2 + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, r.st));
    HIP_TRY(stream_wait(b, r.st));
    if (lp.chunk_work && b->h_chunk_work.p[64]) { // what the chunks' work was this time: half of the next run's shares (small launches measure nothing)
        unsigned long long const began = b->h_chunk_work.p[64];
        bool any = false;
        double w[2][16];
        for (int l = 0; l < 2; ++l)
            for (int c = 0; c < 16; ++c) {
                unsigned long long const left = b->h_chunk_work.p[(l * 16 + c) * 2], n = b->h_chunk_work.p[(l * 16 + c) * 2 + 1];
                // (a wavefront leaves when its chunk's list is dry: n wavefronts busy until the last one left)
                w[l][c] = left > began ? (double)(left - began) * (double)n : 0.0;
                any |= w[l][c] > 0;
            }
        if (any) {
            for (int l = 0; l < 2; ++l)
                for (int c = 0; c < 16; ++c)
                    b->slice_w[l][c] = b->slice_w_valid ? 0.5 * b->slice_w[l][c] + 0.5 * w[l][c] : w[l][c];
            b->slice_w_valid = true;
        }
    }
    size_t const pkg_bytes = b->h_scal.p[2];
    size_t const evt_bytes = b->h_scal.p[3];
    uint32_t const n_slices = want_index ? b->h_scal.p[4] : 0u;
    // the scans saturate at 0xffffffff (k_scan_u32): offsets are 32-bit by format, a larger batch has to be split
    if (evt_bytes > 0xf0000000ull)
        return fail(R433_EOVERFLOW, "event stream of this batch exceeds 3.75 GiB: run it in smaller batches");
    if (pkg_bytes > 0xf0000000ull)
        return fail(R433_EOVERFLOW, "package records of this batch exceed 3.75 GiB: run it in smaller batches");

    if ((rc = b->d_pkg_blob.ensure(pkg_bytes + 16)) || (rc = b->h_pkg_blob.ensure(pkg_bytes + 16))
            || (rc = b->d_events.ensure(evt_bytes + 16)) || (rc = b->h_events.ensure(evt_bytes + 16))
            || (rc = b->h_frame_sums.ensure((size_t)r.n_streams * r.frames_cap)) || (rc = b->h_pkg_off.ensure(max_pkgs + 1))
            || (rc = b->h_rec_off.ensure(max_pkgs + 1)))
        return rc;
    if (r.total_pkgs) {
        launch_gather_packages(b->d_arena.p, b->arena_stride, b->d_dir_stream.p, b->d_dir_off.p, b->d_rec_off.p,
                b->d_scal.p, max_pkgs, b->d_pkg_blob.p, (uint32_t)std::min<size_t>(b->d_pkg_blob.cap, 0xffffffffu),
                r.total_pkgs, r.st);
        if (n_devs && evt_bytes && !placed) {
            lp.events = b->d_events.p;
            lp.events_cap = (uint32_t)std::min<size_t>(b->d_events.cap, 0xffffffffu);
            launch_slice_write(lp, r.total_pkgs, r.st);
        }
        if (want_index && n_slices) { // where every decoder's slices lie, now that the offsets are final (k_dev_prefix)
            if ((rc = b->d_slices.ensure(n_slices)) || (rc = b->h_slices.ensure(n_slices)) || (rc = b->h_slice_start.ensure(n_devs + 16)))
                return rc;
            launch_slice_index_fill(b->d_sizes.p, b->d_dev_off.p, b->d_pkg_off.p, b->d_scal.p, max_pkgs, r.total_pkgs, n_devs, b->d_idx_cnt.p,
                    b->d_slices.p, n_slices, r.st);
        }
        HIP_TRY(hipGetLastError());
    }
    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[5], r.st));
    if (r.turn.owns_lock() && b->exclusive_detect < 3) { // exclusive level 2: the kernels of this pass are done before the next engine's begin
        HIP_TRY(stream_wait(b, r.st));
        r.turn.unlock();
        leg_stamp(b, "sliced");
    }
    if (pkg_bytes)
        HIP_TRY(hipMemcpyAsync(b->h_pkg_blob.p, b->d_pkg_blob.p, pkg_bytes, hipMemcpyDeviceToHost, r.st));
    if (evt_bytes)
        HIP_TRY(hipMemcpyAsync(b->h_events.p, b->d_events.p, evt_bytes, hipMemcpyDeviceToHost, r.st));
    if (r.total_pkgs) {
        HIP_TRY(hipMemcpyAsync(b->h_rec_off.p, b->d_rec_off.p, r.total_pkgs * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
        if (n_devs)
            HIP_TRY(hipMemcpyAsync(b->h_pkg_off.p, b->d_pkg_off.p, r.total_pkgs * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
    }
    HIP_TRY(hipMemcpyAsync(b->h_frame_sums.p, b->d_frame_sums.p, (size_t)r.n_streams * r.frames_cap * sizeof(uint32_t),
            hipMemcpyDeviceToHost, r.st));
    if (lp.pf_counts)
        HIP_TRY(hipMemcpyAsync(b->h_pf_counts.p, b->d_pf_counts.p, (size_t)n_devs * 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
    if (want_index && n_slices) {
        HIP_TRY(hipMemcpyAsync(b->h_slices.p, b->d_slices.p, (size_t)n_slices * sizeof(uint2), hipMemcpyDeviceToHost, r.st));
        HIP_TRY(hipMemcpyAsync(b->h_slice_start.p, b->d_slice_start.p, (size_t)(n_devs + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, r.st));
    }
    if (b->logic_on && r.d_iq)
        HIP_TRY(hipMemcpyAsync(b->h_logic.p, b->d_logic.p, (size_t)r.n_streams * b->logic_stride, hipMemcpyDeviceToHost, r.st));
    if (b->profiling)
        HIP_TRY(hipEventRecord(b->ev[6], r.st));
    HIP_TRY(stream_wait(b, r.st));
    if (r.turn.owns_lock()) // exclusive level 3: the record copies, too, have the device to themselves
        r.turn.unlock();
    leg_stamp(b, "mirrored");
    b->pkg_bytes = pkg_bytes;
    if (b->arena_growth > 1) { // (see arena_growth: a grown stride is given back when the captures stopped needing it)
        uint64_t const slots = std::max<uint32_t>(1u, r.split ? r.n_order : r.n_streams);
        b->calm_runs = pkg_bytes / slots < b->arena_stride / 16u ? b->calm_runs + 1 : 0;
        if (b->calm_runs >= 8) {
            b->arena_growth /= 4;
            b->calm_runs = 0;
        }
    }
    b->evt_bytes = evt_bytes;
    b->n_slices = n_slices;
    b->slices_valid = want_index && n_slices > 0;
    b->pf_ran = lp.pf_counts != nullptr;
    b->events_counted = false;
    b->dispatched = false;
    b->pkg_quality.clear();
    b->pkg_decoded.clear();
    b->h_rec_off.p[r.total_pkgs] = (uint32_t)pkg_bytes;
    b->h_pkg_off.p[r.total_pkgs] = (uint32_t)evt_bytes;
    if (!n_devs)
        for (uint32_t i = 0; i < r.total_pkgs; ++i)
            b->h_pkg_off.p[i] = 0;

    if (b->profiling) {
        r433_batch_timing &t = b->last_timing;
        (void)hipEventElapsedTime(&t.detect_ms, b->ev[0], b->ev[1]);
        (void)hipEventElapsedTime(&t.dir_ms, b->ev[1], b->ev[2]);
        (void)hipEventElapsedTime(&t.count_ms, b->ev[2], b->ev[3]);
        (void)hipEventElapsedTime(&t.scan_ms, b->ev[3], b->ev[4]);
        (void)hipEventElapsedTime(&t.write_ms, b->ev[4], b->ev[5]);
        (void)hipEventElapsedTime(&t.d2h_ms, b->ev[5], b->ev[6]);
        (void)hipEventElapsedTime(&t.total_ms, b->ev[0], b->ev[6]);
    }
    return (int)r.total_pkgs;
}

} // namespace

extern "C" {

int r433_batch_run(r433_batch *b, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes,
        uint32_t n_streams, void *stream)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    DeviceScope on_device(b->device);
    if (n_streams == 0) {
        b->n_streams = 0;
        b->n_pkgs = b->n_events = 0;
        b->pkg_bytes = b->evt_bytes = 0;
        b->slices_valid = false;
        return 0;
    }
    if (!d_iq || (stride_bytes & 15u) || ((uintptr_t)d_iq & 15u))
        return fail(R433_EINVAL, "capture base and stride must be 16-byte aligned");
    if (stride_bytes > 0xfffffff0ull)
        return fail(R433_EINVAL, "captures are limited to 4 GiB each");
    RunCtx r;
    r.b = b;
    r.st = (hipStream_t)stream;
    r.ss = b->cfg.sample_size;
    r.d_iq = d_iq;
    r.stride_bytes = stride_bytes;
    r.stream_bytes = stream_bytes;
    r.n_streams = n_streams;
    int rc;
    if ((rc = run_convert_input(r)) || (rc = run_size_buffers(r)) || (rc = run_autolevel(r)) || (rc = run_plan(r))
            || (rc = run_detect(r)))
        return rc;
    return run_slice_and_mirror(r);
}

void *r433_host_alloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) {
        fail(e == hipErrorNoDevice ? R433_ENODEV : R433_ENOMEM, "hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

void r433_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

int r433_host_register(void *p, size_t bytes)
{
    if (!p || !bytes)
        return fail(R433_EINVAL, "nothing to register");
    hipError_t const e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorNoDevice ? R433_ENODEV : R433_EHIP, "hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    return 0;
}

int r433_host_unregister(void *p)
{
    if (!p)
        return 0;
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(R433_EHIP, "hipHostUnregister");
    }
    return 0;
}

int r433_batch_run_host(r433_batch *b, void const *const *h_captures, uint32_t const *capture_bytes, uint32_t n_captures)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    DeviceScope on_device(b->device);
    if (n_captures == 0)
        return r433_batch_run(b, nullptr, 0, nullptr, 0, nullptr);
    if (!h_captures || !capture_bytes)
        return fail(R433_EINVAL, "null capture list");
    uint64_t max_bytes = 0;
    bool packed = true; // contiguous and equally long: one copy
    for (uint32_t i = 0; i < n_captures; ++i) {
        if (capture_bytes[i] && !h_captures[i])
            return fail(R433_EINVAL, "capture %u is null", i);
        max_bytes = std::max<uint64_t>(max_bytes, capture_bytes[i]);
        if (capture_bytes[i] != capture_bytes[0] || (capture_bytes[0] & 15u)
                || (uint8_t const *)h_captures[i] != (uint8_t const *)h_captures[0] + (size_t)i * capture_bytes[0])
            packed = false;
    }
    if (max_bytes == 0)
        packed = false; // nothing to copy: the pointers may all be null (an empty file is a valid capture)
    uint64_t const stride = std::max<uint64_t>(16, (max_bytes + 15) & ~15ull);
    if (stride > 0xfffffff0ull)
        return fail(R433_EINVAL, "captures are limited to 4 GiB each");
    if (int rc = b->d_input.ensure((size_t)n_captures * stride + 16))
        return rc;
    if (!b->own_stream)
        HIP_TRY(hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking));
    hipStream_t const st = b->own_stream;
    if (packed) {
        HIP_TRY(hipMemcpyAsync(b->d_input.p, h_captures[0], (size_t)n_captures * stride, hipMemcpyHostToDevice, st));
    }
    else {
        for (uint32_t i = 0; i < n_captures; ++i)
            if (capture_bytes[i])
                HIP_TRY(hipMemcpyAsync(b->d_input.p + (size_t)i * stride, h_captures[i], capture_bytes[i], hipMemcpyHostToDevice, st));
    }
    b->host_bytes.assign(capture_bytes, capture_bytes + n_captures);
    return r433_batch_run(b, b->d_input.p, stride, b->host_bytes.data(), n_captures, st);
}

int r433_batch_run_pulses(r433_batch *b, r433_pulse_data const *pulses, uint32_t n_packages, void *stream)
{
    if (!b)
        return fail(R433_EINVAL, "null batch");
    DeviceScope on_device(b->device);
    if (n_packages == 0) {
        b->n_streams = 0;
        b->n_pkgs = b->n_events = 0;
        b->pkg_bytes = b->evt_bytes = 0;
        b->slices_valid = false;
        return 0;
    }
    if (!pulses)
        return fail(R433_EINVAL, "null pulse data");
    RunCtx r;
    r.b = b;
    r.st = (hipStream_t)stream;
    r.ss = b->cfg.sample_size;
    r.d_iq = nullptr;
    r.stride_bytes = 0;
    r.stream_bytes = nullptr;
    r.n_streams = n_packages;
    r.frames_cap = 1;
    r.n_order = n_packages;
    b->stream_samples.clear();
    // one arena slot per package, laid out exactly as the detection kernel leaves a capture with one package
    uint32_t max_pulses = 0;
    for (uint32_t k = 0; k < n_packages; ++k) {
        if (pulses[k].num_pulses > R433_PD_MAX_PULSES)
            return fail(R433_EINVAL, "package %u has %u pulses (at most %u)", k, pulses[k].num_pulses, (unsigned)R433_PD_MAX_PULSES);
        if (pulses[k].sample_rate && pulses[k].sample_rate != b->cfg.samp_rate)
            return fail(R433_EINVAL, "package %u is at %u samples/s, the batch at %u", k, pulses[k].sample_rate, b->cfg.samp_rate);
        max_pulses = std::max(max_pulses, pulses[k].num_pulses);
    }
    uint32_t const stride = ((uint32_t)sizeof(r433_pkg_rec) + 8u * max_pulses + 15u) & ~15u;
    int rc;
    if ((rc = b->d_arena.ensure((size_t)n_packages * stride)) || (rc = b->d_state.ensure(n_packages))
            || (rc = b->d_pkg_base.ensure(n_packages)) || (rc = b->d_frame_sums.ensure(n_packages))
            || (rc = b->h_arena_stage.ensure((size_t)n_packages * stride)) || (rc = b->h_state.ensure(n_packages)))
        return rc;
    b->arena_stride = stride;
    b->frames_cap = 1;
    b->n_streams = n_packages;
    memset(b->h_arena_stage.p, 0, (size_t)n_packages * stride);
    memset(b->h_state.p, 0, (size_t)n_packages * sizeof(StreamState));
    for (uint32_t k = 0; k < n_packages; ++k) {
        r433_pulse_data const &pd = pulses[k];
        uint8_t *rec = b->h_arena_stage.p + (size_t)k * stride;
        r433_pkg_rec h;
        memset(&h, 0, sizeof(h));
        h.total_bytes = (uint32_t)sizeof(h) + 8u * pd.num_pulses;
        h.stream = k;
        h.type = pd.fsk_f2_est ? R433_PKG_FSK : R433_PKG_OOK; // as the reference decides, src/rtl_433.c:1774
        h.num_pulses = pd.num_pulses;
        h.offset = pd.offset;
        h.start_ago = pd.start_ago;
        h.end_ago = pd.end_ago;
        h.ook_low = pd.ook_low_estimate;
        h.ook_high = pd.ook_high_estimate;
        h.fsk_f1 = pd.fsk_f1_est;
        h.fsk_f2 = pd.fsk_f2_est;
        h.sample_rate = pd.sample_rate ? pd.sample_rate : b->cfg.samp_rate;
        memcpy(rec, &h, sizeof(h));
        int32_t *pairs = (int32_t *)(rec + sizeof(h));
        for (uint32_t i = 0; i < pd.num_pulses; ++i) {
            pairs[2 * i] = pd.pulse[i];
            pairs[2 * i + 1] = pd.gap[i];
        }
        b->h_state.p[k].n_pkgs = 1;
        b->h_state.p[k].cursor = h.total_bytes;
    }
    if (b->profiling)
        for (int e = 0; e < 2; ++e)
            HIP_TRY(hipEventRecord(b->ev[e], r.st));
    HIP_TRY(hipMemcpyAsync(b->d_arena.p, b->h_arena_stage.p, (size_t)n_packages * stride, hipMemcpyHostToDevice, r.st));
    HIP_TRY(hipMemcpyAsync(b->d_state.p, b->h_state.p, (size_t)n_packages * sizeof(StreamState), hipMemcpyHostToDevice, r.st));
    HIP_TRY(hipMemsetAsync(b->d_frame_sums.p, 0, (size_t)n_packages * sizeof(uint32_t), r.st));
    launch_pkg_scan(b->d_state.p, nullptr, n_packages, b->d_pkg_base.p, b->d_scal.p, r.st);
    HIP_TRY(hipGetLastError());
    r.total_pkgs = n_packages;
    return run_slice_and_mirror(r);
}

} // extern "C"
