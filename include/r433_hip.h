/* r433_hip.h -- C ABI of librtl433hip.so: the MI355X (gfx950) implementation of rtl_433's
 * IQ -> pulse package -> bitbuffer -> decoder fan-out path.
 *
 * Plain C, pointers and sizes only.  Device pointers are ordinary HIP device addresses (e.g.
 * torch.Tensor.data_ptr()); `stream` arguments are a hipStream_t passed as void* (NULL = default
 * stream).  Every entry point names the reference interface it stands in for (paths relative to the
 * reference tree).  All functions return 0 / a count on success and a negative R433_E* code on
 * failure; r433_last_error() gives the text.  Nothing here falls back to a CPU implementation: without
 * a usable HIP device every compute call fails with R433_ENODEV.
 */
#ifndef R433_HIP_H_
#define R433_HIP_H_

#include <stddef.h>
#include <stdint.h>

#include "r433_abi.h"
#include "r433_records.h"

#ifdef __cplusplus
extern "C" {
#endif

#define R433_OK 0
#define R433_EINVAL (-1)
#define R433_ENODEV (-2)
#define R433_EHIP (-3)
#define R433_ENOMEM (-4)
#define R433_EOVERFLOW (-5)
#define R433_EDECODER (-6) /* a decode_fn returned an invalid code (reference src/pulse_slicer.c:44-47 exits) */

int r433_version(void);
char const *r433_last_error(void);
/* number of visible HIP devices, or R433_ENODEV */
int r433_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Flow configuration: the fields of struct dm_state / r_cfg that parameterise the hot path
 * (reference include/r_private.h:19-86, set per frame in src/rtl_433.c:1094-1120).
 */
typedef struct r433_flow_cfg {
    uint32_t sample_size;   /* 2 = cu8, 4 = cs16 (dm_state.sample_size) */
    uint32_t samp_rate;     /* dm_state.samp_rate */
    uint32_t frame_samples; /* samples per push_sdr_flow call; 0 = file default 262144 / sample_size */
    uint32_t fpdm;          /* resolved FSK detector: 0 classic, 1 minmax (src/rtl_433.c:1094-1102) */
    uint32_t use_mag_est;   /* -Y magest */
    uint32_t enable_fm;     /* dm_state.enable_FM_demod; 0 reproduces the buf.fm/buf.temp aliasing */
    float fm_low_pass;      /* -Y filter, 0 = 0.1 (classic) / 0.2 (minmax), src/r_flow.c:204 */
    float level_limit_db;   /* -Y level, 0 = adaptive */
    float min_level_db;     /* -Y minlevel, default -12.1442 */
    float min_snr_db;       /* -Y minsnr, default 9.0 */
    float auto_level;       /* -Y autolevel > 0 */
    uint32_t center_frequency; /* only for reporting (freq fields of calc_rssi_snr) */
    uint32_t input_format;  /* R433_IN_NATIVE, or a file format the reference converts on load (src/rtl_433.c:1811-1834):
                             * R433_IN_CS8 (sample_size 2: int8 pairs -> cu8) / R433_IN_CF32 (sample_size 4: float pairs -> cs16).
                             * Strides and lengths given to r433_batch_run are then in the input format's bytes.
                             * R433_IN_S16_AM / R433_IN_S16_FM (sample_size 2): the reference's am.s16 / fm.s16 input files
                             * (file_info S16_AM / S16_FM).  Their bytes go through the flow as if they were cu8 pairs -- frame
                             * level, squelch and the other demodulator see exactly that, src/rtl_433.c:1735-1739 -- and the
                             * int16 words then stand in for the filtered envelope / the filtered discriminator in front of the
                             * pulse detector (src/r_flow.c:212-225). */
} r433_flow_cfg;
#define R433_IN_NATIVE 0u
#define R433_IN_CS8 1u
#define R433_IN_CF32 2u
#define R433_IN_S16_AM 3u
#define R433_IN_S16_FM 4u

void r433_flow_cfg_default(r433_flow_cfg *cfg, uint32_t sample_size, uint32_t samp_rate);

/* ------------------------------------------------------------------------------------------------
 * Batch engine: N independent captures ("-r a.cu8 -r b.cu8 ..." with reset_sdr_flow between,
 * reference src/rtl_433.c:1703-1854) decoded in one go.  Stands in for push_sdr_flow /
 * flush_sdr_flow / reset_sdr_flow (include/r_flow.h:19-23) + run_ook_demods / run_fsk_demods
 * (include/r_api.h:50-52) over a whole file list.
 */
typedef struct r433_batch r433_batch;

/* devs: timing rows of the registered r_devices in registration order (may be 0 rows: detection only) */
r433_batch *r433_batch_create(r433_flow_cfg const *cfg, r433_dev_timing const *devs, uint32_t n_devs);
/* The same on GPU `device` (0 .. r433_device_count() - 1) instead of the calling thread's current one.  The engine stays on
 * that GPU: every later call on it switches the calling thread to the engine's device for the time of the call (and back),
 * so one host process can keep an engine per GPU and drive them from any thread -- the batch semantics of the reference's
 * file loop (`-r a -r b ...` with reset_sdr_flow between files, src/rtl_433.c:1703-1854) spread over the GPUs of a node by a
 * plain C host, no collective library involved (dropin/pipeline_host.c --gpus N).  Device pointers handed
 * to r433_batch_run must belong to that GPU; r433_batch_run_host takes host memory and is the portable form.
 * Returns NULL (R433_ENODEV) for a device that is not there. */
r433_batch *r433_batch_create_on(int device, r433_flow_cfg const *cfg, r433_dev_timing const *devs, uint32_t n_devs);
/* the GPU an engine lives on */
int r433_batch_device(r433_batch *b);
void r433_batch_destroy(r433_batch *b);

/* Decode n_streams captures resident in device memory.  Capture s starts at d_iq + s*stride_bytes
 * (stride and base 16-byte aligned) and holds stream_bytes[s] bytes (host array; NULL = all
 * stride_bytes).  Runs detection and the slicer fan-out, leaves package and event records in device
 * memory and mirrors them to pinned host memory.  Returns the number of packages. */
int r433_batch_run(r433_batch *b, void const *d_iq, uint64_t stride_bytes, uint32_t const *stream_bytes,
        uint32_t n_streams, void *stream);

/* The same pass for a C host that keeps its captures in host memory (the `-r` file loop, reference
 * src/rtl_433.c:1703-1859, reads every file into one host buffer): capture s is the capture_bytes[s] bytes at
 * h_captures[s].  The library stages them into device memory of its own (grow-only; one copy when the captures are
 * contiguous and equally long, one per capture otherwise) on an internal stream and runs r433_batch_run there.
 * Memory from r433_host_alloc is pinned, which makes the copy asynchronous and roughly twice as fast; any host
 * memory works.  Returns the number of packages. */
void *r433_host_alloc(size_t bytes);
void r433_host_free(void *p);
/* ... or memory of the host's own, page-locked in place (0 = done): 13-15 ms per 256 MiB of touched memory where allocating the
 * same pinned costs 46-60 ms, and it can be done late, on the thread that starts the pass (dropin/r_flow_hip.c stages into
 * plain memory and registers a buffer the first time a pass reads from it).  Unregister before the memory is freed. */
int r433_host_register(void *p, size_t bytes);
int r433_host_unregister(void *p);
/* Opens the device and loads the library's kernels (what the first call of a process otherwise pays: 70-280 ms), so that a host
 * can do it on a thread beside its own start-up.  0, or R433_ENODEV / R433_EHIP. */
int r433_warmup(void);
int r433_batch_run_host(r433_batch *b, void const *const *h_captures, uint32_t const *capture_bytes, uint32_t n_captures);

/* Host views of the last run (valid until the next run/destroy). */
int r433_batch_packages(r433_batch *b, uint8_t const **blob, size_t *len, uint32_t *count);
int r433_batch_events(r433_batch *b, uint8_t const **blob, size_t *len, uint32_t *count);
/* per (capture, frame) wrapped u32 envelope sums -> avg_db (reference src/baseband.c:39-44) */
int r433_batch_frame_sums(r433_batch *b, uint32_t const **sums, uint32_t *frames_cap);
/* device-side record arenas of the last run (for consumers that stay on the GPU) */
int r433_batch_device_events(r433_batch *b, void const **d_events, size_t *len);

/* The `u8` logic dump (-w file.u8 / U8_LOGIC, reference src/r_flow.c:236-237,271-272,314-315,364-371,480, src/pulse_data.c:58-67):
 * one byte per sample -- 0x01 where a package's pulse/gap list covers the sample, | 0x02 on the pulses of OOK packages, | 0x04
 * on the pulses of FSK packages (and of the FSK candidate that every OOK package carries through its first pulse) -- painted
 * by the detection kernel at the moments the reference paints (package returns, frame ends) with its clipping and overwrite
 * order, so that the bytes equal the reference's file.  With it on, captures are not split (r433_batch_set_split is ignored).
 * r433_batch_logic_dump gives the host copy of the last run: capture s at host + s * stride, as many bytes as it has samples. */
int r433_batch_enable_logic_dump(r433_batch *b, int on);
int r433_batch_logic_dump(r433_batch *b, uint8_t const **host, uint64_t *stride);

/* Optional parity taps: per-sample envelope (u16), low-passed envelope (s16) and FM (s16) of every
 * capture, written to device buffers of n_streams*tap_stride samples.  Pass NULLs to disable. */
int r433_batch_set_taps(r433_batch *b, void *d_env, void *d_am, void *d_fm, uint64_t tap_stride);

/* Long captures: let several wavefronts work on one capture.  A capture longer than segment_samples is cut
 * about every segment_samples where the signal looks idle.  0 = never; R433_SPLIT_AUTO (the default) = only for
 * batches of at most 64 captures with one of at least 2^20 samples, aiming at ~4096 segments of >= 32768 samples.  Cuts are speculative
 * and verified: every later segment is run for both parities of the noise-floor estimate and kept only if the
 * segment before it really ended idle with exactly the floor it assumed; a cut that does not verify is
 * dropped and the piece before it is run again across it, so results never depend on this setting.  There is no reference
 * counterpart: the reference walks a file sample by sample (src/rtl_433.c:1826-1845). */
#define R433_SPLIT_AUTO 1u
int r433_batch_set_split(r433_batch *b, uint32_t segment_samples);
/* Several engines of one process, each on its own stream (a software pipeline over batches): with this set, their
 * detection kernels take turns instead of sharing the compute units -- a launch fills every SIMD by itself, two of them
 * side by side only stretch each other -- while everything after detection (slicers, record copies) still overlaps the
 * next engine's detection.  on = 2: the turn lasts until the slicer kernels of the pass are done as well (they, too, fill
 * the chip by themselves; record copies and the host replay still overlap the next engine's kernels).  on = 3: until the
 * records are on the host -- nothing of two passes overlaps on the device; for profiling (under rocprofv3 a copy is a blit
 * kernel that takes compute units from the kernel beside it; outside the profiler copies run on the SDMA engines and level 2
 * is the faster setting).  Off by default: a lone engine has nobody to wait for. */
int r433_batch_set_exclusive_detect(r433_batch *b, int on);
/* Bytes of the staging slot the slicers build a (package, device)'s records in before they are placed (512 .. 8192 in steps of 512;
 * 0 = the default: 8192, which holds every record there can be -- 50 rows x 132 bytes).  A record that outgrows its slot is
 * sliced a second time by the placing pass: same results, a little more kernel time per pass (+0.8 ms per 8192 packages at
 * 2 KB) -- and a quarter of the device memory: 335 decoders x 2100 packages are 5.6 GB of slots at 8 KB.  For hosts that live as
 * long as one file list: a process that leaves 14 GB of device memory behind slows the NEXT process's start while the driver
 * frees them (the drop-in CLI back to back: 330-1030 ms per run at 8 KB, 250-400 ms at 2 KB: profiles/r06_g_cli_series.txt). */
int r433_batch_set_staging_slot(r433_batch *b, uint32_t bytes);
/* of the last run: wavefront slots planned (segments incl. parity variants), pieces run again after a dropped cut */
int r433_batch_split_stats(r433_batch *b, uint32_t *segments, uint32_t *pieces_rerun);
/* How the last detection pass was launched: 45 = the producers of the grid (filters; filtered tiles to HBM) and its consumers
 * (detector) as two launches, plus a third for the captures that have to run again with every tile filtered (their number is
 * what r433_batch_split_stats reports as pieces_rerun then); 0 = one launch, the roles of a capture in one workgroup.  Results
 * do not depend on it (R433_DEBUG_SPLIT_ROLES / R433_DEBUG_NO_SPLIT_ROLES choose; default: grids of 6144 captures and more). */
int r433_batch_detect_form(r433_batch *b);

/* Kernel timing of the last run measured with HIP events on the caller's stream (ms). */
typedef struct r433_batch_timing {
    float detect_ms;  /* k_wave (+ the grid-ordering look and, for split captures, k_tile_max and the stitch rounds): IQ -> packages */
    float dir_ms;     /* package directory */
    float count_ms;   /* slicer pass 1 */
    float scan_ms;
    float write_ms;   /* slicer pass 2 */
    float d2h_ms;     /* record copies */
    float total_ms;
} r433_batch_timing;
int r433_batch_set_profiling(r433_batch *b, int on);
/* Debugging aid: raw per-capture kernel state of the last run (n_streams records; returns the record size).
 * With R433_DEBUG_TIMING (r433_batch_set_debug) the detection kernel leaves per-phase clock ticks in it (tools/kbench.py). */
int r433_batch_debug_state(r433_batch *b, void *host_buf, size_t bytes);
/* Development switches, off unless asked for here (no environment variable changes what the library computes;
 * R433_TRACE_LEGS=1 only makes r433_batch_run print the stages of a pass with the monotonic clock to stderr).
 * The first three never change results; SKIP_* leave the results of a run INCOMPLETE (kernel timing experiments). */
#define R433_DEBUG_SPLIT_BLIND 1u      /* split captures at fixed distances instead of where they look idle (tests) */
#define R433_DEBUG_TWO_PASS_SLICER 2u  /* count + write slicer passes instead of staging slots */
#define R433_DEBUG_SPLIT_TRACE 4u      /* stderr trace of dropped cuts */
#define R433_DEBUG_DISPATCH_TRACE 32u   /* stderr trace of the ordered replay: index / levels / commit times, the slowest decoders */
#define R433_DEBUG_NO_ORDER 64u         /* large grids in capture order instead of heaviest first (A/B timing) */
#define R433_DEBUG_FORCE_ORDER 128u     /* heaviest first whatever the size of the grid (tests) */
#define R433_DEBUG_ONE_STRETCH 16u     /* the slicer fan-out over all packages at once, however many (A/B timing) */
#define R433_DEBUG_SMALL_STRETCH 8u    /* slicer fan-out three packages at a time (what batches over 1536 packages do 1024 at a time; tests) */
#define R433_DEBUG_SKIP_DETECT 256u
#define R433_DEBUG_SKIP_FILTERS 512u
#define R433_DEBUG_TIMING 1024u        /* per-phase shader clocks into the debug state (tools/kbench.py) */
#define R433_DEBUG_NO_ROLE_SWAP 8192u /* wavefront 0 of every workgroup produces (A/B timing) */
#define R433_DEBUG_NO_PRIO 16384u     /* the consumer wavefront keeps the default issue priority (A/B timing) */
#define R433_DEBUG_PAIR 32768u        /* a producer / consumer pair per capture also in launches of more than 1280 captures */
#define R433_DEBUG_ONE_WAVE 4096u /* one wavefront per capture instead of a producer / consumer pair: same results, for A/B timing */
#define R433_DEBUG_ONE_SLICE_LAUNCH 131072u /* the slicers' sizing pass as one launch (9.6 KB of LDS per workgroup) instead of large and small packages apart (A/B timing) */
#define R433_DEBUG_STATIC_SLICE 65536u /* slicer workgroups take their packages at fixed strides instead of heaviest first from a shared cursor (A/B timing) */
#define R433_DEBUG_NAP_WAIT 2097152u /* wait for the GPU by polling with naps instead of the runtime's spinning hipEventSynchronize: for hosts with fewer CPUs than waiting threads (bench.py sets it for the ranks of a node that share a small CPU quota) */
#define R433_DEBUG_EVEN_SLICE 524288u /* the slicers' sizing pass gives every chunk of devices the same number of workgroups instead of shares by measured work (A/B timing) */
#define R433_DEBUG_SKEW_SLICE 1048576u /* ... and shares by a made-up skew, however small the launch (tests) */
#define R433_DEBUG_NO_LAZY 262144u /* the detection kernel filters every tile, also those that provably cannot move the detector: same results, for A/B timing */
#define R433_DEBUG_NO_TRAIN_ENGINE 2048u /* in-package legs through the older per-leg code: same results, for A/B timing */
#define R433_DEBUG_SPLIT_ROLES 4194304u    /* the detection pass as a launch of producers (filters; filtered tiles to HBM) and a launch of consumers (detector) whatever the size of the grid (default: from 6144 captures on); same results (tests, A/B timing) */
#define R433_DEBUG_NO_SPLIT_ROLES 8388608u /* ... never: producer / consumer pairs in one workgroup as in round 4 (A/B timing) */
int r433_batch_set_debug(r433_batch *b, uint32_t flags);
int r433_batch_get_timing(r433_batch *b, r433_batch_timing *t);

/* ------------------------------------------------------------------------------------------------
 * Decoder dispatch: replays the events of the last run into the registered plugins exactly as
 * run_ook_demods / run_fsk_demods + account_event would (reference src/r_api.c:438-550,
 * src/pulse_slicer.c:26-66): packages in detection order, priority levels ascending until one
 * produced an event, devices in registration order, events in pulse order.  Each event is inflated
 * into a reference-layout bitbuffer_t and handed to r_device.decode_fn on the calling thread;
 * decode_events / decode_ok / decode_messages / decode_fails are updated like the reference does.
 *
 * devices[i] must correspond to timing row i given at r433_batch_create.
 * pkg_cb (may be NULL) is called before the decoders of each package with the inflated
 * pulse_data_t (rssi/snr/freq filled as calc_rssi_snr does, reference src/r_flow.c:35-64).
 * Returns the number of successful decode events (sum of positive decode_fn returns). */
typedef void (*r433_package_fn)(void *user, uint32_t stream, uint32_t type, r433_pulse_data const *pulses);
int r433_batch_dispatch(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices, r433_package_fn pkg_cb,
        void *user);
/* Same, with the packages spread over n_threads host threads (contiguous package ranges, so all
 * reference ordering rules hold inside a package; decode_fn / pkg_cb must be thread-safe and
 * cross-package output order is the caller's business). */
int r433_batch_dispatch_mt(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices, r433_package_fn pkg_cb,
        void *user, uint32_t n_threads);

/* The same replay with every hook point the reference's frame loop has around its decoders (src/r_flow.c:240-340):
 * package_begin before the decoders of a package (pulse_data_t inflated and levelled as above, plus the package record
 * with the frame index and data_counter the reference returned it at), event_done after each decode_fn call and before
 * bitbuffer_clear (account_event's statistics are already updated; this is where its debug printout sits,
 * src/pulse_slicer.c:49-59), package_end with the package's event count (p_events).  Any hook may be NULL.
 * Single-threaded, on the calling thread, in reference order. */
typedef struct r433_dispatch_hooks {
    void *user;
    void (*package_begin)(void *user, r433_pkg_rec const *rec, r433_pulse_data const *pulses);
    void (*event_done)(void *user, r433_r_device *device, int ret, r433_bitbuffer const *bits);
    void (*package_end)(void *user, r433_pkg_rec const *rec, int p_events);
    /* optional: return 0 to leave a package out of the replay altogether (no decoder sees it, no other hook is called for it).
     * dropin/r_flow_hip.c uses it when it has to answer every push at once (-E quit): it then replays a growing prefix of the
     * capture and only lets the packages of the newest frame through. */
    int (*package_filter)(void *user, r433_pkg_rec const *rec);
    /* optional, r433_batch_dispatch_ordered only: called on the replay thread that runs the decoder, at the moment the decoder
     * hands a data_t to r_device.output_fn (src/decoder_util.c decoder_output_data); what it returns takes the data's place
     * in the output_fn call committed later, in reference order, on the calling thread.  For hosts whose output handler is a
     * pure rendering of the data followed by an ordered append (dropin/plugins_shim.c: the JSON line of data_print_jsons):
     * the rendering -- 0.7 us per event, a sixth of a replay when it all ran on the committing thread -- then runs beside
     * the other decoders.  Must be thread-safe; owns the data_t from then on; what it returns (never NULL) is handed to
     * output_fn as is.  Log messages (log_fn) are not rendered. */
    void *(*output_render)(void *user, r433_r_device *device, void *data);
} r433_dispatch_hooks;
int r433_batch_dispatch_hooks(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices,
        r433_dispatch_hooks const *hooks);
/* The same, replayed on n_threads host threads WITHOUT giving up the reference's order where it can be observed:
 *   - a decoder is only ever called by one thread, for its bitbuffers in package order -- decoders that keep state
 *     between calls (e.g. src/devices/secplus_v1.c:142) see exactly the call sequence of the single-threaded replay;
 *   - the priority rule (src/r_api.c:442-451) holds: the devices of a level run for a package only if no lower level
 *     produced an event for it (levels are passes);
 *   - what decoders hand to r_device.output_fn / log_fn during decode_fn is captured and committed on the calling
 *     thread afterwards, package by package, in the order the single-threaded replay would have produced it, between the
 *     package_begin and package_end hooks.
 * hooks->event_done must be NULL (it would need every bitbuffer after its decoder ran; callers that want account_event's
 * debug printout use r433_batch_dispatch_hooks).  r_device.output_fn / log_fn are swapped for the capture during the call
 * and restored before it returns.  Returns the number of successful decode events. */
int r433_batch_dispatch_ordered(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices,
        r433_dispatch_hooks const *hooks, uint32_t n_threads);
/* Decoders whose decode_fn is a function of its arguments alone -- it keeps nothing between calls and writes nothing but the
 * bitbuffer it was handed and what it gives to output_fn / log_fn -- need not stay on one thread: with stateless[d] != 0 the
 * ordered replay spreads the calls of decoder d over its threads in stretches of packages (statistics are added atomically,
 * outputs committed in reference order as for every decoder).  One decoder's calls on one thread are what bounds the replay
 * of a large batch (a TPMS decoder that searches every row of every bitbuffer: 0.6 M calls, 27 ms per 8192 captures), and
 * of the reference's 335 default decoders only four keep state (secplus_v1, secplus_v2, ikea_sparsnas, arad_ms_meter:
 * file-scope statics under src/devices).  Which decoders qualify is the HOST's knowledge about its plugins: the library
 * cannot see it and the default is 0 for every decoder.  NULL: back to the default.  The flags stay with the engine.
 * The flags also tell the pre-filter whom it may ask (round 6; make the statement BEFORE r433_batch_probe_prefilter):
 *   1  stateless: asked as it is;
 *   0  keeps state the library cannot see (file-scope statics): never asked and never filtered -- a refusal of such a decoder
 *      need not be a function of the bitbuffer's head, and the questions would leave made-up messages in its state;
 *   2  (R433_KEEPS_CONTEXT) keeps state, all of it behind r_device.decode_ctx (a create_fn's context: src/decoder_util.c:19-45):
 *      stays on one replay thread like 0, and is asked with decode_ctx pointing at an unreadable page -- a refusal that comes
 *      back without a fault has neither read nor written the decoder's state, so it is a function of the head it was shown
 *      (the same argument as for the payload behind the fence), and the real context is not touched by any question. */
#define R433_KEEPS_CONTEXT 2
int r433_batch_set_stateless(r433_batch *b, uint8_t const *stateless, uint32_t n_devices);
/* per package of the last dispatch: the events its decoders reported (p_events) */
int r433_batch_decoded(r433_batch *b, int const **per_package, uint32_t *count);

/* Device-side decoder pre-filter (reference src/pulse_slicer.c:26-66, the first-line length tests of
 * the decoders under src/devices such as nice_flor_s.c:84).  Most bitbuffers a slicer builds are refused by their decoder on a look at
 * num_rows / bits_per_row[0] alone.  r433_batch_probe_prefilter learns, per registered decoder, for which
 * (num_rows, bits_per_row[0]) that is PROVABLY so: it calls decode_fn on a bitbuffer_t of which only num_rows, free_row and
 * bits_per_row[0] are readable -- everything behind them lies on an inaccessible page -- and keeps a verdict only where the
 * call came back with a failure code (0, DECODE_ABORT_LENGTH .. DECODE_FAIL_SANITY) WITHOUT touching anything else: a
 * decoder that is a function of its arguments then returns that code for every bitbuffer with that head.  From the next run
 * on the slicer kernel drops such records where it builds them (they are neither staged, copied to the host nor replayed)
 * and counts them per decoder and code; the dispatch functions add the counts to decode_events / decode_fails exactly as
 * account_event would have, so `-M stats` is unchanged.  Decoders of later priority levels are not called for every package
 * (src/r_api.c:442-451): what they provably refuse crosses as a 16-byte stub (R433_EVT_STUB, include/r433_records.h) that the
 * replay books without a call where it reaches that level.  Only decoders with verbose == 0 are filtered (account_event prints
 * refused bitbuffers at -vv), and never together with an event_done hook or a package_filter.  One-row bitbuffers of at most 14 bits are asked
 * content by content as well: every one of the 2^n rows on an ordinary cleared bitbuffer_t, twice; where all of them get the
 * same failure code, a bitbuffer of exactly that shape (one row, no sync pulses before it, nothing ever written behind its
 * bits) is dropped under that code too.  decode_fn is called n times per decoder here (~50 000 heads and up to 65 534 tiny
 * rows, outside account_event: no statistics move; output_fn and log_fn are swallowed meanwhile); a decoder that keeps state
 * between calls must not let that state decide what it refuses, and must not remember a bitbuffer it refuses.  Returns the
 * number of decoders with at least one provable refusal. */
int r433_batch_probe_prefilter(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices);
/* What the decoders answered is remembered for the life of the process (several engines over the same decoder objects ask
 * once), keyed by the decoder object, its decode_fn / decode_ctx pointers, protocol number, line code, timings and name.  A
 * host that frees decoders, or changes what lies behind a decode_ctx, calls this before the next probe: everything known
 * is forgotten (engines keep the tables they already have until they are probed again). */
void r433_prefilter_forget(void);
/* Most decoders TOUCH the payload before their length test -- through one of four helpers of the reference's bitbuffer.c:
 * bitbuffer_invert (src/bitbuffer.c:135-149), bitbuffer_search (:228-253), bitbuffer_find_repeated_row / _prefix (:513-533)
 * -- so the fenced call above faults and nothing is learned (Neptune R900, src/devices/neptune_r900.c:88-101: a search for the
 * preamble, THEN "too short" for anything under 224 bits; a quarter of all records that still crossed).  A host that wraps
 * those four helpers of its decoders (ld --wrap, dropin/helper_wrap.c) lets the probe ask further: while `armed` is set on the
 * calling thread the wrappers do not look at the payload but answer from this block --
 *   bitbuffer_invert            nothing (it changes payload bytes only; `inverts` counts)
 *   bitbuffer_search            `answer` for the first call of a question (a bit position, or < 0: "not found" = the row's
 *                               length), and they leave row / start / pattern_bits here; a second call sets `overflow`
 *   bitbuffer_find_repeated_*   on a ONE-row bitbuffer the row compares equal to itself whatever it holds: 0 if it is long
 *                               enough and one repeat suffices, else -1 (`repeats` counts); on several rows -1 where the
 *                               question supposes every row shorter than the call's min_bits (`rows_below`: the verdict is
 *                               then kept for bitbuffers whose longest row is that short, which the slicer kernel knows),
 *                               else the real function (which faults)
 * -- and the probe asks a head once per answer the real helper could have given (every position from `start` to length -
 * pattern_bits, and "not found").  Only where ALL of them make decode_fn return the same failure code, again without a look at
 * anything behind the head, that code becomes the verdict of the head: it holds for every payload.  One-row heads of up to
 * 511 bits are asked this way.  Where the answers differ but "not found" alone leads to a refusal (tpms_imars_t240.c:57-66:
 * "preamble not found" and "found, too short" are two codes), the decoder's search itself becomes the test: for rows of fewer
 * than 64 bits and patterns of at most 32 the slicer kernel looks for the pattern where it finishes the bitbuffer (the row is
 * still in two registers) and drops the record under the "not found" code if it is not there.  The block is the host's (one per thread that runs decoders); the library learns where it is
 * from this call: host_block(+1) when a thread begins to ask, host_block(-1) when it is through (the wrappers of a host that
 * counts these look at their thread's block only while somebody asks: one load of a global on the replay's path), both
 * return the calling thread's block.  Process-wide, before the first probe (or r433_prefilter_forget after it); NULL: back
 * to the fence alone. */
typedef struct r433_helper_probe {
    uint32_t armed;        /* library -> wrappers: a question is being asked on this thread */
    int32_t answer;        /* library -> wrappers: what the first bitbuffer_search of the question returns (< 0: not found) */
    void const *subject;   /* library -> wrappers: the bitbuffer_t the question is about; calls on any other go to the real helper */
    uint32_t searches;     /* wrappers -> library: bitbuffer_search calls of this question */
    uint32_t overflow;     /* ... more than one (the later ones were told "not found": the question does not count) */
    uint32_t row, start, pattern_bits; /* ... arguments of the first */
    uint32_t inverts, repeats;         /* ... bitbuffer_invert / bitbuffer_find_repeated_* calls answered without the payload */
    uint8_t pattern[8];    /* ... and the first bytes of its pattern (as many as pattern_bits needs, at most eight) */
    uint32_t rows_below;   /* library -> wrappers: the question supposes that EVERY row of the subject has fewer bits than this
                            * (0: nothing supposed); bitbuffer_find_repeated_* on several rows then answers -1 where its
                            * min_bits is at least this -- no row can qualify, whatever the rows hold -- */
    uint32_t min_bits;     /* ... and leaves its min_bits here (wrappers -> library) */
    uint32_t reserved[2];
} r433_helper_probe;
typedef r433_helper_probe *(*r433_helper_probe_fn)(int session); /* -> the calling thread's block */
void r433_prefilter_set_helper_probe(r433_helper_probe_fn host_block);
/* on = 0 turns the learned tables off again (records flow as without a probe), 1 back on */
int r433_batch_set_prefilter(r433_batch *b, int on);
/* of the last run: records dropped on the device, counts[device * 5 + code] with code = -(decode_fn return) in 0..4 */
int r433_batch_prefilter_counts(r433_batch *b, uint32_t const **counts, uint32_t *n_devices);

/* What the dispatcher is handing to decode_fn right now, for plugins that want to tag their output
 * (the reference's `output_tag FILE` needs the capture; time stamps need start_ago).  Thread-local.  Inside the package hooks
 * of r433_dispatch_hooks (package_begin / package_end) stream, package, package_type and start_ago describe the hook's package. */
typedef struct r433_dispatch_info {
    uint32_t stream;       /* capture index in the batch */
    uint32_t package;      /* canonical package index */
    uint32_t device;       /* registration index of the r_device */
    uint32_t ordinal;      /* n-th bitbuffer of this (package, device) */
    uint32_t package_type; /* R433_PKG_OOK / R433_PKG_FSK */
    uint32_t start_ago;    /* pulse_data_t.start_ago of the package */
} r433_dispatch_info;
int r433_dispatch_current(r433_dispatch_info *info);

/* A ready-made decode_fn (reference plugin signature) that checksums every bitbuffer it receives:
 * FNV-1a 64 over {package, device, ordinal, num_rows, free_row, per row: bits, syncs, payload bytes},
 * summed mod 2^64 into decode_ctx (a r433_digest_ctx).  bench.py registers it for every device; the
 * oracle computes the same sum from its own records, so a full-size run is parity-checked by one
 * number.  Always returns DECODE_ABORT_LENGTH (no event). */
typedef struct r433_digest_ctx {
    uint64_t sum;
    uint64_t events;
} r433_digest_ctx;
int r433_plugin_digest_decode(r433_r_device *decoder, r433_bitbuffer *bits);

/* ------------------------------------------------------------------------------------------------
 * Function-level seam on device buffers: the prototypes of include/baseband.h with device pointers.
 */
/* envelope_detect (src/baseband.c:36-45): d_sum receives the wrapped u32 sum of the outputs */
int r433_envelope_detect(void const *d_iq, void *d_env, uint32_t n, uint32_t *d_sum, void *stream);
/* magnitude_est_cu8 (src/baseband.c:65-79) / magnitude_est_cs16 (:96-110) */
int r433_magnitude_est_cu8(void const *d_iq, void *d_env, uint32_t n, uint32_t *d_sum, void *stream);
int r433_magnitude_est_cs16(void const *d_iq, void *d_env, uint32_t n, uint32_t *d_sum, void *stream);
/* The two low-passes of include/baseband.h on HOST buffers, one frame with the filter state in and out: what
 * baseband_low_pass_filter (src/baseband.c:145-169), baseband_demod_FM (:210-272) and baseband_demod_FM_cs16 (:303-366) do
 * for one call.  `carry` holds filter_state_t / demodfm_state_t in plain ints (am_y, am_x = y[-1], x[-1] of the AM filter;
 * fm_y, fm_x = yf, xf; last_i, last_q = xr, xi) and is updated to the state after the frame's last sample.  The FM
 * coefficients are the caller's (the reference keeps them in its state struct and recomputes them only when the rate
 * changes): a16/b16 = alp_16[1] / blp_16[0], a32/b32 = alp_32[1] / blp_32[0].  Runs phases A and B of the detection
 * kernel (chunk-parallel exact filters) on the frame; n_samples == 0 is a no-op like in the reference. */
typedef struct r433_filter_carry {
    int32_t am_y, am_x, fm_y, fm_x, last_i, last_q;
} r433_filter_carry;
#define R433_FILTER_AM 1u      /* h_in: u16 envelope  -> h_out: low-passed envelope (s16) */
#define R433_FILTER_FM_CU8 2u  /* h_in: cu8 IQ        -> h_out: low-passed FM discriminator (s16) */
#define R433_FILTER_FM_CS16 3u /* h_in: cs16 IQ       -> h_out: the same from 16-bit samples */
int r433_filter_frame(uint32_t kind, void const *h_in, uint32_t n_samples, int16_t *h_out, r433_filter_carry *carry,
        int32_t a16, int32_t b16, int64_t a32, int64_t b32);
/* envelope_detect / magnitude_est_cu8 / magnitude_est_cs16 and the three evaluation variants the reference keeps next to
 * them (envelope_detect_nolut, magnitude_true_cu8, magnitude_true_cs16, src/baseband.c:50-61,82-93,113-124) on HOST
 * buffers; returns the wrapped u32 sum of the outputs through *sum (the reference derives its dB return value from it). */
#define R433_ENV_AMP_CU8 0u
#define R433_ENV_MAG_CU8 1u
#define R433_ENV_MAG_CS16 2u
#define R433_ENV_TRUE_CU8 3u
#define R433_ENV_TRUE_CS16 4u
int r433_envelope_host(uint32_t kind, void const *h_iq, uint16_t *h_env, uint32_t n_samples, uint32_t *sum);

/* pulse_detect_package() as a call of its own (reference include/pulse_detect.h:37-71, src/pulse_detect.c:199-483): the
 * detector object with its levels, and one visit of a HOST buffer of filtered envelope / discriminator samples per call.
 * Same contract: resumable across buffers, returns R433_PKG_OOK / R433_PKG_FSK at the first package that ends (the next call
 * looks at the same sample again) or 0 at the end of the buffer, len == 0 flushes; `pulses` / `fsk_pulses` hold what the
 * reference's structs hold after the same call (the detector's fields; both are cleared when a package begins).  The state
 * machine runs on the device, one wavefront, sample by sample: this entry point exists for the function-level seam
 * (librtl433seam.so exports it as pulse_detect_package), the fast path is r433_batch_run.  < 0: error. */
typedef struct r433_detector r433_detector;
r433_detector *r433_detector_create(void);
void r433_detector_destroy(r433_detector *d);
void r433_detector_reset(r433_detector *d);
/* pulse_detect_set_levels (src/pulse_detect.c:86-105): dB values as the reference takes them */
void r433_detector_set_levels(r433_detector *d, int use_mag_est, float fixed_high_level, float min_high_level, float high_low_ratio);
int r433_detector_package(r433_detector *d, int16_t const *envelope, int16_t const *fm, int len, uint32_t samp_rate, uint64_t sample_offset,
        r433_pulse_data *pulses, r433_pulse_data *fsk_pulses, unsigned fpdm);

/* The FSK sub-detectors pulse_detect_package drives, one sample per call, as calls of their own (reference
 * include/pulse_detect_fsk.h:46-75, src/pulse_detect_fsk.c:34-221): the caller's pulse_detect_fsk_t and pulse list go to the
 * device, one wavefront runs the step the detection kernel runs (csrc/detect_device.hpp), both come back.  op:
 * R433_FSK_CLASSIC / R433_FSK_MINMAX take the sample fm; R433_FSK_WRAP_UP closes the list (classic detector, :143-156).
 * For the completeness of the function-level seam; the fast path has these fused into k_wave. */
#define R433_FSK_CLASSIC 1
#define R433_FSK_MINMAX 2
#define R433_FSK_WRAP_UP 3
int r433_fsk_step(int op, r433_fsk_state *s, int fm, r433_pulse_data *fsk_pulses);

/* The file loop's input conversions (src/rtl_433.c:1811-1834) on device buffers: n = number of components
 * (2 per IQ sample).  cs8 -> cu8: +128.  cf32 -> cs16: (int)(f * 32767) clamped to +-32767, with C-on-x86
 * semantics for values no int can hold (they become -32767). */
int r433_convert_cs8_cu8(void const *d_in, void *d_out, uint64_t n, void *stream);
int r433_convert_cf32_cs16(void const *d_in, void *d_out, uint64_t n, void *stream);
/* The pulse-data side door (`-r file.ook`, src/rtl_433.c:1755-1794): packages detected elsewhere go straight to the
 * decoder fan-out.  pulses: n_packages structs in the reference's pulse_data_t layout (host memory); a package goes
 * to the FSK decoders if its fsk_f2_est is non-zero, as the reference decides.  Every package must be at the batch's
 * sample rate (the decoders' timings are resolved for it at r433_batch_create; sample_rate 0 = the batch's).  Results are read and dispatched exactly
 * as after r433_batch_run (package k carries stream = k).  Returns the number of packages, negative on error. */
int r433_batch_run_pulses(r433_batch *b, r433_pulse_data const *pulses, uint32_t n_packages, void *stream);
/* pulse_data_load (src/pulse_data.c:122-176) over a whole `.ook` text in memory: every package up to the first empty
 * one, like the file loop reads them, RfRaw lines (AA B0 / AA B1 ..., src/rfraw.c) included -- those carry
 * microseconds and set the package's sample_rate to 1000000 as the reference does.  Returns the number of packages
 * written to out (at most max_packages). */
int r433_pulse_text_load(char const *text, size_t len, uint32_t sample_rate, r433_pulse_data *out, uint32_t max_packages);
/* The sample grabber (`-S all|unknown|known`, src/r_flow.c:136-147,246-252,342-362, src/samp_grab.c:100-165) as a
 * plan: which byte range of which capture the reference would have saved to its g<counter>_<freq>M_<rate>k files, from
 * the package records of the last r433_batch_run (and, for modes 2 and 3, the decode results of the last dispatch).
 * The ranges point into the caller's own IQ buffers -- copying them out is a memcpy.  grab_mode: 1 all, 2 unknown
 * (no decoder reported an event during the frame), 3 known, 4 undecoded (below).  Returns the number of grabs (may exceed max_grabs). */
typedef struct r433_grab {
    uint32_t stream;      /* capture index */
    uint32_t counter;     /* the ### of the file name, counted through the batch */
    uint64_t byte_offset; /* into the capture */
    uint64_t byte_len;    /* multiple of 128 KiB unless cut by the start of the capture or the 3 MiB ring */
    uint32_t n_samples;   /* the padded signal length the reference reports */
    uint32_t clipped;     /* 1: the reference would have read ring memory older than the capture here */
    uint64_t pushed;      /* bytes of the capture pushed when the reference writes the file (its ring holds that much of this capture,
                             plus what came before it: the file is as long as the ring's fill allows, src/samp_grab.c:110-113) */
} r433_grab;
int r433_batch_grab_plan(r433_batch *b, int grab_mode, r433_grab *out, uint32_t max_grabs);
/* grab_mode 4 ("undecoded", src/r_flow.c:290-294,351): a frame nobody decoded is saved if the pulse analyzer thinks one of its
 * packages looks like a transmission.  The analyzer's verdict (pulse_analyzer_check's return value, > 0 = plausible) is the
 * caller's: one value per package of the last run, before r433_batch_grab_plan(b, 4, ...). */
int r433_batch_set_package_quality(r433_batch *b, int32_t const *quality, uint32_t n_packages);
/* The SigMF container the grabber writes with `-S sigmf:...` (src/samp_grab.c:166-232, src/sigmf.c, microtar): the
 * bytes before the data of a grab (meta member + header of the data member) and the bytes after it (record padding +
 * two null records).  Both return the number of bytes needed and write them if cap suffices. */
int r433_sigmf_prefix(uint32_t sample_size, uint32_t sample_rate, uint32_t frequency, uint64_t data_len, uint8_t *buf, size_t cap);
int r433_sigmf_trailer(uint64_t data_len, uint8_t *buf, size_t cap);
/* ... and the reader side (`-r file.sigmf`, sigmf_reader_open, src/sigmf.c:336-434) over an archive in memory: the
 * first stream's meta data and where its samples sit in the buffer.  (The reference then reads the samples as cu8
 * whatever the datatype says, src/rtl_433.c:1719.) */
typedef struct r433_sigmf_info {
    char datatype[32];    /* core:datatype */
    uint32_t sample_rate; /* core:sample_rate */
    uint32_t frequency;   /* captures[0] core:frequency */
    uint32_t sample_start;
    uint32_t reserved;
    uint64_t data_offset; /* of the .sigmf-data member's bytes in the archive */
    uint64_t data_len;
} r433_sigmf_info;
int r433_sigmf_probe(uint8_t const *buf, size_t len, r433_sigmf_info *info);
/* The VCD pulse writer (`-w file.vcd`): pulse_data_print_vcd_header (src/pulse_data.c:77-100; `date` is the text of the
 * $date line) and pulse_data_print_vcd (:102-120; ch_id '\'' for an OOK package, '"' for an FSK one).  snprintf convention. */
int r433_pulse_vcd_header(uint32_t sample_rate, char const *date, char *buf, size_t cap);
int r433_pulse_vcd(r433_pulse_data const *data, int ch_id, char *buf, size_t cap);
/* pulse_data_dump (src/pulse_data.c:193-224): one package as `.ook` text; `received` is the time string of the
 * ";received" line (NULL: no such line).  snprintf convention: returns the full length, writes at most cap bytes. */
int r433_pulse_text_dump(r433_pulse_data const *data, char const *received, char *buf, size_t cap);

/* The pulse analyzer (`-A`, src/pulse_analyzer.c:279-556) over every package of the last run (r433_batch_run or
 * r433_batch_run_pulses): histograms and modulation guess on the device, one wavefront per package; out[k] belongs to
 * package k.  Returns the number of results written (at most max_packages). */
int r433_batch_analyze(r433_batch *b, r433_analysis *out, uint32_t max_packages, void *stream);
/* The reference's report for package pkg of the last run, from "Analyzing pulses..." through the flex-decoder
 * suggestion (what the trial demodulation prints after that is the slicers' business: feed `a->device` to a batch).
 * snprintf convention: returns the full length, writes at most cap bytes. */
int r433_analysis_text(r433_batch *b, uint32_t pkg, r433_analysis const *a, char *buf, size_t cap);

/* The -w dump formats (src/r_flow.c:385-489, named as in include/fileformat.h): what the reference writes next to
 * its input, as one HBM-bound map on device buffers.  sample_size says what d_in holds (2 = cu8 IQ, 4 = cs16 IQ);
 * for R433_DUMP_F32_AM / _FM d_in is the am / fm int16 stream (r433_batch_set_taps, the S16_AM / S16_FM dumps
 * themselves).  n_out = number of output values: IQ components (2 per sample) for the *_IQ formats, samples for
 * the others.  A format that equals the input (cu8 from cu8, cs16 from cs16, S16_AM, S16_FM) is a plain copy. */
#define R433_DUMP_CU8_IQ 1  /* from cs16: x / 256 + 128 */
#define R433_DUMP_CS16_IQ 2 /* from cu8: x * 256 - 32768 */
#define R433_DUMP_CS8_IQ 3  /* cu8: x - 128, cs16: x >> 8 */
#define R433_DUMP_CF32_IQ 4 /* cu8: (x - 128) / 128.0f, cs16: x / 32768.0f */
#define R433_DUMP_S16_AM 5
#define R433_DUMP_S16_FM 6
#define R433_DUMP_F32_AM 7  /* am * (1.0f / 0x8000) */
#define R433_DUMP_F32_FM 8
#define R433_DUMP_F32_I 9   /* cu8: (I - 128) * (1.0f / 0x80), cs16: I * (1.0f / 0x8000) */
#define R433_DUMP_F32_Q 10
int r433_dump_convert(int format, uint32_t sample_size, void const *d_in, void *d_out, uint64_t n_out, void *stream);
/* The same for a frame in HOST memory (the file loop's -w / -W dumpers, dropin/r_flow_hip.c): staged through device buffers of
 * the library's own, synchronous.  No alignment requirement. */
int r433_dump_convert_host(int format, uint32_t sample_size, void const *h_in, void *h_out, uint64_t n_out);
/* AMP_TO_DB / MAG_TO_DB of a frame sum (include/baseband.h:36-37, src/baseband.c:44,78) */
float r433_level_db(uint32_t sum, uint32_t n, int is_magnitude);

#ifdef __cplusplus
}
#endif
#endif /* R433_HIP_H_ */
