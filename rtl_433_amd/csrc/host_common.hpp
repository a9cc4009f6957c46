// host_common.hpp -- what the host-side translation units of librtl433hip.so share: error reporting, device /
// pinned buffers, the dispatch thread pool and the batch object.  Not part of the public C ABI (include/r433_hip.h).
#ifndef R433_HOST_COMMON_HPP_
#define R433_HOST_COMMON_HPP_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "r433_hip.h"
#include "r433_internal.hpp"


// thread-local message behind r433_last_error(); returns `code` so that `return fail(...)` reads well
int fail(int code, char const *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? R433_ENODEV : R433_EHIP, "%s: %s",   \
                    #expr, hipGetErrorString(e_));                                                                     \
    } while (0)

namespace r433 {

// pulse_detect_set_levels and its dB macros on the host (host_api.cpp)
void levels_from_db(DetCfg &c, int use_mag, float fixed_db, float min_db, float ratio_db);
// calc_rssi_snr, reference src/r_flow.c:35-64 (dispatch.cpp)
void fill_levels(r433_flow_cfg const &cfg, r433_pulse_data &p);
} // namespace r433
struct r433_batch;
namespace r433 {
// records the device-side pre-filter dropped in the last run -> decode_events / decode_fails, once per run (prefilter.cpp)
void apply_prefilter_counts(r433_batch *b, r433_r_device *const *devices, uint32_t n_devices);

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0; // elements
    int ensure(size_t n)
    {
        if (n <= cap)
            return 0;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 16;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess)
            (void)hipGetLastError(); // reported here; must not turn up again in a later launch check
        if (e != hipSuccess)
            return fail(R433_ENOMEM, "hipMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T> struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap)
            return 0;
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 16;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess)
            return fail(R433_ENOMEM, "hipHostMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};


// Persistent host workers for the decoder dispatch (spawning 32 threads per batch costs more than
// dispatching a small batch).
class Pool {
  public:
    ~Pool() { stop(); }
    void run(unsigned n, std::function<void(unsigned)> const &job)
    {
        if (n <= 1) {
            job(0);
            return;
        }
        grow(n - 1);
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &job;
            want_ = n - 1;
            pending_ = n - 1;
            ++epoch_;
        }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void grow(unsigned n)
    {
        while (threads_.size() < n) {
            unsigned id = (unsigned)threads_.size();
            threads_.emplace_back([this, id] { loop(id); });
        }
    }
    void loop(unsigned id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(unsigned)> const *job = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return quit_ || (epoch_ != seen && id < want_); });
                if (quit_)
                    return;
                seen = epoch_;
                job = job_;
            }
            (*job)(id + 1);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0)
                    done_.notify_all();
            }
        }
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_)
            t.join();
        threads_.clear();
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    std::function<void(unsigned)> const *job_ = nullptr;
    unsigned want_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool quit_ = false;
};

} // namespace r433

using namespace r433; // internal header of four host translation units; the batch type itself is the C ABI's (global)

// Every engine belongs to one GPU (r433_batch_create_on).  The HIP runtime's current device is a property of the calling
// THREAD: every entry point that touches the device switches to the engine's for the time of the call and puts the
// caller's back, so that a host may drive engines on several GPUs from any of its threads (dropin/pipeline_host.c --gpus).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device)
    {
        if (device < 0)
            return;
        if (hipGetDevice(&prev) == hipSuccess && prev != device)
            switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (switched)
            (void)hipSetDevice(prev);
    }
    DeviceScope(DeviceScope const &) = delete;
    DeviceScope &operator=(DeviceScope const &) = delete;
};

struct r433_batch {
    int device = -1; // the GPU this engine lives on (-1: whatever was current when it was made)
    r433_flow_cfg cfg;
    DetCfg det;
    int a16 = 0, b16 = 0;
    long long a32 = 0, b32 = 0;
    std::vector<r433_dev_timing> timing; // registration order
    std::vector<DevRow> rows;            // sorted for the fan-out
    std::vector<uint32_t> prio_levels;   // distinct priorities ascending

    DevBuf<DevRow> d_rows;
    DevBuf<uint8_t> d_arena;
    DevBuf<int2> d_ring;
    DevBuf<StreamState> d_state;
    DevBuf<uint32_t> d_frame_sums, d_stream_bytes, d_pkg_base, d_scal;
    DevBuf<int> d_frame_min_high;
    std::vector<int> h_frame_min_high;
    // split captures (r433_batch_set_split)
    uint32_t split_samples = R433_SPLIT_AUTO;
    uint32_t debug_flags = 0; // r433_batch_set_debug
    uint32_t stage_slot = 0; // r433_batch_set_staging_slot: bytes per (package, device) staging slot of the slicers (0: 8 KB, every record fits)
    int exclusive_detect = 0; // r433_batch_set_exclusive_detect: 1 the detection kernel, 2 the slicer kernels as well, 3 the record copies too
    bool logic_on = false;         // r433_batch_enable_logic_dump
    DevBuf<uint8_t> d_logic;
    PinBuf<uint8_t> h_logic;
    uint64_t logic_stride = 0;
    DevBuf<uint32_t> d_tile_max, d_order, d_wg;
    // producers and consumers as two launches (StreamParams::tile_store ...): tile records, descriptors, per-slot words, run-again list
    DevBuf<uint8_t> d_tile_store;
    DevBuf<int> d_tile_desc;
    DevBuf<uint32_t> d_tile_words; // [n] over, [n] info, [1] count (+ 3 unused), [n] list, [n] why
    DevBuf<SegDesc> d_segs;
    PinBuf<uint32_t> h_tile_max;
    PinBuf<StreamState> h_state;
    uint32_t last_segments = 0, last_redone = 0;
    bool last_roles = false; // the last detection pass ran producers and consumers as two launches (r433_batch_detect_form)
    DevBuf<uint32_t> d_dir_stream, d_dir_off, d_rec_bytes, d_rec_off, d_sizes, d_dev_off, d_pkg_bytes, d_pkg_off;
    // the slice index (slicer_kernels.hip k_index_*): per decoder the (offset, bytes) of its slices of the event stream
    DevBuf<uint32_t> d_idx_cnt, d_slice_start;
    DevBuf<uint2> d_slices;
    PinBuf<uint32_t> h_slice_start;
    PinBuf<uint2> h_slices;
    uint32_t n_slices = 0;
    bool slices_valid = false;
    DevBuf<uint32_t> d_pkg_order, d_slice_cursor; // the sizing pass of the slicers: packages heaviest first, a cursor per chunk of devices
    // ... and its workgroups shared out by the work each chunk had in the runs before (SliceParams::chunk_work / chunk_deal)
    DevBuf<unsigned long long> d_chunk_work;
    DevBuf<uint8_t> d_chunk_deal;
    PinBuf<unsigned long long> h_chunk_work;
    double slice_w[2][16] = {};
    bool slice_w_valid = false;
    DevBuf<uint8_t> d_pkg_blob, d_events, d_stage, d_converted;
    DevBuf<r433_analysis> d_analysis;
    std::vector<uint32_t> conv_bytes;
    PinBuf<uint32_t> h_scal, h_frame_sums;
    PinBuf<uint8_t> h_pkg_blob, h_events, h_arena_stage;
    PinBuf<uint32_t> h_pkg_off, h_rec_off; // per package: byte offset of its first event / of its record

    uint32_t arena_stride = 0;
    uint32_t arena_growth = 1; // x4 after every arena overflow: what this workload needs per sample; given back after eight runs in a
                               // row that used less than a sixteenth of it on average (one dense batch must not tax the engine for good)
    uint32_t calm_runs = 0;
    uint32_t frames_cap = 0;
    uint32_t n_streams = 0;
    uint32_t n_pkgs = 0, n_events = 0;
    size_t pkg_bytes = 0, evt_bytes = 0;
    bool events_counted = false;
    std::vector<uint32_t> stream_samples; // per capture of the last run (as the detector saw them)
    std::vector<int> pkg_decoded; // per package: events its decoders reported in the last dispatch
    std::vector<uint8_t> stateless; // per device: decode_fn keeps nothing between calls (r433_batch_set_stateless)
    std::vector<int32_t> pkg_quality; // per package: the caller's analyzer verdict (r433_batch_set_package_quality; grab mode 4)
    bool dispatched = false;

    void *tap_env = nullptr, *tap_am = nullptr, *tap_fm = nullptr;
    uint64_t tap_stride = 0;

    // r433_batch_run_host: captures staged from host memory
    DevBuf<uint8_t> d_input;
    hipStream_t own_stream = nullptr;
    hipStream_t slice_stream = nullptr;            // the sizing pass of the small packages runs beside the large ones' (slicer_kernels.hip)
    hipEvent_t slice_forked = nullptr, slice_joined = nullptr;
    std::vector<uint32_t> host_bytes;

    hipEvent_t sync_ev = nullptr; // blocking (sleeping) wait: host threads of other pipeline stages need the cores
    bool profiling = false;
    hipEvent_t ev[8] = {};
    bool ev_made = false;
    r433_batch_timing last_timing = {};

    // decoder pre-filter (prefilter.cpp)
    std::vector<uint8_t> pf_tables;   // kPfTable bytes per filtered decoder
    std::vector<int> pf_index;        // per registered device: its table, or -1
    DevBuf<uint8_t> d_pf_tables;
    DevBuf<uint32_t> d_pf_counts;
    PinBuf<uint32_t> h_pf_counts;     // [device][5] of the last run
    bool pf_on = false;               // tables are in use
    bool pf_ran = false;              // the last run filtered (h_pf_counts is valid)
    bool pf_accounted = false;        // its counts have been added to the decoders' statistics

    // dispatch scratch
    r433_bitbuffer *bits = nullptr;
    r433_pulse_data *pulses = nullptr;
    Pool pool;
};

constexpr unsigned kTurnDevices = 64;
extern std::mutex g_detect_turn[kTurnDevices]; // engines with exclusive_detect take turns on the detection kernel of their GPU (batch_run.cpp)

// Waiting for the stream.  hipEventSynchronize spins on this stack even for an event made with hipEventBlockingSync
// (tools/spin_probe.py on the MI355X box: 10.0 ms of CPU per 9.8 ms of waiting, blocking or not).  With a CPU to spare per GPU
// leg that is the fastest wait there is, and the default.  A host that has fewer CPUs than waiting threads -- eight ranks of a
// node on a 16-CPU quota: sixteen legs in flight -- asks for R433_DEBUG_NAP_WAIT: the event is polled, a short while eagerly
// (the small waits between two kernels of a pass end within microseconds), then asleep in between (a twentieth of what has been
// waited so far, 20-200 us: a 4 ms kernel is noticed at most 0.2 ms late).  Measured on one GPU with CPUs to spare: a GPU leg's
// CPU time 12.5 -> 2.5 ms per step, the leg 0.5 ms longer, nothing faster (profiles/r04_wait_asleep_ab.txt).
inline hipError_t stream_wait(r433_batch *b, hipStream_t st)
{
    if (!b->sync_ev) {
        hipError_t e = hipEventCreateWithFlags(&b->sync_ev, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess)
            return e;
    }
    hipError_t e = hipEventRecord(b->sync_ev, st);
    if (e != hipSuccess)
        return e;
    if (!(b->debug_flags & R433_DEBUG_NAP_WAIT))
        return hipEventSynchronize(b->sync_ev);
    auto const t0 = std::chrono::steady_clock::now();
    for (unsigned polls = 0;; ++polls) {
        e = hipEventQuery(b->sync_ev);
        if (e != hipErrorNotReady)
            return e;
        if (polls < 16)
            continue;
        auto const waited = std::chrono::steady_clock::now() - t0;
        if (waited < std::chrono::microseconds(30))
            continue;
        auto nap = std::chrono::duration_cast<std::chrono::microseconds>(waited) / 20;
        nap = std::max(std::chrono::microseconds(20), std::min(std::chrono::microseconds(200), nap));
        std::this_thread::sleep_for(nap);
    }
}

#endif // R433_HOST_COMMON_HPP_
