// slicer_kernels.hip -- decoder fan-out: every pulse package against every registered r_device.
//
// Work item = (package, chunk of 64 devices); one wavefront per item.  The package's (pulse, gap)
// pairs are staged once in LDS (<= 9.6 KB, coalesced 8-byte loads) and every lane runs the slicer
// of its own device over them (LDS broadcast reads).  Devices are grouped by modulation: a line code
// with 64 decoders and more has wavefronts of its own (the slowest lane of a mixed wavefront pays for
// every slicer present in it), the line codes with a handful of decoders share chunks up to a bound
// on the sum of their walks (host_api.cpp: an item's fixed cost is paid once).  Records are built once into a
// fixed-size staging slot per (package, device); an exclusive scan of their sizes and a compaction
// pass give a dense event stream in canonical (package, device, event) order without atomics on the
// payload.  (A record that outgrows its slot is sliced a second time straight into the stream; if the
// staging arena itself would be too large the classic count + write pair runs instead.)
//
// Replaces run_ook_demods / run_fsk_demods (reference src/r_api.c:438-550) and the ten
// pulse_slicer_* functions (src/pulse_slicer.c) up to, not including, the decode_fn call, which
// stays on the host behind the r_device ABI.
#include <algorithm>
#include <cstdlib>

#include "r433_internal.hpp"
#include "slicer_device.hpp"

namespace r433 {

namespace {

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    uint32_t x = v;
    int const lane = (int)(threadIdx.x & 63);
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
        if (lane >= o)
            x += y;
    }
    total = (uint32_t)__shfl((int)x, 63, 64);
    return x - v;
}

// MODE: what one visit of a (package, 64 devices) work item does
//   M_COUNT    slice, keep only the record sizes                    } the two-pass fallback when the
//   M_WRITE    slice again, records to their final place            } staging arena would not fit
//   M_STAGE    slice ONCE, records into a fixed-size staging slot per (package, row) + their true sizes
//   M_COMPACT  copy staged records to their final place; a (package, device) that outgrew its slot is
//              sliced again straight into the event stream
enum { M_COUNT = 0, M_WRITE = 1, M_STAGE = 2, M_COMPACT = 3 };

// Packages of at most this many pulses are "small": their sizing pass runs as its own launch whose workgroups stage the pulses
// in 2 KB of LDS instead of 9.6 KB -- seven wavefronts to a SIMD instead of four (72 registers allow seven; LDS capped the
// kernel at 16 workgroups per CU) -- beside the launch of the few large packages, on a second stream.  The slicers are a serial
// walk per lane: what hides their latency is more wavefronts.
constexpr uint32_t kSmallW = 52;                 // weight (pulses / 5) from which a package is large
constexpr uint32_t kSmallPulses = kSmallW * 5;   // a small package has fewer pulses than this

// (the small-package form lives on eight wavefronts to a SIMD -- 2 KB of LDS each --, held to the 64 registers that takes with 68
// bytes of scratch a lane: a serial walk waits for every pulse it reads, and more walks in flight beat registers -- six / seven /
// eight wavefronts: 1.62 / 1.52 / 1.50 ms for the sizing pass of a bench step, records identical, profiles/r06_n_slice_waves.txt)
#ifndef R433_SLICE_MIN_WAVES
#define R433_SLICE_MIN_WAVES 8
#endif
#ifdef R433_EMU
#define R433_SLICE_WAVES(cap)
#else
#define R433_SLICE_WAVES(cap) __attribute__((amdgpu_waves_per_eu((cap) < R433_PD_MAX_PULSES ? R433_SLICE_MIN_WAVES : 1, 8)))
#endif
template <int MODE, int CAP = R433_PD_MAX_PULSES> __global__ __launch_bounds__(64) R433_SLICE_WAVES(CAP) void k_slice(SliceParams p)
{
    constexpr bool PLACE = MODE == M_WRITE || MODE == M_COMPACT; // records go to their final offsets
    constexpr bool STORE = MODE != M_COUNT;                       // the sink stores bytes
    constexpr bool SMALL = CAP < R433_PD_MAX_PULSES;              // the launch of the small packages (sizing passes only)
    // The placing pass (M_COMPACT with a token CAP): a copy per record, and only a slot that its records outgrew is sliced again
    // -- rare enough to read the pulses straight from the package record in HBM.  Without the 9.6 KB of LDS six wavefronts fit a
    // SIMD instead of four, and the pass is a chain of dependent loads per item (SIMD IPC 0.04 with 2.9 wavefronts resident,
    // profiles/r06_pmc_slice.json): more of them in flight is what it wants.
    constexpr bool FROM_HBM = MODE == M_COMPACT && CAP < 64;
    __shared__ int2 pairs[CAP];

    uint32_t const n_pkgs = min(min(*p.n_pkgs, p.max_pkgs), p.pkg_end);
    uint32_t const chunks = p.n_rows / 64;
    uint32_t const lane = threadIdx.x;
    // the grid is a multiple of `chunks` (slice_grid): a workgroup keeps its 64 devices for all its packages, so what the
    // pre-filter drops is counted in registers and leaves with one atomic per lane and code at the end
    // (a drawn sizing launch may give the chunks unequal shares of its workgroups: SliceParams::chunk_deal)
    bool const shared_out = !PLACE && p.draw != 0 && p.chunk_deal != nullptr && p.deal_grid == gridDim.x;
    uint32_t chunk = blockIdx.x % chunks;
    if (shared_out) {
        // (workgroups start in the order of their index: the shares are dealt out like a hand of cards, k_deal, so that every
        // chunk's workgroups begin at t = 0 -- as contiguous ranges the last chunk began 2 ms late)
        chunk = min((uint32_t)p.chunk_deal[blockIdx.x], chunks - 1u);
    }
    uint32_t const di = chunk * 64 + lane;
    int const my_pf = p.devs[di].pf, my_orig = p.devs[di].orig;
    uint8_t const *const pf_tab = (p.pf_tables && my_pf >= 0) ? p.pf_tables + (uint64_t)my_pf * kPfTable : nullptr;
    uint32_t dropped0 = 0, dropped1 = 0, dropped2 = 0, dropped3 = 0, dropped4 = 0;

    // Which package next.  The placing pass takes fixed strides (a copy per record: even work).  The sizing pass is a lane per
    // device walking the package's pulses, and its items differ by two orders of magnitude (a 1200-pulse package under a PCM
    // slicer against a short one that fails the first timing test): at fixed strides the launch lasted eight times the mean
    // life of its wavefronts (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES: 1.6 wavefronts resident per SIMD, profiles/r03_slice_pmc.txt).
    // So the workgroups of a chunk of devices DRAW their packages from the chunk's cursor over the packages sorted by pulse
    // count (k_pkg_order, which also rewinds the cursors): every chunk's long items begin at once, whoever is free takes the
    // next, and a chunk whose list has run dry lets its later workgroups go at once, so the chunks with the expensive slicers
    // end up with more of the chip.  8192 bench packages x 335 decoders: 3.70 -> 3.04 ms; a launch of 1024 lasts as long as its
    // longest item either way (0.53 ms).  One cursor over (chunk, package) -- the whole chip on one chunk at a time, a workgroup
    // changing devices as it goes -- was slower (3.24 ms; 0.62 ms for 1024): the expensive chunks' long items then begin late.
    bool const drawn = !PLACE && p.draw != 0;
    uint32_t const n_mine = n_pkgs > p.pkg_begin ? n_pkgs - p.pkg_begin : 0u;
    // a chunk draws the packages of its own kind (k_pkg_order); with two launches sharing the list (p.draw == 2) the large
    // packages of the kind through cursor[chunk], the small ones through cursor[chunks + chunk]
    bool const two = drawn && p.draw == 2;
    uint32_t my_end = 0;
    if (drawn) {
        uint32_t const kind = p.devs[chunk * 64].is_fsk != 0 ? 2u : 0u;
        my_end = min(p.cursor[2 * chunks + kind + (two && !SMALL ? 0u : 1u)], n_mine);
    }
    uint32_t *const my_cursor = p.cursor + (two && SMALL ? chunks : 0u) + chunk;
    for (uint32_t it = blockIdx.x / chunks;; it += gridDim.x / chunks) {
        uint32_t pkg;
        if (drawn) {
            uint32_t i = 0;
            if (lane == 0)
                i = atomicAdd(my_cursor, 1u);
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
            if (i >= my_end)
                break;
            pkg = p.pkg_order[i];
        }
        else {
            pkg = p.pkg_begin + it;
            if (pkg >= n_pkgs)
                break;
        }
        uint8_t const *rec = p.arena + (uint64_t)p.dir_stream[pkg] * p.arena_stride + p.dir_off[pkg];
        uint32_t const type = ((uint32_t const *)rec)[2];
        uint32_t const num = min(((uint32_t const *)rec)[3], (uint32_t)(FROM_HBM ? R433_PD_MAX_PULSES : CAP)); // (a small launch only draws packages that fit)
        int2 const *src = (int2 const *)(rec + sizeof(r433_pkg_rec));
        // The device row is read again for every package ON PURPOSE (the index is made opaque): with the row known to be the
        // same for all packages the compiler unswitches the loop on its modulation -- ten copies of the loop, twice the
        // registers, three times the run time.
        uint32_t di_now = di;
#ifndef R433_EMU
        asm volatile("" : "+v"(di_now));
#endif
        DevRow const t = p.devs[di_now];
        uint32_t my_size = 0;
        if (PLACE && t.orig >= 0)
            my_size = p.sizes[(uint64_t)pkg * p.n_devs + t.orig];
        // a compaction visit only needs the pulses if some record of the item has to be sliced again
        bool const need_pulses = MODE != M_COMPACT || __ballot(my_size > p.stage_cap) != 0;
        if (!FROM_HBM) {
            __syncthreads(); // previous item done with LDS
            if (need_pulses)
                for (uint32_t i = lane; i < num; i += 64)
                    pairs[i] = src[i];
            __syncthreads();
        }

        uint32_t my_bytes = 0;
        uint32_t copy_bytes = 0, copy_base = 0;
        if (t.orig >= 0) { // not a padding row
            bool const run = t.valid && (t.is_fsk != 0) == (type == R433_PKG_FSK);
            uint8_t *const slot = MODE == M_STAGE || MODE == M_COMPACT
                    ? p.stage + ((uint64_t)(pkg - p.pkg_begin) * p.n_devs + (uint32_t)t.orig) * p.stage_cap : nullptr; // (a slot per DEVICE: padding rows have none)
            BitSink<STORE> sink;
            uint8_t *out = nullptr;
            uint32_t limit = 0;
            bool fits = true;
            bool slice = run;
            if (PLACE) {
                uint32_t base = p.pkg_off[pkg] + p.dev_off[(uint64_t)pkg * p.n_devs + t.orig]; // k_dev_prefix
                limit = my_size;
                fits = limit > 0 && (uint64_t)base + limit <= p.events_cap;
                out = p.events + base;
                if (MODE == M_COMPACT && fits && limit <= p.stage_cap) { // the common case: a plain copy (below)
                    copy_bytes = limit;
                    copy_base = base;
                    slice = false;
                }
            }
            else if (MODE == M_STAGE) {
                out = slot;
                limit = p.stage_cap;
            }
            if (slice && fits) {
                PulseView pv{FROM_HBM ? src : pairs, num};
                sink.begin(out, limit, pkg, (uint32_t)t.orig);
                sink.pf = pf_tab;
                slice_dispatch<STORE>(pv, t, sink);
                my_bytes = sink.off;
                if (!PLACE) { // the sizing pass counts; the placing pass only repeats its decisions
                    dropped0 += (uint32_t)(sink.pf_dropped & 0xfffu);
                    dropped1 += (uint32_t)(sink.pf_dropped >> 12) & 0xfffu;
                    dropped2 += (uint32_t)(sink.pf_dropped >> 24) & 0xfffu;
                    dropped3 += (uint32_t)(sink.pf_dropped >> 36) & 0xfffu;
                    dropped4 += (uint32_t)(sink.pf_dropped >> 48) & 0xfffu;
                }
            }
            if (!PLACE)
                p.sizes[(uint64_t)pkg * p.n_devs + t.orig] = my_bytes;
        }
        if (MODE == M_COMPACT) {
            // Staged records -> event stream.  Most records are a header and a row or two: every lane copies its own with
            // 16-byte loads from its (aligned) slot, four in flight, and dword stores (records are 4-byte aligned in the
            // stream; stores do not wait).  One record at a time with the whole wavefront would pay a memory round trip
            // per record, 64 in a row.  The rare long record is left to the whole wavefront below.
            constexpr uint32_t kOwn = 512;
            if (copy_bytes > 0 && copy_bytes <= kOwn) {
                uint8_t const *src_r = p.stage + ((uint64_t)(pkg - p.pkg_begin) * p.n_devs + (uint32_t)t.orig) * p.stage_cap;
                uint32_t *dst = (uint32_t *)(p.events + copy_base);
                uint32_t const words = copy_bytes / 4;
                for (uint32_t w = 0; w < words; w += 16) {
                    uint4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        v[u] = w + 4 * u < words ? *(uint4 const *)(src_r + (w + 4 * u) * 4) : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        uint32_t const at = w + 4 * u;
                        if (at < words) dst[at] = v[u].x;
                        if (at + 1 < words) dst[at + 1] = v[u].y;
                        if (at + 2 < words) dst[at + 2] = v[u].z;
                        if (at + 3 < words) dst[at + 3] = v[u].w;
                    }
                }
            }
            unsigned long long todo = __ballot(copy_bytes > kOwn);
            while (todo) {
                int const r = __ffsll(todo) - 1;
                todo &= todo - 1;
                uint32_t const nb = (uint32_t)__builtin_amdgcn_readlane((int)copy_bytes, r);
                uint32_t const at = (uint32_t)__builtin_amdgcn_readlane((int)copy_base, r);
                uint32_t const orig_r = (uint32_t)__builtin_amdgcn_readlane(t.orig, r);
                uint8_t const *src_r = p.stage + ((uint64_t)(pkg - p.pkg_begin) * p.n_devs + orig_r) * p.stage_cap;
                uint32_t *const dst = (uint32_t *)(p.events + at);
                for (uint32_t w0 = 0; w0 < nb; w0 += 4096) { // 4 KB a round: four 16-byte loads per lane in flight, then the stores
                    uint4 v[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        uint32_t const off = w0 + u * 1024 + lane * 16;
                        v[u] = off < nb ? *(uint4 const *)(src_r + off) : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        uint32_t const off = w0 + u * 1024 + lane * 16;
                        if (off < nb) dst[off / 4] = v[u].x;
                        if (off + 4 < nb) dst[off / 4 + 1] = v[u].y;
                        if (off + 8 < nb) dst[off / 4 + 2] = v[u].z;
                        if (off + 12 < nb) dst[off / 4 + 3] = v[u].w;
                    }
                }
            }
        }
        if (!PLACE) {
            uint32_t tot;
            wave_excl_scan(my_bytes, tot);
            if (lane == 0 && tot)
                atomicAdd(&p.pkg_bytes[pkg], tot);
        }
    }
    // What the next run's shares are made of: when the chunk's last wavefront left (a wavefront leaves when its chunk's list is
    // dry) and how many there were.  One clock read on the way out -- a clock read at the START, kept for the end, cost a launch
    // of 1024 packages half its speed even with the measurement switched off (0.60 -> 1.22 ms, profiles/r04_slice_shares.txt).
#ifndef R433_EMU
    if (!PLACE && p.chunk_work && lane == 0 && chunk < 16) {
        unsigned long long *const w = p.chunk_work + ((SMALL ? 16u : 0u) + chunk) * 2u;
        atomicMax(w, (unsigned long long)wall_clock64());
        atomicAdd(w + 1, 1ull);
    }
#endif
    if (!PLACE && pf_tab && p.pf_counts) {
        uint32_t *const c = p.pf_counts + (uint32_t)my_orig * 5u;
        if (dropped0) atomicAdd(c + 0, dropped0);
        if (dropped1) atomicAdd(c + 1, dropped1);
        if (dropped2) atomicAdd(c + 2, dropped2);
        if (dropped3) atomicAdd(c + 3, dropped3);
        if (dropped4) atomicAdd(c + 4, dropped4);
    }
}

// The packages [pkg_begin, n) by kind and pulse count: the OOK packages first, then the FSK ones, each kind heaviest first -- a
// counting sort in one workgroup of 1024 threads (a launch has a few thousand packages, every key three dependent loads away;
// pulse counts go up to 1200, five to a bucket).  A chunk
// of devices only ever draws the packages of its own kind (a line code is an OOK or an FSK one, src/r_api.c:438-550): an OOK
// package under an FSK slicer is no work, but as an item it still cost a draw, four dependent loads and two barriers -- 38 us of
// a wavefront's life, 8948 times per chunk (five of the thirteen chunks of the default decoders never had anything else to do:
// profiles/r04_slice_shares.txt).  Also rewinds the cursors of the sizing pass that follows:
//   cursor[chunk]            the chunk's next entry of its kind's list (one launch), or of its large part (two launches)
//   cursor[chunks + chunk]   ... of its small part
//   cursor[2 * chunks + k]   k = 0..3: the ends of OOK large / OOK / FSK large / FSK (= everything) in `order`
__global__ __launch_bounds__(1024) void k_pkg_order(uint8_t const *arena, uint32_t arena_stride, uint32_t const *dir_stream,
        uint32_t const *dir_off, uint32_t const *n_pkgs_ptr, uint32_t max_pkgs, uint32_t pkg_begin, uint32_t pkg_end,
        uint32_t *order, uint32_t *cursor, uint32_t chunks, DevRow const *devs, uint32_t by_kind, unsigned long long *work_start)
{
    __shared__ uint32_t count[512], first[512];
    __shared__ uint32_t bound[4];
    uint32_t const n_pkgs = min(min(*n_pkgs_ptr, max_pkgs), pkg_end);
    auto key = [&](uint32_t pkg) -> uint32_t {
        uint32_t const *rec = (uint32_t const *)(arena + (uint64_t)dir_stream[pkg] * arena_stride + dir_off[pkg]);
        return (by_kind && rec[2] == R433_PKG_FSK ? 256u : 0u) + min(rec[3] / 5u, 255u);
    };
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x)
        count[i] = 0;
    __syncthreads();
    for (uint32_t pkg = pkg_begin + threadIdx.x; pkg < n_pkgs; pkg += blockDim.x)
        atomicAdd(&count[key(pkg)], 1u);
    __syncthreads();
    { // where each bucket begins: an exclusive scan over (kind, weight falling) -- eight wavefronts' scans and their totals
      // (one thread walking the 512 buckets through LDS took 45 of the kernel's 51 us)
        __shared__ uint32_t wave_total[8];
        uint32_t const j = threadIdx.x, kind = (j >> 8) & 1u, w = 255u - (j & 255u);
        uint32_t const v = j < 512 ? count[kind * 256 + w] : 0u;
        uint32_t tot;
        uint32_t ex = wave_excl_scan(v, tot);
        if (j < 512 && (j & 63u) == 0)
            wave_total[j >> 6] = tot;
        __syncthreads();
        if (j < 512) {
            for (uint32_t q = 0; q < (j >> 6); ++q)
                ex += wave_total[q];
            first[kind * 256 + w] = ex;
            if (w == kSmallW)
                bound[2 * kind] = ex + v; // the packages of kSmallPulses pulses and more come first
            if (w == 0)
                bound[2 * kind + 1] = ex + v;
        }
    }
    __syncthreads();
    if (!by_kind) { // one list for every chunk (small launches, see launch_slice_count): both kinds end where the list ends
        if (threadIdx.x == 0) {
            bound[2] = bound[0];
            bound[3] = bound[1];
        }
        __syncthreads();
    }
    for (uint32_t c = threadIdx.x; c < chunks; c += blockDim.x) {
        bool const fsk = by_kind && devs[c * 64].is_fsk != 0; // (a chunk is one line code; padding rows follow the real ones)
        cursor[c] = fsk ? bound[1] : 0u;
        cursor[chunks + c] = fsk ? bound[2] : bound[0];
    }
    if (threadIdx.x < 4)
        cursor[2 * chunks + threadIdx.x] = bound[threadIdx.x];
#ifndef R433_EMU
    if (threadIdx.x == 0 && work_start)
        *work_start = (unsigned long long)wall_clock64(); // (the sizing launches begin when this kernel ends)
#endif
    for (uint32_t pkg = pkg_begin + threadIdx.x; pkg < n_pkgs; pkg += blockDim.x)
        order[atomicAdd(&first[key(pkg)], 1u)] = pkg;
}

// Where each device's records of a package begin inside the package's stretch of the event stream: the exclusive prefix
// of sizes[pkg][.] in registration order, once per package (every (package, 64 devices) item of the placing pass needs
// one entry of it; computed inside that pass it was six dependent load + scan rounds per item, most of the pass).
__global__ __launch_bounds__(64) void k_dev_prefix(uint32_t const *sizes, uint32_t *dev_off, uint32_t const *n_pkgs_ptr,
        uint32_t max_pkgs, uint32_t n_devs, uint32_t pkg_begin, uint32_t pkg_end)
{
    uint32_t const n_pkgs = min(min(*n_pkgs_ptr, max_pkgs), pkg_end);
    uint32_t const lane = threadIdx.x;
    for (uint32_t pkg = pkg_begin + blockIdx.x; pkg < n_pkgs; pkg += gridDim.x) {
        uint32_t carry = 0;
        for (uint32_t b = 0; b < n_devs; b += 64) {
            uint32_t const i = b + lane;
            uint32_t const v = i < n_devs ? sizes[(uint64_t)pkg * n_devs + i] : 0u;
            uint32_t tot;
            uint32_t const ex = wave_excl_scan(v, tot);
            if (i < n_devs)
                dev_off[(uint64_t)pkg * n_devs + i] = carry + ex;
            carry += tot;
        }
    }
}

// Exclusive scan of byte counts.  Sums are kept in 64 bits and SATURATE at 0xffffffff on the way out: offsets are
// 32-bit by format, and a batch whose records pass 4 GiB must come back as R433_EOVERFLOW (the host checks the total),
// never as offsets that wrapped around and records that overwrite each other.
__global__ __launch_bounds__(1024) void k_scan_u32(uint32_t const *in, uint32_t *out, uint32_t const *n_ptr,
        uint32_t n_cap, uint32_t *total, uint32_t const *carry_in, uint32_t n_skip)
{
    __shared__ uint64_t part[1024];
    __shared__ uint64_t carry;
    uint32_t const n = min(*n_ptr, n_cap); // elements [n_skip, n) of in / out; the sums start from *carry_in
    int const tid = (int)threadIdx.x;
    if (tid == 0)
        carry = carry_in ? *carry_in : 0u;
    __syncthreads();
    for (uint32_t base = n_skip; base < n; base += 1024) {
        uint32_t i = base + (uint32_t)tid;
        uint64_t v = i < n ? in[i] : 0u;
        part[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint64_t add = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (i < n) {
            uint64_t const off = carry + part[tid] - v;
            out[i] = off > 0xffffffffull ? 0xffffffffu : (uint32_t)off;
        }
        __syncthreads();
        if (tid == 1023)
            carry += part[1023];
        __syncthreads();
    }
    if (tid == 0)
        *total = carry > 0xffffffffull ? 0xffffffffu : (uint32_t)carry;
}

// ---- the slice index: where each decoder's records lie in the (package-major) event stream ----
//
// A decoder's bitbuffers of one package are contiguous in the stream (a "slice": sizes[pkg][dev] bytes at pkg_off[pkg] +
// dev_off[pkg][dev]); the host's ordered replay walks a decoder's slices in package order.  Finding them on the host took
// two passes over the whole stream -- a third of the replay's CPU time, on hosts whose CPU quota is what bounds the
// pipeline.  Here: per decoder the list of its non-empty slices as (offset, bytes), a column compaction of the sizes
// matrix in three small launches (count per block of packages, scan, fill).
constexpr uint32_t kIdxBlock = 256; // packages per block

__global__ __launch_bounds__(64) void k_index_count(uint32_t const *sizes, uint32_t const *n_pkgs_ptr, uint32_t max_pkgs, uint32_t n_devs,
        uint32_t *cnt)
{
    uint32_t const n_pkgs = min(*n_pkgs_ptr, max_pkgs);
    uint32_t const chunks = (n_devs + 63) / 64;
    uint32_t const blk = blockIdx.x / chunks, d = (blockIdx.x % chunks) * 64 + threadIdx.x;
    uint32_t const p0 = blk * kIdxBlock, p1 = min(n_pkgs, p0 + kIdxBlock);
    if (d >= n_devs)
        return;
    // (210 wavefronts for the bench's step: the walk is memory latency, so eight loads in flight)
    uint32_t c = 0;
    for (uint32_t p = p0; p < p1; p += 8) {
        uint32_t sz[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u)
            sz[u] = p + u < p1 ? sizes[(uint64_t)(p + u) * n_devs + d] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u)
            c += sz[u] != 0u;
    }
    cnt[(uint64_t)blk * n_devs + d] = c;
}

// cnt[block][dev]: from a count to the first entry of that block in the decoder's list; start[dev]: first entry of the decoder
__global__ __launch_bounds__(256) void k_index_scan(uint32_t *cnt, uint32_t n_blocks, uint32_t n_devs, uint32_t *start, uint32_t *total)
{
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    int const tid = (int)threadIdx.x;
    if (tid == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t d0 = 0; d0 < n_devs; d0 += 256) {
        uint32_t const d = d0 + (uint32_t)tid;
        uint32_t run = 0;
        if (d < n_devs)
            for (uint32_t b = 0; b < n_blocks; ++b) {
                uint32_t const t = cnt[(uint64_t)b * n_devs + d];
                cnt[(uint64_t)b * n_devs + d] = run;
                run += t;
            }
        part[tid] = run;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) { // Hillis-Steele inclusive scan
            uint32_t const add = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        uint32_t const first = carry + part[tid] - run;
        if (d < n_devs) {
            start[d] = first;
            for (uint32_t b = 0; b < n_blocks; ++b)
                cnt[(uint64_t)b * n_devs + d] += first;
        }
        __syncthreads();
        if (tid == 255)
            carry += part[255];
        __syncthreads();
    }
    if (tid == 0) {
        start[n_devs] = carry;
        *total = carry;
    }
}

__global__ __launch_bounds__(64) void k_index_fill(uint32_t const *sizes, uint32_t const *dev_off, uint32_t const *pkg_off,
        uint32_t const *n_pkgs_ptr, uint32_t max_pkgs, uint32_t n_devs, uint32_t const *base, uint2 *slices, uint32_t cap)
{
    uint32_t const n_pkgs = min(*n_pkgs_ptr, max_pkgs);
    uint32_t const chunks = (n_devs + 63) / 64;
    uint32_t const blk = blockIdx.x / chunks, d = (blockIdx.x % chunks) * 64 + threadIdx.x;
    uint32_t const p0 = blk * kIdxBlock, p1 = min(n_pkgs, p0 + kIdxBlock);
    if (d >= n_devs)
        return;
    uint32_t at = base[(uint64_t)blk * n_devs + d];
    for (uint32_t p = p0; p < p1; p += 8) { // eight packages' sizes and offsets in flight (an offset is read whether its size is zero or not)
        uint32_t sz[8], off[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            bool const in = p + u < p1;
            sz[u] = in ? sizes[(uint64_t)(p + u) * n_devs + d] : 0u;
            off[u] = in ? pkg_off[p + u] + dev_off[(uint64_t)(p + u) * n_devs + d] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u)
            if (sz[u]) {
                if (at < cap)
                    slices[at] = make_uint2(off[u], sz[u]);
                at += 1;
            }
    }
}

uint32_t slice_grid(uint32_t grid_pkgs, uint32_t n_rows, uint32_t cap = 16384)
{
    uint32_t const chunks = n_rows / 64 ? n_rows / 64 : 1;
    uint64_t items = (uint64_t)grid_pkgs * chunks;
    if (items < chunks)
        items = chunks;
    // 256 CUs x 8 wavefront slots per SIMD pair is plenty; the kernel strides over the packages beyond this.  Always a
    // multiple of `chunks`: a workgroup serves one chunk of devices.
    return (uint32_t)(items < cap ? items : cap / chunks * chunks);
}

// The sizing launches DRAW their items: their workgroups should all be resident from the first moment and stay until the lists
// are dry -- one round.  The kernel fits six wavefronts to a SIMD (79 registers): 24 workgroups per CU.  With 16 384 workgroups
// for 6144 places the second round began when the first had drained its lists, paid its prologue for nothing and, worse, took
// its places from workgroups that had work: 8192 bench packages with the pre-filter on, sizing pass 2.69 ms at 16 384
// workgroups, 2.46 at 12 288, 2.04 at 8192, 1.66 at 6144, 1.77 at 4096 (profiles/r04_slice_shares.txt).
uint32_t sizing_grid_cap()
{
    static uint32_t cap = 0;
    if (!cap) {
        if (char const *e = getenv("R433_SLICE_GRID")) // development: A/B timing
            cap = (uint32_t)std::max(256, std::min(16384, atoi(e)));
        else {
            int cus = 256;
#ifndef R433_EMU
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
#endif
            cap = (uint32_t)std::min(16384, cus * 4 * R433_SLICE_MIN_WAVES);
        }
    }
    return cap;
}

} // namespace

// The shares of a launch, interleaved: chunk c's k-th workgroup wants to sit at (k + 1/2) / n_c of the way through the grid; the
// workgroups in the order of these places (ties: the lower chunk first) are the deal.  Every thread ranks one (chunk, k) pair by
// counting, chunk by chunk, the pairs in front of it.
struct DealShares {
    uint32_t first[17];
};
__global__ __launch_bounds__(256) void k_deal(DealShares s, uint32_t chunks, uint32_t grid, uint8_t *deal)
{
    uint32_t const j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= grid)
        return;
    uint32_t c = 0;
    while (c + 1 < chunks && j >= s.first[c + 1])
        ++c;
    uint64_t const k = j - s.first[c], nc = s.first[c + 1] - s.first[c];
    uint64_t rank = 0;
    for (uint32_t c2 = 0; c2 < chunks; ++c2) {
        uint64_t const n2 = s.first[c2 + 1] - s.first[c2];
        if (c2 == c) {
            rank += k;
            continue;
        }
        // pairs (c2, k2) with (2 k2 + 1) / n2 before (2 k + 1) / nc: (2 k2 + 1) nc < (2 k + 1) n2, or <= for a lower chunk
        uint64_t const a = (2 * k + 1) * n2, b2 = 2 * nc;
        uint64_t const cnt = c2 < c ? (a >= nc ? (a - nc) / b2 + 1 : 0) : (a > nc ? (a + nc - 1) / b2 : 0);
        rank += cnt < n2 ? cnt : n2;
    }
    deal[rank] = (uint8_t)c;
}

// Workgroups per chunk of devices in proportion to the work `w` measured for the chunks (at least `least` each); zeros = even.
static void share_out(uint32_t *first, uint32_t chunks, uint32_t grid, double const *w, uint32_t least)
{
    for (int c = 0; c < 17; ++c)
        first[c] = 0;
    // (at least a quarter of an even share, whatever was measured: the next batch may be of another kind -- FSK packages after
    // runs of OOK ones -- and a chunk left with eight workgroups would then take a hundred milliseconds to say so)
    least = std::max(least, least > 1 ? grid / (4 * chunks) : 1u);
    if (!w || chunks < 2 || chunks > 16 || grid < 2 * least * chunks)
        return;
    double sum = 0;
    for (uint32_t c = 0; c < chunks; ++c)
        sum += w[c] > 0 ? w[c] : 0;
    if (!(sum > 0))
        return;
    uint32_t n[16], total = 0, heaviest = 0;
    double const spare = (double)(grid - least * chunks);
    for (uint32_t c = 0; c < chunks; ++c) {
        n[c] = least + (uint32_t)(spare * (w[c] > 0 ? w[c] : 0) / sum);
        total += n[c];
        if (w[c] > w[heaviest])
            heaviest = c;
    }
    n[heaviest] += grid - total; // (rounding down left a few over)
    for (uint32_t c = 0; c < chunks; ++c)
        first[c + 1] = first[c] + n[c];
}

// (grid_pkgs: the packages of THIS launch, p.pkg_end - p.pkg_begin or fewer; shares: [2][16] work per chunk of devices as the
// engine's last run measured it for the large / small launch, or null)
void launch_slice_count(SliceParams const &p_in, uint32_t grid_pkgs, hipStream_t st, SliceFork const *fork, double const *shares, uint32_t least)
{
    // Lists per kind, shares by measured work and one resident round are for the large batches.  A launch of a thousand
    // packages lasts as long as its longest items: a workgroup per item, and every chunk walking the one list -- the chunks of the
    // other kind pass over it doing nothing, and that they hold their places meanwhile is what the busy ones want (1024 packages:
    // 0.59 ms so, 1.25 ms with the places all taken by slicing wavefronts; profiles/r04_slice_shares.txt).
    bool const big = grid_pkgs >= 4096 || least == 1; // (least == 1: R433_DEBUG_SKEW_SLICE, the tests' small launches through the large form)
    SliceParams p = p_in;
    if (!big) {
        shares = nullptr;
        p.chunk_work = nullptr;
    }
    uint32_t const chunks = p.n_rows / 64 ? p.n_rows / 64 : 1u;
    if (p.draw)
        hipLaunchKernelGGL(k_pkg_order, dim3(1), dim3(1024), 0, st, p.arena, p.arena_stride, p.dir_stream, p.dir_off, p.n_pkgs, p.max_pkgs,
                p.pkg_begin, p.pkg_end, p.pkg_order, p.cursor, chunks, p.devs, big ? 1u : 0u, p.chunk_work ? p.chunk_work + 64 : nullptr);
    dim3 const grid(slice_grid(grid_pkgs, p.n_rows, p.draw && grid_pkgs >= 4096 ? sizing_grid_cap() : 16384u));
    if (p.draw == 2) {
        // the small packages on the second stream beside the large ones (both wait for the order and for the deals, which are
        // made on the caller's stream BEFORE the fork: the second stream must not start on a table that is still being dealt)
        SliceParams large = p, small = p;
        small.chunk_deal = p.chunk_deal ? p.chunk_deal + 16384 : nullptr;
        int which = 0;
        for (SliceParams *q : {&large, &small}) {
            DealShares d;
            share_out(d.first, chunks, grid.x, p.chunk_deal && shares ? shares + 16 * which : nullptr, least);
            ++which;
            q->deal_grid = chunks <= 16 && d.first[chunks] == grid.x ? grid.x : 0u; // (first[] has 17 entries: more chunks get no deal)
            if (q->deal_grid)
                hipLaunchKernelGGL(k_deal, dim3((grid.x + 255) / 256), dim3(256), 0, st, d, chunks, grid.x, q->chunk_deal);
        }
        hipStream_t const st2 = fork ? fork->st2 : st;
        if (fork) {
            (void)hipEventRecord(fork->forked, st);
            (void)hipStreamWaitEvent(st2, fork->forked, 0);
        }
        if (p.stage) {
            hipLaunchKernelGGL((k_slice<M_STAGE, R433_PD_MAX_PULSES>), grid, dim3(64), 0, st, large);
            hipLaunchKernelGGL((k_slice<M_STAGE, (int)kSmallPulses>), grid, dim3(64), 0, st2, small);
        }
        else {
            hipLaunchKernelGGL((k_slice<M_COUNT, R433_PD_MAX_PULSES>), grid, dim3(64), 0, st, large);
            hipLaunchKernelGGL((k_slice<M_COUNT, (int)kSmallPulses>), grid, dim3(64), 0, st2, small);
        }
        if (fork) {
            (void)hipEventRecord(fork->joined, st2);
            (void)hipStreamWaitEvent(st, fork->joined, 0);
        }
        return;
    }
    SliceParams one = p;
    DealShares d;
    share_out(d.first, chunks, grid.x, p.draw && p.chunk_deal ? shares : nullptr, least);
    one.deal_grid = chunks <= 16 && d.first[chunks] == grid.x ? grid.x : 0u;
    if (one.deal_grid)
        hipLaunchKernelGGL(k_deal, dim3((grid.x + 255) / 256), dim3(256), 0, st, d, chunks, grid.x, one.chunk_deal);
    if (p.stage)
        hipLaunchKernelGGL(k_slice<M_STAGE>, grid, dim3(64), 0, st, one);
    else
        hipLaunchKernelGGL(k_slice<M_COUNT>, grid, dim3(64), 0, st, one);
}

void launch_scan_u32(uint32_t const *in, uint32_t *out, uint32_t const *n_ptr, uint32_t n_cap, uint32_t *total,
        hipStream_t st, uint32_t const *carry_in, uint32_t n_skip)
{
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, in, out, n_ptr, n_cap, total, carry_in, n_skip);
}

uint32_t slice_index_blocks(uint32_t n_pkgs)
{
    return (n_pkgs + kIdxBlock - 1) / kIdxBlock;
}

void launch_slice_index_count(uint32_t const *sizes, uint32_t const *n_pkgs_ptr, uint32_t max_pkgs, uint32_t n_pkgs, uint32_t n_devs,
        uint32_t *cnt, uint32_t *start, uint32_t *total, hipStream_t st)
{
    uint32_t const blocks = slice_index_blocks(n_pkgs), chunks = (n_devs + 63) / 64;
    if (!blocks || !chunks)
        return;
    hipLaunchKernelGGL(k_index_count, dim3(blocks * chunks), dim3(64), 0, st, sizes, n_pkgs_ptr, max_pkgs, n_devs, cnt);
    hipLaunchKernelGGL(k_index_scan, dim3(1), dim3(256), 0, st, cnt, blocks, n_devs, start, total);
}

void launch_slice_index_fill(uint32_t const *sizes, uint32_t const *dev_off, uint32_t const *pkg_off, uint32_t const *n_pkgs_ptr,
        uint32_t max_pkgs, uint32_t n_pkgs, uint32_t n_devs, uint32_t const *base, uint2 *slices, uint32_t cap, hipStream_t st)
{
    uint32_t const blocks = slice_index_blocks(n_pkgs), chunks = (n_devs + 63) / 64;
    if (!blocks || !chunks)
        return;
    hipLaunchKernelGGL(k_index_fill, dim3(blocks * chunks), dim3(64), 0, st, sizes, dev_off, pkg_off, n_pkgs_ptr, max_pkgs, n_devs, base,
            slices, cap);
}

void launch_slice_write(SliceParams const &p, uint32_t grid_pkgs, hipStream_t st)
{
    hipLaunchKernelGGL(k_dev_prefix, dim3(grid_pkgs < 1 ? 1 : grid_pkgs < 16384 ? grid_pkgs : 16384), dim3(64), 0, st, p.sizes, p.dev_off,
            p.n_pkgs, p.max_pkgs, p.n_devs, p.pkg_begin, p.pkg_end);
    if (p.stage && getenv("R433_PLACE_FROM_LDS")) // development: A/B timing of the form that stages the pulses of every re-sliced item in LDS
        hipLaunchKernelGGL(k_slice<M_COMPACT>, dim3(slice_grid(grid_pkgs, p.n_rows)), dim3(64), 0, st, p);
    else if (p.stage)
        hipLaunchKernelGGL((k_slice<M_COMPACT, 16>), dim3(slice_grid(grid_pkgs, p.n_rows)), dim3(64), 0, st, p);
    else
        hipLaunchKernelGGL(k_slice<M_WRITE>, dim3(slice_grid(grid_pkgs, p.n_rows)), dim3(64), 0, st, p);
}

} // namespace r433
