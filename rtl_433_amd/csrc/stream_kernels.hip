// stream_kernels.hip -- the IQ -> pulse-package kernel: one capture (or one verified segment of a long
// capture, SegDesc in r433_internal.hpp) per wavefront.
//
// A wavefront walks its samples in tiles of 2048 and runs three phases per tile:
//
//   A  sample-parallel   16-byte-per-lane coalesced IQ loads (issued one tile ahead), envelope and FM
//                        discriminator for 8 consecutive samples per lane, parked in LDS.
//   B  chunk-parallel    the two truncating first-order low-passes.  Lane j owns samples
//                        [32j, 32j+32) of the tile.  The recurrences are not associative, so instead of
//                        a scan every lane re-runs the 96 samples before its chunk from BOTH extreme
//                        carries: each step is a monotone map of the carry, the filter contracts, and
//                        when the two tracks meet the carry is exact whatever the history was.  Lanes
//                        whose tracks did not meet (constant input stalls on several fixed points) are
//                        resolved from their left neighbour -- in O(1) when the chunk provably maps
//                        every candidate carry to itself, by an exact re-run otherwise.  Nothing is
//                        assumed: a lane only publishes samples computed from a proven carry.
//   C  state machine     the OOK/FSK pulse detector, wave-uniform (every lane carries the same scalar
//                        state, so it lives in SGPRs).  The wavefront ballots over 64 samples at a time
//                        to find the next sample that can possibly change the state machine (a pulse
//                        start while idle, a falling edge inside a pulse, a rising edge or the
//                        end-of-package count inside a gap), runs lean recurrences over the samples in
//                        between (noise floor -- evaluated lazily --, level and carrier averages,
//                        counters) and the exact general step (detect_device.hpp) on the candidates.
//
// Packages leave as r433_pkg_rec records in a per-wavefront arena.  Frame semantics of the file reader
// (one push_sdr_flow call per 262144 input bytes) are reproduced at their exact sample positions.
//
// Replaces, for file input: envelope_detect / magnitude_est_* (reference src/baseband.c:36-110),
// baseband_low_pass_filter (:145-169), baseband_demod_FM(_cs16) (:210-366), the frame loop of
// push_sdr_flow (src/r_flow.c:149-244) and pulse_detect_package (src/pulse_detect.c:199-483).
#include <type_traits>

#include "dsp_device.hpp"
#include "r433_internal.hpp"

#ifdef R433_EMU_COUNTERS // development aid for the CPU emulator build only: how often each path of phase C runs
extern "C" unsigned long long r433_dbg_counts[32];
unsigned long long r433_dbg_counts[32];
#define CNT(i, n) do { if (lane == 0) r433_dbg_counts[i] += (unsigned long long)(n); } while (0)
#else
#define CNT(i, n) do { } while (0)
#endif

namespace r433 {

namespace {

constexpr int kChunk = 32;          // samples per lane in phase B
constexpr int kTile = 64 * kChunk;  // samples per tile
constexpr int kWarmChunks = 3;      // 96 warm-up samples: 0.854^96 * 2^16 < 1, 0.727^96 * 2^32 < 1 (72 were tried: the bound
                                    // still holds in the reals, but truncation keeps tracks one apart longer -- twice the resolve rounds)
constexpr int kRow = 512;           // samples per phase-A row (8 per lane)
constexpr int kRows = kTile / kRow;
constexpr int kPitch16 = kChunk * 2 + 16;  // LDS pitch of a chunk of 16-bit samples: conflict-free b128 per lane
constexpr int kPitch32 = kChunk * 4 + 16;
constexpr int kFloorWindow = 1024;          // samples a later segment of a split capture walks to find its noise floor
constexpr int kVirtual = 4 * kChunk;        // samples at the start of a filtered tile that follows unfiltered ones: bounds only
constexpr int kHead = 512;                  // ... and how far into the next tile a tile looks before it goes unfiltered (one row)
constexpr int kQuietTile = (int)0x80000000, kPostQuiet = 0x40000000;
constexpr bool kLazySplit = true;           // lazy tiles inside the pieces of a split capture (3.1b / 3.1c)
constexpr int kPitchOut = kChunk * 2;       // filtered samples: time-linear, read by sample index (the padded pitch buys
                                             // nothing there and 8 wavefronts' LDS must fit one CU: 8 x 20 KB = 160 KB)
// FORM 4 -> FORM 5 (the two roles as two kernels): what a tile hands from the filters to the detector, per (capture, tile)
// in HBM -- the filtered envelope and discriminator as the consumer reads them (2 x 64 chunks x 64 bytes, time-linear), the
// per-chunk extrema of the filtered envelope (64 + 64 shorts) -- and one descriptor word per tile beside it (quiet | bound ...)
constexpr int kTileRecAm = 0, kTileRecFm = 64 * kPitchOut, kTileRecMax = 2 * 64 * kPitchOut, kTileRecMin = kTileRecMax + 128;
static_assert(kTileRecMin + 128 == (int)kTileRecBytes, "the tile record of r433_internal.hpp");

__device__ __forceinline__ int rl0(int v)
{
    return __builtin_amdgcn_readlane(v, 0);
}

// A value every lane holds identically, but that the compiler cannot prove uniform (it came out of
// LDS): pin it to an SGPR so that everything computed from it stays on the scalar unit.
__device__ __forceinline__ int uni(int v)
{
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ int wave_sum(int v)
{
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int wave_max(int v)
{
    for (int o = 32; o > 0; o >>= 1)
        v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// The lanes of ONE wavefront exchange data through LDS: its own LDS instructions execute in order, so all that is needed is
// that the compiler keeps them in order (no workgroup barrier -- the other wavefront of the workgroup runs other code).
__device__ __forceinline__ void wave_sync()
{
#ifdef R433_EMU
    (void)emu::exchange<int>(0, 77); // the emulator runs lanes as fibers: a rendezvous makes every lane's stores land
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

__device__ __forceinline__ int ld16(uint8_t const *buf, int i)
{
    return (int)*(int16_t const *)(buf + i * 2);
}

// ---- phase B: one low-pass, both extreme tracks ----
//
// FAST = the host proved that no step can leave the state's integer range and that the feedback
// coefficient is non-negative (a >= 0 and a + 2b <= one in the filter's fixed point -- true for the
// AM filter and for every default FM filter): then each step is a monotone non-decreasing map of the
// carry, the low track stays the low track, and narrowing is the identity.  Otherwise every step is
// checked: an interval whose image could wrap is never trusted.

__device__ __forceinline__ int mul24(int a, int b)
{
    return __mul24(a, b);
}

// AM and cu8-FM filters: 16-bit state, y' = (a*y + k) >> 14 narrowed to int16 (src/baseband.c:161-163, 263)
template <bool FAST> struct Track16 {
    int lo, hi;
    int ok; // checked mode: no int16 wrap on an open interval so far (the sandwich argument holds)

    __device__ __forceinline__ void step(int a, int k)
    {
        int const v0 = (mul24(a, lo) + k) >> 14, v1 = (mul24(a, hi) + k) >> 14;
        if (FAST) {
            lo = v0;
            hi = v1;
        }
        else {
            int const fits = (int)((uint32_t)(v0 + 32768) < 65536u) & (int)((uint32_t)(v1 + 32768) < 65536u);
            ok &= (int)(lo == hi) | fits; // a proven carry may wrap like the reference does; an interval may not
            lo = (int)(int16_t)min(v0, v1);
            hi = (int)(int16_t)max(v0, v1);
        }
    }
    __device__ __forceinline__ bool exact() const { return (FAST || ok) && lo == hi; }
};

// cs16-FM filter: 32-bit state in Q30, y' = (a*y + k) >> 30 truncated to int32 (src/baseband.c:357)
template <bool FAST> struct Track32 {
    int lo, hi;
    int ok;

    __device__ __forceinline__ void step(long long a, long long k)
    {
        long long const v0 = (a * (long long)lo + k) >> 30, v1 = (a * (long long)hi + k) >> 30;
        if (FAST) {
            lo = (int)v0;
            hi = (int)v1;
        }
        else {
            int const fits = (int)(v0 == (long long)(int)v0) & (int)(v1 == (long long)(int)v1);
            ok &= (int)(lo == hi) | fits;
            lo = (int)(v0 < v1 ? v0 : v1);
            hi = (int)(v0 < v1 ? v1 : v0);
        }
    }
    __device__ __forceinline__ bool exact() const { return (FAST || ok) && lo == hi; }
};

// what a lane knows about one filter over its chunk after the first pass
struct ChunkStatus {
    bool start_known; // carry at chunk start proven (outputs published)
    bool end_known;   // carry at chunk end proven
    int ident;        // every carry in [lo0, hi0] is a fixed point of every step of the chunk
    int lo0, hi0;     // carry interval at chunk start
    int y_end;
};

template <int SS> struct Geom {
    static constexpr int f_pitch = SS == 2 ? kPitch16 : kPitch32;
    static constexpr int loads = SS == 2 ? kRows : 2 * kRows; // uint4 per lane per tile
};

// two int16 lanes in one register (v_pk_*_i16)
typedef short v2s __attribute__((vector_size(4)));
__device__ __forceinline__ v2s as_v2s(int w)
{
    v2s r;
    __builtin_memcpy(&r, &w, 4);
    return r;
}
__device__ __forceinline__ v2s pk_max(v2s a, v2s b)
{
    v2s const m = a > b; // -1 where a is larger
    return (a & m) | (b & ~m);
}
__device__ __forceinline__ v2s pk_min(v2s a, v2s b)
{
    v2s const m = a < b;
    return (a & m) | (b & ~m);
}

// inclusive running maximum over the lanes of the wavefront
__device__ __forceinline__ int wave_run_max(int key, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int const t = __shfl_up(key, (unsigned)o, 64);
        if (lane >= o)
            key = max(key, t);
    }
    return key;
}

// C division by 64 / 1024 (truncating toward zero) without a divider
__device__ __forceinline__ int div64(int v)
{
    return (v + ((v >> 31) & 63)) >> 6;
}
__device__ __forceinline__ int div1024(int v)
{
    return (v + ((v >> 31) & 1023)) >> 10;
}

// Eight steps of the two packed averages in their plain form, x += in - (x >> 6), inputs taken from lanes L .. L+7 of
// `rot`.  On the GPU the v_readlane of step u + 2 is issued while step u computes: a value read from another lane into
// an SGPR needs three instructions before a VALU instruction may use it, and left to itself the compiler fills that
// gap with an s_nop per step (5 issue slots per sample); software-pipelined it is 4.
template <int L> __device__ __forceinline__ void ema_group8(v2s &x, int rot)
{
#ifdef R433_EMU
    for (int u = 0; u < 8; ++u) {
        v2s const in = as_v2s(__builtin_amdgcn_readlane(rot, L + u));
        x = x + (in - (x >> 6));
    }
#else
    int xi, q, s0, s1;
    __builtin_memcpy(&xi, &x, 4);
    asm volatile("v_readlane_b32 %[s0], %[rot], %[l0]\n\t"
                 "v_readlane_b32 %[s1], %[rot], %[l1]\n\t"
                 "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\t"
                 "v_pk_sub_i16 %[x], %[x], %[q]\n\t"
                 "v_pk_add_u16 %[x], %[x], %[s0]\n\t"
                 "v_readlane_b32 %[s0], %[rot], %[l2]\n\t"
                 "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\t"
                 "v_pk_sub_i16 %[x], %[x], %[q]\n\t"
                 "v_pk_add_u16 %[x], %[x], %[s1]\n\t"
                 "v_readlane_b32 %[s1], %[rot], %[l3]\n\t"
                 "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\t"
                 "v_pk_sub_i16 %[x], %[x], %[q]\n\t"
                 "v_pk_add_u16 %[x], %[x], %[s0]\n\t"
                 "v_readlane_b32 %[s0], %[rot], %[l4]\n\t"
                 "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\t"
                 "v_pk_sub_i16 %[x], %[x], %[q]\n\t"
                 "v_pk_add_u16 %[x], %[x], %[s1]\n\t"
                 "v_readlane_b32 %[s1], %[rot], %[l5]\n\t"
                 "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\t"
                 "v_pk_sub_i16 %[x], %[x], %[q]\n\t"
                 "v_pk_add_u16 %[x], %[x], %[s0]\n\t"
                 "v_readlane_b32 %[s0], %[rot], %[l6]\n\t"
                 "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\t"
                 "v_pk_sub_i16 %[x], %[x], %[q]\n\t"
                 "v_pk_add_u16 %[x], %[x], %[s1]\n\t"
                 "v_readlane_b32 %[s1], %[rot], %[l7]\n\t"
                 "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\t"
                 "v_pk_sub_i16 %[x], %[x], %[q]\n\t"
                 "v_pk_add_u16 %[x], %[x], %[s0]\n\t"
                 "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\t"
                 "v_pk_sub_i16 %[x], %[x], %[q]\n\t"
                 "v_pk_add_u16 %[x], %[x], %[s1]"
                 : [x] "+v"(xi), [q] "=&v"(q), [s0] "=&s"(s0), [s1] "=&s"(s1)
                 : [rot] "v"(rot), [l0] "n"(L), [l1] "n"(L + 1), [l2] "n"(L + 2), [l3] "n"(L + 3), [l4] "n"(L + 4),
                   [l5] "n"(L + 5), [l6] "n"(L + 6), [l7] "n"(L + 7));
    __builtin_memcpy(&x, &xi, 4);
#endif
}

// up to eight such groups from lane 0 of `rot`
__device__ __forceinline__ void ema_groups(v2s &x, int rot, int nb)
{
    ema_group8<0>(x, rot);
    if (nb < 2) return;
    ema_group8<8>(x, rot);
    if (nb < 3) return;
    ema_group8<16>(x, rot);
    if (nb < 4) return;
    ema_group8<24>(x, rot);
    if (nb < 5) return;
    ema_group8<32>(x, rot);
    if (nb < 6) return;
    ema_group8<40>(x, rot);
    if (nb < 7) return;
    ema_group8<48>(x, rot);
    if (nb < 8) return;
    ema_group8<56>(x, rot);
}

// FORM 1: one wavefront does both halves in turn (workgroups of 64).  FORM 2: a producer and a consumer wavefront per
// capture (workgroups of 128).  FORM 3: a producer and TWO consumers (workgroups of 192) for the later pieces of a split
// capture, which exist in two parity variants of the assumed noise floor: the filters do not depend on the variant.
// Separate kernels, because the forms want different register budgets: in a pair or a triple a wavefront runs one role
// only and fits three to a SIMD, the lone wavefront carries both roles' state across the tile loop.
// FORM 4 and FORM 5 (round 5): the two roles of a pair as TWO LAUNCHES of one-wavefront workgroups.  A producer of a pair
// idles 60 % of its time at the tile barrier once most tiles go by unfiltered, and holds a wavefront slot (and its share of
// the LDS) while it does: of the three wavefronts a SIMD holds, one and a half are consumers.  FORM 4 runs the producers
// alone -- nobody to wait for -- and leaves every tile's filtered samples, chunk extrema and descriptor in HBM
// (StreamParams::tile_store / tile_desc: 8.4 KB per FILTERED tile written once and read once; the chip's HBM is at 3 % of
// its bandwidth in this kernel, its issue slots are what is scarce); FORM 5 runs the consumers alone, three to a SIMD, each
// copying a filtered tile into its own 8 KB of LDS and walking it as ever.  A capture whose unfiltered tiles cannot be
// carried exactly (st_retry in the pair) is started over by the producer itself where the producer finds out, and put on
// StreamParams::retry_list where the consumer does: a third launch (the pair kernel, every tile filtered, workgroups
// that find nothing on the list leave at once) runs those again.
#ifndef R433_CONSUMER_WAVES
#define R433_CONSUMER_WAVES 3
#endif
#ifndef R433_PRODUCER_WAVES
#define R433_PRODUCER_WAVES 4
#endif
#ifdef R433_EMU
#define R433_WAVES_PER_SIMD(n)
#else
#define R433_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif
template <int SS, bool FAST, bool FM, bool SEAM = false, int FORM = 1> __global__ __launch_bounds__((FORM >= 4 ? 1 : FORM) * 64)
        R433_WAVES_PER_SIMD(FORM == 1 ? 2 : FORM == 4 ? (SS == 4 && FM ? 3 : R433_PRODUCER_WAVES) : FORM == 5 ? R433_CONSUMER_WAVES : 3) void k_wave(StreamParams p) // (cs16 at 168 VGPRs spills a dozen dwords; at two per SIMD and 208 VGPRs a 64 Mi-sample stream took 10.4 ms against 8.1)
{
    using G = Geom<SS>;
    constexpr bool to_hbm = FORM == 4, from_hbm = FORM == 5; // the producers / the consumers of a grid as a launch of their own
    __shared__ __attribute__((aligned(16))) uint8_t s_env[64 * kPitch16];
    __shared__ __attribute__((aligned(16))) uint8_t s_f[from_hbm ? 16 : 64 * G::f_pitch]; // (the consumers never stage a discriminator)
    // Two wavefronts per capture when launched with 128 threads: wavefront 0 PRODUCES (phases A + B of tile t + 1 into
    // buffer (t + 1) & 1) while wavefront 1 CONSUMES (phase C of tile t from buffer t & 1); one workgroup barrier per tile.
    // The two sit on different SIMDs of the CU (a workgroup's wavefronts are dealt out round-robin), each next to a
    // wavefront of another capture, so a SIMD issues for two wavefronts instead of one.  Launched with 64 threads the one
    // wavefront does both in turn (filters-only launches; R433_DEBUG_ONE_WAVE for A/B timing).
    // (the tile buffers are dynamic LDS: two of each for a pair, one for a single wavefront -- 8 KB that decide whether
    // five or eight single-wavefront workgroups fit a CU)
#ifdef R433_EMU
    __shared__ __attribute__((aligned(16))) uint8_t s_tiles[4 * 64 * kPitchOut + 64 * kPitch16];
#else
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tiles[];
#endif
    // What else crosses from phase B to phase C lives in the 16 bytes of padding behind every chunk of s_env (phase A only
    // ever stores the 64 bytes in front of them): per chunk and tile buffer the extrema of its filtered envelope, and in
    // chunks 0..2 the producer's flags.  Kept out of arrays of their own, a pair's LDS is 26 KB: six workgroups to a CU.
    auto st_cmax = [&](int buf, int chunk) -> short & { return *(short *)(s_env + chunk * kPitch16 + 2 * kChunk + buf * 2); };
    auto st_cmin = [&](int buf, int chunk) -> short & { return *(short *)(s_env + chunk * kPitch16 + 2 * kChunk + 4 + buf * 2); };
    // producer -> consumer, per buffer: the establishing tile could not be proven
    auto st_pflag = [&](int buf) -> int & { return *(int *)(s_env + buf * kPitch16 + 2 * kChunk + 8); };
    int &s_pover = *(int *)(s_env + 2 * kPitch16 + 2 * kChunk + 8); // producer -> consumer: a filter carry was refused (det.overflow codes 2, 3)
    // Lazy tiles (see `lazy_cfg` below).  producer -> consumer, per buffer: how the tile comes -- kQuietTile | bound: no
    // samples at all, only an upper bound of its filtered envelope; kPostQuiet | bound: filtered samples from sample
    // kVirtual on, a bound for the ones before; 0: as ever.
    auto st_desc = [&](int buf) -> int & { return *(int *)(s_env + (3 + buf) * kPitch16 + 2 * kChunk + 8); };
    // either role -> both, by the parity of the barrier that will publish it: run the capture again with every tile filtered
    auto st_retry = [&](int parity) -> int & { return *(int *)(s_env + (5 + parity) * kPitch16 + 2 * kChunk + 8); };

    int const lane = (int)threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6); // (a scalar: the role branches below are scalar branches)
    constexpr bool solo = FORM == 1; // one wavefront does both halves
    constexpr bool lone = FORM == 1 || FORM >= 4; // workgroups of one wavefront: one tile buffer, no workgroup barriers
    uint8_t *const s_am = s_tiles, *const s_fm = s_tiles + (lone ? 1 : 2) * (64 * kPitchOut);
    // am.s16 / fm.s16 input files (RUN_AM_IS_INPUT / RUN_FM_IS_INPUT; the launch adds the room): the tile's words as they came
    uint8_t *const s_raw = s_tiles + (lone ? 2 : 4) * (64 * kPitchOut);
    bool const raw_in = SS == 2 && !SEAM && (p.flags & (RUN_AM_IS_INPUT | RUN_FM_IS_INPUT)) != 0;
    // role 0 produces, role 1 consumes.  Workgroups alternate which wavefront takes which role, so that the two wavefronts
    // that end up on one SIMD are one of each kind, and the consumer -- the serial critical path -- issues first.
    int const role = lone ? (from_hbm ? 1 : 0) : FORM == 3 ? (wave + (int)(blockIdx.x % 3u)) % 3
                                                           : (wave ^ ((p.flags & RUN_NO_ROLE_SWAP) ? 0 : (int)(blockIdx.x & 1u)));
    if (!lone && role != 0 && !(p.flags & RUN_NO_PRIO))
        __builtin_amdgcn_s_setprio(3);
    // the run-again launch behind a FORM 4 / FORM 5 pass: workgroups beyond the list the consumers left have nothing to do
    if (FORM == 2 && p.wg_count && blockIdx.x >= *p.wg_count)
        return;
    if (threadIdx.x == 0) {
        s_pover = 0;
        st_desc(0) = st_desc(1) = 0;
        st_retry(0) = st_retry(1) = 0;
    }
    if (!lone)
        __syncthreads(); // (whichever wavefront produces may raise these flags in its first tile)
    else
        wave_sync();
    // workgroup = one capture, or one piece of a split capture (in both parity variants: consumers 1 and 2 of a triple)
    uint32_t const wg = p.wg_slot ? p.wg_slot[blockIdx.x] : blockIdx.x;
    bool const idle = FORM == 3 && role == 2 && !(wg >> 31); // a triple whose piece has one variant only: the third wavefront just keeps the barriers
    // (read back through a lane: the slot differs between the wavefronts of a triple, so the compiler takes it -- and the capture,
    // the piece's bounds, every pointer made from them -- for per-lane values; in vector registers they cost the three-wavefront
    // forms 25 spilled registers, 50 MB of scratch written back per config-3 pass)
    // (The builtin, not uni(): same instruction, but at this register pressure the allocation is a coin toss -- 16 bytes of
    // scratch per lane this way, 188 through the inlined helper.  tools/kres.sh after every change of this kernel.)
    uint32_t const s = (uint32_t)__builtin_amdgcn_readfirstlane((int)((wg & 0x7fffffffu) + (FORM == 3 && role == 2 && !idle ? 1u : 0u)));
    uint32_t const cap = p.segs ? p.segs[s].capture : s;
    uint32_t const my_bytes = p.stream_bytes ? p.stream_bytes[cap] : p.uniform_bytes;
    uint32_t const my_n = my_bytes / SS;
    uint32_t const n_tiles = (my_n + kTile - 1) / kTile;
    uint8_t const *const iq = p.iq + (uint64_t)cap * p.stride_bytes;
    uint32_t const F = p.frame_samples;
    uint32_t const seg_flags = p.segs ? p.segs[s].flags : (SEG_FIRST | SEG_LAST | SEG_PRIMARY);
    uint32_t const seg_start = p.segs ? p.segs[s].start : 0u;
    uint32_t const seg_end = p.segs ? min(p.segs[s].end, my_n) : my_n;
    bool const seg_first = (seg_flags & SEG_FIRST) != 0, seg_primary = (seg_flags & SEG_PRIMARY) != 0;
    // a later segment starts one tile early: that tile only establishes the filter carries and the floor
    uint32_t const tile_first = seg_first ? 0u : seg_start / kTile - 1u;
    uint32_t const tile_end = (seg_end + kTile - 1) / kTile;
    int seg_fail = 0, seg_init_low = 0, seg_init_high = 0;
    // FORM 4 / FORM 5: this capture's tile records and descriptors in HBM
    uint8_t *const g_tiles = (to_hbm || from_hbm) ? p.tile_store + (uint64_t)s * p.tiles_cap * kTileRecBytes : nullptr;
    int *const g_desc = (to_hbm || from_hbm) ? p.tile_desc + (uint64_t)s * p.tiles_cap : nullptr;
    int self_retry = 0; // FORM 4: the producer found that the capture cannot be carried across its unfiltered tiles
    int p_active = 0;   // FORM 4: per lane (= chunk) the filtered tiles whose envelope rose above the quiet level there: the consumer's work

    // ---- detector: wave-uniform.  Every lane carries the same scalar state and takes the same
    // branches, in the fast paths and in the general step alike; lane 0 alone touches the arena and
    // the FSK ring (detect_device.hpp).
    DetLane det;
    DetCfg cfg = p.det; // min_high follows p.frame_min_high per frame when -Y autolevel is on
    uint32_t frame;
    uint64_t input_pos;
    int dc, flen;
    // ---- filter carries across tiles (wave-uniform) ----
    int carry_ya, carry_xa; // AM low-pass: y[-1], x[-1]
    int carry_yf, carry_ff; // FM low-pass: y[-1], discriminator[-1]
    int carry_i, carry_q;   // last IQ sample, centred
    int seam_end[4];        // SEAM: the carries after the last sample

    // ---- Lazy tiles.  A tile whose raw envelope -- with the 128 samples before it and the first row of the tile after it --
    // stays below every level at which the detector could start, hold or end a pulse (quiet_bound) cannot change the state
    // machine beyond what counting does: the filtered envelope is a convex mix of the raw one (a + 2b <= 1, floor), so 96
    // samples into such a stretch it is below that level too, a pulse that was open has ended, its debounce is over, and the
    // detector idles or counts a gap.  The producer then skips both filters and the discriminator (4/5 of its instructions go
    // with them) and hands over a bound instead of samples; the consumer counts -- the gap, or the steps of the noise floor,
    // which it evaluates lazily anyway (resolve_low) and now carries across such tiles.  The filter carries stay exact
    // through a tile of identical samples (digital silence: iterate the one map to its fixed point); after any other quiet
    // tile they are established again over the next filtered tile's first kVirtual samples (two tracks, as every chunk's
    // are), which the consumer still takes by their bound.  What cannot be carried exactly -- a floor walk that does not meet
    // in the samples it has, carries that do not settle, |am - low| out of the +-1 regime -- raises st_retry and the whole
    // capture runs again with every tile filtered: nothing is ever published from an assumption.
    // Off for split captures (their pieces are verified against each other by the host), the function seam, taps, the logic
    // dump, sample files that are not IQ, filters outside the FAST class, frames that are not whole tiles.
    // (Pieces of a split capture, round 5: the same inside a piece.  Its establishing tile and -- unless the piece ends the
    // capture -- its last tile are always filtered: the first proves the carries and the floor the piece starts from, the last
    // lets the floor be settled over real samples before the host's stitch compares it with what the next piece assumed.  A piece
    // that cannot be carried starts over with every tile filtered like a whole capture does; kLazySplit = false: as in round 4.)
    bool const lazy_cfg = !SEAM && FAST && (kLazySplit || (FORM != 3 && !p.segs)) && !p.tap_env && !p.tap_am && !p.logic && F % (uint32_t)kTile == 0
            && !(p.flags & (RUN_NO_LAZY | RUN_DBG_SKIP_FILTERS | RUN_AM_IS_INPUT | RUN_FM_IS_INPUT | RUN_ENV_RAW16));
    bool lazy = lazy_cfg;        // this attempt
    int attempts = 0, n_quiet = 0;
    uint32_t sums_done = tile_first; // tiles whose frame sums are in p.frame_sums (a second attempt must not add them again)
    int retry_slot = 0;          // parity of the barrier the running tile ends at
    // producer side
    bool carry_known = true;     // carry_ya / carry_yf are the exact filter states (the other four always are)
    bool prev_quiet = false;     // the tile before this one went by unfiltered
    int prev_bound = 0;          // ... and no filtered sample of it exceeded this
    int tail_max = 0;            // raw envelope maximum over the last 128 samples before the tile
    uint32_t carry_w = 0;        // the last sample as it came (all samples of a tile equal to it: a constant tile)
    // consumer side: pending steps of the noise floor (see resolve_low)
    int lz_n = 0, lz_from = 0, lz_min = 0x7fffffff, lz_max = -0x7fffffff;
    bool lz_carried = false;     // some of them over samples that never were filtered: only a two-sided walk can settle them
    int lzc_min = 0x7fffffff, lzc_max = -0x7fffffff; // ... and the bounds of those samples alone
    int lz_fail = 0;             // why the capture has to run again (the codes of r433_batch_debug_state: 4.. the consumer's)
    int retry_why = 0;
    int p_fail = 0, p_over = 0;  // producer side of seg_fail / det.overflow
    if (FORM == 2 && (p.flags & RUN_RETRY_PASS)) { // the run-again launch behind a FORM 4 / FORM 5 pass: what the first attempt found
        attempts = 1;
        retry_why = (int)p.retry_why[s];
    }

    // (Two of them: what a role keeps across tiles must be dead in the other role's loop, or the pair kernel does not fit its
    // 168 registers -- with one shared start-over loop every scalar of both roles was alive around its back edge, the
    // consumer spilled eighteen registers at every entry to the train engine and a grid of 8192 captures wrote 900 MB of
    // scratch.)
    auto init_consumer = [&]() {
        // ---- detector: wave-uniform.  Every lane carries the same scalar state and takes the same
        // branches, in the fast paths and in the general step alike; lane 0 alone touches the arena and
        // the FSK ring (detect_device.hpp).
        cfg = p.det;
        det_reset(det);
        det.arena = p.arena + (uint64_t)s * p.arena_stride;
        det.fsk_ring = p.fsk_ring + (uint64_t)s * R433_PD_MAX_PULSES; // HBM scratch, touched by lane 0 on FSK pulses only
        det.arena_cap = p.arena_stride;
        det.stream = cap;
        det.writer = lane == 0;
        det.cursor = 0;
        det.ook_base = 0;
        det.n_pkgs = 0;
        det.overflow = 0;
        frame = seg_start / F;
        input_pos = (uint64_t)frame * F;
        dc = (int)(seg_start - frame * F);
        flen = (int)min(my_n - frame * F, F); // only meaningful when the segment starts inside a frame
        if (!seg_first)
            det.lead_in = 1025; // saturated for good after the first 1025 idle samples of a capture
        if (p.frame_min_high) // -Y autolevel: the level of the frame the segment starts in
            cfg.min_high = p.frame_min_high[(uint64_t)cap * p.frames_cap + min(frame, p.frames_cap - 1)];
        seg_fail = seg_init_low = seg_init_high = 0;
        lz_n = lz_from = 0;
        lz_min = 0x7fffffff, lz_max = -0x7fffffff;
        lz_carried = false;
        lzc_min = 0x7fffffff, lzc_max = -0x7fffffff;
        lz_fail = 0;
        n_quiet = 0;
    };
    auto init_producer = [&]() {
        carry_ya = carry_xa = carry_yf = carry_ff = carry_i = carry_q = 0;
        if (SEAM && p.seam_init) { // a frame that continues a stream: filter_state_t / demodfm_state_t of the caller
            carry_ya = p.seam_init[0], carry_xa = p.seam_init[1];
            carry_yf = p.seam_init[2], carry_ff = p.seam_init[3];
            carry_i = p.seam_init[4], carry_q = p.seam_init[5];
        }
        seam_end[0] = carry_ya, seam_end[1] = carry_xa, seam_end[2] = carry_yf, seam_end[3] = carry_ff;
        carry_known = true;
        prev_quiet = false;
        prev_bound = 0;
        tail_max = 0;
        carry_w = SS == 2 ? 0x8080u : 0u; // the centred (0, 0) the discriminator starts from
        p_fail = p_over = 0;
    };

    // profiling aid (RUN_DBG_TIMING): shader-clock ticks per phase, returned in unused StreamState slots
#ifdef R433_KERNEL_TIMING // a development build (python -m rtl_433_amd.build --timing): the counters cost registers in the hot loops
    bool const timing = (p.flags & RUN_DBG_TIMING) != 0;
#else
    constexpr bool timing = false;
#endif
    // phase timing (RUN_DBG_TIMING only): A+B, idle, gap, pulse, gap-start, general step, resolve, iterations.
    // In LDS behind a scalar branch: an array in registers costs a select chain per update even when unused.
#ifdef R433_KERNEL_TIMING
    __shared__ long long s_tk[16]; // 8..15: inside the train engine (window loads, pulse prologue, averages, candidate check, debounce, gap, chunk skip, legs)
    if (timing && lane < 16)
        s_tk[lane] = 0;
#else
    long long s_tk[16] = {0}; // never touched: `timing` is a constant false
#endif
    auto tick = [&](int slot, long long since) {
        if (timing) {
            long long const d = (long long)clock64() - since;
            if (lane == 0)
                s_tk[slot] += d;
        }
    };
    auto now = [&]() -> long long { return timing ? (long long)clock64() : 0ll; };

    // ---- `u8` logic dump (optional): the reference paints one byte per sample into a per-frame buffer -- 0x01 | bits over the
    // pulses of a package, 0x01 over its gaps -- when a package is returned, and again for whatever its two pulse_data_t hold
    // when the frame ends (src/r_flow.c:271-272,314-315,364-371; bounded_memset clips to the frame; later paints overwrite
    // earlier ones, src/pulse_data.c:45-67).  Same paints, same order, same clipping; lanes share the pairs of a paint.
    uint8_t *const logic = (!SEAM && p.logic && seg_primary) ? p.logic + (uint64_t)cap * p.logic_stride : nullptr;
    auto paint = [&](uint64_t offset, uint32_t n_pairs, int2 const *pairs, int bits, uint64_t win0, int win_len) {
        long long run = (long long)offset - (long long)win0; // position of the pair being painted, relative to the frame
        for (uint32_t b0 = 0; b0 < n_pairs; b0 += 64) {
            uint32_t const kx = b0 + (uint32_t)lane;
            int pw = 0, gw = 0;
            if (kx < n_pairs) { // lane 0 wrote these: read them past the vector cache
                pw = __hip_atomic_load(&pairs[kx].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gw = __hip_atomic_load(&pairs[kx].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int incl = pw + gw;
            for (int o = 1; o < 64; o <<= 1) {
                int const t = __shfl_up(incl, (unsigned)o, 64);
                if (lane >= o)
                    incl += t;
            }
            long long const start = run + (long long)(incl - (pw + gw));
            auto span = [&](long long from, int len, int value) { // bounded_memset
                long long a = from < 0 ? 0 : from, z = from + len > win_len ? win_len : from + len;
                for (long long x = a; x < z; ++x)
                    logic[win0 + (uint64_t)x] = (uint8_t)value;
            };
            span(start, pw, 0x01 | bits);
            span(start + pw, gw, 0x01);
            run += (long long)__shfl(incl, 63, 64);
        }
    };

    uint4 pf[G::loads];
    uint4 hd[SS == 2 ? 1 : 2]; // lazy tiles look one row past their end: row 0 of the tile after the one in `pf`
    // (have_head: `hd` already holds this tile's first row -- it was looked at when the tile before was judged)
    auto issue_loads = [&](uint32_t tile, bool have_head = false) {
#pragma unroll
        for (int k = 0; k < G::loads; ++k) {
            if (have_head && k < (SS == 2 ? 1 : 2)) {
                pf[k] = hd[k];
                continue;
            }
            uint64_t samp = (uint64_t)tile * kTile + (uint64_t)(SS == 2 ? k : k / 2) * kRow + (uint64_t)lane * 8;
            uint64_t off = samp * SS + (SS == 2 ? 0 : (k & 1) * 16);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tile < n_tiles && off + 16 <= p.stride_bytes)
                v = *(uint4 const *)(iq + off);
            pf[k] = v;
        }
    };
    auto issue_head = [&](uint32_t tile) {
#pragma unroll
        for (int k = 0; k < (SS == 2 ? 1 : 2); ++k) {
            uint64_t const off = ((uint64_t)tile * kTile + (uint64_t)lane * 8) * SS + (uint64_t)k * 16;
            uint4 v = make_uint4(0, 0, 0, 0); // (past the capture: as loud as can be for cu8, and nothing to look at anyway)
            if (tile < n_tiles && off + 16 <= p.stride_bytes)
                v = *(uint4 const *)(iq + off);
            hd[k] = v;
        }
    };
    // raw envelope of one sample as it came
    auto env_of = [&](uint32_t w) -> int {
        if (SS == 2)
            return (int)(p.use_mag ? env_mag_cu8(w & 0xffu, (w >> 8) & 0xffu) : env_amp_cu8(w & 0xffu, (w >> 8) & 0xffu));
        return (int)env_mag_cs16((int)(int16_t)(w & 0xffffu), (int)(int16_t)(w >> 16));
    };
    // No filtered sample at or below this can start, hold or end a pulse, whatever the detector's levels are: the threshold is
    // (low + min(high, max)) / 2 with low >= -1 (the floor follows samples that are >= 0) and high >= min_high at every sample
    // (src/pulse_detect.c:283,300-304,332-334,362-363), thr - thr / 8 grows with thr.  -1: no such level.
    auto quiet_bound = [&](int min_high) -> int {
        int thr = (int)(int16_t)((-1 + min(min_high, p.det.max_high)) / 2);
        if (p.det.fixed_high != 0)
            thr = (int)(int16_t)p.det.fixed_high;
        return thr > 0 ? thr - (int)(int16_t)(thr / 8) - 1 : -1;
    };

    // ======================================= the producer: phases A and B of one tile =======================================
    auto produce = [&](uint32_t tile, int buf) {
        uint32_t const t0 = tile * kTile;                        // absolute sample index of the tile
        int const n_t = (int)min((uint32_t)kTile, seg_end - t0); // valid samples in it
        bool const warm = !seg_first && tile == tile_first;      // the establishing tile of a later segment
        uint8_t *const p_am = to_hbm ? g_tiles + (uint64_t)tile * kTileRecBytes + kTileRecAm : s_am + buf * (64 * kPitchOut);
        uint8_t *const p_fm = to_hbm ? g_tiles + (uint64_t)tile * kTileRecBytes + kTileRecFm : s_fm + buf * (64 * kPitchOut);
        auto put_desc = [&](int d) { // (lane 0)
            if (to_hbm)
                g_desc[tile] = d;
            else
                st_desc(buf) = d;
        };
        int p_retry = 0;
        bool const post_q = lazy && prev_quiet;  // the tile before went by unfiltered: this one's first kVirtual samples are taken by their bound
        bool const est = lazy && !carry_known;   // ... and left the two filter states unknown: every lane warms up from extremes
        int vbound = 0;                          // bound of those kVirtual samples

        // ================= lazy tiles: a first pass over the registers, raw envelope only =================
        if (lazy) {
            int lmax = 0, row0 = 0, rlast = 0;
            uint32_t lsum = 0, diff = 0;
            uint32_t const refw = SS == 2 ? carry_w * 0x10001u : carry_w;
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                int rmax = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    uint32_t w;
                    if (SS == 2) {
                        uint32_t const ww[4] = {pf[r].x, pf[r].y, pf[r].z, pf[r].w};
                        if ((j & 1) == 0)
                            diff |= ww[j >> 1] ^ refw;
                        w = (ww[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                    }
                    else {
                        uint32_t const ww[8] = {pf[2 * r].x, pf[2 * r].y, pf[2 * r].z, pf[2 * r].w, pf[2 * r + 1].x, pf[2 * r + 1].y, pf[2 * r + 1].z, pf[2 * r + 1].w};
                        w = ww[j];
                        diff |= w ^ refw;
                    }
                    int const e = env_of(w);
                    rmax = max(rmax, e);
                    lsum += (uint32_t)e;
                }
                lmax = max(lmax, rmax);
                if (r == 0)
                    row0 = rmax;
                if (r == kRows - 1)
                    rlast = rmax;
            }
            int hmax = 0; // the first row of the tile after this one
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t w;
                if (SS == 2) {
                    uint32_t const ww[4] = {hd[0].x, hd[0].y, hd[0].z, hd[0].w};
                    w = (ww[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                }
                else {
                    uint32_t const ww[8] = {hd[0].x, hd[0].y, hd[0].z, hd[0].w, hd[SS == 2 ? 0 : 1].x, hd[SS == 2 ? 0 : 1].y, hd[SS == 2 ? 0 : 1].z, hd[SS == 2 ? 0 : 1].w};
                    w = ww[j];
                }
                hmax = max(hmax, env_of(w));
            }
            int E = quiet_bound(p.det.min_high);
            if (p.frame_min_high) { // -Y autolevel: the lowest level among the frames the stretch touches
                uint32_t const fl = p.frames_cap - 1;
                int const *const mh = p.frame_min_high + (uint64_t)cap * p.frames_cap;
                E = min(min(quiet_bound(mh[min(t0 ? (t0 - 1u) / F : 0u, fl)]), quiet_bound(mh[min(t0 / F, fl)])), quiet_bound(mh[min((t0 + (uint32_t)kTile) / F, fl)]));
            }
            bool const must_filter = warm || (p.segs && !(seg_flags & SEG_LAST) && tile + 1u >= tile_end); // (a piece's first and last tile)
            bool const quiet = uni((int)(n_t == kTile && !must_filter && tail_max <= E && !__ballot(lmax > E || hmax > E))) != 0;
            if (post_q) // (the head rule of the tile before this one: these samples are below its level)
                vbound = max(prev_bound, wave_max(lane < kVirtual / 8 ? row0 : 0));
            if (!quiet) {
                tail_max = wave_max(lane >= 64 - 128 / 8 ? rlast : 0);
            }
            else {
                int const qmax = wave_max(lmax);
                bool const constant = !__ballot(diff != 0u); // every sample of the tile is the sample before it again
                if (p.frame_sums && seg_primary && tile >= sums_done) { // (a tile lies in one frame: F is a multiple of the tile)
                    int const part = wave_sum((int)lsum);
                    uint32_t const f = t0 / F;
                    if (lane == 0 && f < p.frames_cap)
                        atomicAdd(&p.frame_sums[(uint64_t)cap * p.frames_cap + f], (uint32_t)part);
                }
                sums_done = max(sums_done, tile + 1u);
                // the four carries that are plain samples: the last sample, its envelope, its discriminator value
                uint32_t const w7 = SS == 2 ? pf[kRows - 1].w >> 16 : pf[2 * kRows - 1].w;
                uint32_t const w6 = SS == 2 ? pf[kRows - 1].w & 0xffffu : pf[2 * kRows - 1].z;
                int f_last = 0;
                if (FM) {
                    int ci, cq, pi_, pq_;
                    if (SS == 2) {
                        ci = (int)(w7 & 0xffu) - 128, cq = (int)((w7 >> 8) & 0xffu) - 128;
                        pi_ = (int)(w6 & 0xffu) - 128, pq_ = (int)((w6 >> 8) & 0xffu) - 128;
                        f_last = atan2_q15(cq * pi_ - ci * pq_, ci * pi_ + cq * pq_);
                    }
                    else {
                        ci = (int)(int16_t)(w7 & 0xffffu), cq = (int)(int16_t)(w7 >> 16);
                        pi_ = (int)(int16_t)(w6 & 0xffffu), pq_ = (int)(int16_t)(w6 >> 16);
                        long long const dot = (long long)ci * pi_ + (long long)cq * pq_;
                        long long const crs = (long long)cq * pi_ - (long long)ci * pq_;
                        f_last = atan2_q31((int)crs, (int)dot);
                    }
                    f_last = __builtin_amdgcn_readlane(f_last, 63);
                }
                uint32_t const last_w = (uint32_t)__builtin_amdgcn_readlane((int)w7, 63);
                int const x_last = uni(env_of(last_w));
                if (uni((int)(carry_known && constant)) != 0) {
                    // Every step of the tile is the same map of the state (but the first, which still sees the sample before
                    // the tile): iterate to its fixed point -- at once where the silence has lasted, some hundred steps
                    // right after a burst.  An exact state stays exact.
                    int y = uni(carry_ya);
                    y = (mul24(kLpfA, y) + mul24(kLpfB, x_last + uni(carry_xa))) >> 14;
                    int const k2 = mul24(kLpfB, 2 * x_last);
                    for (int n = 1; n < kTile; ++n) {
                        int const y2 = (mul24(kLpfA, y) + k2) >> 14;
                        if (y2 == y)
                            break;
                        y = y2;
                    }
                    carry_ya = y;
                    if (FM && SS == 2) {
                        int yf = uni(carry_yf);
                        yf = (int)(int16_t)((mul24(p.a16, yf) + mul24(p.b16, f_last + uni(carry_ff))) >> 14);
                        int const kf = mul24(p.b16, 2 * f_last);
                        for (int n = 1; n < kTile; ++n) {
                            int const y2 = (int)(int16_t)((mul24(p.a16, yf) + kf) >> 14);
                            if (y2 == yf)
                                break;
                            yf = y2;
                        }
                        carry_yf = yf;
                    }
                    else if (FM) {
                        int yf = uni(carry_yf);
                        yf = (int)((p.a32 * (long long)yf + p.b32 * ((long long)f_last + uni(carry_ff))) >> 30);
                        long long const kf = p.b32 * (2ll * f_last);
                        for (int n = 1; n < kTile; ++n) {
                            int const y2 = (int)((p.a32 * (long long)yf + kf) >> 30);
                            if (y2 == yf)
                                break;
                            yf = y2;
                        }
                        carry_yf = yf;
                    }
                }
                else {
                    carry_known = false;
                }
                carry_xa = x_last;
                carry_ff = f_last;
                carry_w = SS == 2 ? last_w & 0xffffu : last_w;
                if (SS == 2) {
                    carry_i = (int)(last_w & 0xffu) - 128;
                    carry_q = (int)((last_w >> 8) & 0xffu) - 128;
                }
                else {
                    carry_i = (int)(int16_t)(last_w & 0xffffu);
                    carry_q = (int)(int16_t)(last_w >> 16);
                }
                tail_max = qmax;
                prev_quiet = true;
                prev_bound = qmax;
                if (lane == 0)
                    put_desc(kQuietTile | qmax);
                issue_loads(tile + 1, true);
                issue_head(tile + 2);
                return;
            }
        }

        // ================= phase A: envelope + discriminator, 8 samples per lane and row =================
        long long const t_tile = now();
        wave_sync(); // phase B of the tile before this one is done with s_env / s_f
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            uint32_t wd[8];
            if (SS == 2) {
                uint4 w = pf[r];
                wd[0] = w.x & 0xffffu, wd[1] = w.x >> 16, wd[2] = w.y & 0xffffu, wd[3] = w.y >> 16;
                wd[4] = w.z & 0xffffu, wd[5] = w.z >> 16, wd[6] = w.w & 0xffffu, wd[7] = w.w >> 16;
            }
            else {
                uint4 w0 = pf[2 * r], w1 = pf[2 * r + 1];
                wd[0] = w0.x, wd[1] = w0.y, wd[2] = w0.z, wd[3] = w0.w;
                wd[4] = w1.x, wd[5] = w1.y, wd[6] = w1.z, wd[7] = w1.w;
            }
            int const prev_w = __shfl_up((int)wd[7], 1, 64);
            int pi_, pq_;
            if (lane == 0) {
                pi_ = carry_i;
                pq_ = carry_q;
            }
            else if (SS == 2) {
                pi_ = (prev_w & 0xff) - 128;
                pq_ = ((prev_w >> 8) & 0xff) - 128;
            }
            else {
                pi_ = (int)(int16_t)(prev_w & 0xffff);
                pq_ = (int)(int16_t)((uint32_t)prev_w >> 16);
            }
            uint32_t ev[8];
            int fv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int ci, cq; // centred sample as the FM demodulator sees it
                if (SS == 2) {
                    uint32_t bi = wd[j] & 0xffu, bq = (wd[j] >> 8) & 0xffu;
                    ev[j] = p.use_mag ? env_mag_cu8(bi, bq) : env_amp_cu8(bi, bq);
                    if (SEAM && (p.flags & RUN_ENV_RAW16))
                        ev[j] = wd[j]; // the caller's envelope as it is
                    ci = (int)bi - 128;
                    cq = (int)bq - 128;
                }
                else {
                    ci = (int)(int16_t)(wd[j] & 0xffffu);
                    cq = (int)(int16_t)(wd[j] >> 16);
                    ev[j] = env_mag_cs16(ci, cq);
                }
                if (FM) {
                    if (SS == 2) {
                        int dot = ci * pi_ + cq * pq_;
                        int crs = cq * pi_ - ci * pq_;
                        fv[j] = atan2_q15(crs, dot);
                    }
                    else {
                        long long dot = (long long)ci * pi_ + (long long)cq * pq_;
                        long long crs = (long long)cq * pi_ - (long long)ci * pq_;
                        fv[j] = atan2_q31((int)crs, (int)dot);
                    }
                }
                else {
                    fv[j] = 0;
                }
                pi_ = ci;
                pq_ = cq;
            }
            // the row's last sample is the next row's (or tile's) predecessor
            int const last_w = __builtin_amdgcn_readlane((int)wd[7], 63);
            carry_w = (uint32_t)last_w;
            if (SS == 2) {
                carry_i = (last_w & 0xff) - 128;
                carry_q = ((last_w >> 8) & 0xff) - 128;
            }
            else {
                carry_i = (int)(int16_t)(last_w & 0xffff);
                carry_q = (int)(int16_t)((uint32_t)last_w >> 16);
            }
            int const chunk = r * (kRow / kChunk) + (lane >> 2);
            int const sub = lane & 3; // 8-sample group inside the chunk
            *(uint4 *)(s_env + chunk * kPitch16 + sub * 16) = make_uint4(ev[0] | (ev[1] << 16), ev[2] | (ev[3] << 16),
                    ev[4] | (ev[5] << 16), ev[6] | (ev[7] << 16));
            if (SS == 2 && raw_in)
                *(uint4 *)(s_raw + chunk * kPitch16 + sub * 16) = pf[SS == 2 ? r : 0];
            if (SS == 2) {
                *(uint4 *)(s_f + chunk * G::f_pitch + sub * 16) = make_uint4(((uint32_t)fv[0] & 0xffffu) | ((uint32_t)fv[1] << 16),
                        ((uint32_t)fv[2] & 0xffffu) | ((uint32_t)fv[3] << 16), ((uint32_t)fv[4] & 0xffffu) | ((uint32_t)fv[5] << 16),
                        ((uint32_t)fv[6] & 0xffffu) | ((uint32_t)fv[7] << 16));
            }
            else {
                *(uint4 *)(s_f + chunk * G::f_pitch + sub * 32) = make_uint4((uint32_t)fv[0], (uint32_t)fv[1], (uint32_t)fv[2], (uint32_t)fv[3]);
                *(uint4 *)(s_f + chunk * G::f_pitch + sub * 32 + 16) = make_uint4((uint32_t)fv[4], (uint32_t)fv[5], (uint32_t)fv[6], (uint32_t)fv[7]);
            }
            if (p.tap_env) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int idx = r * kRow + lane * 8 + j;
                    if (idx < n_t && seg_primary && !warm)
                        p.tap_env[(uint64_t)cap * p.tap_stride + t0 + (uint32_t)idx] = (uint16_t)ev[j];
                }
            }
        }
        issue_loads(tile + 1, lazy); // in flight while phase B runs
        if (lazy)
            issue_head(tile + 2);
        wave_sync();
        if (p.flags & RUN_DBG_SKIP_FILTERS)
            return;

        // ================= phase B: the two low-passes, lane = chunk of 32 samples =================
        int const cs = lane * kChunk;                       // chunk start inside the tile
        int const cnt = max(0, min(kChunk, n_t - cs));      // valid samples of my chunk
        int const first = max(0, lane - kWarmChunks);       // first chunk I read
        bool const from_carry = lane <= kWarmChunks && !warm && !est; // my warm-up reaches the tile start: exact carry
        // chunks at which a frame (= a push_sdr_flow call) starts
        unsigned long long const fs_mask = __ballot((t0 + (uint32_t)cs) % F == 0);
        Track16<FAST> ta, tf16;
        Track32<FAST> tf32;
        int xa1, ff1; // previous envelope / discriminator sample
        if (from_carry) {
            ta.lo = ta.hi = carry_ya;
            tf16.lo = tf16.hi = carry_yf;
            tf32.lo = tf32.hi = carry_yf;
            xa1 = carry_xa;
            ff1 = carry_ff;
        }
        else {
            ta.lo = -32768, ta.hi = 32767;
            tf16.lo = -32768, tf16.hi = 32767;
            tf32.lo = INT32_MIN, tf32.hi = INT32_MAX;
            xa1 = ff1 = 0; // lanes 0..2 of an establishing tile have no history at all: never proven, never used
            if (first > 0) {
                xa1 = (int)*(uint16_t const *)(s_env + (first - 1) * kPitch16 + (kChunk - 1) * 2);
                ff1 = SS == 2 ? (int)*(int16_t const *)(s_f + (first - 1) * G::f_pitch + (kChunk - 1) * 2)
                              : *(int const *)(s_f + (first - 1) * G::f_pitch + (kChunk - 1) * 4);
            }
            else if (est) { // after unfiltered tiles: the samples before the tile are known, the two filter states are not --
                xa1 = carry_xa; // but the envelope's lies between zero and the bound of the tile before
                ff1 = carry_ff;
                ta.lo = 0, ta.hi = prev_bound;
            }
        }
        ta.ok = tf16.ok = tf32.ok = 1;

        ChunkStatus sa, sf; // AM, FM
        sa.start_known = sf.start_known = false;
        sa.ident = sf.ident = 1;
        sa.lo0 = sa.hi0 = sf.lo0 = sf.hi0 = 0;
        int csum = 0; // envelope sum of my chunk (frame average)
        int cmax = -0x7fffffff, cmin = 0x7fffffff;
        int cap_ya = 0, cap_xa = 0, cap_yf = 0, cap_ff = 0; // SEAM: the state right after my chunk's last valid sample

        // one chunk of 32 steps for both filters; MAIN = my own chunk (publish, statistics)
        auto chunk_pass = [&](int c, auto main_tag) {
            constexpr bool MAIN = decltype(main_tag)::value;
            // a frame starts here: the AM filter state keeps x[-1] in an int16 slot (baseband.c:166-168)
            if ((fs_mask >> c) & 1ull)
                xa1 = (int)(int16_t)xa1;
            if (MAIN) {
                sa.start_known = ta.exact();
                sa.lo0 = ta.lo, sa.hi0 = ta.hi;
                if (SS == 2) {
                    sf.start_known = tf16.exact();
                    sf.lo0 = tf16.lo, sf.hi0 = tf16.hi;
                }
                else {
                    sf.start_known = tf32.exact();
                    sf.lo0 = tf32.lo, sf.hi0 = tf32.hi;
                }
            }
#pragma unroll 1
            for (int g = 0; g < kChunk / 8; ++g) {
                uint4 const e4 = *(uint4 const *)(s_env + c * kPitch16 + g * 16);
                uint32_t const ew[4] = {e4.x, e4.y, e4.z, e4.w};
                uint4 f4a = make_uint4(0, 0, 0, 0), f4b = make_uint4(0, 0, 0, 0);
                if (FM) {
                    if (SS == 2) {
                        f4a = *(uint4 const *)(s_f + c * G::f_pitch + g * 16);
                    }
                    else {
                        f4a = *(uint4 const *)(s_f + c * G::f_pitch + g * 32);
                        f4b = *(uint4 const *)(s_f + c * G::f_pitch + g * 32 + 16);
                    }
                }
                uint32_t const fw[8] = {f4a.x, f4a.y, f4a.z, f4a.w, f4b.x, f4b.y, f4b.z, f4b.w};
                uint32_t oa[4] = {0, 0, 0, 0}, of[4] = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    // samples past the end of the capture are stepped too (their results are never
                    // read); only the chunk statistics have to leave them out
                    int const lv = MAIN ? (int)(g * 8 + u < cnt) : 1;
                    int const x = (int)((ew[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
                    int const alo = ta.lo, ahi = ta.hi;
                    ta.step(kLpfA, mul24(kLpfB, x + xa1));
                    xa1 = x;
                    int fm_out;
                    int f_same = 1;
                    if (FM) {
                        if (SS == 2) {
                            int const f = (int)(int16_t)((fw[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
                            int const flo = tf16.lo, fhi = tf16.hi;
                            tf16.step(p.a16, mul24(p.b16, f + ff1));
                            ff1 = f;
                            fm_out = tf16.lo;
                            f_same = (int)(tf16.lo == flo) & (int)(tf16.hi == fhi);
                        }
                        else {
                            int const f = (int)fw[u];
                            int const flo = tf32.lo, fhi = tf32.hi;
                            tf32.step(p.a32, p.b32 * ((long long)f + ff1));
                            ff1 = f;
                            fm_out = (int)(int16_t)(tf32.lo >> 16);
                            f_same = (int)(tf32.lo == flo) & (int)(tf32.hi == fhi);
                        }
                    }
                    else {
                        fm_out = (int)(int16_t)x; // buf.fm aliases the raw envelope (include/r_private.h:32-36)
                    }
                    if (MAIN && SEAM) {
                        bool const last = g * 8 + u == cnt - 1;
                        cap_ya = last ? ta.lo : cap_ya;
                        cap_xa = last ? x : cap_xa;
                        cap_yf = last ? (SS == 2 ? tf16.lo : tf32.lo) : cap_yf;
                        cap_ff = last ? ff1 : cap_ff;
                    }
                    if (MAIN) {
                        sa.ident &= ((int)(ta.lo == alo) & (int)(ta.hi == ahi)) | (lv ^ 1);
                        sf.ident &= f_same | (lv ^ 1);
                        csum += lv ? x : 0;
                        cmax = lv ? max(cmax, ta.lo) : cmax;
                        cmin = lv ? min(cmin, ta.lo) : cmin;
                        oa[u >> 1] |= ((uint32_t)ta.lo & 0xffffu) << ((u & 1) * 16);
                        of[u >> 1] |= ((uint32_t)fm_out & 0xffffu) << ((u & 1) * 16);
                    }
                }
                if (MAIN) {
                    *(uint4 *)(p_am + lane * kPitchOut + g * 16) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
                    *(uint4 *)(p_fm + lane * kPitchOut + g * 16) = make_uint4(of[0], of[1], of[2], of[3]);
                }
            }
        };
#pragma unroll 1
        for (int q = 0; q < kWarmChunks; ++q) {
            int const c = lane - kWarmChunks + q; // chunk being read
            if (c >= 0)
                chunk_pass(c, std::false_type{});
        }
        chunk_pass(lane, std::true_type{});
        sa.end_known = ta.exact();
        sa.y_end = ta.lo;
        if (!FM) {
            sf.start_known = sf.end_known = true;
            sf.y_end = 0;
        }
        else if (SS == 2) {
            sf.end_known = tf16.exact();
            sf.y_end = tf16.lo;
        }
        else {
            sf.end_known = tf32.exact();
            sf.y_end = tf32.lo;
        }
        if (cnt == 0) { // past the end of the capture: nothing to prove
            sa.start_known = sf.start_known = true;
            sa.end_known = sf.end_known = false;
        }
        if (warm) {
            // An establishing tile is wanted for two things only: proven carries at its end and proven samples in its last
            // thirty-two chunks (the floor walk).  Its first lanes have no history: take what the tracks say, proven or not, and
            // never wait for it.  The same goes for any lane before those thirty-two chunks whose warm-up did not collapse (two
            // AM tracks one apart stay apart with probability 0.854 per step: 1-2 % of the lanes): its left neighbour may be
            // one of the unproven first lanes, then nobody could ever settle it and the whole cut would be given up.
            bool const early = lane < 64 - kFloorWindow / kChunk;
            if (lane < kWarmChunks || (early && !sa.start_known)) {
                sa.start_known = true;
                sa.ident = 0;
            }
            if (lane < kWarmChunks || (early && !sf.start_known)) {
                sf.start_known = true;
                sf.ident = 0;
            }
        }
        if (est && lane < kWarmChunks) { // no history to warm up on: what the tracks say, proven or not (the consumer takes
            sa.start_known = sf.start_known = true; // the first kVirtual samples by their bound); nobody waits for these lanes
            sa.ident = sf.ident = 0;
        }
        // the fixed-point argument needs a feedback coefficient in [0, 1]: monotone map, slope <= 1
        sa.ident &= ta.ok;
        sf.ident &= SS == 2 ? (tf16.ok & (int)(p.a16 >= 0 && p.a16 <= 16384)) : (tf32.ok & (int)(p.a32 >= 0 && p.a32 <= (1ll << 30)));

        bool const seam_main_a = sa.start_known, seam_main_f = sf.start_known; // SEAM: my captures came from a proven carry
        // ---- resolve the lanes whose warm-up did not collapse, left to right ----
        // which: 0 = AM, 1 = FM.  Wave-uniform control flow; every round settles everything up to and
        // including the first lane of each run that needs an exact re-run (lanes 0..3 always start from
        // the proven tile carry).
        for (int which = 0; which < (FM ? 2 : 1); ++which) {
            ChunkStatus &st = which == 0 ? sa : sf;
            for (int round = 0;; ++round) {
                unsigned long long const open = __ballot(!st.start_known);
                if (!open)
                    break;
                if (round > 64) { // cannot happen in a regular tile (the first open lane settles every round)
                    if (warm)
                        p_fail |= 8; // a stall reaching back past the establishing tile: the carry is not provable here
                    else if (lazy)
                        p_retry = 1; // (carries that do not settle after unfiltered tiles: the capture runs again, every tile filtered)
                    else
                        p_over = 2;
                    break;
                }
                // What does the carry look like when it leaves each lane?  CONST(y_end) where the end is
                // proven, PASS where the chunk maps every candidate to itself, BLOCK where only a re-run
                // can tell.  An inclusive scan of these transfers (PASS o x = x) carries proven values
                // across whole runs of stalled lanes in six shuffle steps.
                enum { T_PASS = 0, T_CONST = 1, T_BLOCK = 2 };
                int tk = st.end_known ? T_CONST : (st.ident ? T_PASS : T_BLOCK);
                int tv = st.y_end;
                for (int o = 1; o < 64; o <<= 1) {
                    int const qk = __shfl_up(tk, o, 64);
                    int const qv = __shfl_up(tv, o, 64);
                    if (lane >= o && tk == T_PASS) {
                        tk = qk;
                        tv = qv;
                    }
                }
                int const pk = __shfl_up(tk, 1, 64);
                int const pv = __shfl_up(tv, 1, 64);
                bool const take = !st.start_known && pk == T_CONST && lane > 0;
                bool rerun = false, bad = false;
                int y0 = 0;
                if (take) {
                    y0 = pv;
                    st.start_known = true;
                    if (st.ident && y0 >= st.lo0 && y0 <= st.hi0) {
                        // every step of my chunk leaves y0 where it is: outputs are constant
                        int const out = (which == 0 || SS == 2) ? y0 : (int)(int16_t)(y0 >> 16);
                        uint32_t const w2 = ((uint32_t)out & 0xffffu) * 0x10001u;
                        uint8_t *dst = (which == 0 ? p_am : p_fm) + lane * kPitchOut;
                        for (int g = 0; g < kChunk / 8; ++g)
                            *(uint4 *)(dst + g * 16) = make_uint4(w2, w2, w2, w2);
                        st.y_end = y0;
                        st.end_known = true;
                        if (which == 0)
                            cmax = cmin = y0;
                    }
                    else {
                        bad = st.ident != 0; // a proven interval that does not hold the carry
                        rerun = true;
                    }
                }
                if (__ballot(bad)) {
                    if (lazy)
                        p_retry = 2;
                    else
                        p_over = 3; // refuse the result (the host reports it)
                }
                if (__ballot(rerun)) {
                    CNT(15, 1); // resolve rounds with an exact re-run
                    // exact re-run of my chunk from the proven carry
                    int x1 = 0, f1 = 0;
                    if (rerun) {
                        if (which == 0) {
                            x1 = cs == 0 ? carry_xa : (int)*(uint16_t const *)(s_env + (lane - 1) * kPitch16 + (kChunk - 1) * 2);
                            if ((fs_mask >> lane) & 1ull)
                                x1 = (int)(int16_t)x1;
                            cmax = -0x7fffffff, cmin = 0x7fffffff;
                        }
                        else {
                            f1 = cs == 0 ? carry_ff
                                         : (SS == 2 ? (int)*(int16_t const *)(s_f + (lane - 1) * G::f_pitch + (kChunk - 1) * 2)
                                                    : *(int const *)(s_f + (lane - 1) * G::f_pitch + (kChunk - 1) * 4));
                        }
                    }
                    int y = y0;
#pragma unroll 1
                    for (int i = 0; i < kChunk; ++i) {
                        if (rerun && i < cnt) {
                            int out;
                            if (which == 0) {
                                int const x = (int)*(uint16_t const *)(s_env + lane * kPitch16 + i * 2);
                                y = (int)(int16_t)((mul24(kLpfA, y) + mul24(kLpfB, x + x1)) >> 14);
                                x1 = x;
                                out = y;
                                cmax = max(cmax, y);
                                cmin = min(cmin, y);
                            }
                            else if (SS == 2) {
                                int const f = (int)*(int16_t const *)(s_f + lane * G::f_pitch + i * 2);
                                y = (int)(int16_t)((mul24(p.a16, y) + mul24(p.b16, f + f1)) >> 14);
                                f1 = f;
                                out = y;
                            }
                            else {
                                int const f = *(int const *)(s_f + lane * G::f_pitch + i * 4);
                                y = (int)((p.a32 * (long long)y + p.b32 * ((long long)f + f1)) >> 30);
                                f1 = f;
                                out = (int)(int16_t)(y >> 16);
                            }
                            *(int16_t *)((which == 0 ? p_am : p_fm) + lane * kPitchOut + i * 2) = (int16_t)out;
                        }
                    }
                    if (rerun) {
                        st.y_end = y;
                        st.end_known = true;
                    }
                }
            }
        }

        if (raw_in) {
            // "The IQ buffer is really AM (FM) demodulated data" (src/r_flow.c:212-225): everything above ran on the file's
            // bytes taken as cu8 pairs -- the frame level and the squelch see that envelope, like the reference's do -- and
            // then the file's int16 words replace the filtered envelope (am.s16) or the filtered discriminator (fm.s16).
            bool const am_in = (p.flags & RUN_AM_IS_INPUT) != 0;
            uint8_t *const dst = (am_in ? p_am : p_fm) + lane * kPitchOut;
            if (am_in)
                cmax = -0x7fffffff, cmin = 0x7fffffff;
#pragma unroll 1
            for (int g = 0; g < kChunk / 8; ++g) {
                uint4 const w = *(uint4 const *)(s_raw + lane * kPitch16 + g * 16);
                *(uint4 *)(dst + g * 16) = w;
                uint32_t const ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int const v = (int)(int16_t)((ww[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
                    bool const in = am_in && g * 8 + u < cnt;
                    cmax = in ? max(cmax, v) : cmax;
                    cmin = in ? min(cmin, v) : cmin;
                }
            }
        }
        // carries for the next tile (only meaningful when this tile is full)
        carry_ya = __builtin_amdgcn_readlane(sa.y_end, 63);
        carry_yf = __builtin_amdgcn_readlane(sf.y_end, 63);
        carry_xa = __builtin_amdgcn_readlane(xa1, 63);
        carry_ff = __builtin_amdgcn_readlane(ff1, 63);
        if (lazy) {
            carry_known = true;
            prev_quiet = false;
            if (post_q && lane < kVirtual / kChunk) // the consumer never looks at these samples
                cmax = vbound, cmin = 0;
            if (lane == 0) {
                put_desc(post_q ? (kPostQuiet | vbound) : 0);
                if (p_retry && !to_hbm)
                    st_retry(retry_slot) = p_retry;
            }
            if (to_hbm)
                self_retry |= p_retry;
        }
        else if (to_hbm && lane == 0) {
            put_desc(0);
        }
        if (to_hbm) {
            short *const ext = (short *)(g_tiles + (uint64_t)tile * kTileRecBytes);
            p_active += cmax > quiet_bound(p.det.min_high) ? 1 : 0;
            ext[kTileRecMax / 2 + lane] = (short)max(cmax, -32768);
            ext[kTileRecMin / 2 + lane] = (short)min(cmin, 32767);
            if (p_over && lane == 0)
                p.tile_over[s] = p_over;
        }
        else {
            st_cmax(buf, lane) = (short)max(cmax, -32768); // (an empty chunk keeps its sentinels, clamped to 16 bits)
            st_cmin(buf, lane) = (short)min(cmin, 32767);
            if (p_over && lane == 0)
                s_pover = p_over;
        }

        if (warm) {
            // ---- establishing tile of a later segment: no detection here.  The carries just taken must be
            // proven, and so must the last thirty-two chunks (the floor is walked over their samples).
            bool const tail_ok = lane < 64 - kFloorWindow / kChunk || (sa.start_known && sa.end_known && sf.start_known && sf.end_known);
            if (__ballot(!tail_ok))
                p_fail |= 16;
            if (lane == 0)
                st_pflag(buf) = p_fail;
            return; // the consumer walks the floor over this tile
        }
        // per-frame envelope sums (u32, wraps like the reference's accumulator, baseband.c:39-44)
        if (p.frame_sums && seg_primary && tile >= sums_done) {
            uint32_t const f_first = t0 / F, f_last = (t0 + (uint32_t)n_t - 1) / F;
            uint32_t const my_frame = (t0 + (uint32_t)cs) / F;
            for (uint32_t f = f_first; f <= f_last; ++f) {
                int const part = wave_sum(cnt > 0 && my_frame == f ? csum : 0);
                if (lane == 0 && f < p.frames_cap) // several segments of a capture may share a frame
                    atomicAdd(&p.frame_sums[(uint64_t)cap * p.frames_cap + f], (uint32_t)part);
            }
        }
        sums_done = max(sums_done, tile + 1u);
        wave_sync();

        if (p.tap_am && seg_primary) {
            for (int idx = lane; idx < n_t; idx += 64) {
                uint64_t o = (uint64_t)cap * p.tap_stride + t0 + (uint32_t)idx;
                p.tap_am[o] = (int16_t)ld16(p_am, idx);
                p.tap_fm[o] = (int16_t)ld16(p_fm, idx);
            }
        }

        if (SEAM) { // filters only: remember the carries after the frame's last sample (this tile may hold it)
            int const L = (n_t - 1) >> 5;
            int const ea = seam_main_a ? cap_ya : sa.y_end, ef = seam_main_f ? cap_yf : sf.y_end;
            seam_end[0] = __builtin_amdgcn_readlane(ea, L);
            seam_end[1] = __builtin_amdgcn_readlane(cap_xa, L);
            seam_end[2] = __builtin_amdgcn_readlane(ef, L);
            seam_end[3] = __builtin_amdgcn_readlane(cap_ff, L);
            return;
        }
        tick(0, t_tile);
    };


    // The consumer's side of an unfiltered stretch: n samples from the tile's first on, none of which, filtered, exceeds qmax.
    // The detector idles (the steps of its noise floor are counted, see resolve_low) or counts a gap, whose end-of-package
    // count may fall here (src/pulse_detect.c:441-468: no sample value enters but through `am > threshold`, which is false).
    auto quiet_span = [&](uint32_t t0, int n, int qmax) {
        int q = 0;
        while (q < n) {
            if (dc == 0) { // a new frame == a new push_sdr_flow call
                flen = (int)min(my_n - (t0 + (uint32_t)q), F);
                if (p.frame_min_high)
                    cfg.min_high = uni(p.frame_min_high[(uint64_t)cap * p.frames_cap + min(frame, p.frames_cap - 1)]);
                det_call_entry(det, cfg, flen, 0);
            }
            int const lim = min(n, q + (flen - dc));
            int const st = uni(det.state), low = uni(det.low);
            if (st == ST_IDLE) {
                int const l_lo = min(low, min(lz_min, 0)) - 1;
                int const l_hi = max(low, max(lz_max, qmax)) + 1;
                int thr = (int)(int16_t)((l_lo + min(cfg.min_high, cfg.max_high)) / 2);
                if (cfg.fixed_high != 0)
                    thr = (int)(int16_t)cfg.fixed_high;
                int const hys = (int)(int16_t)(thr / 8);
                if (l_hi - l_lo >= 1000 || qmax > thr + hys) { // steps of more than one, or (never: quiet_bound) a pulse could start
                    lz_fail = l_hi - l_lo >= 1000 ? 4 : 8;
                    return;
                }
                int const cnt = lim - q;
                lz_n += cnt;
                lz_min = min(lz_min, 0);
                lz_max = max(lz_max, qmax);
                lzc_min = min(lzc_min, 0);
                lzc_max = max(lzc_max, qmax);
                lz_carried = true;
                det.lead_in = min(1025, uni(det.lead_in) + cnt);
                q += cnt;
                dc += cnt;
            }
            else if (st == ST_GAP) {
                int thr = (int)(int16_t)((low + min(uni(det.high), cfg.max_high)) / 2);
                if (cfg.fixed_high != 0)
                    thr = (int)(int16_t)cfg.fixed_high;
                int const hys = (int)(int16_t)(thr / 8);
                if (qmax > thr + hys) {
                    lz_fail = 8;
                    return;
                }
                int const lim_eop = 10 * min(max(uni(det.max_pulse), cfg.per_ms), 10 * cfg.per_ms);
                int const togo = uni(det.eop_spurious) ? 0 : max(0, lim_eop - uni(det.run));
                int const cnt = min(togo, lim - q);
                det.run = uni(det.run) + cnt;
                q += cnt;
                dc += cnt;
                if (q < lim) { // the count ends the package at this sample; the idle state then looks at the sample again
                    int const r = det_step(det, cfg, 0, 0, flen, dc, input_pos, frame);
                    if (r) {
                        det_call_entry(det, cfg, flen, dc);
                    }
                    else {
                        q += 1;
                        dc += 1;
                    }
                }
            }
            else { // (never: a pulse cannot be open 32 samples into such a stretch)
                lz_fail = 16;
                return;
            }
            if (dc == flen) { // the end of a frame (no logic dump with lazy tiles)
                input_pos += (uint64_t)flen;
                frame += 1;
                dc = 0;
            }
        }
    };

    // ======================================= the consumer: phase C of one tile =======================================
    auto consume = [&](uint32_t tile, int buf) {
        uint32_t const t0 = tile * kTile;
        int const n_t = (int)min((uint32_t)kTile, seg_end - t0);
        bool const warm = !seg_first && tile == tile_first;
        uint8_t const *const c_am = s_am + buf * (64 * kPitchOut), *const c_fm = s_fm + buf * (64 * kPitchOut);
        if (p.flags & (RUN_DBG_SKIP_FILTERS | RUN_DBG_SKIP_DETECT)) // (profiling: the producers alone -- tools/pmc_issue.py splits the
            return;                                                   // instruction counts by role this way)
        if (warm) {
            if (st_pflag(buf)) // (bits: 1 the filters -- 8 a carry that cannot be proven inside the tile, 16 an unproven chunk among
                seg_fail |= 1 | st_pflag(buf); // those the floor is walked over --, 2 floor range too wide, 4 the walks did not meet)
            // Noise floor at the segment's first sample: the detector is assumed idle over the last 1024
            // samples with a floor of the assumed parity somewhere inside the tile's sample range; both
            // extremes of that parity are walked and must meet (see the lazy floor below).
            int rmax = lane >= 4 ? st_cmax(buf, lane) : -0x7fffffff, rmin = lane >= 4 ? st_cmin(buf, lane) : 0x7fffffff;
            for (int o = 32; o > 0; o >>= 1) {
                rmax = max(rmax, __shfl_xor(rmax, o, 64));
                rmin = min(rmin, __shfl_xor(rmin, o, 64));
            }
            int const par = (seg_flags & SEG_ODD) ? 1 : 0;
            int a = rmin - 1, b = rmax + 1;
            if (b - a >= 1000)
                seg_fail |= 2; // steps of more than one are possible: not the regime the cut assumes
            a += (a ^ par) & 1;
            b -= (b ^ par) & 1;
            for (int w = kTile - kFloorWindow; w < kTile; w += 64) { // chunks 32..63: proven above
                int const v = ld16(c_am, w + lane);
#pragma unroll 8
                for (int u = 0; u < 64; ++u) {
                    int const x = __builtin_amdgcn_readlane(v, u);
                    a += x > a ? 1 : -1;
                    b += x > b ? 1 : -1;
                }
            }
            if (a != b)
                seg_fail |= 4;
            det.low = a;
            // (the high estimate a sample meets was made on the sample before it, pulse_detect.c:265-268,300-304: under -Y autolevel
            // a piece that starts on a frame boundary inherits the level of the frame before)
            int const min_high_before = p.frame_min_high ? p.frame_min_high[(uint64_t)cap * p.frames_cap + min((seg_start - 1u) / F, p.frames_cap - 1)]
                                                         : cfg.min_high;
            det.high = max(cfg.ratio * a, min_high_before);
            seg_init_low = a;
            seg_init_high = det.high;
            return;
        }

        int vq = 0; // samples at the start of the tile that come as a bound instead of samples
        if (lazy) {
            int const desc = uni(st_desc(buf));
            if (desc & (kQuietTile | kPostQuiet)) {
                bool const whole = (desc & kQuietTile) != 0;
                vq = whole ? n_t : min(kVirtual, n_t);
                quiet_span(t0, vq, desc & 0xffff);
                if (whole)
                    n_quiet += 1;
                if (lz_fail || whole) {
                    if (lz_fail && lane == 0)
                        st_retry(retry_slot) = lz_fail;
                    return;
                }
            }
        }

        // ================= phase C: pulse detector =================
        int i = (p.flags & RUN_DBG_SKIP_DETECT) ? n_t : vq;
        // chunk statistics (lane = chunk) and their suffix extrema, for jumping over whole chunks
        // (Kept in LDS, not in registers: four values per lane that live as long as the tile does were what pushed the
        // consumer over its 168 registers -- 900 MB of scratch written per grid of 8192 captures.  The chunk extrema are
        // where the producer left them; their suffix extrema go into the last free bytes of the padding, written here.)
        auto my_cmax = [&]() -> int { return (int)st_cmax(buf, lane); };
        auto my_cmin = [&]() -> int { return (int)st_cmin(buf, lane); };
        auto st_sfx_max = [&](int chunk) -> short & { return *(short *)(s_env + chunk * kPitch16 + 2 * kChunk + 12); };
        auto st_sfx_min = [&](int chunk) -> short & { return *(short *)(s_env + chunk * kPitch16 + 2 * kChunk + 14); };
        int sfx_max = my_cmax(), sfx_min = my_cmin();
        for (int o = 1; o < 64; o <<= 1) {
            int const qa = __shfl_down(sfx_max, o, 64), qb = __shfl_down(sfx_min, o, 64);
            if (lane + o < 64) {
                sfx_max = max(sfx_max, qa);
                sfx_min = min(sfx_min, qb);
            }
        }
        st_sfx_max(lane) = (short)sfx_max;
        st_sfx_min(lane) = (short)sfx_min;
        wave_sync();
        auto sfx_max_at = [&](int chunk) -> int { return uni((int)st_sfx_max(chunk)); };
        auto sfx_min_at = [&](int chunk) -> int { return uni((int)st_sfx_min(chunk)); };
        // Lazy noise floor.  While the detector idles over samples that cannot start a pulse, every
        // step moves `low` by exactly +-1 towards the sample (pulse_detect.c:326-329 with |am-low| < 1024),
        // so (a) its parity after n steps is known without walking, (b) it never leaves
        // [min(low0, samples) - 1, max(low0, samples) + 1], and (c) two walks of the same parity keep
        // their order and close in by two whenever a sample falls between them.  The steps are therefore
        // only counted; when the exact value is needed (a pulse may start, the tile ends) the last 128
        // samples are walked from both extremes of the right parity and must meet -- else the whole
        // stretch is walked.  lz_n pending steps start at lz_from; lz_min/lz_max bound their samples.
        // (lz_n, lz_from, lz_min, lz_max live across tiles: over unfiltered tiles the steps stay pending -- lz_carried --,
        // and only a two-sided walk over filtered samples of a later tile can settle them)
        auto resolve_low = [&](int upto) {
            if (lz_n == 0)
                return;
            int lo_est = det.low;
            bool done = false;
            if (lz_carried) {
                // The walk from the first pending step is not to be had.  Both extremes of the right parity over the filtered
                // samples before `upto` -- the last 128, then up to 384 -- must meet; else the capture runs again.
                int const avail = uni(upto - vq);
                bool met = false;
                for (int W = 128; avail > 0; W = 384) {
                    int const w0 = upto - min(W, avail);
                    int const par = (det.low + (lz_n - (upto - w0))) & 1;
                    // where the floor can be at w0: between the extremes of what it has seen up to there -- the bound of the
                    // unfiltered samples and the chunks of this tile that begin before w0 (lz_min / lz_max also cover chunks
                    // behind the window, the signal that made the tile worth filtering among them)
                    bool const mine = lane >= (vq >> 5) && lane < ((w0 + kChunk - 1) >> 5);
                    int const seen_hi = wave_max(mine ? my_cmax() : -0x7fffffff), seen_lo = -wave_max(mine ? -my_cmin() : -0x7fffffff);
                    int a = min(det.low, min(lzc_min, seen_lo)) - 1, b = max(det.low, max(lzc_max, seen_hi)) + 1;
                    a += (a ^ par) & 1;
                    b -= (b ^ par) & 1;
                    for (int j0 = uni(w0); j0 < upto; j0 += 64) {
                        int const v = j0 + lane < upto ? ld16(c_am, j0 + lane) : 0;
                        int const cntj = uni(min(64, upto - j0));
                        if (cntj == 64) {
#pragma unroll 8
                            for (int u = 0; u < 64; ++u) {
                                int const x = __builtin_amdgcn_readlane(v, u);
                                a += x > a ? 1 : -1;
                                b += x > b ? 1 : -1;
                            }
                        }
                        else {
                            for (int u = 0; u < cntj; ++u) {
                                int const x = __builtin_amdgcn_readlane(v, u);
                                a += x > a ? 1 : -1;
                                b += x > b ? 1 : -1;
                            }
                        }
                    }
#ifdef R433_EMU_DEBUG_LAZY
                    if (lane == 0) fprintf(stderr, "resolve carried: t0 %u upto %d vq %d W %d w0 %d lz_n %d low %d lz_min %d lz_max %d -> a %d b %d\n", t0, upto, vq, W, w0, lz_n, det.low, lz_min, lz_max, a, b);
#endif
                    if (uni(a) == uni(b)) {
                        lo_est = a;
                        met = true;
                    }
                    if (met || W == 384 || avail <= 128)
                        break;
                }
                if (!met)
                    lz_fail = avail > 0 ? 32 : 64;
                done = true;
            }
            else if (lz_n > 160) {
                int const w0 = upto - 128;
                int const par = (det.low + (w0 - lz_from)) & 1;
                int a = min(det.low, lz_min) - 1, b = max(det.low, lz_max) + 1;
                a += (a ^ par) & 1; // lowest / highest candidate of that parity
                b -= (b ^ par) & 1;
                if (uni(b - a) <= 256) { // the two close in by at most 2 per sample: further apart they cannot meet in 128
                    int const v0 = ld16(c_am, w0 + lane), v1 = ld16(c_am, w0 + 64 + lane); // w0 + 127 < upto <= n_t
#pragma unroll 8
                    for (int u = 0; u < 64; ++u) {
                        int const x = __builtin_amdgcn_readlane(v0, u);
                        a += x > a ? 1 : -1;
                        b += x > b ? 1 : -1;
                    }
#pragma unroll 8
                    for (int u = 0; u < 64; ++u) {
                        int const x = __builtin_amdgcn_readlane(v1, u);
                        a += x > a ? 1 : -1;
                        b += x > b ? 1 : -1;
                    }
                    if (a == b) {
                        lo_est = a;
                        done = true;
                    }
                }
            }
            if (!done) {
                lo_est = uni(lo_est);
                for (int j0 = uni(lz_from); j0 < upto; j0 += 64) {
                    int const v = j0 + lane < upto ? ld16(c_am, j0 + lane) : 0;
                    int const cntj = uni(min(64, upto - j0));
                    if (cntj == 64) { // whole blocks: constant lane numbers, no loop control (a noisy floor walks every sample)
#pragma unroll
                        for (int u = 0; u < 64; ++u)
                            lo_est += __builtin_amdgcn_readlane(v, u) > lo_est ? 1 : -1;
                    }
                    else {
                        for (int u = 0; u < cntj; ++u)
                            lo_est += __builtin_amdgcn_readlane(v, u) > lo_est ? 1 : -1;
                    }
                }
            }
            det.low = lo_est = uni(lo_est);
            det.high = max(cfg.ratio * lo_est, cfg.min_high);
            lz_n = 0;
            lz_min = 0x7fffffff;
            lz_max = -0x7fffffff;
            lz_carried = false;
            lzc_min = 0x7fffffff, lzc_max = -0x7fffffff;
        };
        // Every lane holds the same detector state, but the general step and call entry are per-lane-looking code;
        // after them, pin what the fast paths loop on to SGPRs so that those loops run on the scalar unit with
        // scalar branches (the fast paths themselves keep these values scalar).
        auto pin_state = [&]() {
            i = uni(i);
            dc = uni(dc);
            flen = uni(flen);
            det.state = uni(det.state);
            det.lead_in = uni(det.lead_in);
            det.low = uni(det.low);
            det.high = uni(det.high);
            det.run = uni(det.run);
            det.max_pulse = uni(det.max_pulse);
            det.ook_num = (uint32_t)uni((int)det.ook_num);
            det.eop_spurious = uni(det.eop_spurious);
            det.cur_pulse = uni(det.cur_pulse);
            det.ook_f1 = uni(det.ook_f1);
            det.fsk_num = (uint32_t)uni((int)det.fsk_num);
        };
        pin_state();
        // the end of a frame (= of a push_sdr_flow call): "Dump partial pulse data, might overlap with the last complete
        // package" (src/r_flow.c:364-371) -- whatever the two structs hold right now, OOK first, FSK over it
        auto frame_done = [&]() {
            if (logic) {
                __threadfence();
                paint(det.offset, det.ook_num, (int2 const *)(det.arena + det.ook_base + sizeof(r433_pkg_rec)), 0x02, input_pos, flen);
                paint(det.fsk_offset, det.fsk_num, det.fsk_ring, 0x04, input_pos, flen);
            }
            input_pos += (uint64_t)flen;
            frame += 1;
            dc = 0;
        };
        while (i < n_t && !lz_fail) {
            CNT(0, 1); // outer iterations
            long long const t_it = now();
            int const st_it = det.state;
            if (timing && lane == 0)
                s_tk[7] += 1;
            if (dc == 0) { // a new frame == a new push_sdr_flow call
                flen = (int)min(my_n - (t0 + (uint32_t)i), F);
                if (p.frame_min_high) // pulse_detect_set_levels before this frame's detection, r_flow.c:180-186
                    cfg.min_high = uni(p.frame_min_high[(uint64_t)cap * p.frames_cap + min(frame, p.frames_cap - 1)]);
                det_call_entry(det, cfg, flen, 0);
                pin_state();
            }
            // ---- whole chunks at a time: while idle or inside a gap, nothing can happen before the first
            // chunk whose maximum reaches the (conservative) threshold, the end-of-package count, or the
            // end of the frame ----
            if ((det.state == ST_IDLE && det.lead_in > 1024) || det.state == ST_GAP) {
                int const ci = i >> 5;
                int const lim_i = min(n_t, i + (flen - dc));
                int jump_to = i;
                if (det.state == ST_IDLE) {
                    int const rmin = sfx_min_at(ci), rmax = sfx_max_at(ci);
                    int const l_lo = min(det.low, min(lz_min, rmin)) - 1;
                    int const l_hi = max(det.low, max(lz_max, rmax)) + 1;
                    if (l_hi - l_lo < 1000) {
                        int thr = (int)(int16_t)((l_lo + min(cfg.min_high, cfg.max_high)) / 2);
                        if (cfg.fixed_high != 0)
                            thr = (int)(int16_t)cfg.fixed_high;
                        int const hys = (int)(int16_t)(thr / 8);
                        unsigned long long const m = __ballot(lane >= ci && my_cmax() > thr + hys);
                        jump_to = min(m ? (__ffsll(m) - 1) * kChunk : n_t, lim_i);
                        if (jump_to > i) {
                            if (lz_n == 0)
                                lz_from = i;
                            lz_n += jump_to - i;
                            lz_min = min(lz_min, rmin);
                            lz_max = max(lz_max, rmax);
                        }
                    }
                }
                else {
                    int thr = (int)(int16_t)((det.low + min(det.high, cfg.max_high)) / 2);
                    if (cfg.fixed_high != 0)
                        thr = (int)(int16_t)cfg.fixed_high;
                    int const hys = (int)(int16_t)(thr / 8);
                    unsigned long long const m = __ballot(lane >= ci && my_cmax() > thr + hys);
                    // min(max(10 max_pulse, 10 per_ms), 100 per_ms) without leaving 32 bits (100 per_ms < 2^29)
                    int const lim = 10 * min(max(det.max_pulse, cfg.per_ms), 10 * cfg.per_ms);
                    int const togo = det.eop_spurious ? 0 : max(0, lim - det.run);
                    int const je = togo < lim_i - i ? i + togo : lim_i;
                    jump_to = min(min(m ? (__ffsll(m) - 1) * kChunk : n_t, je), lim_i);
                    if (jump_to > i)
                        det.run += jump_to - i;
                }
                if (jump_to > i) {
                    int const done = jump_to - i;
                    i += done;
                    dc += done;
                    if (dc == flen)
                        frame_done();
                    tick(st_it == ST_IDLE ? 1 : 2, t_it);
                    continue;
                }
            }
            int const lim = min(n_t, i + (flen - dc)); // end of the tile or of the frame, whichever comes first
            int base = i & ~63;
            int e = min(base + 64, lim);
            // (the block's values live one trip of this loop, not across trips: seven registers per lane that the compiler
            // otherwise keeps -- and spills -- around the loop's head for the rare trip that meets the same block again)
            int loaded = -1;           // block whose samples the lanes hold
            int am_l = 0, fm_l = 0;    // my sample of that block
            int a64_l = 0, f64_l = 0;  // am / 64, fm / 64 (C division) for the level and carrier averages
            int in_pk_l = 0;           // the two of them packed as 16-bit halves
            int in_pkn_l = 0;          // the same with the carrier half negated (|f1| form of the average)
            int ff_pk_l = 0;           // fm / 64 in both halves (the package's and the FSK detector's carrier averages)
            int bmax = 0, bmin = 0;
            auto load_block = [&]() { // the 64 samples at `base`, one per lane, and what the fast paths want of them
                if (loaded == base)
                    return;
                CNT(1, 1); // block loads
                int const il = base + lane;
                am_l = il < n_t ? ld16(c_am, il) : 0;
                fm_l = il < n_t ? ld16(c_fm, il) : 0;
                a64_l = div64(am_l);
                f64_l = div64(fm_l);
                in_pk_l = (a64_l & 0xffff) | (f64_l << 16);
                in_pkn_l = (a64_l & 0xffff) | (-f64_l << 16);
                ff_pk_l = (f64_l & 0xffff) | (f64_l << 16);
                bmax = uni(max((int)st_cmax(buf, base >> 5), (int)st_cmax(buf, (base >> 5) + 1)));
                bmin = uni(min((int)st_cmin(buf, base >> 5), (int)st_cmin(buf, (base >> 5) + 1)));
                loaded = base;
            };
            load_block();
            // Legs inside this block.  A fast path covers [i0, k) and, where it can, also what happens at k (a
            // regular pulse end, the end of the debounce, the next pulse's start): then the next leg starts
            // right there, without going round the outer loop.  Otherwise the general step takes over at k.
            int k = i;
            bool settled = false;
            // ---- Inside a regular package: the train engine.  Between the second pulse of a package and its end the
            // state machine does three things: wait for a falling edge while the level and carrier averages advance
            // (pulse), count to ten (debounce), wait for a rising edge or the end-of-package count (gap).  With the
            // FSK candidate out of the picture and nothing spurious pending, the legs of a block chain on a handful
            // of scalars: the rising-edge mask of a block is one ballot per frozen level, the falling-edge candidates
            // one ballot per pulse leg against the highest level the tile can produce, block changes reload two
            // values per lane.  Anything irregular (first pulse, spurious pulse, package end, the 1200-pulse cap,
            // levels outside int16) leaves the engine for the legs and the exact general step below.
            bool const regular = det.state != ST_IDLE && det.ook_num >= 1 && det.fsk_num <= 16 && !det.eop_spurious
                    && det.ook_num + 70 < R433_PD_MAX_PULSES && det.high >= 0 && det.high <= 32767 && cfg.min_high >= 0
                    && cfg.min_high <= 32767 && det.low >= -1 && det.low <= 32767 && cfg.min_high >= 1 && cfg.fixed_high >= 0
                    && cfg.fixed_high <= 32767; // (the floor sits at -1 half of the time over digital silence)
            bool engine_ran = false;
            if (uni((int)regular) && !(p.flags & RUN_NO_TRAIN_ENGINE)) {
                engine_ran = true;
                // every loop condition below is a scalar: one value the compiler takes for per-lane (the detector's
                // fields are, after the general step) would turn the whole loop into exec-masked vector code
                int st = uni(det.state), run = uni(det.run), cur = uni(det.cur_pulse), mx = uni(det.max_pulse);
                int n_pairs = uni((int)det.ook_num);
                int const lim_u = uni(lim), n_tu = uni(n_t);
                k = uni(k);
                base = uni(base);
                e = uni(e);
                int const low = uni(det.low);
                int const fl6 = uni(cfg.min_high >> 6);
                int const amax_ub = sfx_max_at(0); // no filtered sample of this tile is larger
                v2s hv = {(short)det.high, (short)det.ook_f1};
                v2s const floor_v = {(short)cfg.min_high, (short)-32768};
                v2s const m63 = {63, 63};
                // thresholds (pulse_detect.c:300-304): the floor is >= -1 and every level >= min_high >= 1 here, so the sums are
                // in [0, 49150]: the C divisions are shifts and the int16 narrowing is the identity
                int const thr_fixed = uni(cfg.fixed_high), max_high = uni(cfg.max_high);
                auto thr_of = [&](int level) -> int { return thr_fixed != 0 ? thr_fixed : (low + min(level, max_high)) >> 1; };
                int eop_lim = 10 * min(max(mx, cfg.per_ms), 10 * cfg.per_ms); // pulse_detect.c:446-450
                int thi = 0;                   // rising-edge level of the frozen averages (debounce and gap)
                unsigned long long m_hi = 0;   // lanes above it in this block
                bool m_hi_ok = false;
                unsigned long long vmask = e - base >= 64 ? ~0ull : ((1ull << (e - base)) - 1ull);
                unsigned long long okp = __ballot(a64_l >= fl6 && f64_l >= 0);
                unsigned long long okn = __ballot(a64_l >= fl6 && f64_l <= 0 && f64_l > -512);
                bool need_general = false;
                // The engine's window: 64 samples from k on, wherever k is (the legs below keep the blocks aligned; here a
                // window that begins with a pulse lets the averages run from lane 0 without a rotation).
                auto load_window = [&]() {
                    long long const t_w = now();
                    CNT(14, 1); // engine window loads
                    base = k;
                    e = min(base + 64, lim_u);
                    int const il = base + lane;
                    am_l = il < n_tu ? ld16(c_am, il) : 0;
                    fm_l = il < n_tu ? ld16(c_fm, il) : 0;
                    a64_l = div64(am_l);
                    f64_l = div64(fm_l);
                    in_pk_l = (a64_l & 0xffff) | (f64_l << 16);
                    in_pkn_l = (a64_l & 0xffff) | (-f64_l << 16);
                    loaded = -1; // not a block as the legs below know them
                    vmask = e - base >= 64 ? ~0ull : ((1ull << (e - base)) - 1ull);
                    okp = __ballot(a64_l >= fl6 && f64_l >= 0);
                    okn = __ballot(a64_l >= fl6 && f64_l <= 0 && f64_l > -512);
                    m_hi_ok = false;
                    tick(8, t_w);
                };
                // Windows outside, legs inside: the window registers change at one place only (a conditional reload in the
                // middle of the leg loop costs a register copy per value and leg).
                bool leave = false;
                for (;;) {
                load_window();
                for (;;) {
                    CNT(13, 1); // engine legs
                    if (timing && lane == 0)
                        s_tk[15] += 1;
                    long long const t_leg = now();
                    // a pulse that begins in the last third of a window gets a window of its own
                    if (st == ST_PULSE && k - base > 40 && k < e && e < lim_u)
                        break;
                    if (st == ST_PULSE) {
                        CNT(23, 1); // engine: pulse legs
                        int const h0 = uni((int)hv[0]);
                        int const thr_ub = thr_of(max(h0, amax_ub) + 1);
                        int const tlo_ub = thr_ub - (thr_ub >> 3);
                        unsigned long long cand = __ballot(am_l < tlo_ub) & vmask & (~0ull << (k - base));
                        int const i0 = k;
                        int j = k;
                        bool fall = false;
                        m_hi_ok = false;
                        tick(9, t_leg);
                        for (;;) {
                            k = cand ? base + (__ffsll(cand) - 1) : e;
                            int const kk = k;
                            long long const t_ema = now();
                            while (j < kk) { // the averages over [j, kk): see the pulse leg below for the three forms
                                CNT(16, 1); // engine: runs of the averages
                                int const f1s = uni((int)hv[1]);
                                bool const neg = f1s < 0;
                                unsigned long long const bad = ~((neg ? okn : okp) >> (j - base));
                                int const len = uni(f1s == -32768 ? 0 : min(kk - j, bad ? (int)__builtin_ctzll(bad) : 64));
                                if (len >= 8) {
                                    v2s const sv = {1, (short)(neg ? -1 : 1)};
                                    int const sel = neg ? in_pkn_l : in_pk_l;
                                    // windows begin where pulses begin: most runs start at lane 0 and need no rotation
                                    int const rot = j == base ? sel : __builtin_amdgcn_ds_bpermute(((int)lane + (j - base)) << 2, sel);
                                    int const nb = len >> 3;
                                    v2s x = hv * sv;
                                    ema_groups(x, rot, nb);
                                    CNT(17, nb * 8);         // engine: samples in groups of eight
                                    CNT(18, len - nb * 8);   // engine: the rest of such runs
                                    CNT(22, j != base);      // engine: runs that needed a rotation
                                    j += nb * 8;
                                    int const rem = len - nb * 8; // the rest of the run right away: same form, lane numbers computed
                                    for (int u = 0; u < rem; ++u) {
                                        v2s const in = as_v2s(__builtin_amdgcn_readlane(sel, j - base + u));
                                        x = x + (in - (x >> 6));
                                    }
                                    j += rem;
                                    hv = x * sv;
                                    continue;
                                }
                                if (len > 0) {
                                    v2s const sv = {1, (short)(neg ? -1 : 1)};
                                    int const in_sel = neg ? in_pkn_l : in_pk_l;
                                    v2s x = hv * sv;
                                    for (int u = 0; u < len; ++u) {
                                        v2s const in = as_v2s(__builtin_amdgcn_readlane(in_sel, j - base + u));
                                        x = x + (in - (x >> 6));
                                    }
                                    hv = x * sv;
                                    CNT(19, len); // engine: samples of short runs
                                    j += len;
                                    continue;
                                }
                                int const cnt = uni(min(min(8, kk - j), f1s == -32768 ? 8 : (int)__builtin_ctzll(~bad | (1ull << 63))));
                                for (int u = 0; u < cnt; ++u) {
                                    v2s const in = as_v2s(__builtin_amdgcn_readlane(in_pk_l, j - base + u));
                                    v2s const q = (hv + ((hv >> 15) & m63)) >> 6; // hv / 64, truncating toward zero
                                    hv = pk_max(hv - q + in, floor_v);
                                }
                                CNT(20, cnt); // engine: samples through the general form
                                j += cnt;
                            }
                            tick(10, t_ema);
                            if (k >= e)
                                break;
                            long long const t_cand = now();
                            // candidate: decide with the exact level.  Not an edge -> it is one more pulse sample.
                            CNT(21, 1); // engine: candidates checked
                            int const thr = thr_of(uni((int)hv[0]));
                            int const am_k = __builtin_amdgcn_readlane(am_l, k - base);
                            if (am_k < thr - (thr >> 3)) {
                                fall = true;
                                tick(11, t_cand);
                                break;
                            }
                            v2s const in = as_v2s(__builtin_amdgcn_readlane(in_pk_l, k - base));
                            v2s const q = (hv + ((hv >> 15) & m63)) >> 6;
                            hv = pk_max(hv - q + in, floor_v);
                            j = k + 1;
                            cand &= cand - 1;
                            tick(11, t_cand);
                        }
                        long long const t_pe = now();
                        run += k - i0;
                        if (fall) {
                            if (run + 1 < 10) { // a spurious short pulse: the general step knows what that means
                                need_general = true;
                                leave = true;
                                break;
                            }
                            cur = run + 1; // pulse_detect.c:340-357: the width is known, the debounce begins
                            mx = max(cur, mx);
                            eop_lim = 10 * min(max(mx, cfg.per_ms), 10 * cfg.per_ms);
                            run = 0;
                            st = ST_GAP_START;
                            k += 1;
                        }
                        tick(9, t_pe);
                    }
                    else {
                        int const st_in = st;
                        if (!m_hi_ok) {
                            int const thr = thr_of(uni((int)hv[0]));
                            thi = thr + (thr >> 3);
                            m_hi = __ballot(am_l > thi) & vmask;
                            m_hi_ok = true;
                        }
                        unsigned long long const m = m_hi & (~0ull << (k - base));
                        int const ka = m ? base + (__ffsll(m) - 1) : e; // first sample above the level again
                        if (st == ST_GAP_START) { // pulse_detect.c:376-421 without the FSK candidate
                            int const kg = k + max(0, 9 - run); // sample at which the count reaches 10
                            if (ka <= kg && ka < e) {
                                run += (ka - k) + 1 + cur;
                                st = ST_PULSE;
                                k = ka + 1;
                            }
                            else if (kg < e) {
                                run += (kg - k) + 1;
                                st = ST_GAP;
                                k = kg + 1;
                            }
                            else {
                                run += e - k;
                                k = e;
                            }
                        }
                        if (st == ST_GAP && k < e) { // pulse_detect.c:422-470; also right after the debounce above (ka is still the one)
                            int const togo = max(0, eop_lim - run);
                            int const ke = togo < e - k ? k + togo : e; // first sample whose count ends the package
                            if (ka <= ke && ka < e) {
                                run += ka - k;
                                det.cur_pulse = cur;
                                ook_push_pair(det, run + 1); // the next pulse begins at ka: the pair is complete
                                n_pairs += 1;
                                cur = 0;
                                run = 0;
                                st = ST_PULSE;
                                k = ka + 1;
                            }
                            else {
                                run += ke - k;
                                k = ke;
                                if (ke < e) { // the package ends here: the general step emits it
                                    need_general = true;
                                    leave = true;
                                    break;
                                }
                            }
                        }
                        tick(st_in == ST_GAP_START ? 12 : 13, t_leg);
                    }
                    if (k < e)
                        continue;
                    long long const t_bu = now();
                    // the window is used up
                    leave = true;
                    if (k >= lim_u)
                        break;
                    if (st == ST_GAP) { // whole chunks without a sample above the level cannot end the gap
                        if (!m_hi_ok) {
                            int const thr = thr_of(uni((int)hv[0]));
                            thi = thr + (thr >> 3);
                        }
                        unsigned long long const m = __ballot(lane >= (k >> 5) && my_cmax() > thi);
                        int const togo = max(0, eop_lim - run);
                        int const je = togo < lim_u - k ? k + togo : lim_u;
                        int const jump_to = min(min(m ? (__ffsll(m) - 1) * kChunk : n_tu, je), lim_u);
                        if (jump_to > k) {
                            run += jump_to - k;
                            k = jump_to;
                        }
                        if (k >= lim_u)
                            break;
                    }
                    if (n_pairs + 70 >= R433_PD_MAX_PULSES) // the 1200-pulse cap is the general step's business
                        break;
                    tick(14, t_bu);
                    leave = false;
                    break;
                }
                if (leave)
                    break;
                }
                det.state = st;
                det.run = run;
                det.cur_pulse = cur;
                det.max_pulse = mx;
                det.high = uni((int)hv[0]);
                det.ook_f1 = uni((int)hv[1]);
                settled = !need_general;
            }
            if (!engine_ran)
            for (;;) {
                int const i0 = k;
                bool const in_seg = base + lane >= i0 && base + lane < e;
                int const st = det.state;
                settled = false;
                CNT(2 + (st & 3), 1); // legs by state
                if (st == ST_IDLE) {
                    if (det.lead_in <= 1024) { // no pulse can start during the lead-in (pulse_detect.c:310)
                        resolve_low(i0); // (pending steps here: only after an unfiltered stretch shorter than the lead-in)
                        k = min(e, i0 + (1025 - det.lead_in));
                        det.lead_in += k - i0;
                        int lo_est = det.low; // idle arm, pulse_detect.c:326-334
                        for (int j = i0; j < k; ++j) {
                            int const dl = __builtin_amdgcn_readlane(am_l, j - base) - lo_est;
                            lo_est += div1024(dl);
                            lo_est += dl > 0 ? 1 : -1;
                        }
                        det.low = lo_est = uni(lo_est);
                        det.high = max(cfg.ratio * lo_est, cfg.min_high);
                    }
                    else {
                        // lowest threshold the idle state can present while it chases the noise floor in this block
                        int const l_lo = min(det.low, min(lz_min, bmin)) - 1;
                        int const l_hi = max(det.low, max(lz_max, bmax)) + 1;
                        int thr = (int)(int16_t)((l_lo + min(cfg.min_high, cfg.max_high)) / 2);
                        if (cfg.fixed_high != 0)
                            thr = (int)(int16_t)cfg.fixed_high;
                        int const hys = (int)(int16_t)(thr / 8);
                        unsigned long long const m = __ballot(in_seg && am_l > thr + hys);
                        k = m ? base + (__ffsll(m) - 1) : e;
                        if (l_hi - l_lo < 1000) { // |am - low| < 1024 over everything pending: count, do not walk
                            if (lz_n == 0)
                                lz_from = i0;
                            lz_n += k - i0;
                            lz_min = min(lz_min, bmin);
                            lz_max = max(lz_max, bmax);
                        }
                        else {
                            resolve_low(i0);
                            int lo_est = det.low; // idle arm without the (impossible) pulse start, pulse_detect.c:326-334
                            for (int j = i0; j < k; ++j) {
                                int const dl = __builtin_amdgcn_readlane(am_l, j - base) - lo_est;
                                lo_est += div1024(dl);
                                lo_est += dl > 0 ? 1 : -1;
                            }
                            if (k > i0) {
                                det.low = lo_est = uni(lo_est);
                                det.high = max(cfg.ratio * lo_est, cfg.min_high);
                            }
                        }
                        if (k < e)
                            resolve_low(k); // a pulse may start at k: the general step needs the exact floor
                    }
                }
                else if (st == ST_GAP) {
                    int thr = (int)(int16_t)((det.low + min(det.high, cfg.max_high)) / 2);
                    if (cfg.fixed_high != 0)
                        thr = (int)(int16_t)cfg.fixed_high;
                    int const hys = (int)(int16_t)(thr / 8);
                    unsigned long long const m = __ballot(in_seg && am_l > thr + hys);
                    int const ka = m ? base + (__ffsll(m) - 1) : e;
                    // first sample whose gap count ends the package (pulse_detect.c:446-450)
                    int const lim = 10 * min(max(det.max_pulse, cfg.per_ms), 10 * cfg.per_ms);
                    int const togo = det.eop_spurious ? 0 : max(0, lim - det.run);
                    int const ke = togo < e - i0 ? i0 + togo : e;
                    k = min(ka, ke);
                    det.run += k - i0;
                    if (ka <= ke && ka < e && !det.eop_spurious && det.ook_num + 1 < R433_PD_MAX_PULSES) {
                        // the next pulse begins at ka (pulse_detect.c:425-440): the pair is complete
                        ook_push_pair(det, det.run + 1);
                        det.run = 0;
                        det.state = ST_PULSE;
                        k = ka + 1;
                        settled = true;
                    }
                }
                else if (st == ST_PULSE) {
                    // the level estimate cannot climb above max(high, block max); below the threshold that
                    // belongs to it no sample can be a falling edge
                    int const h_ub = max(det.high, bmax) + 1;
                    int thr_ub = (int)(int16_t)((det.low + min(h_ub, cfg.max_high)) / 2);
                    if (cfg.fixed_high != 0)
                        thr_ub = (int)(int16_t)cfg.fixed_high;
                    int const hys_ub = (int)(int16_t)(thr_ub / 8);
                    unsigned long long cand = __ballot(in_seg && am_l < thr_ub - hys_ub);
                    int h = det.high, f1 = det.ook_f1;
                    bool const feed = det.ook_num == 0; // first pulse of a package: the FSK sub-detector listens (pulse_detect.c:368-375)
                    bool const packed = uni((int)(!feed && h >= 0 && h <= 32767 && cfg.min_high >= 0 && cfg.min_high <= 32767)) != 0;
                    int const fl6 = uni(cfg.min_high >> 6);
                    unsigned long long const okp = __ballot(a64_l >= fl6 && f64_l >= 0);
                    unsigned long long const okn = __ballot(a64_l >= fl6 && f64_l <= 0 && f64_l > -512);
                    int j = i0;
                    bool fall = false;
                    for (;;) {
                        k = cand ? base + (__ffsll(cand) - 1) : e;
                        // pulse arm without the falling edge, pulse_detect.c:359-366: the level and carrier
                        // averages v += in/64 - v/64 (C division).  Both live in 16 bits (an average never
                        // leaves the hull of its start value and its inputs), so one packed instruction
                        // stream advances the two of them together.
                        if (packed) {
                            // Where the plain form of the two averages is exact: the level cannot fall below its
                            // floor while in/64 >= floor/64 (v - v/64 is monotone in v), and the carrier average
                            // keeps its sign while its inputs have that sign -- then, with |f1| in the second half,
                            // both C divisions are plain shifts (trunc(v / 64) = sign * (|v| >> 6)) and the clamp
                            // never acts: 3 packed instructions per sample instead of 7.  The inputs of such a run
                            // are rotated to lane 0 first: v_readlane with a constant lane costs a lone wavefront
                            // far less than one with a computed lane (tools/ubench/ema.hip: 23 vs 35 clocks/sample).
                            j = uni(j);
                            int const kk = uni(k);
                            v2s hv = {(short)h, (short)f1};
                            v2s const floor_v = {(short)cfg.min_high, (short)-32768};
                            v2s const m63 = {63, 63};
                            while (j < kk) {
                                CNT(6, 1); // packed-loop iterations
                                int const f1s = uni((int)hv[1]);
                                bool const neg = f1s < 0;
                                unsigned long long const bad = ~((neg ? okn : okp) >> (j - base));
                                int const run = uni(f1s == -32768 ? 0 : min(kk - j, bad ? (int)__builtin_ctzll(bad) : 64));
                                if (run >= 8) {
                                    CNT(7, (run >> 3) * 8); // samples in unrolled groups
                                    v2s const sv = {1, (short)(neg ? -1 : 1)};
                                    int const rot = __builtin_amdgcn_ds_bpermute(((int)lane + (j - base)) << 2, neg ? in_pkn_l : in_pk_l);
                                    int const nb = run >> 3;
                                    v2s x = hv * sv;
    #pragma unroll
                                    for (int b8 = 0; b8 < 8; ++b8) {
                                        if (b8 >= nb)
                                            break;
    #pragma unroll
                                        for (int u = 0; u < 8; ++u) {
                                            v2s const in = as_v2s(__builtin_amdgcn_readlane(rot, b8 * 8 + u));
                                            x = x + (in - (x >> 6));
                                        }
                                    }
                                    hv = x * sv; // the sign cannot have flipped (the magnitude may have reached 0: either sign then)
                                    j += nb * 8;
                                    continue;
                                }
                                if (run > 0) { // the tail of such a run: same plain form, lane numbers computed
                                    CNT(8, run);
                                    v2s const sv = {1, (short)(neg ? -1 : 1)};
                                    int const in_sel = neg ? in_pkn_l : in_pk_l;
                                    v2s x = hv * sv;
                                    for (int u = 0; u < run; ++u) {
                                        v2s const in = as_v2s(__builtin_amdgcn_readlane(in_sel, j - base + u));
                                        x = x + (in - (x >> 6));
                                    }
                                    hv = x * sv;
                                    j += run;
                                    continue;
                                }
                                // the samples the plain form must not take (and only those), at most 8 at a time
                                int const cnt = uni(min(min(8, kk - j), f1s == -32768 ? 8 : (int)__builtin_ctzll(~bad | (1ull << 63))));
                                CNT(9, cnt); // samples through the general 7-instruction form
                                if (cnt == 8) {
    #pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        v2s const in = as_v2s(__builtin_amdgcn_readlane(in_pk_l, j - base + u));
                                        v2s const q = (hv + ((hv >> 15) & m63)) >> 6; // hv / 64, truncating toward zero
                                        hv = pk_max(hv - q + in, floor_v);
                                    }
                                }
                                else {
                                    for (int u = 0; u < cnt; ++u) {
                                        v2s const in = as_v2s(__builtin_amdgcn_readlane(in_pk_l, j - base + u));
                                        v2s const q = (hv + ((hv >> 15) & m63)) >> 6;
                                        hv = pk_max(hv - q + in, floor_v);
                                    }
                                }
                                j += cnt;
                            }
                            h = hv[0];
                            f1 = hv[1];
                        }
                        else if (feed && cfg.fpdm != 0) {
                            // First pulse of a package with the min/max FSK detector listening (pulse_detect_fsk.c:145-221):
                            // between two of its events (a frequency toggle, or a sample exactly at the mid level) the
                            // detector only lets its upper bound sink by 10 per sample towards the signal (or its lower
                            // bound rise) -- vmax'_r = max(vmax_0 - 10 r, max_{t<=r}(v_t - 10 (r - t))), a running maximum,
                            // so every lane can tell at once whether the detector would still be in the same state at its
                            // sample.  The stretch up to the first lane that says no is taken in one go (bounds in closed
                            // form, the three averages in a short loop); the event sample itself goes through the exact step.
                            j = uni(j);
                            int const kk = uni(k);
                            while (j < kk) {
                                int const fstate = uni(det.f_state);
                                int const M0 = uni(det.f_vmax), m0 = uni(det.f_vmin);
                                int run = 0;
                                if (uni(det.f_skip) == 0 && fstate != 0 && abs(M0) <= 32000 && abs(m0) <= 32000 && h >= 0 && cfg.min_high >= 0) {
                                    int const r = lane - (j - base); // my place in the stretch
                                    bool const mine = r >= 0 && base + lane < kk;
                                    int const v = fm_l;
                                    int const key = wave_run_max(!mine ? INT32_MIN : fstate == 1 ? v + 10 * r : -(v - 10 * r), lane);
                                    int const Mr = fstate == 1 ? max(M0, key) - 10 * r : max(v, M0); // bounds as the sample sees them
                                    int const mr = fstate == 1 ? min(v, m0) : min(m0, -key) + 10 * r;
                                    int const mid = (int)(int16_t)((Mr + mr) / 2);
                                    bool const stay = (fstate == 1 ? v > mid : v < mid) && abs(v) <= 32000;
                                    unsigned long long const bad = __ballot(mine && !stay);
                                    int const stop = bad ? base + (__ffsll(bad) - 1) : kk;
                                    run = uni(stop - j);
                                    if (run > 0) {
                                        int const last = j - base + run - 1;
                                        if (fstate == 1)
                                            det.f_vmax = __builtin_amdgcn_readlane(Mr, last) - 10;
                                        else
                                            det.f_vmin = __builtin_amdgcn_readlane(mr, last) + 10;
                                        det.f_run += (uint32_t)run;
                                        // the level average alone, the two carrier averages (package and FSK detector:
                                        // same input, (sic) f2 while high, f1 while low) as a packed pair
                                        int fx = fstate == 1 ? det.f_f2 : det.f_f1;
                                        v2s fv = {(short)f1, (short)fx};
                                        v2s const m63 = {63, 63};
                                        // scalar lane numbers, four samples per trip: a lane number that sits in a vector register
                                        // costs a v_readfirstlane and its wait states per sample (19 -> 13 instructions per sample)
                                        int const idx0 = uni(j - base);
                                        auto step = [&](int at) {
                                            h = max(h - (h >> 6) + __builtin_amdgcn_readlane(a64_l, at), cfg.min_high);
                                            v2s const in = as_v2s(__builtin_amdgcn_readlane(ff_pk_l, at));
                                            v2s const q = (fv + ((fv >> 15) & m63)) >> 6; // v / 64, truncating toward zero
                                            fv = fv - q + in;
                                        };
                                        int u = 0;
                                        for (; u + 4 <= run; u += 4) {
                                            step(idx0 + u);
                                            step(idx0 + u + 1);
                                            step(idx0 + u + 2);
                                            step(idx0 + u + 3);
                                        }
                                        for (; u < run; ++u)
                                            step(idx0 + u);
                                        f1 = fv[0];
                                        fx = fv[1];
                                        if (fstate == 1)
                                            det.f_f2 = fx;
                                        else
                                            det.f_f1 = fx;
                                        j += run;
                                    }
                                }
                                if (run == 0) { // the detector's skip samples, its first decision, an event, values at the edge of int16
                                    h += __builtin_amdgcn_readlane(a64_l, j - base) - div64(h);
                                    h = max(h, cfg.min_high);
                                    f1 += __builtin_amdgcn_readlane(f64_l, j - base) - div64(f1);
                                    fsk_feed(det, cfg, __builtin_amdgcn_readlane(fm_l, j - base));
                                    j += 1;
                                }
                            }
                        }
                        else if (feed) {
                            // ... and with the classic FSK detector listening (pulse_detect_fsk.c:34-141): while it sits on
                            // one of its two frequency estimates, that estimate follows the signal (v/16 or v/64 steps,
                            // never leaving the hull of where it started and what it has seen since -- truncating division
                            // is monotone), and the detector toggles when the sample is closer to the other estimate.  Against
                            // the running hull of the samples before it every lane knows whether a toggle is possible at its
                            // sample; up to the first lane that cannot rule it out only the averages advance.
                            constexpr int kNone = -0x40000000;
                            j = uni(j);
                            int const kk = uni(k);
                            while (j < kk) {
                                int const fstate = uni(det.f_state);
                                int run = 0;
                                // (before its first decision the detector follows the signal with its first estimate in v/16
                                // steps from the 10th sample on and decides when a sample is more than 3000 away, :44-68)
                                if ((fstate != 0 || uni((int)det.f_run) >= 9) && h >= 0 && cfg.min_high >= 0) {
                                    int const F0 = uni(fstate == 2 ? det.f_f2 : det.f_f1); // the estimate the detector sits on
                                    int const other = uni(fstate == 1 ? det.f_f2 : det.f_f1);
                                    int const r = lane - (j - base);
                                    bool const mine = r >= 0 && base + lane < kk;
                                    int const v = fm_l;
                                    int up = __shfl_up(wave_run_max(mine ? v : kNone, lane), 1u, 64);   // max of the samples before mine
                                    int dn = __shfl_up(wave_run_max(mine ? -v : kNone, lane), 1u, 64);  // -min
                                    if (r <= 0 || lane == 0)
                                        up = dn = kNone;
                                    int const U = max(F0, up), L = min(F0, -max(dn, -0x3fffffff));
                                    bool const possible = max(abs(v - L), abs(v - U)) > (fstate == 0 ? 3000 : abs(v - other));
                                    unsigned long long const bad = __ballot(mine && possible);
                                    run = uni((bad ? base + (__ffsll(bad) - 1) : kk) - j);
                                    if (run > 0) {
                                        det.f_run += (uint32_t)run;
                                        int F = F0;
                                        int const idx0 = uni(j - base);
                                        // one copy of the loop per state, no branch inside: a taken branch costs a lone
                                        // wavefront more than the few instructions it would skip
                                        auto advance = [&](auto mode_tag) {
                                            constexpr int MODE = decltype(mode_tag)::value; // the detector's state
                                            auto step = [&](int at) {
                                                int const x = __builtin_amdgcn_readlane(fm_l, at);
                                                int const x64 = __builtin_amdgcn_readlane(f64_l, at);
                                                int const x16 = (x + ((x >> 31) & 15)) >> 4; // x / 16, C division
                                                h = max(h - (h >> 6) + __builtin_amdgcn_readlane(a64_l, at), cfg.min_high);
                                                f1 += x64 - div64(f1);
                                                bool const fast = MODE == 0 || (MODE == 1 ? x > F : x < F); // towards the outside: 1/16 steps
                                                int const sh = fast ? 4 : 6, bias = fast ? 15 : 63;
                                                F += (fast ? x16 : x64) - ((F + ((F >> 31) & bias)) >> sh); // v/16 - F/16 or v/64 - F/64, C division
                                            };
                                            int u = 0;
                                            for (; u + 4 <= run; u += 4) { // four samples per trip: the loop bookkeeping is a fifth of a sample's work
                                                step(idx0 + u);
                                                step(idx0 + u + 1);
                                                step(idx0 + u + 2);
                                                step(idx0 + u + 3);
                                            }
                                            for (; u < run; ++u)
                                                step(idx0 + u);
                                        };
                                        if (fstate == 0)
                                            advance(std::integral_constant<int, 0>{});
                                        else if (fstate == 1)
                                            advance(std::integral_constant<int, 1>{});
                                        else
                                            advance(std::integral_constant<int, 2>{});
                                        if (fstate == 2)
                                            det.f_f2 = F;
                                        else
                                            det.f_f1 = F;
                                        j += run;
                                    }
                                }
                                if (run == 0) { // start-up of the detector, a possible toggle
                                    h += __builtin_amdgcn_readlane(a64_l, j - base) - div64(h);
                                    h = max(h, cfg.min_high);
                                    f1 += __builtin_amdgcn_readlane(f64_l, j - base) - div64(f1);
                                    fsk_feed(det, cfg, __builtin_amdgcn_readlane(fm_l, j - base));
                                    j += 1;
                                }
                            }
                        }
                        else {
                            CNT(12, k - j); // samples through the scalar fallback
                            for (; j < k; ++j) {
                                h += __builtin_amdgcn_readlane(a64_l, j - base) - div64(h);
                                h = max(h, cfg.min_high);
                                f1 += __builtin_amdgcn_readlane(f64_l, j - base) - div64(f1);
                            }
                        }
                        if (k >= e)
                            break;
                        CNT(10, 1); // candidates checked
                        // candidate: decide with the exact level.  Not an edge -> it is one more pulse sample.
                        int thr = (int)(int16_t)((det.low + min(h, cfg.max_high)) / 2);
                        if (cfg.fixed_high != 0)
                            thr = (int)(int16_t)cfg.fixed_high;
                        int const hys = (int)(int16_t)(thr / 8);
                        int const am_k = __builtin_amdgcn_readlane(am_l, k - base);
                        if (am_k < thr - hys) {
                            fall = true;
                            break;
                        }
                        h += __builtin_amdgcn_readlane(a64_l, k - base) - div64(h);
                        h = max(h, cfg.min_high);
                        f1 += __builtin_amdgcn_readlane(f64_l, k - base) - div64(f1);
                        if (feed)
                            fsk_feed(det, cfg, __builtin_amdgcn_readlane(fm_l, k - base));
                        j = k + 1;
                        cand &= cand - 1; // next candidate
                    }
                    det.high = uni(h);
                    det.ook_f1 = uni(f1);
                    det.run += k - i0;
                    if (fall && !feed && det.run + 1 >= 10) {
                        // the pulse ends at k (pulse_detect.c:340-357, the regular case): its width is known, the
                        // debounce of the gap begins.  (Spurious short pulses and the first pulse of a package,
                        // which the FSK detector listens to, go through the general step.)
                        det.cur_pulse = det.run + 1;
                        det.max_pulse = max(det.cur_pulse, det.max_pulse);
                        det.run = 0;
                        det.state = ST_GAP_START;
                        k += 1;
                        settled = true;
                    }
                }
                else if (st == ST_GAP_START && det.ook_num > 0 && det.fsk_num <= 16) {
                    // debouncing the end of a pulse (pulse_detect.c:376-421) once the FSK candidate is out of
                    // the picture: either the signal comes back before the count reaches 10, or the gap begins.
                    // (The candidate can still be in the picture after the first pulse: the sample that ends the
                    // first debounce is fed to the FSK detector AFTER the `> 16 pulses` test, so the 17th FSK
                    // pulse may arrive there and the reference then returns the FSK package at the end of the
                    // NEXT pulse -- the general step does that.)
                    int thr = (int)(int16_t)((det.low + min(det.high, cfg.max_high)) / 2);
                    if (cfg.fixed_high != 0)
                        thr = (int)(int16_t)cfg.fixed_high;
                    int const hys = (int)(int16_t)(thr / 8);
                    unsigned long long const m = __ballot(in_seg && am_l > thr + hys);
                    int const ka = m ? base + (__ffsll(m) - 1) : e;   // first sample above the threshold again
                    int const kg = i0 + max(0, 9 - det.run);            // sample at which the count reaches 10
                    if (ka <= kg && ka < e) {
                        det.run += (ka - i0) + 1 + det.cur_pulse;
                        det.state = ST_PULSE;
                        k = ka + 1;
                    }
                    else if (kg < e) {
                        det.run += (kg - i0) + 1;
                        det.state = ST_GAP;
                        k = kg + 1;
                    }
                    else {
                        det.run += e - i0;
                        k = e;
                    }
                    settled = true;
                }
                if (k < e) {
                    if (!settled)
                        break; // the general step takes over at k
                    continue;
                }
                // The block is used up and nothing is pending.  Inside a package (pulse, debounce, gap) the next block
                // follows at once -- a gap first skips whole chunks, like the outer loop does -- ; idle stretches and the
                // ends of the frame and of the tile go round the outer loop.
                settled = true;
                if (k >= lim || det.state == ST_IDLE)
                    break;
                if (det.state == ST_GAP) {
                    int thr = (int)(int16_t)((det.low + min(det.high, cfg.max_high)) / 2);
                    if (cfg.fixed_high != 0)
                        thr = (int)(int16_t)cfg.fixed_high;
                    int const hys = (int)(int16_t)(thr / 8);
                    unsigned long long const m = __ballot(lane >= (k >> 5) && my_cmax() > thr + hys);
                    int const lim_eop = 10 * min(max(det.max_pulse, cfg.per_ms), 10 * cfg.per_ms);
                    int const togo = det.eop_spurious ? 0 : max(0, lim_eop - det.run);
                    int const je = togo < lim - k ? k + togo : lim;
                    int const jump_to = min(min(m ? (__ffsll(m) - 1) * kChunk : n_t, je), lim);
                    if (jump_to > k) {
                        det.run += jump_to - k;
                        k = jump_to;
                    }
                    if (k >= lim)
                        break;
                }
                base = k & ~63;
                e = min(base + 64, lim);
                load_block();
            }
            if (settled) { // everything up to k is done: no general step this round
                int const done = k - i;
                i += done;
                dc += done;
                if (dc == flen)
                    frame_done();
                tick(st_it == ST_IDLE ? 1 : st_it == ST_GAP ? 2 : st_it == ST_PULSE ? 3 : 4, t_it);
                continue;
            }

            tick(st_it == ST_IDLE ? 1 : st_it == ST_GAP ? 2 : st_it == ST_PULSE ? 3 : 4, t_it);
            long long const t_fast = now();
            int consumed = k - i;
            if (k < e) { // the exact general step: candidate samples, and the states that need every sample
                int j = k;
                int local_dc = dc + (k - i);
                do {
                    int const am = __builtin_amdgcn_readlane(am_l, j - base), fm = __builtin_amdgcn_readlane(fm_l, j - base);
                    CNT(11, 1); // general steps
                    int const r = det_step(det, cfg, am, fm, flen, local_dc, input_pos, frame);
                    if (r && logic) { // the returned struct, before the re-examined sample may start the next package
                        __threadfence();
                        if (r == R433_PKG_OOK)
                            paint(det.offset, det.ook_num, (int2 const *)(det.arena + det.ook_base + sizeof(r433_pkg_rec)), 0x02, input_pos, flen);
                        else
                            paint(det.fsk_offset, det.fsk_num, det.fsk_ring, 0x04, input_pos, flen);
                    }
                    if (r) { // package returned: the next call starts at the same sample, in the idle state
                        det_call_entry(det, cfg, flen, local_dc);
                        det_idle(det, cfg, am, flen, local_dc, input_pos);
                    }
                    ++j;
                    ++local_dc;
                } while (j < e && det.ook_num == 0 && (det.state == ST_GAP_START || det.state == ST_PULSE));
                consumed += j - k;
                pin_state();
            }
            i += consumed;
            dc += consumed;
            if (dc == flen)
                frame_done();
            tick(5, t_fast);
        }
        long long const t_res = now();
        if (!lz_fail)
            resolve_low(n_t);
        if (lz_fail && lane == 0)
            st_retry(retry_slot) = lz_fail;
        tick(6, t_res); // the samples leave LDS with the tile
    };

    // tile t + 1 is produced while tile t is consumed; one barrier per tile hands a buffer over and takes one back.  The two
    // roles are two loops: what one role keeps across tiles is dead in the other's loop (registers, scalar and vector).
    // (Lazy tiles: either role may find that the capture cannot be carried exactly across its unfiltered tiles and raises
    // st_retry in the slot of the barrier its tile ends at; both roles read that slot behind that barrier -- nobody writes it
    // before the barrier after the next -- and start the capture over with every tile filtered.)
    // what both roles do between two attempts: everybody has seen the flag, it is cleared, every tile is filtered from now on
    auto start_over = [&]() {
        if (!lone)
            __syncthreads();
        if (threadIdx.x == 0) {
            st_retry(0) = st_retry(1) = 0;
            st_desc(0) = st_desc(1) = 0;
            s_pover = 0;
        }
        lazy = false;
        attempts += 1;
        if (lone)
            wave_sync();
        else
            __syncthreads();
    };
    if constexpr (to_hbm) {
        // ---- the producers of a grid as a launch of their own: every tile's record and descriptor to HBM, nobody to wait for
        for (;;) {
            init_producer();
            self_retry = 0;
            p_active = 0;
            if (lane == 0)
                p.tile_over[s] = 0;
            issue_loads(tile_first);
            if (lazy)
                issue_head(tile_first + 1u);
            for (uint32_t tile = tile_first; tile < tile_end && !self_retry; ++tile)
                produce(tile, 0);
            if (!self_retry)
                break;
            retry_why |= self_retry; // (carries that do not settle after unfiltered tiles: again, every tile filtered)
            lazy = false;
            attempts += 1;
        }
        // what the consumers' launch sorts its captures by (k_order_falling): the chunks the detector cannot skip
        p_active = wave_sum(p_active);
        if (lane == 0) {
            p.tile_info[s] = (uint32_t)attempts | ((uint32_t)retry_why << 8);
            if (p.cons_weight)
                p.cons_weight[s] = min((uint32_t)p_active >> 3, 254u) + (tile_end > tile_first ? 1u : 0u);
        }
        return;
    }
    else if constexpr (from_hbm) {
        // ---- the consumers of a grid as a launch of their own, behind the producers' launch
        init_consumer();
        for (uint32_t tile = tile_first; tile < tile_end; ++tile) {
            int const desc = uni(g_desc[tile]);
            if (lane == 0)
                st_desc(0) = desc;
            if (!(desc & kQuietTile)) { // a filtered tile: its record into this wavefront's LDS, laid out as a pair's buffer 0
                uint8_t const *const rec = g_tiles + (uint64_t)tile * kTileRecBytes;
#pragma unroll
                for (int k = 0; k < 2 * 64 * kPitchOut / 1024; ++k)
                    *(uint4 *)(s_tiles + k * 1024 + lane * 16) = *(uint4 const *)(rec + k * 1024 + lane * 16);
                st_cmax(0, lane) = ((short const *)rec)[kTileRecMax / 2 + lane];
                st_cmin(0, lane) = ((short const *)rec)[kTileRecMin / 2 + lane];
            }
            wave_sync();
            consume(tile, 0);
            wave_sync(); // (the tile is done with before the next one's record lands on it)
            if (lz_fail) {
                // the capture cannot be carried exactly across its unfiltered tiles: onto the list of the run-again launch
                if (lane == 0) {
                    uint32_t const at = atomicAdd(p.retry_count, 1u);
                    p.retry_list[at] = s;
                    p.retry_why[s] = (uint32_t)lz_fail;
                }
                return; // (that launch writes this capture's state and packages afresh)
            }
        }
        uint32_t const info = p.tile_info[s];
        attempts = (int)(info & 0xffu);
        retry_why = (int)(info >> 8);
        if (lane == 0)
            s_pover = p.tile_over[s];
        wave_sync();
    }
    else if constexpr (solo) {
        for (;;) {
            init_consumer();
            init_producer();
            bool again = false;
            issue_loads(tile_first);
            if (lazy)
                issue_head(tile_first + 1u);
            for (uint32_t tile = tile_first; tile < tile_end; ++tile) {
                produce(tile, 0);
                wave_sync();
                if (!SEAM)
                    consume(tile, 0);
                if (lazy) {
                    wave_sync();
                    if (int const why = uni(st_retry(0))) {
                        retry_why |= why;
                        again = true;
                        break;
                    }
                }
            }
            if (!again)
                break;
            start_over();
        }
    }
    else if (role == 0) {
        for (;;) {
            init_producer();
            bool again = false;
            issue_loads(tile_first);
            if (lazy)
                issue_head(tile_first + 1u);
            for (uint32_t it = tile_first; it <= tile_end; ++it) {
                retry_slot = (int)(it & 1u);
                if (it < tile_end)
                    produce(it, (int)(it & 1u));
                __syncthreads();
                if (lazy && uni(st_retry((int)(it & 1u)))) {
                    again = true;
                    break;
                }
            }
            if (!again)
                break;
            start_over();
        }
        return; // the consumer wavefronts report
    }
    else {
        for (;;) {
            init_consumer();
            bool again = false;
            for (uint32_t it = tile_first; it <= tile_end; ++it) {
                retry_slot = (int)(it & 1u);
                if (it > tile_first && !idle)
                    consume(it - 1, (int)((it - 1) & 1u));
                __syncthreads();
                if (int const why = lazy ? uni(st_retry((int)(it & 1u))) : 0) {
                    retry_why |= why;
                    again = true;
                    break;
                }
            }
            if (!again)
                break;
            start_over();
        }
    }
    if (idle)
        return; // (a triple's third wavefront whose piece has one variant only)
    if (s_pover)
        det.overflow = (uint32_t)s_pover;


    if (SEAM) {
        if (lane == 0) {
            StreamState &S = p.state[s];
            S.lpf_y = seam_end[0], S.lpf_x = seam_end[1], S.fm_yf = seam_end[2], S.fm_xf = seam_end[3];
            S.overflow = det.overflow;
        }
        return;
    }
    int const end_state = det.state, end_high = det.high; // before the flush: what the next segment has to agree with
    if (!(p.flags & RUN_NOFLUSH) && (seg_flags & SEG_LAST))
        det_flush(det, cfg, frame);
    if (lane == 0) {
        StreamState &S = p.state[s];
        S.seg_init_low = seg_init_low;
        S.seg_init_high = seg_init_high;
        S.seg_fail = seg_fail;
        S.seg_end_state = end_state;
        S.seg_end_lead = det.lead_in;
        S.seg_end_low = det.low;
        S.seg_end_high = end_high;
        S.cursor = det.cursor;
        S.n_pkgs = det.n_pkgs;
        S.overflow = det.overflow;
        S.input_pos = input_pos;
        S.frame = frame;
        S.fm_xr = n_quiet;  // (statistics, r433_batch_debug_state: tiles that went by unfiltered; attempts beyond the first)
        S.fm_xi = attempts;
        S.fm_xf = retry_why; // 1, 2 filter carries after unfiltered tiles; 4 floor steps of more than one; 32, 64 the floor walks did not meet / had no samples
        if (timing) {
            S.lpf_y = (int)(s_tk[0] >> 6), S.lpf_x = (int)(s_tk[1] >> 6), S.fm_xr = (int)(s_tk[2] >> 6), S.fm_xi = (int)(s_tk[3] >> 6);
            S.fm_xf = (int)(s_tk[4] >> 6), S.fm_yf = (int)(s_tk[5] >> 6), S.state = (int)(s_tk[6] >> 6), S.run = (int)s_tk[7];
            S.max_pulse = (int)(s_tk[8] >> 6), S.lead_in = (int)(s_tk[9] >> 6), S.low = (int)(s_tk[10] >> 6), S.high = (int)(s_tk[11] >> 6);
            S.f_state = (int)(s_tk[12] >> 6), S.f_f1 = (int)(s_tk[13] >> 6), S.f_f2 = (int)(s_tk[14] >> 6), S.f_vmax = (int)s_tk[15];
        }
    }
}

// ---- package directory: canonical (capture, detection order) numbering ----

__global__ __launch_bounds__(1024) void k_pkg_scan(StreamState const *state, uint32_t const *order, uint32_t n_streams,
        uint32_t *pkg_base, uint32_t *scal)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    __shared__ uint32_t any_overflow;
    int const tid = (int)threadIdx.x;
    if (tid == 0) {
        carry = 0;
        any_overflow = 0;
    }
    __syncthreads();
    for (uint32_t base = 0; base < n_streams; base += 1024) {
        uint32_t i = base + (uint32_t)tid;
        uint32_t const slot = i < n_streams ? (order ? order[i] : i) : 0u;
        uint32_t v = i < n_streams ? state[slot].n_pkgs : 0u;
        if (i < n_streams && state[slot].overflow)
            any_overflow = 1;
        part[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) { // Hillis-Steele inclusive scan
            uint32_t add = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (i < n_streams)
            pkg_base[i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023)
            carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) {
        scal[0] = carry;
        scal[1] = any_overflow;
    }
}

__global__ __launch_bounds__(256) void k_pkg_directory(uint8_t const *arena, uint32_t arena_stride,
        StreamState const *state, uint32_t const *order, uint32_t n_streams, uint32_t const *pkg_base,
        uint32_t *dir_stream, uint32_t *dir_off, uint32_t *rec_bytes, uint32_t max_pkgs)
{
    uint32_t const k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_streams)
        return;
    uint32_t const s = order ? order[k] : k; // arena slot
    uint32_t n = state[s].n_pkgs;
    uint32_t at = 0;
    uint8_t const *a = arena + (uint64_t)s * arena_stride;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t g = pkg_base[k] + i;
        uint32_t sz = *(uint32_t const *)(a + at);
        if (g < max_pkgs) {
            dir_stream[g] = s;
            dir_off[g] = at;
            rec_bytes[g] = sz;
        }
        at += sz;
    }
}

// one wavefront per package: records are multiples of 8 bytes and 8-byte aligned on both sides
__global__ __launch_bounds__(64) void k_gather_packages(uint8_t const *arena, uint32_t arena_stride,
        uint32_t const *dir_stream, uint32_t const *dir_off, uint32_t const *rec_off, uint32_t const *n_pkgs,
        uint32_t max_pkgs, uint8_t *dst, uint32_t dst_cap)
{
    uint32_t const n = min(*n_pkgs, max_pkgs);
    for (uint32_t g = blockIdx.x; g < n; g += gridDim.x) {
        uint8_t const *src = arena + (uint64_t)dir_stream[g] * arena_stride + dir_off[g];
        uint32_t bytes = *(uint32_t const *)src;
        uint32_t off = rec_off[g];
        if ((uint64_t)off + bytes > dst_cap)
            continue;
        uint2 const *s2 = (uint2 const *)src;
        uint2 *d2 = (uint2 *)(dst + off);
        for (uint32_t i = threadIdx.x; i < bytes / 8; i += 64)
            d2[i] = s2[i];
    }
}

// The consumers' launch (FORM 5) takes its captures heaviest first BY THEIR OWN WORK: what the producers left -- per capture the
// chunks of its filtered tiles whose envelope rises above the level below which the detector only idles or counts (quiet_bound:
// everything else the consumer skips by the chunk maxima); every producer leaves that count as it goes (StreamParams::
// cons_weight).  The producers' own order (a look at 16 samples of the raw capture, k_capture_weight) says little about it, and a
// launch of 2.7 rounds of consumers lasts as long as its last round's slowest capture.  Order only: every capture is still one
// workgroup.
// captures by falling weight (a counting sort over 256 weights, one workgroup)
__global__ __launch_bounds__(256) void k_order_falling(uint32_t const *weight, uint32_t n, uint32_t *order)
{
    __shared__ uint32_t count[256], first[256];
    count[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        atomicAdd(&count[weight[i] & 255u], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 255; w >= 0; --w) {
            first[w] = run;
            run += count[w];
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        order[atomicAdd(&first[weight[i] & 255u], 1u)] = i;
}

} // namespace

bool launch_stream(StreamParams const &p, uint32_t sample_size, hipStream_t st)
{
    if (p.n_streams == 0)
        return false;
    // Two wavefronts per capture (producer + consumer), three of them to a SIMD (the pair kernel is built for 168 VGPRs, a
    // workgroup's 26 KB of LDS make six workgroups to a CU): 1536 captures in flight.  One wavefront doing both in turn needs
    // the registers of both roles (two to a SIMD, 2048 captures in flight) and only wins where exactly that many more
    // captures fit the chip at once.  Per launch on an MI355X, pair / single, round 3: 1024 captures 1.34 / 1.60 ms,
    // 1536: 1.42 / 1.86, 2048: 1.99 / 1.88, 3072: 2.63 / 2.93, 4096: 3.09 / 3.33, 8192: 5.56 / 5.73
    // (profiles/r03_b_pair_vs_single.txt).  RUN_ONE_WAVE / RUN_PAIR force either form (A/B timing, tests).
    bool const pair = (p.flags & RUN_PAIR) || (!(p.flags & RUN_ONE_WAVE) && !(p.n_streams > 1536u && p.n_streams <= 2304u));
    // split captures come with their workgroup list: a producer and (where a piece has both parity variants) two consumers
    // (a workgroup list without pieces is only an order: launch_capture_order)
    bool const triple = p.wg_slot != nullptr && p.segs != nullptr && !(p.flags & RUN_ONE_WAVE);
    // the two roles as two launches (FORM 4, FORM 5) and a third for what has to run again: whole captures only, no taps,
    // no logic dump, IQ input
    bool const split_roles = (p.flags & RUN_SPLIT_ROLES) && p.tile_store && !p.segs && !p.tap_env && !p.tap_am && !p.logic
            && !(p.flags & (RUN_AM_IS_INPUT | RUN_FM_IS_INPUT | RUN_ENV_RAW16 | RUN_ONE_WAVE | RUN_DBG_TIMING));
    dim3 grid(p.wg_slot ? p.n_wgs : p.n_streams), block(triple ? 192 : pair ? 128 : 64);
    uint32_t lds = (pair || triple ? 4u : 2u) * 64u * (uint32_t)kPitchOut // the tile buffers (s_tiles)
            + ((p.flags & (RUN_AM_IS_INPUT | RUN_FM_IS_INPUT)) ? 64u * (uint32_t)kPitch16 : 0u); // + s_raw
    // FAST: no filter step can wrap and both feedback coefficients are non-negative (see Track16).
    // The AM filter always qualifies (13993 + 2*1195 <= 16384); the FM filter does for every cutoff
    // up to half the Nyquist rate, which includes the defaults.
    bool fast;
    if (sample_size == 2)
        fast = !p.enable_fm || (p.a16 >= 0 && p.b16 >= 0 && p.a16 + 2 * p.b16 <= 16384);
    else
        fast = !p.enable_fm || (p.a32 >= 0 && p.b32 >= 0 && p.a32 + 2 * p.b32 <= (1ll << 30));
    bool const fm = p.enable_fm != 0;
#define R433_LAUNCH_WAVE(SS, FORM, P)                                                                                  \
    do {                                                                                                               \
        if (fast && fm)                                                                                                \
            hipLaunchKernelGGL((k_wave<SS, true, true, false, FORM>), grid, block, lds, st, P);                          \
        else if (fast)                                                                                                 \
            hipLaunchKernelGGL((k_wave<SS, true, false, false, FORM>), grid, block, lds, st, P);                         \
        else if (fm)                                                                                                   \
            hipLaunchKernelGGL((k_wave<SS, false, true, false, FORM>), grid, block, lds, st, P);                         \
        else                                                                                                           \
            hipLaunchKernelGGL((k_wave<SS, false, false, false, FORM>), grid, block, lds, st, P);                        \
    } while (0)
    if (split_roles) {
        (void)hipMemsetAsync(p.retry_count, 0, sizeof(uint32_t), st);
        block = dim3(64);
        lds = 0; // the producers stage in the static arrays only
        if (sample_size == 2)
            R433_LAUNCH_WAVE(2, 4, p);
        else
            R433_LAUNCH_WAVE(4, 4, p);
        lds = 2u * 64u * (uint32_t)kPitchOut; // the consumers: one tile
        StreamParams c = p;
        if (p.cons_weight && p.cons_order) { // ... heaviest first by what the producers left them
            hipLaunchKernelGGL(k_order_falling, dim3(1), dim3(256), 0, st, p.cons_weight, p.n_streams, p.cons_order);
            c.wg_slot = p.cons_order;
            c.n_wgs = p.n_streams;
        }
        if (sample_size == 2)
            R433_LAUNCH_WAVE(2, 5, c);
        else
            R433_LAUNCH_WAVE(4, 5, c);
        // what a consumer could not carry across its unfiltered tiles runs again as a pair, every tile filtered (the frame
        // sums were added by the first launch); a grid of the chip's size, strided over the list by further launches only if
        // it ever were longer (it never is: one capture in tens of thousands)
        StreamParams q = p;
        q.flags = (p.flags | RUN_NO_LAZY | RUN_RETRY_PASS) & ~(uint32_t)RUN_SPLIT_ROLES;
        q.frame_sums = nullptr;
        q.wg_slot = p.retry_list;
        q.n_wgs = p.n_streams;
        q.wg_count = p.retry_count;
        grid = dim3(p.n_streams);
        block = dim3(128);
        lds = 4u * 64u * (uint32_t)kPitchOut;
        if (sample_size == 2)
            R433_LAUNCH_WAVE(2, 2, q);
        else
            R433_LAUNCH_WAVE(4, 2, q);
        return true;
    }
    else if (sample_size == 2 && triple)
        R433_LAUNCH_WAVE(2, 3, p);
    else if (sample_size == 2 && pair)
        R433_LAUNCH_WAVE(2, 2, p);
    else if (sample_size == 2)
        R433_LAUNCH_WAVE(2, 1, p);
    else if (triple)
        R433_LAUNCH_WAVE(4, 3, p);
    else if (pair)
        R433_LAUNCH_WAVE(4, 2, p);
    else
        R433_LAUNCH_WAVE(4, 1, p);
#undef R433_LAUNCH_WAVE
    return false;
}

void launch_filters(StreamParams const &p, uint32_t sample_size, hipStream_t st)
{
    if (p.n_streams == 0)
        return;
    dim3 grid(p.n_streams), block(64);
    uint32_t const lds = 2u * 64u * (uint32_t)kPitchOut; // the tile buffers (s_tiles) of a single wavefront
    bool const fm = p.enable_fm != 0;
    bool fast;
    if (sample_size == 2)
        fast = !fm || (p.a16 >= 0 && p.b16 >= 0 && p.a16 + 2 * p.b16 <= 16384);
    else
        fast = !fm || (p.a32 >= 0 && p.b32 >= 0 && p.a32 + 2 * p.b32 <= (1ll << 30));
    if (sample_size == 2) {
        if (!fm)
            hipLaunchKernelGGL((k_wave<2, true, false, true>), grid, block, lds, st, p);
        else if (fast)
            hipLaunchKernelGGL((k_wave<2, true, true, true>), grid, block, lds, st, p);
        else
            hipLaunchKernelGGL((k_wave<2, false, true, true>), grid, block, lds, st, p);
    }
    else {
        if (fast)
            hipLaunchKernelGGL((k_wave<4, true, true, true>), grid, block, lds, st, p);
        else
            hipLaunchKernelGGL((k_wave<4, false, true, true>), grid, block, lds, st, p);
    }
}

void launch_pkg_scan(StreamState const *state, uint32_t const *order, uint32_t n, uint32_t *pkg_base, uint32_t *scal,
        hipStream_t st)
{
    hipLaunchKernelGGL(k_pkg_scan, dim3(1), dim3(1024), 0, st, state, order, n, pkg_base, scal);
}

void launch_directory(uint8_t const *arena, uint32_t arena_stride, StreamState const *state, uint32_t const *order,
        uint32_t n, uint32_t const *pkg_base, uint32_t *dir_stream, uint32_t *dir_off, uint32_t *rec_bytes,
        uint32_t max_pkgs, hipStream_t st)
{
    hipLaunchKernelGGL(k_pkg_directory, dim3((n + 255) / 256), dim3(256), 0, st, arena, arena_stride, state, order, n,
            pkg_base, dir_stream, dir_off, rec_bytes, max_pkgs);
}

void launch_gather_packages(uint8_t const *arena, uint32_t arena_stride, uint32_t const *dir_stream,
        uint32_t const *dir_off, uint32_t const *rec_off, uint32_t const *n_pkgs, uint32_t max_pkgs, uint8_t *dst,
        uint32_t dst_cap, uint32_t grid_pkgs, hipStream_t st)
{
    uint32_t grid = grid_pkgs < 1 ? 1 : (grid_pkgs > 8192 ? 8192 : grid_pkgs);
    hipLaunchKernelGGL(k_gather_packages, dim3(grid), dim3(64), 0, st, arena, arena_stride, dir_stream, dir_off, rec_off,
            n_pkgs, max_pkgs, dst, dst_cap);
}

} // namespace r433
