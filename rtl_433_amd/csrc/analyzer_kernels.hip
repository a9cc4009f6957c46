// analyzer_kernels.hip -- the pulse analyzer (`-A`, reference src/pulse_analyzer.c:279-430) per package.
//
// One wavefront per package.  The five width histograms are tolerance clusterings with at most 16 bins
// (src/pulse_analyzer.c:38-66): every lane below 16 owns one bin in registers, a value is matched against all
// bins at once (ballot, lowest matching lane takes it; no match: the next free lane opens a bin), so the 6 x
// num_pulses sequential insertions cost a handful of instructions each and nothing spills.  Fusing, the two
// bubble sorts (whose exchange order decides ties) and the modulation guess work on <= 16 bins and are done by
// lane 0 on the LDS copy, exactly in the reference's order of operations.  Output: one r433_analysis per package.
#include "r433_hip.h"
#include "r433_internal.hpp"

namespace r433 {

namespace {

constexpr float kTolerance = 0.2f; // src/pulse_analyzer.c:211

__device__ __forceinline__ bool within(int bn, int bm)
{
    return (float)abs(bn - bm) < kTolerance * (float)max(bn, bm); // :46, :142
}

// one bin per lane (lanes >= 16 idle)
struct LaneBin {
    uint32_t count = 0;
    int sum = 0, mean = 0, mn = 0, mx = 0;
};

// histogram_sum for one value, src/pulse_analyzer.c:42-65
__device__ __forceinline__ void hist_add(LaneBin &b, uint32_t &bins_count, int v, uint32_t lane)
{
    bool const match = lane < bins_count && within(v, b.mean);
    unsigned long long const m = __ballot(match);
    if (m) {
        if (lane == (uint32_t)(__ffsll(m) - 1)) {
            b.count += 1;
            b.sum += v;
            b.mean = b.sum / (int)b.count;
            b.mn = min(v, b.mn);
            b.mx = max(v, b.mx);
        }
    }
    else if (bins_count < R433_HIST_BINS) {
        if (lane == bins_count) {
            b.count = 1;
            b.sum = b.mean = b.mn = b.mx = v;
        }
        bins_count += 1;
    }
}

__device__ __forceinline__ void hist_store(r433_histogram &h, LaneBin const &b, uint32_t bins_count, uint32_t lane)
{
    if (lane < R433_HIST_BINS) {
        r433_hist_bin o;
        o.count = lane < bins_count ? b.count : 0u;
        o.sum = lane < bins_count ? b.sum : 0;
        o.mean = lane < bins_count ? b.mean : 0;
        o.min = lane < bins_count ? b.mn : 0;
        o.max = lane < bins_count ? b.mx : 0;
        h.bins[lane] = o;
    }
    if (lane == 0)
        h.bins_count = bins_count;
}

// ---- lane 0 only: the small serial parts ----
__device__ void hist_delete_bin(r433_histogram &h, uint32_t index) // :69-82
{
    if (h.bins_count < 1)
        return;
    for (uint32_t n = index; n + 1 < h.bins_count; ++n)
        h.bins[n] = h.bins[n + 1];
    h.bins_count -= 1;
    h.bins[h.bins_count] = r433_hist_bin{0, 0, 0, 0, 0};
}

__device__ void hist_fuse(r433_histogram &h) // :130-154
{
    if (h.bins_count < 2)
        return;
    for (uint32_t n = 0; n + 1 < h.bins_count; ++n) {
        for (uint32_t m = n + 1; m < h.bins_count; ++m) {
            if (within(h.bins[n].mean, h.bins[m].mean)) {
                h.bins[n].count += h.bins[m].count;
                h.bins[n].sum += h.bins[m].sum;
                h.bins[n].mean = h.bins[n].sum / (int)h.bins[n].count;
                h.bins[n].min = min(h.bins[n].min, h.bins[m].min);
                h.bins[n].max = max(h.bins[n].max, h.bins[m].max);
                hist_delete_bin(h, m);
                m--; // compare the bin that moved into this place
            }
        }
    }
}

template <bool BY_COUNT> __device__ void hist_sort(r433_histogram &h) // :96-127, same exchange order
{
    if (h.bins_count < 2)
        return;
    for (uint32_t n = 0; n + 1 < h.bins_count; ++n)
        for (uint32_t m = n + 1; m < h.bins_count; ++m) {
            bool const less = BY_COUNT ? h.bins[m].count < h.bins[n].count : h.bins[m].mean < h.bins[n].mean;
            if (less) {
                r433_hist_bin const t = h.bins[m];
                h.bins[m] = h.bins[n];
                h.bins[n] = t;
            }
        }
}

// the guess, :349-430; P and G are scratch copies that get sorted
__device__ void guess_modulation(r433_analysis &a, r433_histogram &P, r433_histogram &G, uint32_t type, uint32_t sample_rate)
{
    double const to_us = 1e6 / sample_rate;
    hist_sort<false>(P);
    hist_sort<false>(G);
    if (P.bins[0].mean == 0)
        hist_delete_bin(P, 0); // FSK initial zero-bin
    r433_dev_timing d;
    d.modulation = 0;
    d.short_width = d.long_width = d.reset_limit = d.gap_limit = d.sync_width = d.tolerance = 0.0f;
    d.priority = 0;
    bool const fsk = type == R433_PKG_FSK;
    uint32_t const np = P.bins_count, ng = G.bins_count;
    int const big = ng ? G.bins[ng - 1].max + 1 : 1; // above the biggest gap
    uint32_t guess;
    if (a.num_pulses == 1) {
        guess = R433_GUESS_SINGLE_PULSE;
    }
    else if (np == 1 && ng == 1) {
        guess = R433_GUESS_UNMODULATED;
    }
    else if (np == 1 && ng > 1) {
        guess = R433_GUESS_PPM;
        d.modulation = 5; // OOK_PULSE_PPM, include/r_device.h:28
        d.short_width = (float)(to_us * G.bins[0].mean);
        d.long_width = (float)(to_us * G.bins[1].mean);
        d.gap_limit = (float)(to_us * (G.bins[1].max + 1));
        d.reset_limit = (float)(to_us * big);
    }
    else if ((np == 2 && ng == 1) || (np == 2 && ng == 2 && a.periods_pg.bins_count == 1)) {
        guess = ng == 1 ? R433_GUESS_PWM_FIXED_GAP : R433_GUESS_PWM_FIXED_PERIOD;
        d.modulation = fsk ? 17 : 6; // FSK_PULSE_PWM / OOK_PULSE_PWM
        d.short_width = (float)(to_us * P.bins[0].mean);
        d.long_width = (float)(to_us * P.bins[1].mean);
        d.tolerance = (float)((d.long_width - d.short_width) * 0.4);
        d.reset_limit = (float)(to_us * big);
    }
    else if (np == 2 && ng == 2 && a.periods_pg.bins_count == 3) {
        guess = R433_GUESS_MANCHESTER;
        d.modulation = fsk ? 18 : 3; // FSK_PULSE_MANCHESTER_ZEROBIT / OOK_PULSE_MANCHESTER_ZEROBIT
        d.short_width = (float)(to_us * min(P.bins[0].mean, P.bins[1].mean));
        d.long_width = 0.0f;
        d.reset_limit = (float)(to_us * big);
    }
    else if (np == 2 && ng >= 3) {
        guess = R433_GUESS_PWM_MULTI;
        d.modulation = fsk ? 17 : 6;
        d.short_width = (float)(to_us * P.bins[0].mean);
        d.long_width = (float)(to_us * P.bins[1].mean);
        d.gap_limit = (float)(to_us * (G.bins[1].max + 1));
        d.tolerance = (float)((d.long_width - d.short_width) * 0.4);
        d.reset_limit = (float)(to_us * big);
    }
    else if (np >= 3 && ng >= 3 && abs(P.bins[1].mean - 2 * P.bins[0].mean) <= P.bins[0].mean / 8
            && abs(P.bins[2].mean - 3 * P.bins[0].mean) <= P.bins[0].mean / 8 && abs(G.bins[0].mean - P.bins[0].mean) <= P.bins[0].mean / 8
            && abs(G.bins[1].mean - 2 * P.bins[0].mean) <= P.bins[0].mean / 8 && abs(G.bins[2].mean - 3 * P.bins[0].mean) <= P.bins[0].mean / 8) {
        guess = R433_GUESS_NRZ;
        d.modulation = fsk ? 16 : 4; // FSK_PULSE_PCM / OOK_PULSE_PCM
        d.short_width = (float)(to_us * P.bins[0].mean);
        d.long_width = (float)(to_us * P.bins[0].mean);
        d.reset_limit = (float)(to_us * P.bins[0].mean * 1024);
    }
    else if (np == 3) {
        guess = R433_GUESS_PWM_SYNC;
        hist_sort<true>(P); // lowest count first: probably the delimiter
        int const p1 = P.bins[1].mean, p2 = P.bins[2].mean;
        d.modulation = fsk ? 17 : 6;
        d.short_width = (float)(to_us * (p1 < p2 ? p1 : p2));
        d.long_width = (float)(to_us * (p1 < p2 ? p2 : p1));
        d.sync_width = (float)(to_us * P.bins[0].mean);
        d.reset_limit = (float)(to_us * big);
    }
    else {
        guess = R433_GUESS_NO_CLUE;
    }
    a.guess = guess;
    a.device = d;
}

__global__ __launch_bounds__(64) void k_analyze(uint8_t const *arena, uint32_t arena_stride, uint32_t const *dir_stream,
        uint32_t const *dir_off, uint32_t n_pkgs, r433_analysis *out)
{
    __shared__ r433_analysis A;
    __shared__ r433_histogram sP, sG;
    uint32_t const lane = threadIdx.x;
    for (uint32_t pkg = blockIdx.x; pkg < n_pkgs; pkg += gridDim.x) {
        uint8_t const *rec = arena + (uint64_t)dir_stream[pkg] * arena_stride + dir_off[pkg];
        uint32_t const type = ((uint32_t const *)rec)[2];
        uint32_t const num = min(((uint32_t const *)rec)[3], (uint32_t)R433_PD_MAX_PULSES);
        uint32_t const rate = ((uint32_t const *)rec)[14];
        int2 const *pairs = (int2 const *)(rec + sizeof(r433_pkg_rec));
        __syncthreads(); // the previous package has left LDS
        LaneBin hp, hg, hpg, hgp, ht;
        uint32_t cp = 0, cg = 0, cpg = 0, cgp = 0, ct = 0;
        int total = 0, prev_gap = 0;
        for (uint32_t n = 0; n < num; ++n) { // src/pulse_analyzer.c:291-318, one walk for the four per-pulse histograms
            int2 const pg = pairs[n];
            hist_add(hp, cp, pg.x, lane);
            if (n + 1 < num) { // the last gap (end of package) is left out of these two
                hist_add(hg, cg, pg.y, lane);
                hist_add(hpg, cpg, pg.x + pg.y, lane);
            }
            hist_add(hgp, cgp, n ? pg.x + prev_gap : pg.x, lane);
            hist_add(ht, ct, pg.x, lane);
            total += pg.x + pg.y;
            prev_gap = pg.y;
        }
        for (uint32_t n = 0; n < num; ++n) // timings: all pulses first, then all gaps (:317-318)
            hist_add(ht, ct, pairs[n].y, lane);
        if (num)
            total -= prev_gap;
        hist_store(A.pulses, hp, cp, lane);
        hist_store(A.gaps, hg, cg, lane);
        hist_store(A.periods_pg, hpg, cpg, lane);
        hist_store(A.periods_gp, hgp, cgp, lane);
        hist_store(A.timings, ht, ct, lane);
        __syncthreads();
        if (lane == 0) {
            A.num_pulses = num;
            A.total_period = total;
            A.reserved = 0;
            hist_fuse(A.pulses); // :321-324
            hist_fuse(A.gaps);
            hist_fuse(A.periods_pg);
            hist_fuse(A.timings);
            sP = A.pulses;
            sG = A.gaps;
            if (num)
                guess_modulation(A, sP, sG, type, rate);
            else {
                A.guess = R433_GUESS_NO_PULSES;
                A.device = r433_dev_timing{0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0};
            }
        }
        __syncthreads();
        uint32_t const *src = (uint32_t const *)&A;
        uint32_t *dst = (uint32_t *)&out[pkg];
        for (uint32_t w = lane; w < sizeof(r433_analysis) / 4; w += 64)
            dst[w] = src[w];
    }
}

} // namespace

void launch_analyze(uint8_t const *arena, uint32_t arena_stride, uint32_t const *dir_stream, uint32_t const *dir_off, uint32_t n_pkgs,
        r433_analysis *out, hipStream_t st)
{
    uint32_t const blocks = n_pkgs < 1 ? 1u : (n_pkgs < 8192u ? n_pkgs : 8192u);
    hipLaunchKernelGGL(k_analyze, dim3(blocks), dim3(64), 0, st, arena, arena_stride, dir_stream, dir_off, n_pkgs, out);
}

} // namespace r433
