// Microbenchmark: cost per sample of the pulse-arm averages (k_wave phase C) for a lone wavefront.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ema tools/ubench/ema.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v2s __attribute__((vector_size(4)));
__device__ __forceinline__ v2s as_v2s(int w) { v2s r; __builtin_memcpy(&r, &w, 4); return r; }
__device__ __forceinline__ int as_int(v2s w) { int r; __builtin_memcpy(&r, &w, 4); return r; }
__device__ __forceinline__ v2s pk_max(v2s a, v2s b) { v2s const m = a > b; return (a & m) | (b & ~m); }
#define N 512
template <int MODE> __global__ __launch_bounds__(64) void k_ema(int *out, int seed)
{
    int in_l = (seed * (threadIdx.x + 1)) & 0x00ff00ff;
    v2s hv = as_v2s(__builtin_amdgcn_readfirstlane(seed) & 0x0fff0fff);
    v2s const floor_v = {100, -32768};
    v2s const m63 = {63, 63};
    int h = seed & 0xfff, f = (seed >> 12) & 0xfff;
    long long t0 = clock64();
    for (int it = 0; it < N; ++it) {
        int const base = __builtin_amdgcn_readfirstlane(it & 7) * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { // general packed
                v2s const in = as_v2s(__builtin_amdgcn_readlane(in_l, base + u));
                v2s const q = (hv + ((hv >> 15) & m63)) >> 6;
                hv = pk_max(hv - q + in, floor_v);
            }
            else if (MODE == 1) { // plain packed
                v2s const in = as_v2s(__builtin_amdgcn_readlane(in_l, base + u));
                hv = hv + (in - (hv >> 6));
            }
            else if (MODE == 2) { // plain packed, constant lane
                v2s const in = as_v2s(__builtin_amdgcn_readlane(in_l, u));
                hv = hv + (in - (hv >> 6));
            }
            else if (MODE == 3) { // scalar unit
                int const in = __builtin_amdgcn_readlane(in_l, base + u);
                h = h - (h >> 6) + (in & 0xffff);
                f = f - (f >> 6) + (in >> 16);
            }
            else if (MODE == 4) { // plain packed, inputs pre-broadcast to SGPRs by one v_readlane burst
                v2s const in = as_v2s(__builtin_amdgcn_readlane(in_l, base + u));
                hv = hv - (hv >> 6) + in;
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = as_int(hv) + h + f;
    if (threadIdx.x == 0) out[gridDim.x * 64 + blockIdx.x] = (int)(t1 - t0);
}
template <typename K> void run(char const *name, K k, int blocks)
{
    int *d;
    (void)hipMalloc(&d, (blocks * 64 + blocks) * sizeof(int));
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 12345);
    (void)hipDeviceSynchronize();
    std::vector<int> h(blocks * 64 + blocks);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += h[blocks * 64 + b];
    printf("%-28s blocks=%5d  ticks/sample=%7.2f\n", name, blocks, s / blocks / (N * 8));
    (void)hipFree(d);
}
int main()
{
    for (int blocks : {1024, 2048}) {
        run("general packed (7 dep)", k_ema<0>, blocks);
        run("plain packed (3 dep)", k_ema<1>, blocks);
        run("plain packed const lane", k_ema<2>, blocks);
        run("scalar unit", k_ema<3>, blocks);
        run("plain packed sub-first", k_ema<4>, blocks);
    }
    return 0;
}
