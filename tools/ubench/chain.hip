// Microbenchmark: what does one wavefront alone on a SIMD pay per dependent instruction?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/chain tools/ubench/chain.hip   (binary is git-ignored)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 4096
__global__ __launch_bounds__(64) void k_valu_dep(int *out, int seed)
{
    int v = seed + threadIdx.x;
    long long t0 = clock64();
#pragma unroll 64
    for (int i = 0; i < N; ++i)
        v = (v >> 6) + v + i;            // 2 dependent VALU (shift, add3?) per iteration
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = v;
    if (threadIdx.x == 0) out[gridDim.x * 64 + blockIdx.x] = (int)(t1 - t0);
}
__global__ __launch_bounds__(64) void k_salu_dep(int *out, int seed)
{
    int v = __builtin_amdgcn_readfirstlane(seed);
    long long t0 = clock64();
#pragma unroll 64
    for (int i = 0; i < N; ++i)
        v = (v >> 6) + v + i;
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = v;
    if (threadIdx.x == 0) out[gridDim.x * 64 + blockIdx.x] = (int)(t1 - t0);
}
__global__ __launch_bounds__(64) void k_readlane_salu(int *out, int seed)
{
    int x = seed * (threadIdx.x + 1);
    int v = __builtin_amdgcn_readfirstlane(seed);
    long long t0 = clock64();
    for (int i = 0; i < N / 64; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            int a = __builtin_amdgcn_readlane(x, u);
            v += a > v ? 1 : -1;
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = v;
    if (threadIdx.x == 0) out[gridDim.x * 64 + blockIdx.x] = (int)(t1 - t0);
}
__global__ __launch_bounds__(64) void k_valu_indep(int *out, int seed)
{
    int a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        a = (a >> 6) + a + i;
        b = (b >> 5) + b + i;
        c = (c >> 4) + c + i;
        d = (d >> 3) + d + i;
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) out[gridDim.x * 64 + blockIdx.x] = (int)(t1 - t0);
}

template <typename K> void run(char const *name, K k, int blocks, int insts_per_iter)
{
    int *d;
    hipMalloc(&d, (blocks * 64 + blocks) * sizeof(int));
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 12345);
    hipDeviceSynchronize();
    std::vector<int> h(blocks * 64 + blocks);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += h[blocks * 64 + b];
    printf("%-18s blocks=%5d  ticks/iter=%7.2f  (%d source ops per iter)\n", name, blocks, s / blocks / N, insts_per_iter);
    hipFree(d);
}

int main()
{
    for (int blocks : {256, 1024, 2048, 4096}) {
        run("valu dependent", k_valu_dep, blocks, 3);
        run("salu dependent", k_salu_dep, blocks, 3);
        run("readlane+salu", k_readlane_salu, blocks, 4);
        run("valu 4 chains", k_valu_indep, blocks, 12);
    }
    return 0;
}
