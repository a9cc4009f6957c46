// selftest_kernels.hip -- TEST INFRASTRUCTURE: kernels that exercise the emulator itself (tests/emu), not the product.
// A workgroup of two wavefronts running different code: wavefront 0 produces a tile into LDS with a varying number of
// cross-lane primitives per step, wavefront 1 consumes the previous tile; one __syncthreads() per step.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void k_emu_two_waves(int *out, int steps)
{
    __shared__ int buf[2][64];
    int const wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    int acc = 0;
    for (int t = 0; t <= steps; ++t) {
        if (wave == 0 && t < steps) { // producer: t % 3 + 1 shuffles before the store
            int v = lane * 3 + t;
            for (int k = 0; k <= t % 3; ++k)
                v += __shfl_xor(v, 1 << k, 64);
            buf[t & 1][lane] = v;
        }
        if (wave == 1 && t > 0) { // consumer of step t - 1: a ballot and a variable number of broadcasts
            int const v = buf[(t - 1) & 1][lane];
            unsigned long long const m = __ballot(v & 1);
            acc += v + (int)__popcll(m);
            for (int k = 0; k < (t & 1); ++k)
                acc += __builtin_amdgcn_readlane(v, 7);
        }
        __syncthreads();
    }
    if (wave == 1)
        out[lane] = acc;
}

extern "C" int r433emu_selftest_two_waves(int *host_out, int steps)
{
    int *d = nullptr;
    if (hipMalloc((void **)&d, 64 * sizeof(int)) != hipSuccess)
        return -1;
    hipLaunchKernelGGL(k_emu_two_waves, dim3(2), dim3(128), 0, 0, d, steps);
    (void)hipMemcpy(host_out, d, 64 * sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return 0;
}
