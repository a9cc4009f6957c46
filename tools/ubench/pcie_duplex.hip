// Is the PCIe link of the box full duplex, and for whom?  H2D of 1 GiB and D2H of 200 MiB (the sizes of one bench step), each leg
// either as a DMA copy (hipMemcpyAsync on its own stream) or as a kernel that reads / writes host-mapped pinned memory.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/pcie_duplex tools/ubench/pcie_duplex.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += step) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t n_in = 1ull << 30, n_out = 200ull << 20;
    int blocks_out = argc > 1 ? atoi(argv[1]) : 64;   // workgroups of the writing kernel (it should not need the chip)
    int blocks_in = argc > 2 ? atoi(argv[2]) : 256;
    void *h_in, *h_out, *d_in, *d_out, *hd_in, *hd_out;
    CK(hipHostMalloc(&h_in, n_in, hipHostMallocMapped));
    CK(hipHostMalloc(&h_out, n_out, hipHostMallocMapped));
    memset(h_in, 1, n_in);
    memset(h_out, 2, n_out);
    CK(hipHostGetDevicePointer(&hd_in, h_in, 0));
    CK(hipHostGetDevicePointer(&hd_out, h_out, 0));
    CK(hipMalloc(&d_in, n_in));
    CK(hipMalloc(&d_out, n_out));
    CK(hipMemset(d_out, 3, n_out));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto in_dma = [&] { CK(hipMemcpyAsync(d_in, h_in, n_in, hipMemcpyHostToDevice, s1)); };
    auto out_dma = [&] { CK(hipMemcpyAsync(h_out, d_out, n_out, hipMemcpyDeviceToHost, s2)); };
    auto in_krn = [&] { hipLaunchKernelGGL(k_copy, dim3(blocks_in), dim3(256), 0, s1, (const uint4*)hd_in, (uint4*)d_in, n_in / 16); };
    auto out_krn = [&] { hipLaunchKernelGGL(k_copy, dim3(blocks_out), dim3(256), 0, s2, (const uint4*)d_out, (uint4*)hd_out, n_out / 16); };
    auto out_dma_chunks = [&] { for (size_t o = 0; o < n_out; o += 25ull << 20) CK(hipMemcpyAsync((char*)h_out + o, (char*)d_out + o, 25ull << 20, hipMemcpyDeviceToHost, s2)); };
    struct { const char* name; int a, b; } cases[] = {
        {"h2d dma alone", 0, -1}, {"d2h dma alone", -1, 0}, {"h2d dma + d2h dma", 0, 0},
        {"d2h kernel alone", -1, 1}, {"h2d dma + d2h kernel", 0, 1},
        {"h2d kernel alone", 1, -1}, {"h2d kernel + d2h dma", 1, 0}, {"h2d kernel + d2h kernel", 1, 1},
        {"h2d dma + d2h dma in 8 chunks", 0, 2},
    };
    for (auto& c : cases) {
        double best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            if (c.b == 0) out_dma(); else if (c.b == 1) out_krn(); else if (c.b == 2) out_dma_chunks();
            if (c.a == 0) in_dma(); else if (c.a == 1) in_krn();
            CK(hipDeviceSynchronize());
            double dt = now() - t0;
            if (dt < best) best = dt;
        }
        double bytes = (c.a >= 0 ? n_in : 0) + (c.b >= 0 ? n_out : 0);
        printf("%-34s %7.2f ms  %6.1f GB/s (both directions summed)\n", c.name, best * 1e3, bytes / best / 1e9);
    }
    if (((unsigned char*)h_out)[12345] != 3) { printf("BAD: host buffer not written\n"); return 1; }
    return 0;
}
