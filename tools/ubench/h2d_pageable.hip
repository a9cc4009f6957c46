// What a cold process pays before its first kernel has run, and what a host -> device copy costs from pageable against pinned
// memory (the drop-in CLI's staging: is pinning 256 MiB at 0.2 ms per MiB worth it for bytes that cross once?)
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/h2d_pageable tools/ubench/h2d_pageable.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(unsigned char *p) { p[threadIdx.x] += 1; }
int main() {
    double t0 = now();
    int n = 0;
    CK(hipGetDeviceCount(&n));
    double t1 = now();
    void *d;
    size_t const N = 256u << 20;
    CK(hipMalloc(&d, N));
    double t2 = now();
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, 0, (unsigned char *)d);
    CK(hipDeviceSynchronize());
    double t3 = now();
    printf("hipGetDeviceCount %.1f ms, first hipMalloc %.1f ms, first launch + sync %.1f ms\n", t1 - t0, t2 - t1, t3 - t2);
    void *pg = malloc(N);
    memset(pg, 1, N);
    void *pin;
    double t4 = now();
    CK(hipHostMalloc(&pin, N, hipHostMallocDefault));
    double t5 = now();
    memset(pin, 2, N);
    printf("hipHostMalloc 256 MiB %.1f ms\n", t5 - t4);
    double t6 = now();
    CK(hipHostRegister(pg, N, hipHostRegisterDefault));
    double t7 = now();
    CK(hipHostUnregister(pg));
    printf("hipHostRegister 256 MiB of touched malloc memory %.1f ms\n", t7 - t6);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; rep++) {
        double a = now();
        CK(hipMemcpyAsync(d, pg, N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double b = now();
        CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double c = now();
        printf("H2D 256 MiB: pageable %.1f ms (%.1f GB/s), pinned %.1f ms (%.1f GB/s)\n", b - a, N / (b - a) / 1e6, c - b, N / (c - b) / 1e6);
    }
    // a bounce of our own: 2 x 16 MiB pinned, memcpy into one while the other crosses
    size_t const C = 16u << 20;
    void *bounce[2];
    CK(hipHostMalloc(&bounce[0], C, hipHostMallocDefault));
    CK(hipHostMalloc(&bounce[1], C, hipHostMallocDefault));
    hipEvent_t ev[2];
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    for (int rep = 0; rep < 3; rep++) {
        double a = now();
        for (size_t o = 0, k = 0; o < N; o += C, k++) {
            if (k >= 2)
                CK(hipEventSynchronize(ev[k & 1]));
            memcpy(bounce[k & 1], (char *)pg + o, C);
            CK(hipMemcpyAsync((char *)d + o, bounce[k & 1], C, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ev[k & 1], st));
        }
        CK(hipStreamSynchronize(st));
        double b = now();
        printf("H2D 256 MiB through a 2 x 16 MiB pinned bounce (one thread): %.1f ms (%.1f GB/s)\n", b - a, N / (b - a) / 1e6);
    }
    return 0;
}
