// hw_queues.hip -- how many of a process's streams really run side by side: a long kernel on stream 0, then a short one on each of
// streams 1..N-1; a short kernel that ends before the long one ran beside it, one that ends after it shared its hardware queue.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/hw_queues.hip -o /tmp/hw_queues && /tmp/hw_queues [streams]
//   GPU_MAX_HW_QUEUES=8 /tmp/hw_queues                      (the runtime's default is 4)
//   hipcc ... -DWITH_LIB -Iinclude -Lrtl_433_amd/lib -lrtl433hip -Wl,-rpath,... : the same with librtl433hip.so loaded first
//   (its constructor asks for eight queues: rtl_433_amd/csrc/host_api.cpp)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef WITH_LIB
#include "r433_hip.h"
#endif

__global__ void k_spin(long long ticks, int *out)
{
    long long const t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
    if (out)
        *out = 1;
}

int main(int argc, char **argv)
{
#ifdef WITH_LIB
    char const *e0 = r433_last_error(); // (a reference into the library: it is loaded, its constructor has run)
    (void)e0;
#endif
    int const n = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<hipStream_t> st(n);
    for (auto &s : st)
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
            return 1;
    for (auto &s : st) // warm every stream (queues are made lazily)
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 1000LL, (int *)nullptr);
    (void)hipDeviceSynchronize();
    long long const long_ticks = 100000000LL * 30 / 1000; // 30 ms at the 100 MHz wall clock
    int beside = 0;
    printf("GPU_MAX_HW_QUEUES=%s, %d streams: ", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)", n);
    for (int i = 1; i < n; ++i) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[0], long_ticks, (int *)nullptr);
        auto const t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], 1000LL, (int *)nullptr);
        (void)hipStreamSynchronize(st[i]);
        double const ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        (void)hipDeviceSynchronize();
        printf("%d:%s ", i, ms < 15.0 ? "beside" : "BEHIND");
        beside += ms < 15.0;
    }
    printf("| %d of %d streams ran beside stream 0\n", beside, n - 1);
    return 0;
}
