// What device and pinned allocations of the sizes an engine's first pass makes cost on this box (the drop-in CLI's first pass).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/malloc_cost tools/ubench/malloc_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    CK(hipFree(0));
    size_t const sizes[] = {64ull << 20, 256ull << 20, 1ull << 30, 2ull << 30, 6ull << 30, 6ull << 30, 16ull << 30};
    for (size_t s : sizes) {
        void *p;
        double t0 = now();
        CK(hipMalloc(&p, s));
        double t1 = now();
        CK(hipMemset(p, 0, 4096));
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipFree(p));
        double t3 = now();
        printf("hipMalloc %6zu MiB: %8.2f ms, first touch %6.2f ms, hipFree %8.2f ms\n", s >> 20, t1 - t0, t2 - t1, t3 - t2);
    }
    for (size_t s : {16ull << 20, 64ull << 20, 256ull << 20}) {
        void *p;
        double t0 = now();
        CK(hipHostMalloc(&p, s, hipHostMallocDefault));
        double t1 = now();
        CK(hipHostFree(p));
        printf("hipHostMalloc %4zu MiB: %8.2f ms, hipHostFree %6.2f ms\n", s >> 20, t1 - t0, now() - t1);
    }
    return 0;
}
