// Microbenchmark: forms of the eight-sample group of the packed averages x += in - (x >> 6), one wavefront per SIMD and two.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ema2 tools/ubench/ema2.hip   (binary is git-ignored)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 512
#define STEP3(S) "v_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\tv_pk_sub_i16 %[x], %[x], %[q]\n\tv_pk_add_u16 %[x], %[x], " S "\n\t"
#define STEP2(S) "v_pk_add_u16 %[a], %[x], " S "\n\tv_pk_ashrrev_i16 %[q], 6, %[x] op_sel_hi:[0,1]\n\tv_pk_sub_i16 %[x], %[a], %[q]\n\t"
#define STEP32(S) "v_add_u32 %[a], " S ", %[x]\n\tv_pk_lshrrev_b16 %[q], 6, %[x] op_sel_hi:[0,1]\n\tv_sub_u32 %[x], %[a], %[q]\n\t"
#define RL(S, L) "v_readlane_b32 " S ", %[rot], " #L "\n\t"
template <int MODE> __global__ __launch_bounds__(64) void k(int *out, int seed)
{
    int rot = (seed * (threadIdx.x + 1)) & 0x00ff00ff;
    int x = __builtin_amdgcn_readfirstlane(seed) & 0x0fff0fff, q, a, s0, s1;
    long long t0 = clock64();
    for (int it = 0; it < N; ++it) {
        if (MODE == 0) // three deep, software-pipelined readlanes (the product's form)
            asm volatile(RL("%[s0]", 0) RL("%[s1]", 1) STEP3("%[s0]") RL("%[s0]", 2) STEP3("%[s1]") RL("%[s1]", 3) STEP3("%[s0]") RL("%[s0]", 4)
                         STEP3("%[s1]") RL("%[s1]", 5) STEP3("%[s0]") RL("%[s0]", 6) STEP3("%[s1]") RL("%[s1]", 7) STEP3("%[s0]") STEP3("%[s1]")
                         : [x] "+v"(x), [q] "=&v"(q), [a] "=&v"(a), [s0] "=&s"(s0), [s1] "=&s"(s1) : [rot] "v"(rot));
        else if (MODE == 1) // two deep: x + in and x >> 6 side by side
            asm volatile(RL("%[s0]", 0) RL("%[s1]", 1) STEP2("%[s0]") RL("%[s0]", 2) STEP2("%[s1]") RL("%[s1]", 3) STEP2("%[s0]") RL("%[s0]", 4)
                         STEP2("%[s1]") RL("%[s1]", 5) STEP2("%[s0]") RL("%[s0]", 6) STEP2("%[s1]") RL("%[s1]", 7) STEP2("%[s0]") STEP2("%[s1]")
                         : [x] "+v"(x), [q] "=&v"(q), [a] "=&v"(a), [s0] "=&s"(s0), [s1] "=&s"(s1) : [rot] "v"(rot));
        else if (MODE == 2) // two deep, 32-bit add / sub around one packed shift
            asm volatile(RL("%[s0]", 0) RL("%[s1]", 1) STEP32("%[s0]") RL("%[s0]", 2) STEP32("%[s1]") RL("%[s1]", 3) STEP32("%[s0]") RL("%[s0]", 4)
                         STEP32("%[s1]") RL("%[s1]", 5) STEP32("%[s0]") RL("%[s0]", 6) STEP32("%[s1]") RL("%[s1]", 7) STEP32("%[s0]") STEP32("%[s1]")
                         : [x] "+v"(x), [q] "=&v"(q), [a] "=&v"(a), [s0] "=&s"(s0), [s1] "=&s"(s1) : [rot] "v"(rot));
        else if (MODE == 3) { // three deep, inputs already in SGPRs (as a scalar load would leave them)
            s0 = seed & 0xff00ff, s1 = (seed >> 1) & 0xff00ff;
            asm volatile(STEP3("%[s0]") STEP3("%[s1]") STEP3("%[s0]") STEP3("%[s1]") STEP3("%[s0]") STEP3("%[s1]") STEP3("%[s0]") STEP3("%[s1]")
                         : [x] "+v"(x), [q] "=&v"(q), [a] "=&v"(a) : [s0] "s"(s0), [s1] "s"(s1));
        }
        else if (MODE == 4) { // two deep, inputs already in SGPRs
            s0 = seed & 0xff00ff, s1 = (seed >> 1) & 0xff00ff;
            asm volatile(STEP2("%[s0]") STEP2("%[s1]") STEP2("%[s0]") STEP2("%[s1]") STEP2("%[s0]") STEP2("%[s1]") STEP2("%[s0]") STEP2("%[s1]")
                         : [x] "+v"(x), [q] "=&v"(q), [a] "=&v"(a) : [s0] "s"(s0), [s1] "s"(s1));
        }
        else if (MODE == 5) { // two deep 32-bit, inputs already in SGPRs
            s0 = seed & 0xff00ff, s1 = (seed >> 1) & 0xff00ff;
            asm volatile(STEP32("%[s0]") STEP32("%[s1]") STEP32("%[s0]") STEP32("%[s1]") STEP32("%[s0]") STEP32("%[s1]") STEP32("%[s0]") STEP32("%[s1]")
                         : [x] "+v"(x), [q] "=&v"(q), [a] "=&v"(a) : [s0] "s"(s0), [s1] "s"(s1));
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) out[gridDim.x * 64 + blockIdx.x] = (int)(t1 - t0);
}
template <typename K> void run(char const *name, K kk, int blocks)
{
    int *d;
    (void)hipMalloc(&d, (blocks * 64 + blocks) * sizeof(int));
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL(kk, dim3(blocks), dim3(64), 0, 0, d, 12345);
    (void)hipDeviceSynchronize();
    std::vector<int> h(blocks * 64 + blocks);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += h[blocks * 64 + b];
    printf("%-44s blocks=%5d  ticks/sample=%7.2f  x=%08x\n", name, blocks, s / blocks / (N * 8), h[0]);
    (void)hipFree(d);
}
int main()
{
    for (int blocks : {1, 1024, 2048}) {
        run("3 deep + readlane (product)", k<0>, blocks);
        run("2 deep + readlane", k<1>, blocks);
        run("2 deep 32-bit + readlane", k<2>, blocks);
        run("3 deep, inputs in SGPRs", k<3>, blocks);
        run("2 deep, inputs in SGPRs", k<4>, blocks);
        run("2 deep 32-bit, inputs in SGPRs", k<5>, blocks);
    }
    return 0;
}
