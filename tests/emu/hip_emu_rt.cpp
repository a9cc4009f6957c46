// hip_emu_rt.cpp -- TEST INFRASTRUCTURE: fiber scheduler + runtime stubs behind tests/emu/include/hip/hip_runtime.h.
//
// One workgroup at a time; each HIP thread is a fiber with its own small stack, switched by a
// 20-instruction x86-64 context switch.  A rendezvous parks the calling fiber; the scheduler resumes
// fibers round-robin, so after one full round every live fiber has reached the same rendezvous and
// the values deposited in the (double-buffered) slot array can be read by all of them.
#include <map>
#include <mutex>
#include <vector>
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <time.h>

#include <vector>

#if !defined(__x86_64__)
#error "the wave emulator's context switch is x86-64 only"
#endif

extern "C" void r433emu_switch(void **from_sp, void *to_sp);
asm(R"(
.text
.globl r433emu_switch
.type r433emu_switch,@function
r433emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size r433emu_switch,.-r433emu_switch
)");

namespace emu {

constexpr size_t kStack = 256 * 1024;

struct Fiber {
    void *sp = nullptr;
    uint8_t *stack = nullptr;
    unsigned tid = 0;
    bool done = false;
    unsigned seq = 0; // rendezvous count
};

struct Block {
    unsigned n = 0, bid = 0, nblocks = 0;
    std::vector<Fiber> fibers;
    std::vector<uint8_t> slots[2];
    std::vector<uint8_t> live[2];
    std::vector<unsigned> tags[2];
    std::function<void()> const *body = nullptr;
    void *sched_sp = nullptr;
    Fiber *cur = nullptr;
};

static Block g_blk;
static std::vector<uint8_t *> g_stacks;

Fiber &cur() { return *g_blk.cur; }
Block &blk() { return g_blk; }
unsigned tid() { return g_blk.cur->tid; }
unsigned nthreads() { return g_blk.n; }
unsigned bid() { return g_blk.bid; }
unsigned nblocks() { return g_blk.nblocks; }

static void trampoline()
{
    Fiber *f = g_blk.cur;
    (*g_blk.body)();
    f->done = true;
    r433emu_switch(&f->sp, g_blk.sched_sp);
    abort();
}

uint8_t *rendezvous(void const *val, unsigned bytes, unsigned tag, uint8_t const **live)
{
    Block &B = g_blk;
    Fiber *f = B.cur;
    unsigned par = f->seq & 1u;
    memcpy(B.slots[par].data() + (size_t)f->tid * kSlot, val, bytes);
    B.live[par][f->tid] = 1;
    B.tags[par][f->tid] = tag;
    f->seq += 1;
    r433emu_switch(&f->sp, B.sched_sp);
    // resumed: every live fiber has deposited its value for this rendezvous
    unsigned wave0 = f->tid & ~63u;
    for (unsigned i = wave0; i < wave0 + 64 && i < B.n; ++i)
        if (B.live[par][i] && B.tags[par][i] != tag && !(tag == 1 || B.tags[par][i] == 1)) {
            fprintf(stderr, "hip_emu: divergent cross-lane primitive in block %u (thread %u tag %u vs thread %u tag %u)\n",
                    B.bid, f->tid, tag, i, B.tags[par][i]);
            abort();
        }
    if (live)
        *live = B.live[par].data();
    return B.slots[par].data();
}

bool all_at_barrier()
{
    Block &B = g_blk;
    unsigned const par = (B.cur->seq - 1u) & 1u; // the round this fiber has just come back from
    for (unsigned i = 0; i < B.n; ++i)
        if (!B.fibers[i].done && !(B.live[par][i] && B.tags[par][i] == 1))
            return false;
    return true;
}

static uint8_t *get_stack(unsigned i)
{
    while (g_stacks.size() <= i) {
        void *p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            perror("hip_emu: mmap");
            abort();
        }
        g_stacks.push_back((uint8_t *)p);
    }
    return g_stacks[i];
}

static std::mutex g_launch_lock; // one kernel at a time, whichever host thread (engines on several pretend devices launch side by side)

void launch(dim3 grid, dim3 block, std::function<void()> const &body)
{
    std::lock_guard<std::mutex> one_at_a_time(g_launch_lock);
    Block &B = g_blk;
    if (B.cur) {
        fprintf(stderr, "hip_emu: nested launch\n");
        abort();
    }
    unsigned const n = block.x * block.y * block.z;
    unsigned const nb = grid.x * grid.y * grid.z;
    B.n = n;
    B.nblocks = nb;
    B.body = &body;
    B.fibers.assign(n, Fiber());
    for (int p = 0; p < 2; ++p) {
        B.slots[p].assign((size_t)n * kSlot, 0);
        B.live[p].assign(n, 0);
        B.tags[p].assign(n, 0);
    }
    for (unsigned b = 0; b < nb; ++b) {
        B.bid = b;
        for (unsigned t = 0; t < n; ++t) {
            Fiber &f = B.fibers[t];
            f.tid = t;
            f.done = false;
            f.seq = 0;
            f.stack = get_stack(t);
            uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
            void **sp = (void **)top;
            *--sp = nullptr;                 // alignment pad
            *--sp = (void *)&trampoline;     // return address of the first switch
            for (int k = 0; k < 6; ++k)
                *--sp = nullptr;             // rbp rbx r12 r13 r14 r15
            f.sp = sp;
        }
        unsigned live_count = n;
        unsigned round = 0;
        while (live_count) {
            // what was deposited two rounds ago is dead now; clear the buffer the coming round fills
            unsigned par = round & 1u;
            std::fill(B.live[par].begin(), B.live[par].end(), 0);
            for (unsigned t = 0; t < n; ++t) {
                Fiber &f = B.fibers[t];
                if (f.done)
                    continue;
                B.cur = &f;
                r433emu_switch(&B.sched_sp, f.sp);
                if (f.done)
                    live_count--;
            }
            round++;
        }
        B.cur = nullptr;
    }
    B.body = nullptr;
}

} // namespace emu

// ---- runtime stubs ----
// Device memory is not zeroed, and a block that was freed comes back with what its last owner left in it: a fresh block
// is poisoned (0xA5), a freed one is kept by size class and handed out again AS IT IS.  (A slot nobody ran must not be
// read: with poison alone a stale detector state looked invalid and hid exactly that.)
namespace {
std::mutex g_heap_lock;
std::map<size_t, std::vector<void *>> g_heap_free; // size class -> blocks
std::map<void *, size_t> g_heap_class;             // live or cached block -> its size class
size_t heap_class(size_t n)
{
    size_t c = 256;
    while (c < n)
        c *= 2;
    return c;
}
} // namespace

hipError_t hipMalloc(void **p, size_t n)
{
    size_t const c = heap_class(n);
    {
        std::lock_guard<std::mutex> g(g_heap_lock);
        auto it = g_heap_free.find(c);
        if (it != g_heap_free.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            return hipSuccess;
        }
    }
    void *q = nullptr;
    if (posix_memalign(&q, 256, c) != 0)
        return hipErrorOutOfMemory;
    memset(q, 0xA5, c);
    {
        std::lock_guard<std::mutex> g(g_heap_lock);
        g_heap_class[q] = c;
    }
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p)
        return hipSuccess;
    std::lock_guard<std::mutex> g(g_heap_lock);
    auto it = g_heap_class.find(p);
    if (it == g_heap_class.end()) {
        free(p);
        return hipSuccess;
    }
    std::vector<void *> &cache = g_heap_free[it->second];
    if (it->second > (size_t(64) << 20) || cache.size() >= 4) { // keep the cache small: the large arenas go back to the system
        g_heap_class.erase(it);
        free(p);
        return hipSuccess;
    }
    cache.push_back(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; } // (the emulator's device reads host memory as it is)
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipMemcpy(void *d, void const *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, void const *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)64 << 30; *t = (size_t)64 << 30; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned) { *st = (hipStream_t)(uintptr_t)1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
static int emu_devices()
{
    char const *e = getenv("R433_EMU_DEVICES");
    int const n = e ? atoi(e) : 1;
    return n < 1 ? 1 : n;
}
static thread_local int t_device = 0;
hipError_t hipGetDeviceCount(int *n) { *n = emu_devices(); return hipSuccess; }
hipError_t hipSetDevice(int device)
{
    if (device < 0 || device >= emu_devices())
        return (hipError_t)101; // hipErrorInvalidDevice
    t_device = device;
    return hipSuccess;
}
hipError_t hipGetDevice(int *device) { *device = t_device; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
char const *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated error"; }
static double now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new EmuEvent{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; } // (launches are synchronous here)
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
