// cusim.cpp — block/thread scheduler of the CPU SIMT emulator (see include/cusim.h).  TEST INFRASTRUCTURE ONLY.
//
// Threads of a block are fibers when the translation unit uses __syncthreads().  The reference's reductions (reduce.cu:90-185) pass a
// barrier a few hundred times per thread, so a tracked frame is ~10^8 fiber switches: on x86-64 the switch is a dozen instructions of
// our own (callee-saved registers + stack pointer; ucontext's swapcontext makes a sigprocmask system call per switch and was 98 % of
// the emulator's run time).  Other targets, and -DCUSIM_UCONTEXT, keep <ucontext.h>.
#include "cusim.h"
#include <vector>

#if defined(__x86_64__) && !defined(CUSIM_UCONTEXT)
#define CUSIM_ASM_SWITCH 1
extern "C" void cusim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.hidden cusim_switch
.globl cusim_switch
.type cusim_switch,@function
cusim_switch:
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
.size cusim_switch,.-cusim_switch
)");
#else
#include <ucontext.h>
#endif

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
struct Fiber {
#ifdef CUSIM_ASM_SWITCH
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr;
    bool done = false, waiting = false;
    uint3 tid{};
};
constexpr size_t kStack = 256 << 10;
std::vector<Fiber> g_fibers;
#ifdef CUSIM_ASM_SWITCH
void* g_sched_sp = nullptr;
#else
ucontext_t g_sched;
#endif
Fiber* g_cur = nullptr;
void (*g_entry)(void*) = nullptr;
void* g_arg = nullptr;

inline void to_scheduler(Fiber* f)
{
#ifdef CUSIM_ASM_SWITCH
    cusim_switch(&f->sp, g_sched_sp);
#else
    swapcontext(&f->ctx, &g_sched);
#endif
}
inline void to_fiber(Fiber* f)
{
#ifdef CUSIM_ASM_SWITCH
    cusim_switch(&g_sched_sp, f->sp);
#else
    swapcontext(&g_sched, &f->ctx);
#endif
}

void trampoline()
{
    g_entry(g_arg);
    g_cur->done = true;
    to_scheduler(g_cur);
    abort();  // a finished fiber is never resumed
}

void arm(Fiber& f)
{
#ifdef CUSIM_ASM_SWITCH
    // the frame cusim_switch pops: r15 r14 r13 r12 rbx rbp, then `ret` into trampoline with rsp = 8 (mod 16) as after a call
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8 * 8);
    for (int i = 0; i < 6; i++) sp[i] = nullptr;
    sp[6] = (void*)&trampoline;
    sp[7] = nullptr;
    f.sp = sp;
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, trampoline, 0);
#endif
}
}  // namespace

void __syncthreads()
{
    if (!g_cur) { fprintf(stderr, "cusim: __syncthreads() in a launch compiled without fibers\n"); abort(); }
    g_cur->waiting = true;
    to_scheduler(g_cur);
}

void cusim::run_grid(dim3 grid, dim3 block, bool fibers, void (*entry)(void*), void* arg)
{
    gridDim = grid; blockDim = block;
    const unsigned nt = block.x * block.y * block.z;
    if (fibers && g_fibers.size() < nt) {
        size_t old = g_fibers.size();
        g_fibers.resize(nt);
        for (size_t i = old; i < nt; i++) g_fibers[i].stack = (char*)malloc(kStack);
    }
    g_entry = entry; g_arg = arg;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        blockIdx = uint3{bx, by, bz};
        if (!fibers) {
            g_cur = nullptr;
            for (unsigned tz = 0; tz < block.z; tz++)
            for (unsigned ty = 0; ty < block.y; ty++)
            for (unsigned tx = 0; tx < block.x; tx++) { threadIdx = uint3{tx, ty, tz}; entry(arg); }
            continue;
        }
        unsigned i = 0;
        for (unsigned tz = 0; tz < block.z; tz++)
        for (unsigned ty = 0; ty < block.y; ty++)
        for (unsigned tx = 0; tx < block.x; tx++, i++) {
            Fiber& f = g_fibers[i];
            f.done = f.waiting = false; f.tid = uint3{tx, ty, tz};
            arm(f);
        }
        unsigned live = nt;
        while (live) {
            unsigned waiting = 0;
            for (i = 0; i < nt; i++) {
                Fiber& f = g_fibers[i];
                if (f.done) continue;
                if (f.waiting) { waiting++; continue; }
                g_cur = &f; threadIdx = f.tid;
                to_fiber(&f);
                if (f.done) live--; else waiting++;
            }
            // every live thread of the block has arrived at the barrier (exited threads do not take part)
            if (live && waiting == live)
                for (i = 0; i < nt; i++) g_fibers[i].waiting = false;
        }
        g_cur = nullptr;
    }
}
