// cusim.cpp — block/thread scheduler of the CPU SIMT emulator (see include/cusim.h).  TEST INFRASTRUCTURE ONLY.
#include "cusim.h"
#include <ucontext.h>
#include <vector>

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false, waiting = false;
    uint3 tid{};
};
constexpr size_t kStack = 256 << 10;
std::vector<Fiber> g_fibers;
ucontext_t g_sched;
Fiber* g_cur = nullptr;
void (*g_entry)(void*) = nullptr;
void* g_arg = nullptr;

void trampoline()
{
    g_entry(g_arg);
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_sched);
}
}  // namespace

void __syncthreads()
{
    if (!g_cur) { fprintf(stderr, "cusim: __syncthreads() in a launch compiled without fibers\n"); abort(); }
    g_cur->waiting = true;
    swapcontext(&g_cur->ctx, &g_sched);
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
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, trampoline, 0);
        }
        unsigned live = nt;
        while (live) {
            unsigned waiting = 0;
            for (i = 0; i < nt; i++) {
                Fiber& f = g_fibers[i];
                if (f.done) continue;
                if (f.waiting) { waiting++; continue; }
                g_cur = &f; threadIdx = f.tid;
                swapcontext(&g_sched, &f.ctx);
                if (f.done) live--; else waiting++;
            }
            // every live thread of the block has arrived at the barrier (exited threads do not take part)
            if (live && waiting == live)
                for (i = 0; i < nt; i++) g_fibers[i].waiting = false;
        }
        g_cur = nullptr;
    }
}
