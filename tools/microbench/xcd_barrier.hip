// Micro-benchmark: what does a GRID BARRIER cost on MI355X when the participating workgroups all sit on ONE XCD
// (one L2: arrive with an atomic that executes in that L2, poll with loads that bypass the CU's vector L1 but not the L2) -- against the
// same barrier spanning all eight XCDs (device-scope atomics: they go through the fabric to memory), and against a kernel boundary
// (tools/microbench/launch_floor.hip: ~2.5 us dependent launch)?
//
// The question behind it: levels 1-2 of the Gauss-Newton loop are 15 of its 19 iterations on 76 800 / 19 200 pixels; every persistent
// variant so far spanned the chip and paid ~8 us per barrier (csrc/track_reduce.hip, launch_gn_track).  A one-XCD loop would pay one
// L2 round trip per barrier instead -- IF that is well under a launch boundary, and at an eighth of the chip's issue rate.
//
// Variants (K barriers inside one launch, in-kernel wall_clock64 of workgroup 0 and hipEvents around the launch):
//   xcd1/agent   32 workgroups on XCD 0 (grid of 256, blockIdx.x & 7 != 0 leave), atomics + polling at agent (device) scope
//   xcd1/wg      the same workgroups, atomics at WORKGROUP scope (on gfx942/950 they execute in the L2; within one XCD that is coherent),
//                polling with a RETURNING atomic (OR 0), which cannot be answered by the CU's vector L1.  (-DPOLL_WITH_LOADS polls with
//                `global_load ... sc0` instead: that is a group-scope load, the L1 may serve it, and the so3-like kernel below then
//                spins on a stale arrival count -- measured: 16 of 16 workgroups timed out in the second meeting.)
//   chip/agent   256 workgroups over all XCDs, agent scope
// Each barrier is followed by a tiny amount of "work" on data another workgroup wrote before the barrier (a rotating read of a per-
// workgroup word): the result is checked, so a barrier that does not order memory shows as an error count, not as a fast number.
//
// Build: hipcc -O2 --offload-arch=gfx950 xcd_barrier.hip -o xcd_barrier ; run on the GPU box (tools/gpu_micro.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

struct Sync { unsigned arrive; unsigned pad[63]; unsigned words[256]; unsigned errors; unsigned long long ticks; };

template <int SCOPE>  // __HIP_MEMORY_SCOPE_WORKGROUP / _AGENT
__device__ __forceinline__ unsigned load_past_l1(const unsigned* p)
{
    if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v;
#ifdef POLL_WITH_LOADS
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");  // group-scope load: MAY be served by the CU's L1
#else
    // a returning read-modify-write (OR with 0) always executes in the L2: no vector-L1 copy can answer it.  (A load marked sc0 is a
    // GROUP-scope load: on one CU the L1 may serve it -- the so3-like kernel below spun on a stale arrival count with it.)
    const unsigned zero = 0;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
#endif
    return v;
}

template <int SCOPE>
__global__ void barrier_kernel(Sync* s, int K, int members, int xcd_only)
{
    if (xcd_only && (blockIdx.x & 7) != 0) return;  // hardware workgroup b lands on XCD b % 8
    const int me = xcd_only ? (blockIdx.x >> 3) : blockIdx.x;
    unsigned errors = 0;
    bool dead = false;
    __shared__ int s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int k = 1; k <= K; k++) {
        if (threadIdx.x == 0 && !dead) {
            // publish a word, then arrive: the store must be visible (in L2 / memory) before the arrival is
            s->words[me] = (unsigned)(k * 1000 + me);
            if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // write-through L1: the store has reached the L2
            __hip_atomic_fetch_add(&s->arrive, 1u, __ATOMIC_RELAXED, SCOPE);
            const unsigned target = (unsigned)(k * members);
            unsigned spins = 0;
            while (load_past_l1<SCOPE>(&s->arrive) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) { dead = true; break; }   // never hang the GPU: a barrier that does not complete is reported
            }
            if (dead) { atomicAdd(&s->errors, 1000000u); s_dead = 1; }
            if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int other = (me + k) % members;
            const unsigned w = load_past_l1<SCOPE>(&s->words[other]);
            // the neighbour may already have published round k + 1 (it left the barrier first): both values prove round k's store arrived
            if (w != (unsigned)(k * 1000 + other) && w != (unsigned)((k + 1) * 1000 + other)) errors++;
        }
        __syncthreads();
        if (s_dead) break;
    }
    if (threadIdx.x == 0) {
        if (errors) atomicAdd(&s->errors, errors);
        if (me == 0) s->ticks = wall_clock64() - t0;
    }
}

// The shape of csrc/track_reduce.hip's so3_prealign_kernel: grid (8 x 16, models), the workgroups of model y are those with x mod 8 ==
// y mod 8; per iteration 11 conditional u64 atomics into the iteration's slot, wait, arrive, poll, 16 u64 loads; last one out resets.
struct So3Like { unsigned long long acc[10][16]; unsigned arrive, depart; };
__global__ void __launch_bounds__(256) so3_like(So3Like* syncs, unsigned* report)
{
    if ((blockIdx.x & 7) != (blockIdx.y & 7)) return;
    So3Like* sync = syncs + blockIdx.y;
    const int bx = blockIdx.x >> 3;
    const unsigned G = gridDim.x >> 3;
    __shared__ unsigned long long totals[16];
    __shared__ int s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    for (int it = 0; it < 10; it++) {
        if (threadIdx.x < 16) totals[threadIdx.x] = (threadIdx.x < 11) ? (unsigned long long)(bx + 1) * (it + 1) : 0ull;
        __syncthreads();
        if (threadIdx.x < 64) {
            unsigned long long* slot = sync->acc[it];
            if (threadIdx.x < 11 && totals[threadIdx.x] != 0)
                __hip_atomic_fetch_add(&slot[threadIdx.x], totals[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&sync->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned target = (unsigned)(it + 1) * G;
                unsigned spins = 0, seen = 0;
                while ((seen = load_past_l1<__HIP_MEMORY_SCOPE_WORKGROUP>(&sync->arrive)) < target) {
                    if (++spins > (1u << 20)) { s_dead = 1; report[4 + blockIdx.y * 4 + 0] = it; report[4 + blockIdx.y * 4 + 1] = bx; report[4 + blockIdx.y * 4 + 2] = seen; report[4 + blockIdx.y * 4 + 3] = target; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (threadIdx.x < 16) {
                unsigned long long v;
                const unsigned long long zero64 = 0;
                asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&slot[threadIdx.x]), "v"(zero64) : "memory");
                const unsigned long long want = threadIdx.x < 11 ? (unsigned long long)(G * (G + 1) / 2) * (it + 1) : 0ull;
                if (v != want) atomicAdd(&report[0], 1u);
            }
        }
        __syncthreads();
        if (s_dead) { if (threadIdx.x == 0) atomicAdd(&report[1], 1u); break; }
    }
    if (threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(&sync->depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == G - 1) {
            for (int it = 0; it < 10; it++) for (int w = 0; w < 16; w++) sync->acc[it][w] = 0;
            sync->arrive = 0; sync->depart = 0;
        }
    }
}
static int run_so3_like(int models)
{
    So3Like* d; unsigned* rep;
    CK(hipMalloc(&d, sizeof(So3Like) * models)); CK(hipMemset(d, 0, sizeof(So3Like) * models));
    CK(hipMalloc(&rep, 4 * (1 + models) * sizeof(unsigned))); CK(hipMemset(rep, 0, 4 * (1 + models) * sizeof(unsigned)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep_i = 0; rep_i < 4; rep_i++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(so3_like, dim3(128, models), dim3(256), 0, 0, d, rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[4 * 17]; CK(hipMemcpy(h, rep, 4 * (1 + models) * sizeof(unsigned), hipMemcpyDeviceToHost));
        printf("so3-like, %d models, launch %d: %.1f us for 10 meetings of 16 workgroups; wrong totals %u, timed-out workgroups %u", models, rep_i, 1e3 * ms, h[0], h[1]);
        if (h[1]) printf(" (model 0: iteration %u, workgroup %u saw %u of %u)", h[4], h[5], h[6], h[7]);
        printf("\n");
    }
    return 0;
}

template <int SCOPE>
static int run(const char* name, Sync* d, int grid, int members, int xcd_only, int K)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Sync h{};
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(d, 0, sizeof(Sync)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(barrier_kernel<SCOPE>, dim3(grid), dim3(256), 0, 0, d, K, members, xcd_only);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&h, d, sizeof(Sync), hipMemcpyDeviceToHost));
        if (rep == 2)
            printf("%-11s %3d workgroups, %d barriers: %.3f us per barrier (launch %.1f us total, in-kernel clock %.3f us per barrier at 100 MHz), "
                   "ordering errors %u\n", name, members, K, 1e3 * ms / K, 1e3 * ms, (double)h.ticks / 100.0 / K, h.errors);
    }
    return 0;
}

int main()
{
    Sync* d; CK(hipMalloc(&d, sizeof(Sync)));
    const int K = 2000;
    if (run<__HIP_MEMORY_SCOPE_AGENT>("xcd1/agent", d, 256, 32, 1, K)) return 1;
    if (run<__HIP_MEMORY_SCOPE_WORKGROUP>("xcd1/wg", d, 256, 32, 1, K)) return 1;
    if (run<__HIP_MEMORY_SCOPE_AGENT>("chip/agent", d, 256, 256, 0, K)) return 1;
    if (run<__HIP_MEMORY_SCOPE_AGENT>("xcd1/agent8", d, 64, 8, 1, K)) return 1;
    if (run<__HIP_MEMORY_SCOPE_WORKGROUP>("xcd1/wg8", d, 64, 8, 1, K)) return 1;
    if (run_so3_like(1)) return 1;
    if (run_so3_like(5)) return 1;
    return 0;
}
