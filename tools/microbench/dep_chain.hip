// Micro-benchmark: what ONE dependent VALU / SALU instruction of a lone wave costs on this part, and whether that cost depends on what the
// rest of the chip is doing (DESIGN-NOTES R4.5 left "a single busy workgroup on an otherwise idle chip does not run at the clock the
// arithmetic assumes" as an untested guess: the sequential f32 sums of the segmentation and the solve's f64 chains are priced by it).
// Build: hipcc -O2 --offload-arch=gfx950 dep_chain.hip -o dep_chain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz constant clock

constexpr int kRept = 4096, kOuter = 8;   // 32768 dependent instructions per chain

template <int MODE>
__global__ void __launch_bounds__(256) chain_kernel(float* out, unsigned long long* stamps, float seed)
{
    float x = seed + (float)threadIdx.x, y = seed * 0.5f, t = 1.0f / 3.0f;
    double d = (double)seed, e = 1.0 / 3.0;
    unsigned s = (unsigned)blockIdx.x + 7u;
    __syncthreads();
    const unsigned long long t0 = wall();
    for (int o = 0; o < kOuter; o++) {
        if constexpr (MODE == 0) asm volatile(".rept 4096\n\tv_add_f32 %0, %0, %1\n\t.endr" : "+v"(x) : "v"(t));
        if constexpr (MODE == 1) asm volatile(".rept 4096\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr" : "+&v"(x) : "v"(t));
        if constexpr (MODE == 2) asm volatile(".rept 4096\n\tv_fma_f64 %0, %0, %1, %1\n\t.endr" : "+v"(d) : "v"(e));
        if constexpr (MODE == 3) asm volatile(".rept 2048\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\t.endr" : "+v"(x), "+v"(y) : "v"(t));   // two chains
        if constexpr (MODE == 4) asm volatile(".rept 4096\n\ts_add_u32 %0, %0, 3\n\t.endr" : "+s"(s) : : "scc");
        if constexpr (MODE == 5) asm volatile(".rept 4096\n\tv_mul_f32 %0, %0, %1\n\t.endr" : "+v"(x) : "v"(t));
        if constexpr (MODE == 6) asm volatile(".rept 4096\n\tv_add_f64 %0, %0, %1\n\t.endr" : "+v"(d) : "v"(e));
    }
    const unsigned long long t1 = wall();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + (float)d + (float)s;
}

// keeps every CU busy with packed-f32 work for about `iters` x 0.5 us
__global__ void __launch_bounds__(256) heater_kernel(float* out, int iters, volatile int* stop)
{
    float a = (float)threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters && !*stop; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) { a = a * b + c; d = d * b + a; c = c * b + d; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d;
}

template <int MODE>
static int run(const char* name, int threads, int blocks, hipStream_t s, float* out, unsigned long long* st, unsigned long long* hst, int n_instr_per_lane_chain)
{
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(chain_kernel<MODE>, dim3(blocks), dim3(threads), 0, s, out, st, 1.0f);
        CK(hipStreamSynchronize(s));
    }
    CK(hipMemcpy(hst, st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int b = 0; b < blocks; b++) { const double ns = (double)(hst[b * 2 + 1] - hst[b * 2]) * 10.0; if (ns > worst) worst = ns; }
    printf("  %-44s %4d thr x %3d wg : %7.2f ns per dependent instruction\n", name, threads, blocks, worst / n_instr_per_lane_chain);
    return 0;
}

static int suite(const char* title, hipStream_t s, float* out, unsigned long long* st, unsigned long long* hst)
{
    const int N = kRept * kOuter;
    printf("%s\n", title);
    if (run<0>("v_add_f32 (one chain)", 64, 1, s, out, st, hst, N)) return 1;
    if (run<0>("v_add_f32, four waves on the CU", 256, 1, s, out, st, hst, N)) return 1;
    if (run<1>("s_nop 1 + v_add_f32_dpp wave_shr:1", 64, 1, s, out, st, hst, N)) return 1;
    if (run<3>("v_add_f32, two interleaved chains (per pair)", 64, 1, s, out, st, hst, N / 2)) return 1;
    if (run<5>("v_mul_f32", 64, 1, s, out, st, hst, N)) return 1;
    if (run<2>("v_fma_f64", 64, 1, s, out, st, hst, N)) return 1;
    if (run<6>("v_add_f64", 64, 1, s, out, st, hst, N)) return 1;
    if (run<4>("s_add_u32", 64, 1, s, out, st, hst, N)) return 1;
    return 0;
}

int main()
{
    float* out; unsigned long long* st; int* stop;
    CK(hipMalloc(&out, 4 << 20)); CK(hipMalloc(&st, 4096 * 16)); CK(hipMalloc(&stop, 4)); CK(hipMemset(stop, 0, 4));
    std::vector<unsigned long long> hst(4096 * 2);
    hipStream_t s, h; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&h));
    if (suite("[cold: first kernels of the process]", s, out, st, hst.data())) return 1;
    // the chip kept busy on another stream while the chain runs (240 workgroups of 256: one wave per SIMD on most CUs, the chain finds a free one)
    hipLaunchKernelGGL(heater_kernel, dim3(240), dim3(256), 0, h, out + (1 << 19), 400000, stop);
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    if (suite("[a heater kernel on every CU beside the chain]", s, out, st, hst.data())) return 1;
    CK(hipStreamSynchronize(h));
    // right behind a busy period
    hipLaunchKernelGGL(heater_kernel, dim3(2048), dim3(256), 0, s, out + (1 << 19), 40000, stop);
    if (suite("[right behind 0.1 s of a busy chip, same stream]", s, out, st, hst.data())) return 1;
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    if (suite("[after 0.5 s of idle]", s, out, st, hst.data())) return 1;
    return 0;
}
