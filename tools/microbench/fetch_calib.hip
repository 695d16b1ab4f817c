// fetch_calib.hip -- calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on this part.
// Streams a buffer of known size through (a) 4 B/lane loads, (b) 16 B/lane loads, and writes a buffer of known size with 4 B/lane and
// 16 B/lane stores; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) the per-kernel counter values
// divided by the known bytes give the correction factors that profiles/*_icp_traffic.json applies.  The buffer (256 MiB by
// default) is larger than the 256 MB Infinity Cache slice a single pass can keep, and every launch reads a DIFFERENT half of a 2x
// allocation, so the reads come from HBM.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ;  ./fetch_calib [MiB] [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) read4_kernel(const float* __restrict__ p, size_t n, float* __restrict__ out)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) read16_kernel(const float4* __restrict__ p, size_t n4, float* __restrict__ out)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) write4_kernel(float* __restrict__ p, size_t n, float v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ void __launch_bounds__(256) write16_kernel(float4* __restrict__ p, size_t n4, float v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 256;
    const int launches = argc > 2 ? atoi(argv[2]) : 6;
    const size_t bytes = mib << 20, n = bytes / 4;
    float *buf, *out;
    CK(hipMalloc(&buf, 2 * bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, 2 * bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 16;
    for (int kind = 0; kind < 4; kind++) {
        float ms_sum = 0;
        for (int l = 0; l < launches; l++) {
            float* p = buf + (size_t)(l & 1) * n;   // alternate halves: the previous launch's half is what the caches hold
            CK(hipEventRecord(e0));
            if (kind == 0) read4_kernel<<<grid, 256>>>(p, n, out);
            else if (kind == 1) read16_kernel<<<grid, 256>>>((const float4*)p, n / 4, out);
            else if (kind == 2) write4_kernel<<<grid, 256>>>(p, n, 1.0f);
            else write16_kernel<<<grid, 256>>>((float4*)p, n / 4, 2.0f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (l > 0) ms_sum += ms;
        }
        const char* names[4] = {"read4_kernel", "read16_kernel", "write4_kernel", "write16_kernel"};
        printf("%s: %zu bytes per launch, %.1f us, %.0f GB/s\n", names[kind], bytes, 1e3 * ms_sum / (launches - 1), bytes / (ms_sum / (launches - 1) * 1e-3) / 1e9);
    }
    return 0;
}
