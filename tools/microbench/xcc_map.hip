// Micro-benchmark: which XCD does hardware workgroup (x, y) of a launch run on?  HW_REG_XCC_ID (id 20) per workgroup, for 1-D and 2-D
// grids, small and LDS-heavy workgroups, with and without most workgroups leaving at once.  The kernels that meet workgroups in one
// XCD's L2 (csrc/track_reduce.hip: so3_prealign_kernel) rely on "linear workgroup id mod 8".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS_WORDS>
__global__ void __launch_bounds__(256) probe(unsigned* out, int spin)
{
    __shared__ unsigned lds[LDS_WORDS];
    lds[threadIdx.x % LDS_WORDS] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
        const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID (id 4): cu / se / sh fields
        out[(blockIdx.y * gridDim.x + blockIdx.x) * 2] = xcc;
        out[(blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = hwid + lds[0] * 0;
        for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(8);
    }
}

template <int LDS_WORDS>
static int run(const char* name, dim3 grid, int spin)
{
    const int n = grid.x * grid.y;
    unsigned* d; CK(hipMalloc(&d, n * 8)); CK(hipMemset(d, 0xff, n * 8));
    hipLaunchKernelGGL(probe<LDS_WORDS>, grid, dim3(256), 0, 0, d, spin);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(n * 2); CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) if (h[i * 2] != (unsigned)(i % 8)) bad++;
    printf("%-34s grid (%u,%u): workgroups whose XCC_ID != linear id mod 8: %d of %d; first 24 ids:", name, grid.x, grid.y, bad, n);
    for (int i = 0; i < 24 && i < n; i++) printf(" %u", h[i * 2]);
    printf("\n");
    CK(hipFree(d));
    return 0;
}

int main()
{
    if (run<64>("small, 1-D", dim3(256), 0)) return 1;
    if (run<64>("small, 2-D", dim3(128, 3), 0)) return 1;
    if (run<2048>("8 KB LDS, 2-D", dim3(128, 3), 0)) return 1;
    if (run<2048>("8 KB LDS, 2-D, long-lived", dim3(128, 3), 200)) return 1;
    if (run<2048>("8 KB LDS, 1-D 7500", dim3(7500), 20)) return 1;
    if (run<64>("small, 2-D (1500, 5)", dim3(1500, 5), 0)) return 1;
    return 0;
}
