// track_prep.hip -- map-preparation kernels of the tracking path (gfx950).
//
// MI355X-native replacements of Core/Cuda/cudafuncs.cu:109-751.  All are pure streaming
// kernels: one thread per output pixel, 256-thread workgroups walking rows so every wave
// issues 256 B coalesced accesses on each plane; no pitch (planes are dense [3H x W]).
// Results are bit-identical to the CPU oracle (same IEEE operation order, -ffp-contract=off).
#include "cf_device.h"
#include "cf_kernels.h"

namespace cf {

static constexpr int kBlock = 256;
static inline int grid_for(int n) { return (n + kBlock - 1) / kBlock; }

// computeVmapKernel, cudafuncs.cu:109-134 (mask test commented out at :119; NaN to x plane only)
__global__ void __launch_bounds__(kBlock) vmap_kernel(const float* __restrict__ depth, int cols, int rows, float fx_inv,
                                                      float fy_inv, float cx, float cy, float cutoff,
                                                      float* __restrict__ vmap)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= cols * rows) return;
    const int v = i / cols, u = i - v * cols;
    const float z = depth[i];
    if (z != 0 && z < cutoff) {
        vmap[i] = z * (u - cx) * fx_inv;
        vmap[i + rows * cols] = z * (v - cy) * fy_inv;
        vmap[i + 2 * rows * cols] = z;
    } else {
        vmap[i] = qnan();
    }
}

// computeNmapKernel, cudafuncs.cu:152-189
__global__ void __launch_bounds__(kBlock) nmap_kernel(const float* __restrict__ vmap, int cols, int rows,
                                                      float* __restrict__ nmap)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int N = cols * rows;
    if (i >= N) return;
    const int v = i / cols, u = i - v * cols;
    if (u == cols - 1 || v == rows - 1) { nmap[i] = qnan(); return; }
    const float x00 = vmap[i], x01 = vmap[i + 1], x10 = vmap[i + cols];
    if (!is_nan(x00) && !is_nan(x01) && !is_nan(x10)) {
        f3 v00 = {x00, vmap[i + N], vmap[i + 2 * N]};
        f3 v01 = {x01, vmap[i + 1 + N], vmap[i + 1 + 2 * N]};
        f3 v10 = {x10, vmap[i + cols + N], vmap[i + cols + 2 * N]};
        f3 r = normalized(cross(v01 - v00, v10 - v00));
        nmap[i] = r.x; nmap[i + N] = r.y; nmap[i + 2 * N] = r.z;
    } else {
        nmap[i] = qnan();
    }
}

// copyMapsKernel, cudafuncs.cu:271-311: RGBA32F -> planar, z == 0 -> NaN in all planes.
// float4 loads (16 B/lane) on the interleaved side, dword stores on the planar side.
__global__ void __launch_bounds__(kBlock) copy_maps_kernel(const float4* __restrict__ v4, const float4* __restrict__ n4,
                                                           int N, float* __restrict__ vmap, float* __restrict__ nmap)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float4 vs = v4[i], ns = n4[i];
    f3 vd = {qnan(), qnan(), qnan()}, nd = vd;
    if (!(vs.z == 0)) { vd = f3{vs.x, vs.y, vs.z}; nd = f3{ns.x, ns.y, ns.z}; }
    vmap[i] = vd.x; vmap[i + N] = vd.y; vmap[i + 2 * N] = vd.z;
    nmap[i] = nd.x; nmap[i + N] = nd.y; nmap[i + 2 * N] = nd.z;
}

// resizeMapKernel<normalize>, cudafuncs.cu:366-417
template <bool NORMALIZE>
__global__ void __launch_bounds__(kBlock) resize_map_kernel(const float* __restrict__ in, int in_cols, int in_rows,
                                                            float* __restrict__ out)
{
    const int dcols = in_cols / 2, drows = in_rows / 2;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= dcols * drows) return;
    const int y = i / dcols, x = i - y * dcols;
    const int s = (2 * y) * in_cols + 2 * x, SN = in_cols * in_rows, DN = dcols * drows;
    const float2 a0 = *reinterpret_cast<const float2*>(in + s);
    const float2 a1 = *reinterpret_cast<const float2*>(in + s + in_cols);
    if (is_nan(a0.x) || is_nan(a0.y) || is_nan(a1.x) || is_nan(a1.y)) { out[i] = qnan(); return; }
    const float2 b0 = *reinterpret_cast<const float2*>(in + s + SN);
    const float2 b1 = *reinterpret_cast<const float2*>(in + s + SN + in_cols);
    const float2 c0 = *reinterpret_cast<const float2*>(in + s + 2 * SN);
    const float2 c1 = *reinterpret_cast<const float2*>(in + s + 2 * SN + in_cols);
    f3 n;
    n.x = (a0.x + a0.y + a1.x + a1.y) / 4;
    n.y = (b0.x + b0.y + b1.x + b1.y) / 4;
    n.z = (c0.x + c0.y + c1.x + c1.y) / 4;
    if (NORMALIZE) n = normalized(n);
    out[i] = n.x; out[i + DN] = n.y; out[i + 2 * DN] = n.z;
}

// tranformMapsKernel, cudafuncs.cu:207-249 (in place)
__global__ void __launch_bounds__(kBlock) transform_maps_kernel(float* __restrict__ vmap, float* __restrict__ nmap, int N,
                                                                m33 R, f3 t)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float vx = vmap[i];
    float outx = qnan();
    if (!is_nan(vx)) {
        f3 vd = mul(R, f3{vx, vmap[i + N], vmap[i + 2 * N]}) + t;
        vmap[i + N] = vd.y; vmap[i + 2 * N] = vd.z; outx = vd.x;
    }
    vmap[i] = outx;
    const float nx = nmap[i];
    outx = qnan();
    if (!is_nan(nx)) {
        f3 nd = mul(R, f3{nx, nmap[i + N], nmap[i + 2 * N]});
        nmap[i + N] = nd.y; nmap[i + 2 * N] = nd.z; outx = nd.x;
    }
    nmap[i] = outx;
}

// verticesToDepthKernel, cudafuncs.cu:602-613
__global__ void __launch_bounds__(kBlock) vertices_to_depth_kernel(const float4* __restrict__ v4, int N, float cutoff,
                                                                   float* __restrict__ depth)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float z = v4[i].z;
    depth[i] = (z > cutoff || z <= 0) ? qnan() : z;
}

constexpr int kGaussI[5] = {1, 4, 6, 4, 1};
__constant__ float kGauss25[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1};

// pyrDownKernelGaussF, cudafuncs.cu:333-364 (window excludes last row/col, weights anchored at the
// clamped window END, int count).  The reference cudaMalloc/Free's the weights per call (:523-531).
__global__ void __launch_bounds__(kBlock) pyrdown_f32_kernel(const float* __restrict__ src, int scols, int srows,
                                                             float* __restrict__ dst)
{
    const int dcols = scols / 2, drows = srows / 2;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= dcols * drows) return;
    const int y = i / dcols, x = i - y * dcols;
    float sum = 0; int count = 0;
    if (2 * x - 2 >= 0 && 2 * y - 2 >= 0 && 2 * x + 3 <= scols - 1 && 2 * y + 3 <= srows - 1) {
        // interior: the window is the full 5x5, all 25 loads are independent and the weights are literals
        // (anchored at the window end: index 4-r, 4-c == r, c by symmetry); same row-major summation order
        const float* __restrict__ p0 = src + (2 * y - 2) * scols + (2 * x - 2);
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float s = p0[r * scols + c];
                if (!is_nan(s)) {
                    const int wi = kGaussI[r] * kGaussI[c];
                    sum += s * (float)wi;
                    count += wi;
                }
            }
    } else {
        const int tx = min(2 * x - 2 + 5, scols - 1), ty = min(2 * y - 2 + 5, srows - 1);
        for (int cy = max(0, 2 * y - 2); cy < ty; ++cy)
            for (int cx = max(0, 2 * x - 2); cx < tx; ++cx) {
                const float s = src[cy * scols + cx];
                if (!is_nan(s)) {
                    const float w = kGauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                    sum += s * w;
                    count += (int)w;
                }
            }
    }
    dst[i] = sum / (float)count;
}

// pyrDownKernelIntensityGauss, cudafuncs.cu:534-564
__global__ void __launch_bounds__(kBlock) pyrdown_u8_kernel(const uint8_t* __restrict__ src, int scols, int srows,
                                                            uint8_t* __restrict__ dst)
{
    const int dcols = scols / 2, drows = srows / 2;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= dcols * drows) return;
    const int y = i / dcols, x = i - y * dcols;
    float sum = 0; int count = 0;
    if (2 * x - 2 >= 0 && 2 * y - 2 >= 0 && 2 * x + 3 <= scols - 1 && 2 * y + 3 <= srows - 1) {
        const uint8_t* __restrict__ p0 = src + (2 * y - 2) * scols + (2 * x - 2);
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const uint8_t s = p0[r * scols + c];
                if (s > 0) {
                    const int wi = kGaussI[r] * kGaussI[c];
                    sum += (float)s * (float)wi;
                    count += wi;
                }
            }
    } else {
        const int tx = min(2 * x - 2 + 5, scols - 1), ty = min(2 * y - 2 + 5, srows - 1);
        for (int cy = max(0, 2 * y - 2); cy < ty; ++cy)
            for (int cx = max(0, 2 * x - 2); cx < tx; ++cx) {
                const uint8_t s = src[cy * scols + cx];
                if (s > 0) {
                    const float w = kGauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                    sum += (float)s * w;
                    count += (int)w;
                }
            }
    }
    const float q = sum / (float)count;
    dst[i] = is_nan(q) ? (uint8_t)0 : (uint8_t)(int)q;
}

// bgr2IntensityKernel, cudafuncs.cu:626-639 (texel is R,G,B: .114 R + .299 G + .587 B)
__global__ void __launch_bounds__(kBlock) intensity_kernel(const uchar4* __restrict__ rgba, int N, uint8_t* __restrict__ dst)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const uchar4 s = rgba[i];
    const int value = (int)((float)s.x * 0.114f + (float)s.y * 0.299f + (float)s.z * 0.587f);
    dst[i] = (uint8_t)value;
}

__constant__ float kSobelX[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
__constant__ float kSobelY[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};

// applyKernel, cudafuncs.cu:658-683 (kernelIndex counts down from 8 over the CLAMPED window)
__global__ void __launch_bounds__(kBlock) sobel_kernel(const uint8_t* __restrict__ src, int cols, int rows,
                                                       int16_t* __restrict__ dx, int16_t* __restrict__ dy)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= cols * rows) return;
    const int y = i / cols, x = i - y * cols;
    float dxv = 0, dyv = 0;
    constexpr float sx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    constexpr float sy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
    if (x >= 1 && y >= 1 && x <= cols - 2 && y <= rows - 2) {  // interior: literal weights, independent loads
        const uint8_t* __restrict__ p0 = src + (y - 1) * cols + (x - 1);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float s = (float)p0[r * cols + c];
                dxv += s * sx[8 - (r * 3 + c)];
                dyv += s * sy[8 - (r * 3 + c)];
            }
    } else {
        int k = 8;
        for (int j = max(y - 1, 0); j <= min(y + 1, rows - 1); j++)
            for (int c = max(x - 1, 0); c <= min(x + 1, cols - 1); c++) {
                const float s = (float)src[j * cols + c];
                dxv += s * kSobelX[k];
                dyv += s * kSobelY[k];
                --k;
            }
    }
    dx[i] = (int16_t)(int)dxv;
    dy[i] = (int16_t)(int)dyv;
}

// projectPointsKernel, cudafuncs.cu:718-736
__global__ void __launch_bounds__(kBlock) cloud_kernel(const float* __restrict__ depth, int cols, int rows, float invFx,
                                                       float invFy, float cx, float cy, float* __restrict__ cloud3)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= cols * rows) return;
    const int y = i / cols, x = i - y * cols;
    const float z = depth[i];
    cloud3[i * 3 + 0] = (x - cx) * z * invFx;
    cloud3[i * 3 + 1] = (y - cy) * z * invFy;
    cloud3[i * 3 + 2] = z;
}

// ------------------------------------------------------------------ launchers ----
void launch_vmap(hipStream_t s, const float* depth, int cols, int rows, cf_cam intr, float cutoff, float* vmap)
{
    vmap_kernel<<<grid_for(cols * rows), kBlock, 0, s>>>(depth, cols, rows, 1.f / intr.fx, 1.f / intr.fy, intr.cx, intr.cy,
                                                         cutoff, vmap);
}
void launch_nmap(hipStream_t s, const float* vmap, int cols, int rows, float* nmap)
{
    nmap_kernel<<<grid_for(cols * rows), kBlock, 0, s>>>(vmap, cols, rows, nmap);
}
void launch_copy_maps(hipStream_t s, const float* v4, const float* n4, int cols, int rows, float* vmap, float* nmap)
{
    copy_maps_kernel<<<grid_for(cols * rows), kBlock, 0, s>>>(reinterpret_cast<const float4*>(v4),
                                                              reinterpret_cast<const float4*>(n4), cols * rows, vmap, nmap);
}
void launch_resize_map(hipStream_t s, const float* in, int in_cols, int in_rows, float* out, bool normalize)
{
    const int n = (in_cols / 2) * (in_rows / 2);
    if (normalize) resize_map_kernel<true><<<grid_for(n), kBlock, 0, s>>>(in, in_cols, in_rows, out);
    else resize_map_kernel<false><<<grid_for(n), kBlock, 0, s>>>(in, in_cols, in_rows, out);
}
void launch_transform_maps(hipStream_t s, float* vmap, float* nmap, int cols, int rows, const float R[9], const float t[3])
{
    m33 Rm; for (int i = 0; i < 9; i++) Rm.m[i] = R[i];
    transform_maps_kernel<<<grid_for(cols * rows), kBlock, 0, s>>>(vmap, nmap, cols * rows, Rm, f3{t[0], t[1], t[2]});
}
void launch_vertices_to_depth(hipStream_t s, const float* v4, int cols, int rows, float cutoff, float* depth)
{
    vertices_to_depth_kernel<<<grid_for(cols * rows), kBlock, 0, s>>>(reinterpret_cast<const float4*>(v4), cols * rows,
                                                                      cutoff, depth);
}
void launch_pyrdown_f32(hipStream_t s, const float* src, int scols, int srows, float* dst)
{
    pyrdown_f32_kernel<<<grid_for((scols / 2) * (srows / 2)), kBlock, 0, s>>>(src, scols, srows, dst);
}
void launch_pyrdown_u8(hipStream_t s, const uint8_t* src, int scols, int srows, uint8_t* dst)
{
    pyrdown_u8_kernel<<<grid_for((scols / 2) * (srows / 2)), kBlock, 0, s>>>(src, scols, srows, dst);
}
void launch_intensity(hipStream_t s, const uint8_t* rgba, int cols, int rows, uint8_t* dst)
{
    intensity_kernel<<<grid_for(cols * rows), kBlock, 0, s>>>(reinterpret_cast<const uchar4*>(rgba), cols * rows, dst);
}
void launch_sobel(hipStream_t s, const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy)
{
    sobel_kernel<<<grid_for(cols * rows), kBlock, 0, s>>>(src, cols, rows, dx, dy);
}
void launch_cloud(hipStream_t s, const float* depth, int cols, int rows, cf_cam il, float* cloud3)
{
    cloud_kernel<<<grid_for(cols * rows), kBlock, 0, s>>>(depth, cols, rows, 1.0f / il.fx, 1.0f / il.fy, il.cx, il.cy, cloud3);
}

}  // namespace cf
