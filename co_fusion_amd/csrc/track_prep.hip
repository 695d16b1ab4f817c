// track_prep.hip -- map-preparation kernels of the tracking path (gfx950).
//
// MI355X-native replacements of Core/Cuda/cudafuncs.cu:109-751.  All are pure streaming
// kernels: one thread per output pixel, 256-thread workgroups walking rows so every wave
// issues 256 B coalesced accesses on each plane; no pitch (planes are dense [3H x W]).
// Results are bit-identical to the CPU oracle (same IEEE operation order, -ffp-contract=off).
#include "cf_device.h"
#include "cf_kernels.h"
#include "track_prep_dev.h"

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

// 1-D binomial weight {1, 4, 6, 4, 1}[k]; the 5x5 kernel of the reference is their outer product (exact integers)
__device__ __forceinline__ int gauss_w(int k) { return k == 2 ? 6 : ((k == 0 || k == 4) ? 1 : 4); }

// pyrDownKernelGaussF, cudafuncs.cu:333-364 (window excludes last row/col, weights anchored at the clamped window
// END, int count).  The reference cudaMalloc/Free's the weights per call (:523-531).  Fully unrolled 5x5 walks (25
// independent loads, row-major summation order as the reference's loops): literal weights in the interior, weights
// computed from the window end (ty - cy - 1, tx - cx - 1) and predicated taps on the border.
__device__ __forceinline__ void pyrdown_f32_px(const float* __restrict__ src, int scols, int srows, float* __restrict__ dst, int i)
{
    const int dcols = scols / 2;
    const int y = i / dcols, x = i - y * dcols;
    float sum = 0; int count = 0;
    if (2 * x - 2 >= 0 && 2 * y - 2 >= 0 && 2 * x + 3 <= scols - 1 && 2 * y + 3 <= srows - 1) {
        // interior: full window, literal weights (index 4-r, 4-c == r, c by symmetry)
        constexpr int g[5] = {1, 4, 6, 4, 1};
        const float* __restrict__ p0 = src + (2 * y - 2) * scols + (2 * x - 2);
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const float s = p0[r * scols + c];
                if (!is_nan(s)) { sum += s * (float)(g[r] * g[c]); count += g[r] * g[c]; }
            }
    } else {
        // border: same walk, predicated, weights from the clamped window end (no serial 25-iteration loop)
        const int tx = min(2 * x - 2 + 5, scols - 1), ty = min(2 * y - 2 + 5, srows - 1);
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int cy = 2 * y - 2 + r;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const int cx = 2 * x - 2 + c;
                const bool in = cy >= 0 && cy < ty && cx >= 0 && cx < tx;
                const float s = src[in ? cy * scols + cx : 0];
                if (in && !is_nan(s)) {
                    const int wi = gauss_w(ty - cy - 1) * gauss_w(tx - cx - 1);
                    sum += s * (float)wi;
                    count += wi;
                }
            }
        }
    }
    dst[i] = sum / (float)count;
}
__global__ void __launch_bounds__(kBlock) pyrdown_f32_kernel(const float* __restrict__ src, int scols, int srows,
                                                             float* __restrict__ dst)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < (scols / 2) * (srows / 2)) pyrdown_f32_px(src, scols, srows, dst, i);
}

// pyrDownKernelIntensityGauss, cudafuncs.cu:534-564
__device__ __forceinline__ uint8_t pyrdown_u8_val(const uint8_t* __restrict__ src, int scols, int srows, int i)
{
    const int dcols = scols / 2;
    const int y = i / dcols, x = i - y * dcols;
    float sum = 0; int count = 0;
    if (2 * x - 2 >= 0 && 2 * y - 2 >= 0 && 2 * x + 3 <= scols - 1 && 2 * y + 3 <= srows - 1) {
        constexpr int g[5] = {1, 4, 6, 4, 1};
        const uint8_t* __restrict__ p0 = src + (2 * y - 2) * scols + (2 * x - 2);
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const uint8_t s = p0[r * scols + c];
                if (s > 0) { sum += (float)s * (float)(g[r] * g[c]); count += g[r] * g[c]; }
            }
    } else {
        const int tx = min(2 * x - 2 + 5, scols - 1), ty = min(2 * y - 2 + 5, srows - 1);
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int cy = 2 * y - 2 + r;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const int cx = 2 * x - 2 + c;
                const bool in = cy >= 0 && cy < ty && cx >= 0 && cx < tx;
                const uint8_t s = src[in ? cy * scols + cx : 0];
                if (in && s > 0) {
                    const int wi = gauss_w(ty - cy - 1) * gauss_w(tx - cx - 1);
                    sum += (float)s * (float)wi;
                    count += wi;
                }
            }
        }
    }
    const float q = sum / (float)count;
    return is_nan(q) ? (uint8_t)0 : (uint8_t)(int)q;
}
__device__ __forceinline__ void pyrdown_u8_px(const uint8_t* __restrict__ src, int scols, int srows, uint8_t* __restrict__ dst, int i)
{
    dst[i] = pyrdown_u8_val(src, scols, srows, i);
}
// a chain's intensity value of pixel i at `level` to its own pyramid and to those of the trackers it also serves (RgbdBatch::fan_owner)
__device__ __forceinline__ void rgbd_store_image(const RgbdBatch& b, int by, int level, int i, uint8_t v)
{
    uint8_t* const own = b.c[by].image[level];
    own[i] = v;
#pragma unroll
    for (int t = 0; t < kPrepBatch; t++)
        if (b.fan_owner[t] == by + 1 && b.fan_image[level][t] != own) b.fan_image[level][t][i] = v;   // (uniform)
}
__global__ void __launch_bounds__(kBlock) pyrdown_u8_kernel(const uint8_t* __restrict__ src, int scols, int srows,
                                                            uint8_t* __restrict__ dst)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < (scols / 2) * (srows / 2)) pyrdown_u8_px(src, scols, srows, dst, i);
}

// RGBDOdometry::populateRGBDData (RGBDOdometry.cpp:177-194) issues a depth chain and an intensity chain that do not
// depend on each other: each pyramid step of both shares one launch (workgroups [0, nd) depth, the rest intensity).
__device__ __forceinline__ void rgbd_base_body(const RgbdBatch& b, int N, float cutoff, int bx, int by)
{
    const RgbdChain& c = b.c[by];  // one (depth, intensity) chain per grid row: models x {prediction, frame}
    const bool alt = c.sel && (float)c.sel[0] / (float)c.sel[1] < c.sel_ratio;  // (CoFusion::requiresFillIn, decided here: cf_kernels.h)
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(alt ? c.alt_v4 : c.v4);
    const uchar4* __restrict__ rgba = reinterpret_cast<const uchar4*>(alt ? c.alt_rgba : c.rgba);
    const int nb = (N + kBlock - 1) / kBlock;
    if (bx < nb) {  // verticesToDepthKernel, cudafuncs.cu:602-613
        const int i = bx * kBlock + threadIdx.x;
        if (i >= N || !c.depth[0]) return;  // (a chain without a depth pyramid: the intensity half only)
        const float z = v4[i].z;
        c.depth[0][i] = (z > cutoff || z <= 0) ? qnan() : z;
    } else {                     // bgr2IntensityKernel, cudafuncs.cu:626-639
        const int i = (bx - nb) * kBlock + threadIdx.x;
        if (i >= N) return;
        const uchar4 s = rgba[i];
        rgbd_store_image(b, by, 0, i, (uint8_t)(int)((float)s.x * 0.114f + (float)s.y * 0.299f + (float)s.z * 0.587f));
    }
}
__global__ void __launch_bounds__(kBlock) rgbd_base_kernel(const RgbdBatch b, int N, float cutoff) { rgbd_base_body(b, N, cutoff, (int)blockIdx.x, (int)blockIdx.y); }
__global__ void __launch_bounds__(kBlock) rgbd_pyrdown_kernel(const RgbdBatch b, int level, int scols, int srows)
{
    const RgbdChain& c = b.c[blockIdx.y];
    const int n = (scols / 2) * (srows / 2), nb = (n + kBlock - 1) / kBlock;
    if ((int)blockIdx.x < nb) {
        const int i = blockIdx.x * kBlock + threadIdx.x;
        if (i < n && c.depth[level]) pyrdown_f32_px(c.depth[level], scols, srows, c.depth[level + 1], i);
    } else {
        const int i = (blockIdx.x - nb) * kBlock + threadIdx.x;
        if (i < n) rgbd_store_image(b, (int)blockIdx.y, level + 1, i, pyrdown_u8_val(c.image[level], scols, srows, i));
    }
}

// bgr2IntensityKernel, cudafuncs.cu:626-639 (texel is R,G,B: .114 R + .299 G + .587 B)
// 3-byte RGB (FrameData.rgb, the host image of CoFusion.cpp:179) -> RGBA8 with alpha 255 (the GL_RGBA texture upload)
__global__ void __launch_bounds__(kBlock) rgb_expand_kernel(const uint8_t* __restrict__ rgb, int N, uchar4* __restrict__ rgba)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    rgba[i] = make_uchar4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 255);
}

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

// =================================================================================================
// Fused per-frame preparation.  The reference (and the one-function-per-entry C-ABI above) issues ~38 tiny
// launches per tracked model and frame; each is 3-8 us of launch + latency for a few hundred KB of data.  The
// tracker object uses the three kernels below instead: every pyramid level in ONE launch (workgroup ranges
// per level) and dependent stages recomputed in registers with the same expressions, so the outputs -- including
// which planes are left untouched for invalid pixels -- are bit-identical to the chain of single kernels.
// =================================================================================================

// vmap_kernel + nmap_kernel for the three levels (RGBDOdometry::initICP, RGBDOdometry.cpp:116-143)
__global__ void __launch_bounds__(kBlock) frame_maps_kernel(const FrameMapsArgs a)
{
    int lb;
    const int lv = level_of(a.L, blockIdx.x, lb);
    const int cols = a.L.cols[lv], rows = a.L.rows[lv], N = cols * rows;
    const int i = lb * kBlock + threadIdx.x;
    const float* __restrict__ depth = a.depth[lv];
    const float cutoff = a.cutoff;
    if (a.zrange[lv]) {
        // depth interval of this wave's 64 consecutive pixels (valid vertices only): the ICP reduction skips a run whose interval
        // misses the depth interval a culled model can match in (screen_box)
        const float inf = __int_as_float(0x7f800000);
        const float zz = i < N ? depth[i] : 0.f;
        const bool ok = zz != 0 && zz < cutoff;
        float lo = ok ? zz : inf, hi = ok ? zz : -inf;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
        if ((threadIdx.x & 63) == 0 && i < N) a.zrange[lv][i >> 6] = make_float2(lo, hi);
    }
    if (i >= N) return;
    float* __restrict__ vmap = a.vmap[lv];
    float* __restrict__ nmap = a.nmap[lv];
    const float fx_inv = a.fx_inv[lv], fy_inv = a.fy_inv[lv], cx = a.cx[lv], cy = a.cy[lv];
    const int v = i / cols, u = i - v * cols;
    const bool edge = (u == cols - 1 || v == rows - 1);
    const float z = depth[i];
    const float z01 = edge ? 0.f : depth[i + 1], z10 = edge ? 0.f : depth[i + cols];
    float x00 = qnan();
    if (z != 0 && z < cutoff) {
        x00 = z * (u - cx) * fx_inv;
        vmap[i] = x00;
        vmap[i + N] = z * (v - cy) * fy_inv;
        vmap[i + 2 * N] = z;
    } else {
        vmap[i] = qnan();
    }
    if (edge) { nmap[i] = qnan(); return; }
    const float x01 = (z01 != 0 && z01 < cutoff) ? z01 * ((u + 1) - cx) * fx_inv : qnan();
    const float x10 = (z10 != 0 && z10 < cutoff) ? z10 * (u - cx) * fx_inv : qnan();
    if (!is_nan(x00) && !is_nan(x01) && !is_nan(x10)) {
        const f3 v00 = {x00, z * (v - cy) * fy_inv, z};
        const f3 v01 = {x01, z01 * (v - cy) * fy_inv, z01};
        const f3 v10 = {x10, z10 * ((v + 1) - cy) * fy_inv, z10};
        const f3 r = normalized(cross(v01 - v00, v10 - v00));
        nmap[i] = r.x; nmap[i + N] = r.y; nmap[i + 2 * N] = r.z;
    } else {
        nmap[i] = qnan();
    }
}

// sobel_kernel + rgb_cand_kernel + cloud_kernel for the three levels: rgb_prep_body (track_prep_dev.h)
__global__ void __launch_bounds__(kBlock) rgb_prep_kernel(const RgbPrepBatch b) { rgb_prep_body(b.m[blockIdx.y], (int)blockIdx.x); }

// copy_maps + resize_map<false/true> x2 + transform_maps x3 (RGBDOdometry::initICPModel, RGBDOdometry.cpp:145-174):
// one thread per level-2 pixel owns its 4x4 level-0 block.  The resize chain runs on the untransformed values,
// every level is stored transformed; invalid pixels keep the reference's partial writes (see `emit`).
struct MapVal { f3 p; bool have; };  // have == false: the resize bailed out on a NaN source (only the x plane is written)

__device__ __forceinline__ void emit_map(float* __restrict__ map, int idx, int N, const MapVal& m, const m33& R, const f3& t, bool add_t)
{
    if (!m.have) { map[idx] = qnan(); return; }
    if (!is_nan(m.p.x)) {
        f3 d = mul(R, m.p);
        if (add_t) d = d + t;
        map[idx] = d.x; map[idx + N] = d.y; map[idx + 2 * N] = d.z;
    } else {  // the stage before the transform wrote all planes, the transform only overwrites x with NaN
        map[idx] = qnan(); map[idx + N] = m.p.y; map[idx + 2 * N] = m.p.z;
    }
}

__device__ __forceinline__ MapVal resize4(const MapVal& a00, const MapVal& a01, const MapVal& a10, const MapVal& a11, bool normalize)
{
    // resize_map_kernel: the NaN test looks at the x planes of the four sources (a source whose own resize bailed
    // out has x == NaN in memory)
    MapVal o; o.have = false; o.p = f3{qnan(), qnan(), qnan()};
    const float x00 = a00.have ? a00.p.x : qnan(), x01 = a01.have ? a01.p.x : qnan();
    const float x10 = a10.have ? a10.p.x : qnan(), x11 = a11.have ? a11.p.x : qnan();
    if (is_nan(x00) || is_nan(x01) || is_nan(x10) || is_nan(x11)) return o;
    f3 n;
    n.x = (a00.p.x + a01.p.x + a10.p.x + a11.p.x) / 4;
    n.y = (a00.p.y + a01.p.y + a10.p.y + a11.p.y) / 4;
    n.z = (a00.p.z + a01.p.z + a10.p.z + a11.p.z) / 4;
    if (normalize) n = normalized(n);
    o.have = true; o.p = n;
    return o;
}

__global__ void __launch_bounds__(64) model_maps_kernel(const ModelMapsBatch b)
{
    const ModelMapsArgs& a = b.m[blockIdx.y];  // one tracked model per grid row
    const int c2 = a.cols >> 2, r2 = a.rows >> 2;
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= c2 * r2) return;
    bool any_valid = false;  // does this thread's 4x4 block hold any predicted surface?
    const int Y2 = t / c2, X2 = t - Y2 * c2;
    const int cols = a.cols, rows = a.rows, N0 = cols * rows, c1 = cols >> 1, N1 = N0 >> 2, N2 = N0 >> 4;
    const bool alt = a.sel && (float)a.sel[0] / (float)a.sel[1] < a.sel_ratio;
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(alt ? a.alt_v4 : a.pred_v4);
    const float4* __restrict__ n4 = reinterpret_cast<const float4*>(alt ? a.alt_n4 : a.pred_n4);
    float4* __restrict__ snap = reinterpret_cast<float4*>(a.snapshot);
    m33 R; for (int k = 0; k < 9; k++) R.m[k] = a.R[k];
    const f3 tr = {a.t[0], a.t[1], a.t[2]};
    MapVal v1[4], n1[4];
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {  // the four level-1 pixels of this block
        const int y1 = 2 * Y2 + (qd >> 1), x1 = 2 * X2 + (qd & 1);
        MapVal v0[4], n0[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int yy = 2 * y1 + (s >> 1), xx = 2 * x1 + (s & 1);
            const int i0 = yy * cols + xx;
            const float4 vs = v4[i0], ns = n4[i0];
            snap[i0] = vs;
            v0[s].have = true; n0[s].have = true;
            if (!(vs.z == 0)) { v0[s].p = f3{vs.x, vs.y, vs.z}; n0[s].p = f3{ns.x, ns.y, ns.z}; any_valid = true; }
            else { v0[s].p = f3{qnan(), qnan(), qnan()}; n0[s].p = v0[s].p; }
            emit_map(a.vmap[0], i0, N0, v0[s], R, tr, true);
            emit_map(a.nmap[0], i0, N0, n0[s], R, tr, false);
        }
        v1[qd] = resize4(v0[0], v0[1], v0[2], v0[3], false);
        n1[qd] = resize4(n0[0], n0[1], n0[2], n0[3], true);
        const int i1 = y1 * c1 + x1;
        emit_map(a.vmap[1], i1, N1, v1[qd], R, tr, true);
        emit_map(a.nmap[1], i1, N1, n1[qd], R, tr, false);
    }
    const MapVal v2 = resize4(v1[0], v1[1], v1[2], v1[3], false);
    const MapVal n2 = resize4(n1[0], n1[1], n1[2], n1[3], true);
    const int i2 = Y2 * c2 + X2;
    emit_map(a.vmap[2], i2, N2, v2, R, tr, true);
    emit_map(a.nmap[2], i2, N2, n2, R, tr, false);
    // occupancy map of the prediction, one byte per 4x4 block: the ICP reduction skips the gathers of pixels that project into an
    // empty block (every level's model-map pixel inside such a block is invalid)
    if (a.occ) a.occ[t] = any_valid ? 1 : 0;
}

// The same pass with one lane per level-0 pixel: a wave owns a 16 x 4 tile (rows of 16 float4 = 256 contiguous bytes per load, 64
// contiguous bytes per planar store), the 2x2 averages of levels 1 and 2 take their four sources from the neighbouring lanes in
// the order the reference adds them (resize4 is the same function).  16x the parallelism of model_maps_kernel, whose lanes walk
// a 4x4 block each.  Needs cols % 16 == 0 and rows % 4 == 0.
__device__ __forceinline__ MapVal lane_mapval(const MapVal& m, int src_lane)
{
    MapVal o;
    o.p.x = __shfl(m.p.x, src_lane, 64); o.p.y = __shfl(m.p.y, src_lane, 64); o.p.z = __shfl(m.p.z, src_lane, 64);
    o.have = __shfl((int)m.have, src_lane, 64) != 0;
    return o;
}
__device__ __forceinline__ void model_maps_tiled_body(const ModelMapsBatch& b, int bx, int by)
{
    const ModelMapsArgs& a = b.m[by];  // one tracked model per grid row
    const int cols = a.cols, rows = a.rows, N0 = cols * rows, c1 = cols >> 1, c2 = cols >> 2, N1 = N0 >> 2, N2 = N0 >> 4;
    const int lane = threadIdx.x & 63, lx = lane & 15, ly = lane >> 4;
    const int tiles_x = cols >> 4;
    const int tile = bx * 4 + (threadIdx.x >> 6);
    if (tile >= tiles_x * (rows >> 2)) return;  // whole waves leave
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = tx * 16 + lx, y = ty * 4 + ly, i0 = y * cols + x;
    const bool alt = a.sel && (float)a.sel[0] / (float)a.sel[1] < a.sel_ratio;  // (CoFusion::requiresFillIn, decided here: cf_kernels.h)
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(alt ? a.alt_v4 : a.pred_v4);
    const float4* __restrict__ n4 = reinterpret_cast<const float4*>(alt ? a.alt_n4 : a.pred_n4);
    float4* __restrict__ snap = reinterpret_cast<float4*>(a.snapshot);
    m33 R; for (int k = 0; k < 9; k++) R.m[k] = a.R[k];
    const f3 tr = {a.t[0], a.t[1], a.t[2]};
    const float4 vs = v4[i0], ns = n4[i0];
    snap[i0] = vs;
    MapVal v0, n0;
    v0.have = true; n0.have = true;
    const bool valid = !(vs.z == 0);
    if (valid) { v0.p = f3{vs.x, vs.y, vs.z}; n0.p = f3{ns.x, ns.y, ns.z}; }
    else { v0.p = f3{qnan(), qnan(), qnan()}; n0.p = v0.p; }
    emit_map(a.vmap[0], i0, N0, v0, R, tr, true);
    emit_map(a.nmap[0], i0, N0, n0, R, tr, false);
    // level 1: the lane at the even corner of a 2x2 block combines (y, x), (y, x+1), (y+1, x), (y+1, x+1)
    const MapVal v01 = lane_mapval(v0, lane + 1), v10 = lane_mapval(v0, lane + 16), v11 = lane_mapval(v0, lane + 17);
    const MapVal n01 = lane_mapval(n0, lane + 1), n10 = lane_mapval(n0, lane + 16), n11 = lane_mapval(n0, lane + 17);
    MapVal v1, n1;
    v1.have = false; n1.have = false; v1.p = f3{qnan(), qnan(), qnan()}; n1.p = v1.p;
    if (((lx | ly) & 1) == 0) {
        v1 = resize4(v0, v01, v10, v11, false);
        n1 = resize4(n0, n01, n10, n11, true);
        const int i1 = (y >> 1) * c1 + (x >> 1);
        emit_map(a.vmap[1], i1, N1, v1, R, tr, true);
        emit_map(a.nmap[1], i1, N1, n1, R, tr, false);
    }
    // level 2: the lane at the corner of a 4x4 block combines the four level-1 values of the block
    const MapVal w01 = lane_mapval(v1, lane + 2), w10 = lane_mapval(v1, lane + 32), w11 = lane_mapval(v1, lane + 34);
    const MapVal m01 = lane_mapval(n1, lane + 2), m10 = lane_mapval(n1, lane + 32), m11 = lane_mapval(n1, lane + 34);
    const unsigned long long valid_bits = __ballot(valid);
    if ((lx & 3) == 0 && ly == 0) {
        const MapVal v2 = resize4(v1, w01, w10, w11, false);
        const MapVal n2 = resize4(n1, m01, m10, m11, true);
        const int i2 = (y >> 2) * c2 + (x >> 2);
        emit_map(a.vmap[2], i2, N2, v2, R, tr, true);
        emit_map(a.nmap[2], i2, N2, n2, R, tr, false);
        // occupancy of this 4x4 block: lanes lx..lx+3 of the four tile rows
        if (a.occ) a.occ[i2] = ((valid_bits >> lx) & 0x000F000F000F000Full) != 0 ? 1 : 0;
    }
    // Bounding FRUSTUM of the prediction, in its own camera: the pixel rectangle of the valid level-0 vertices and the interval of their
    // depths (the coarser levels are averages of these).  A pixel of the current frame finds a correspondence only if its vertex, taken
    // into this camera, projects onto a valid pixel -- it lies in that pixel's pyramid -- at a depth within distThres of the model's
    // (reduce.cu:321-325): inside the rectangle's frustum between zmin - distThres and zmax + distThres.  The Gauss-Newton loop projects
    // that frustum piece into the current camera (screen_box) and culls everything outside.  (Until round 5: the axis-aligned box of the
    // vertices in the GLOBAL frame, dilated by distThres in every direction -- 53 pixels sideways at one metre.)
    // Six order-preserving keys, atomicMax each (the lower bounds as complemented keys), zero = empty.
    if (a.aabb && valid_bits != 0) {
        const float inf = __int_as_float(0x7f800000);
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        if (valid && !is_nan(v0.p.x) && !is_nan(vs.z)) { lo[0] = hi[0] = (float)x; lo[1] = hi[1] = (float)y; lo[2] = hi[2] = vs.z; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64)); }
        // lanes 0..5 own one key each; a wave that does not extend the box (nearly all of them, after the first few) only READS the
        // current keys -- six same-address atomics per wave had doubled the duration of this kernel
        if (lane < 6 && lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) {
            const float val = lane == 0 ? lo[0] : lane == 1 ? lo[1] : lane == 2 ? lo[2] : lane == 3 ? hi[0] : lane == 4 ? hi[1] : hi[2];
            const unsigned key = lane < 3 ? ~fkey(val) : fkey(val);
            if (key > __hip_atomic_load(&a.aabb[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&a.aabb[lane], key);
        }
    }
}
__global__ void __launch_bounds__(256) model_maps_tiled_kernel(const ModelMapsBatch b) { model_maps_tiled_body(b, (int)blockIdx.x, (int)blockIdx.y); }
// {model maps || base level of the depth / intensity pyramids} of a batch of trackers in ONE launch: the two passes read the same
// predictions and write different buffers; alone each is a 10-25 us kernel in the chain in front of the Gauss-Newton loop.  A
// one-dimensional grid: [model-map workgroups of all models | pyramid-base workgroups of all chains].
__global__ void __launch_bounds__(256) prep_fused_kernel(const ModelMapsBatch mb, int mm_bx, int mm_total, const RgbdBatch rb, int N, float cutoff, int rb_bx)
{
    const int b = blockIdx.x;
    if (b < mm_total) { const int by = b / mm_bx; model_maps_tiled_body(mb, b - by * mm_bx, by); }
    else { const int r = b - mm_total, by = r / rb_bx; rgbd_base_body(rb, N, cutoff, r - by * rb_bx, by); }
}

// ------------------------------------------------------------------ launchers ----
static Level3 levels3(int W, int H)
{
    Level3 L; int acc = 0;
    for (int i = 0; i < 3; i++) { L.cols[i] = W >> i; L.rows[i] = H >> i; acc += grid_for(L.cols[i] * L.rows[i]); L.blk_end[i] = acc; }
    return L;
}
void launch_frame_maps(hipStream_t s, FrameMapsArgs a, int W, int H)
{
    a.L = levels3(W, H);
    frame_maps_kernel<<<a.L.blk_end[2], kBlock, 0, s>>>(a);
}
void rgb_prep_levels(RgbPrepBatch& b, int n, int W, int H)
{
    const Level3 L = levels3(W, H);
    for (int m = 0; m < n; m++) b.m[m].L = L;
}
void launch_rgb_prep(hipStream_t s, RgbPrepBatch b, int n, int W, int H)
{
    const Level3 L = levels3(W, H);
    for (int m = 0; m < n; m++) b.m[m].L = L;
    rgb_prep_kernel<<<dim3(L.blk_end[2], n), kBlock, 0, s>>>(b);
}
void launch_model_maps(hipStream_t s, const ModelMapsBatch& b, int n)
{
    const int cols = b.m[0].cols, rows = b.m[0].rows;
    if (cols % 16 == 0 && rows % 4 == 0) {
        const int tiles = (cols >> 4) * (rows >> 2);
        model_maps_tiled_kernel<<<dim3((tiles + 3) / 4, n), 256, 0, s>>>(b);
        return;
    }
    const int t = (cols >> 2) * (rows >> 2);
    model_maps_kernel<<<dim3((t + 63) / 64, n), 64, 0, s>>>(b);
}
void launch_model_maps_and_pyramids(hipStream_t s, const ModelMapsBatch& mb, int n, const RgbdBatch& rb, int n_chains, int W, int H, float cutoff)
{
    static_assert(kBlock == 256, "the fused launch runs both bodies with 256 threads");
    const int cols = mb.m[0].cols, rows = mb.m[0].rows;
    if (cols % 16 == 0 && rows % 4 == 0) {
        const int mm_bx = ((cols >> 4) * (rows >> 2) + 3) / 4, rb_bx = 2 * grid_for(W * H);
        prep_fused_kernel<<<mm_bx * n + rb_bx * n_chains, 256, 0, s>>>(mb, mm_bx, mm_bx * n, rb, W * H, cutoff, rb_bx);
    } else {
        launch_model_maps(s, mb, n);
        rgbd_base_kernel<<<dim3(2 * grid_for(W * H), n_chains), kBlock, 0, s>>>(rb, W * H, cutoff);
    }
    for (int i = 0; i + 1 < 3; i++)
        rgbd_pyrdown_kernel<<<dim3(2 * grid_for(((W >> i) / 2) * ((H >> i) / 2)), n_chains), kBlock, 0, s>>>(rb, i, W >> i, H >> i);
}
void launch_rgbd_pyramids(hipStream_t s, const RgbdBatch& b, int n_chains, int W, int H, float cutoff)
{
    rgbd_base_kernel<<<dim3(2 * grid_for(W * H), n_chains), kBlock, 0, s>>>(b, W * H, cutoff);
    for (int i = 0; i + 1 < 3; i++)
        rgbd_pyrdown_kernel<<<dim3(2 * grid_for(((W >> i) / 2) * ((H >> i) / 2)), n_chains), kBlock, 0, s>>>(b, i, W >> i, H >> i);
}
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
void launch_rgb_expand(hipStream_t s, const uint8_t* rgb, int n, uint8_t* rgba)
{
    rgb_expand_kernel<<<grid_for(n), kBlock, 0, s>>>(rgb, n, reinterpret_cast<uchar4*>(rgba));
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
