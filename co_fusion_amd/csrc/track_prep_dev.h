// Device bodies of the per-frame preparation that more than one translation unit launches (track_prep.hip on its own,
// track_reduce.hip fused beside the SO3 pre-alignment).  256 threads per workgroup.
#pragma once
#include "cf_device.h"
#include "cf_kernels.h"

namespace cf {
__device__ __forceinline__ int level_of(const Level3& L, int b, int& lb)
{
    const int lv = b < L.blk_end[0] ? 0 : (b < L.blk_end[1] ? 1 : 2);
    lb = b - (lv ? L.blk_end[lv - 1] : 0);
    return lv;
}

// sobel_kernel + rgb_cand_kernel + cloud_kernel for the three levels (RGBDOdometry.cpp:231-235, :333)
__device__ __forceinline__ void rgb_prep_body(const RgbPrepArgs& a, int bx)
{
    int lb;
    const int lv = level_of(a.L, bx, lb);
    const int cols = a.L.cols[lv], rows = a.L.rows[lv];
    const int k = lb * 256 + (int)threadIdx.x;
    uint8_t ok = 0;
    if (k < cols * rows) {
    const int y = k / cols, x = k - y * cols;
    const uint8_t* __restrict__ src = a.nextImage[lv];
    // cloud (independent of the rest)
    {
        const float z = a.lastDepth[lv][k];
        float* __restrict__ cloud3 = a.cloud[lv];
        cloud3[k * 3 + 0] = (x - a.cx[lv]) * z * a.fx_inv[lv];
        cloud3[k * 3 + 1] = (y - a.cy[lv]) * z * a.fy_inv[lv];
        cloud3[k * 3 + 2] = z;
    }
    // Sobel
    float dxv = 0, dyv = 0;
    constexpr float sx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    constexpr float sy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
    if (x >= 1 && y >= 1 && x <= cols - 2 && y <= rows - 2) {
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
        int kk = 8;
        for (int j = max(y - 1, 0); j <= min(y + 1, rows - 1); j++)
            for (int c = max(x - 1, 0); c <= min(x + 1, cols - 1); c++) {
                const float s = (float)src[j * cols + c];
                dxv += s * sx[kk];
                dyv += s * sy[kk];
                --kk;
            }
    }
    const int16_t dx16 = (int16_t)(int)dxv, dy16 = (int16_t)(int)dyv;
    a.dIdx[lv][k] = dx16; a.dIdy[lv][k] = dy16;
    // candidate mask
    if (x < cols - 5 && y < rows - 1) {
        bool valid = true;
        for (int u = max(y - 2, 0); u < min(y + 2, rows); u++)
            for (int v = max(x - 2, 0); v < min(x + 2, cols); v++) valid = valid && (src[u * cols + v] > 0);
        if (valid) {
            const int valx = dx16, valy = dy16;
            const float mTwo = (float)((valx * valx) + (valy * valy));
            if (mTwo >= a.minScale[lv] && !is_nan(a.nextDepth[lv][k])) ok = 1;
        }
    }
    a.cand[lv][k] = ok;
    }
    // Culled trackers (object models): the first and the last 256-pixel chunk of the level that holds a candidate at all.  The residual
    // workgroups of the Gauss-Newton loop are dealt the record slots between the two (rgb_residual_body) instead of the whole image's --
    // an object's mask is empty outside its prediction.  Two atomics per chunk WITH a candidate; words: ~first (0 = none seen), last + 1.
    if (a.res_range) {
        if (__syncthreads_or(ok)) {
            if (threadIdx.x == 0) { atomicMax(&a.res_range[2 * lv], ~(unsigned)lb); atomicMax(&a.res_range[2 * lv + 1], (unsigned)lb + 1u); }
        }
    }
}
}  // namespace cf
