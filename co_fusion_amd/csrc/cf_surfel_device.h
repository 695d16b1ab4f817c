// cf_surfel_device.h -- device helpers shared by the surfel kernels (surfel.hip).
//
// These restate the GLSL helper files of the reference (Core/Shaders/surfels.glsl,
// color_encoding.glsl, geometry.glsl) and pin what OpenGL leaves to the driver, exactly as the
// CPU oracle does (oracle/orc_surfel.c header): NEAREST = floor(u*size) clamped, LINEAR = f32
// bilinear weights from u*size-0.5 clamped to edge, exp/acos = fixed polynomial forms, GLSL
// round() = half away from zero, mat4*vec4 row-wise without FMA.
#pragma once

#include "cf_device.h"

namespace cf {

struct Surfel { float4 pos_conf, col_time, norm_rad; };  // 48 B, Core/Shaders/Vertex.cpp:21-43

__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int nearest_texel(float u, int size) { return iclamp((int)floorf(u * (float)size), 0, size - 1); }

__device__ __forceinline__ float det_expf(float x)
{
    if (!(x > -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.0f) return __int_as_float(0x7f800000);
    const float n = rintf(x * 1.44269504088896341f);
    float r = x - n * 0.693145751953125f;
    r = r - n * 1.42860682030941723212e-6f;
    float p = 1.0f / 720.0f;
    p = p * r + 1.0f / 120.0f;
    p = p * r + 1.0f / 24.0f;
    p = p * r + 1.0f / 6.0f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    return ldexpf(p, (int)n);
}
__device__ __forceinline__ float det_acos_r(float z)
{
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    const float p = z * (pS0 + z * (pS1 + z * pS2));
    const float q = 1.0f + z * qS1;
    return p / q;
}
__device__ __forceinline__ float det_acosf(float x)
{
    const float pio2 = 1.57079637050628662109375f, pi = 3.1415927410125732421875f;
    if (x != x || x > 1.0f || x < -1.0f) return qnan();
    if (fabsf(x) < 0.5f) return pio2 - (x + x * det_acos_r(x * x));
    if (x < 0.0f) {
        const float z = (1.0f + x) * 0.5f, s = sqrtf(z);
        return pi - 2.0f * (s + s * det_acos_r(z));
    }
    const float z = (1.0f - x) * 0.5f, s = sqrtf(z);
    return 2.0f * (s + s * det_acos_r(z));
}

// surfels.glsl:19-34
__device__ __forceinline__ float get_radius(float depth, float norm_z, float inv_fx, float inv_fy)
{
    const float meanFocal = ((1.0f / fabsf(inv_fx)) + (1.0f / fabsf(inv_fy))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    const float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius;
    radius_n = radius_n / fabsf(norm_z);
    radius_n = fminf(2.0f * radius, radius_n);
    return radius_n;
}
// surfels.glsl:36-46
__device__ __forceinline__ float confidence(float x, float y, float cx, float cy, float weighting)
{
    const float dx = x - cx, dy = y - cy;
    const float radialDist = sqrtf(dx * dx + dy * dy) / 400.0f;
    return det_expf((-(radialDist * radialDist) / 0.72f)) * weighting;
}
// color_encoding.glsl
__device__ __forceinline__ float glsl_round(float v) { return (v < 0) ? -floorf(-v + 0.5f) : floorf(v + 0.5f); }
__device__ __forceinline__ float encode_color(float r, float g, float b)
{
    int rgb = (int)glsl_round(r * 255.0f);
    rgb = (rgb << 8) + (int)glsl_round(g * 255.0f);
    rgb = (rgb << 8) + (int)glsl_round(b * 255.0f);
    return (float)rgb;
}
__device__ __forceinline__ f3 decode_color(float c)
{
    const int ci = (int)c;
    return f3{(float)((ci >> 16) & 0xFF) / 255.0f, (float)((ci >> 8) & 0xFF) / 255.0f, (float)(ci & 0xFF) / 255.0f};
}

struct Mat4 { float m[16]; };  // row-major
__device__ __forceinline__ f3 xform_point(const Mat4& T, f3 p)
{
    return f3{T.m[0] * p.x + T.m[1] * p.y + T.m[2] * p.z + T.m[3], T.m[4] * p.x + T.m[5] * p.y + T.m[6] * p.z + T.m[7],
              T.m[8] * p.x + T.m[9] * p.y + T.m[10] * p.z + T.m[11]};
}
__device__ __forceinline__ f3 xform_dir(const Mat4& T, f3 p)
{
    return f3{T.m[0] * p.x + T.m[1] * p.y + T.m[2] * p.z, T.m[4] * p.x + T.m[5] * p.y + T.m[6] * p.z,
              T.m[8] * p.x + T.m[9] * p.y + T.m[10] * p.z};
}

// LINEAR RGBA32F fetch, clamp to edge
__device__ __forceinline__ float4 tex4_linear(const float4* __restrict__ img, int cols, int rows, float u, float v)
{
    const float fu = u * (float)cols - 0.5f, fv = v * (float)rows - 0.5f;
    const float x0f = floorf(fu), y0f = floorf(fv);
    const float wx = fu - x0f, wy = fv - y0f;
    const int x0 = iclamp((int)x0f, 0, cols - 1), x1 = iclamp((int)x0f + 1, 0, cols - 1);
    const int y0 = iclamp((int)y0f, 0, rows - 1), y1 = iclamp((int)y0f + 1, 0, rows - 1);
    const float4 a = img[y0 * cols + x0], b = img[y0 * cols + x1], c = img[y1 * cols + x0], d = img[y1 * cols + x1];
    float4 r;
    {
        const float top = a.x * (1.0f - wx) + b.x * wx, bot = c.x * (1.0f - wx) + d.x * wx;
        r.x = top * (1.0f - wy) + bot * wy;
    }
    {
        const float top = a.y * (1.0f - wx) + b.y * wx, bot = c.y * (1.0f - wx) + d.y * wx;
        r.y = top * (1.0f - wy) + bot * wy;
    }
    {
        const float top = a.z * (1.0f - wx) + b.z * wx, bot = c.z * (1.0f - wx) + d.z * wx;
        r.z = top * (1.0f - wy) + bot * wy;
    }
    {
        const float top = a.w * (1.0f - wx) + b.w * wx, bot = c.w * (1.0f - wx) + d.w * wx;
        r.w = top * (1.0f - wy) + bot * wy;
    }
    return r;
}

// geometry.glsl:19-37 (float x/y overloads, NEAREST neighbour fetches clamped to the edge)
__device__ __forceinline__ f3 get_vertex(const float* __restrict__ depth, int cols, int rows, int px, int py, float x, float y,
                                         float cx, float cy, float inv_fx, float inv_fy)
{
    const float z = depth[iclamp(py, 0, rows - 1) * cols + iclamp(px, 0, cols - 1)];
    return f3{(x - cx) * z * inv_fx, (y - cy) * z * inv_fy, z};
}
__device__ __forceinline__ f3 half_sum(f3 a, f3 b) { return f3{(a.x + b.x) / 2, (a.y + b.y) / 2, (a.z + b.z) / 2}; }
__device__ __forceinline__ f3 get_normal_central(f3 p, const float* __restrict__ depth, int cols, int rows, int px, int py, float x,
                                                 float y, float cx, float cy, float inv_fx, float inv_fy)
{
    const f3 xf = get_vertex(depth, cols, rows, px + 1, py, x + 1, y, cx, cy, inv_fx, inv_fy);
    const f3 xb = get_vertex(depth, cols, rows, px - 1, py, x - 1, y, cx, cy, inv_fx, inv_fy);
    const f3 yf = get_vertex(depth, cols, rows, px, py + 1, x, y + 1, cx, cy, inv_fx, inv_fy);
    const f3 yb = get_vertex(depth, cols, rows, px, py - 1, x, y - 1, cx, cy, inv_fx, inv_fy);
    const f3 del_x = half_sum(xb, p) - half_sum(xf, p);
    const f3 del_y = half_sum(yb, p) - half_sum(yf, p);
    return normalized(cross(del_x, del_y));
}

}  // namespace cf
