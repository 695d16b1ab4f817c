/*
 * orc_surfel.c -- CPU ORACLE for the surfel half of the hot path (projection, fusion, cleaning,
 * bootstrap, bilateral filter, fill-in).  TEST INFRASTRUCTURE ONLY (see orc.h).  PINNED bit-exactly against the reference's
 * own shader sources compiled to C++ (oracle/ref_shim/ref_gl.cpp, tests/test_cpu_refpin.py); the fixed-function GL
 * choices listed below are shared with that harness and remain this oracle's own.
 *
 * Restates the GLSL passes of the reference as plain loops:
 *   depth_bilateral_metric.frag                    -> orc_bilateral
 *   vertex_feedback.{vert,geom} + init_unstable    -> orc_vertex_feedback / orc_model_initialise
 *   index_map.{vert,frag}                          -> orc_predict_indices
 *   splat.vert + combo_splat.frag                  -> orc_combined_predict
 *   fill_{vertex,normal,rgb}.frag                  -> orc_fill_in
 *   data.{vert,geom,frag} + update.vert            -> orc_fuse
 *   copy_unstable.{vert,geom}                      -> orc_clean
 *
 * OpenGL leaves several things to the driver; this oracle PINS them:
 *   - depth test GL_LESS, ties keep the first primitive (lowest surfel index / earliest pixel in the
 *     column-major uv order of Model.cpp:166-170);
 *   - point -> pixel: floor(fx*X/Z + cx) (no NDC round trip, no sub-pixel snapping);
 *   - NEAREST sampling at normalised coordinate u: texel floor(u*size) clamped to the edge;
 *     LINEAR sampling: f32 bilinear weights from u*size - 0.5, clamp to edge;
 *   - which textures are LINEAR follows the `draw` constructor flag incl. the string-literal->bool
 *     accident of ModelProjection.cpp:28-38 (the sparse vertConf/colorTime/normRad maps are LINEAR,
 *     the integer index map cannot be and is NEAREST);
 *   - texture fetches addressed with float(cx)/cols in the bilateral filter hit texel cx;
 *   - mat4*vec4 is evaluated row-wise left to right without FMA; normalize(v) = v * (1/sqrt(v.v));
 *   - exp / acos are the fixed polynomial forms of orc_math.h; GLSL round() is half away from zero.
 */
#include "orc.h"
#include "orc_math.h"

#include <stdlib.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* texcoord of pixel column/row i as the reference builds it on the host (Model.cpp:166-170) */
static inline float tex_coord(int i, int size)
{
    return (float)((double)((float)i / (float)size) + 1.0 / (2.0 * (double)(float)size));
}

/* ------------------------------------------------------------------ sampling ---- */
static inline int nearest_texel(float u, int size) { return iclamp((int)floorf(u * (float)size), 0, size - 1); }

typedef struct { float v[4]; } f4;

static inline f4 tex4_linear(const float *img, int cols, int rows, float u, float v)
{
    const float fu = u * (float)cols - 0.5f, fv = v * (float)rows - 0.5f;
    const float x0f = floorf(fu), y0f = floorf(fv);
    const float wx = fu - x0f, wy = fv - y0f;
    const int x0 = iclamp((int)x0f, 0, cols - 1), x1 = iclamp((int)x0f + 1, 0, cols - 1);
    const int y0 = iclamp((int)y0f, 0, rows - 1), y1 = iclamp((int)y0f + 1, 0, rows - 1);
    const float *a = img + ((size_t)y0 * cols + x0) * 4, *b = img + ((size_t)y0 * cols + x1) * 4;
    const float *c = img + ((size_t)y1 * cols + x0) * 4, *d = img + ((size_t)y1 * cols + x1) * 4;
    f4 r;
    for (int k = 0; k < 4; k++) {
        const float top = a[k] * (1.0f - wx) + b[k] * wx;
        const float bot = c[k] * (1.0f - wx) + d[k] * wx;
        r.v[k] = top * (1.0f - wy) + bot * wy;
    }
    return r;
}

/* surfels.glsl:19-34 (cam.z/w are 1/fx, 1/fy) */
static inline float get_radius(float depth, float norm_z, float inv_fx, float inv_fy)
{
    const float meanFocal = ((1.0f / fabsf(inv_fx)) + (1.0f / fabsf(inv_fy))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    const float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius;
    radius_n = radius_n / fabsf(norm_z);
    radius_n = fminf(2.0f * radius, radius_n);
    return radius_n;
}

/* surfels.glsl:36-46 */
static inline float confidence(float x, float y, float cx, float cy, float weighting)
{
    const float maxRadDist = 400.0f, twoSigmaSquared = 0.72f;
    const float dx = x - cx, dy = y - cy;
    const float radialDist = sqrtf(dx * dx + dy * dy) / maxRadDist;
    return orc_expf((-(radialDist * radialDist) / twoSigmaSquared)) * weighting;
}

/* color_encoding.glsl */
static inline float glsl_round(float v) { return (v < 0) ? -floorf(-v + 0.5f) : floorf(v + 0.5f); }
static inline float encode_color(float r, float g, float b)
{
    int rgb = (int)glsl_round(r * 255.0f);
    rgb = (rgb << 8) + (int)glsl_round(g * 255.0f);
    rgb = (rgb << 8) + (int)glsl_round(b * 255.0f);
    return (float)rgb;
}
static inline void decode_color(float c, float out[3])
{
    const int ci = (int)c;
    out[0] = (float)((ci >> 16) & 0xFF) / 255.0f;
    out[1] = (float)((ci >> 8) & 0xFF) / 255.0f;
    out[2] = (float)(ci & 0xFF) / 255.0f;
}

/* inverse of a rigid/affine 4x4 (row-major f32): linear part by cofactors */
static void inv44f(const float a[16], float o[16])
{
    float L[9] = {a[0], a[1], a[2], a[4], a[5], a[6], a[8], a[9], a[10]}, Li[9];
    orc_inv33f(L, Li);
    for (int i = 0; i < 3; i++) {
        o[i * 4 + 0] = Li[i * 3 + 0]; o[i * 4 + 1] = Li[i * 3 + 1]; o[i * 4 + 2] = Li[i * 3 + 2];
        o[i * 4 + 3] = -(Li[i * 3 + 0] * a[3] + Li[i * 3 + 1] * a[7] + Li[i * 3 + 2] * a[11]);
    }
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}
void orc_inverse_pose(const float pose[16], float out[16]) { inv44f(pose, out); }

static inline orc_f3 xform_point(const float T[16], orc_f3 p)
{
    return orc_f3_make(T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3], T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7],
                       T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11]);
}
static inline orc_f3 xform_dir(const float T[16], orc_f3 p)
{
    return orc_f3_make(T[0] * p.x + T[1] * p.y + T[2] * p.z, T[4] * p.x + T[5] * p.y + T[6] * p.z,
                       T[8] * p.x + T[9] * p.y + T[10] * p.z);
}

/* ============================ bilateral filter ===================================
 * depth_bilateral_metric.frag:30-75, driven by CoFusion::filterDepth (CoFusion.cpp:567-574) */
void orc_bilateral(const float *depth, int cols, int rows, float maxD, float *out)
{
    const float sigma_space2_inv_half = 0.024691358f, sigma_color2_inv_half = 555.556f;
    const int R = 6, D = R * 2 + 1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const float value = depth[y * cols + x];
            if (value > maxD || value < 0.3f) { out[y * cols + x] = 0; continue; }
            const int tx = imin(x - D / 2 + D, cols), ty = imin(y - D / 2 + D, rows);
            float sum1 = 0, sum2 = 0;
            for (int cy = imax(y - D / 2, 0); cy < ty; ++cy)
                for (int cx = imax(x - D / 2, 0); cx < tx; ++cx) {
                    const float tmp = depth[cy * cols + cx];
                    const float space2 = ((float)x - (float)cx) * ((float)x - (float)cx) + ((float)y - (float)cy) * ((float)y - (float)cy);
                    const float color2 = (value - tmp) * (value - tmp);
                    const float weight = orc_expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
                    sum1 += tmp * weight;
                    sum2 += weight;
                }
            out[y * cols + x] = sum1 / sum2;
        }
}

/* ---- geometry.glsl (float x/y overloads: central differences) ---- */
static inline orc_f3 get_vertex(const float *depth, int cols, int rows, int px, int py, float x, float y, orc_cam cam, float inv_fx, float inv_fy)
{
    const float z = depth[iclamp(py, 0, rows - 1) * cols + iclamp(px, 0, cols - 1)];
    return orc_f3_make((x - cam.cx) * z * inv_fx, (y - cam.cy) * z * inv_fy, z);
}
static inline orc_f3 f3_half_sum(orc_f3 a, orc_f3 b) { return orc_f3_make((a.x + b.x) / 2, (a.y + b.y) / 2, (a.z + b.z) / 2); }
static inline orc_f3 get_normal_central(orc_f3 p, const float *depth, int cols, int rows, int px, int py, float x, float y, orc_cam cam, float inv_fx, float inv_fy)
{ /* geometry.glsl:25-37; the neighbour fetch is NEAREST at texcoord +- 1/cols -> texel px+-1, clamped */
    const orc_f3 xf = get_vertex(depth, cols, rows, px + 1, py, x + 1, y, cam, inv_fx, inv_fy);
    const orc_f3 xb = get_vertex(depth, cols, rows, px - 1, py, x - 1, y, cam, inv_fx, inv_fy);
    const orc_f3 yf = get_vertex(depth, cols, rows, px, py + 1, x, y + 1, cam, inv_fx, inv_fy);
    const orc_f3 yb = get_vertex(depth, cols, rows, px, py - 1, x, y - 1, cam, inv_fx, inv_fy);
    const orc_f3 del_x = orc_f3_sub(f3_half_sum(xb, p), f3_half_sum(xf, p));
    const orc_f3 del_y = orc_f3_sub(f3_half_sum(yb, p), f3_half_sum(yf, p));
    return orc_f3_normalized(orc_f3_cross(del_x, del_y));
}

/* ============================ frame-1 bootstrap ==================================
 * FeedbackBuffer::compute (FeedbackBuffer.cpp:78-128) + vertex_feedback.{vert,geom}: one vertex per
 * pixel in COLUMN-major order, kept when 0 < z <= maxDepth.  out must hold cols*rows*12 floats and is
 * zero-filled beyond the returned count (the reference's VBO is zero-initialised, :30-38). */
int orc_vertex_feedback(const uint8_t *rgba, const float *depth, int cols, int rows, orc_cam cam, int time, float maxDepth, float *out)
{
    const float inv_fx = (float)(1.0 / (double)cam.fx), inv_fy = (float)(1.0 / (double)cam.fy);
    int n = 0;
    memset(out, 0, sizeof(float) * 12 * (size_t)cols * rows);
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < rows; j++) {
            const float x = tex_coord(i, cols) * (float)cols, y = tex_coord(j, rows) * (float)rows;
            const orc_f3 p = get_vertex(depth, cols, rows, i, j, x, y, cam, inv_fx, inv_fy);
            if (p.z <= 0 || p.z > maxDepth) continue;
            const orc_f3 nrm = get_normal_central(p, depth, cols, rows, i, j, x, y, cam, inv_fx, inv_fy);
            const uint8_t *c = rgba + ((size_t)j * cols + i) * 4;
            float *o = out + (size_t)n * 12;
            o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = confidence(x, y, cam.cx, cam.cy, 1.0f);
            o[4] = encode_color((float)c[0] / 255.0f, (float)c[1] / 255.0f, (float)c[2] / 255.0f);
            o[5] = 0; o[6] = (float)c[2] / 255.0f; /* vColor.z keeps the blue channel (vertex_feedback.vert:49-66) */
            o[7] = (float)time;
            o[8] = nrm.x; o[9] = nrm.y; o[10] = nrm.z; o[11] = get_radius(p.z, nrm.z, inv_fx, inv_fy);
            n++;
        }
    return n;
}

/* Model::initialise (Model.cpp:227-272) + init_unstable.vert: attributes 0/1 from the RAW feedback,
 * attribute 2 from the FILTERED feedback at the same index (misaligned when the two compactions
 * differ -- reproduced literally); raw_count vertices are drawn. */
int orc_model_initialise(const float *raw_fb, int raw_count, const float *filtered_fb, float *surfels)
{
    for (int k = 0; k < raw_count; k++) {
        const float *r = raw_fb + (size_t)k * 12, *f = filtered_fb + (size_t)k * 12;
        float *o = surfels + (size_t)k * 12;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
        o[4] = r[4]; o[5] = 0; o[6] = 1; o[7] = r[7];
        o[8] = f[8]; o[9] = f[9]; o[10] = f[10]; o[11] = f[11];
    }
    return raw_count;
}

/* ============================ index map ==========================================
 * ModelProjection::predictIndices (ModelProjection.cpp:105-157) + index_map.vert:38-63 / .frag */
void orc_predict_indices(const float *surfels, int count, const float pose[16], orc_cam cam, int cols, int rows, float maxDepth,
                         int time, int timeDelta, uint32_t *index, float *vertConf4, float *colorTime4, float *normRad4)
{
    float t_inv[16];
    inv44f(pose, t_inv);
    const size_t N = (size_t)cols * rows;
    float *zbuf = malloc(sizeof(float) * N);
    for (size_t i = 0; i < N; i++) zbuf[i] = maxDepth; /* depth buffer cleared to 1.0 == z of maxDepth; GL_LESS */
    memset(index, 0, sizeof(uint32_t) * N);
    memset(vertConf4, 0, sizeof(float) * 4 * N); memset(colorTime4, 0, sizeof(float) * 4 * N); memset(normRad4, 0, sizeof(float) * 4 * N);
    for (int id = 0; id < count; id++) {
        const float *s = surfels + (size_t)id * 12;
        const orc_f3 ph = xform_point(t_inv, orc_f3_make(s[0], s[1], s[2]));
        if (ph.z > maxDepth || ph.z < 0 || (float)time - s[7] > (float)timeDelta) continue;
        const float u = ((cam.fx * ph.x) / ph.z) + cam.cx, v = ((cam.fy * ph.y) / ph.z) + cam.cy;
        if (!(u >= 0.0f && v >= 0.0f && u < (float)cols && v < (float)rows)) continue; /* clipped; NaN (z == 0) too */
        const int px = (int)floorf(u), py = (int)floorf(v);
        const size_t q = (size_t)py * cols + px;
        if (!(ph.z < zbuf[q])) continue; /* GL_LESS: first (lowest id) wins ties */
        zbuf[q] = ph.z;
        index[q] = (uint32_t)id;
        const orc_f3 n = orc_f3_normalized(xform_dir(t_inv, orc_f3_make(s[8], s[9], s[10])));
        float *a = vertConf4 + q * 4, *b = colorTime4 + q * 4, *c = normRad4 + q * 4;
        a[0] = ph.x; a[1] = ph.y; a[2] = ph.z; a[3] = s[3];
        b[0] = s[4]; b[1] = s[5]; b[2] = s[6]; b[3] = s[7];
        c[0] = n.x; c[1] = n.y; c[2] = n.z; c[3] = s[11];
    }
    free(zbuf);
}

/* ============================ splat prediction ===================================
 * ModelProjection::combinedPredict (ModelProjection.cpp:192-273) + splat.vert:54-88 + combo_splat.frag:37-65.
 * A point sprite of size s centred at (u,v) covers the pixels whose centres lie in [u-s/2, u+s/2). */
void orc_combined_predict(const float *surfels, int count, const float pose[16], orc_cam cam, int cols, int rows, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, uint8_t *image_rgba, float *vertexConf4,
                          float *normalRad4, uint16_t *time16)
{
    float t_inv[16];
    inv44f(pose, t_inv);
    const size_t N = (size_t)cols * rows;
    float *zbuf = malloc(sizeof(float) * N);
    for (size_t i = 0; i < N; i++) zbuf[i] = maxDepth; /* depth buffer cleared to 1.0 == z of maxDepth; GL_LESS */
    memset(image_rgba, 0, 4 * N); memset(vertexConf4, 0, 16 * N); memset(normalRad4, 0, 16 * N); memset(time16, 0, 2 * N);
    for (int id = 0; id < count; id++) {
        const float *s = surfels + (size_t)id * 12;
        const orc_f3 ph = xform_point(t_inv, orc_f3_make(s[0], s[1], s[2]));
        if (ph.z > maxDepth || ph.z < 0 || s[3] < confThreshold || (float)time - s[7] > (float)timeDelta || s[7] > (float)maxTime) continue;
        const orc_f3 n = orc_f3_normalized(xform_dir(t_inv, orc_f3_make(s[8], s[9], s[10])));
        const float rad = s[11];
        const orc_f3 x1n = orc_f3_normalized(orc_f3_make(n.y - n.z, -n.x, n.x));
        const orc_f3 x1 = orc_f3_make(x1n.x * rad * 1.41421356f, x1n.y * rad * 1.41421356f, x1n.z * rad * 1.41421356f);
        const orc_f3 y1 = orc_f3_cross(n, x1);
        const orc_f3 c[4] = {orc_f3_add(ph, x1), orc_f3_add(ph, y1), orc_f3_sub(ph, y1), orc_f3_sub(ph, x1)};
        float px_[4], py_[4];
        for (int k = 0; k < 4; k++) { px_[k] = ((cam.fx * c[k].x) / c[k].z) + cam.cx; py_[k] = ((cam.fy * c[k].y) / c[k].z) + cam.cy; }
        const float xmin = fminf(px_[0], fminf(px_[1], fminf(px_[2], px_[3]))), xmax = fmaxf(px_[0], fmaxf(px_[1], fmaxf(px_[2], px_[3])));
        const float ymin = fminf(py_[0], fminf(py_[1], fminf(py_[2], py_[3]))), ymax = fmaxf(py_[0], fmaxf(py_[1], fmaxf(py_[2], py_[3])));
        const float size = fmaxf(0.0f, fmaxf(fabsf(xmax - xmin), fabsf(ymax - ymin)));
        if (!(size > 0.0f) || !(size <= 4096.0f)) continue; /* degenerate / NaN sprites draw nothing */
        const float u = ((cam.fx * ph.x) / ph.z) + cam.cx, v = ((cam.fy * ph.y) / ph.z) + cam.cy;
        if (!(u >= 0.0f && v >= 0.0f && u <= (float)cols && v <= (float)rows)) continue; /* points are clipped by their centre */
        const float half = size * 0.5f;
        /* pixel centre px+0.5 in [u-half, u+half) */
        const int x_lo = imax((int)ceilf(u - half - 0.5f), 0), x_hi = imin((int)ceilf(u + half - 0.5f) - 1, cols - 1);
        const int y_lo = imax((int)ceilf(v - half - 0.5f), 0), y_hi = imin((int)ceilf(v + half - 0.5f) - 1, rows - 1);
        const float sqrRad = rad * rad;
        const float pn = orc_f3_dot(ph, n);
        float col[3];
        decode_color(s[4], col);
        for (int py = y_lo; py <= y_hi; py++)
            for (int px = x_lo; px <= x_hi; px++) {
                const float fx_ = (float)px + 0.5f, fy_ = (float)py + 0.5f;
                const orc_f3 l = orc_f3_normalized(orc_f3_make((fx_ - cam.cx) / cam.fx, (fy_ - cam.cy) / cam.fy, 1.0f));
                const float k = pn / orc_f3_dot(l, n);
                const orc_f3 cp = orc_f3_make(k * l.x, k * l.y, k * l.z);
                const orc_f3 diff = orc_f3_sub(cp, ph);
                if (!(orc_f3_dot(diff, diff) <= sqrRad)) continue; /* discard (NaN discards too) */
                const size_t q = (size_t)py * cols + px;
                if (!(cp.z < zbuf[q])) continue; /* gl_FragDepth = z/(2 maxDepth)+0.5, GL_LESS */
                zbuf[q] = cp.z;
                uint8_t *im = image_rgba + q * 4;
                im[0] = (uint8_t)glsl_round(col[0] * 255.0f); im[1] = (uint8_t)glsl_round(col[1] * 255.0f);
                im[2] = (uint8_t)glsl_round(col[2] * 255.0f); im[3] = 255;
                float *vc = vertexConf4 + q * 4, *nr = normalRad4 + q * 4;
                const float z = cp.z;
                vc[0] = (fx_ - cam.cx) * z * (1.f / cam.fx); vc[1] = (fy_ - cam.cy) * z * (1.f / cam.fy); vc[2] = z; vc[3] = s[3];
                nr[0] = n.x; nr[1] = n.y; nr[2] = n.z; nr[3] = rad;
                time16[q] = (uint16_t)(uint32_t)s[6];
            }
    }
    free(zbuf);
}

/* ============================ fill-in ============================================
 * Model::performFillIn (Model.cpp:901-909) + fill_{vertex,normal,rgb}.frag; rawDepth is the FILTERED
 * depth (CoFusion.cpp:541).  In place on copies: outputs are the FillIn textures. */
void orc_fill_in(const float *pred_vertex4, const float *pred_normal4, const uint8_t *pred_image, const float *depth,
                 const uint8_t *rgba, int cols, int rows, orc_cam cam, int passthrough_geom, int passthrough_rgb,
                 float *out_vertex4, float *out_normal4, uint8_t *out_image)
{
    const float inv_fx = (float)(1.0 / (double)cam.fx), inv_fy = (float)(1.0 / (double)cam.fy);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const size_t q = (size_t)y * cols + x;
            /* vertex */
            if (pred_vertex4[q * 4 + 2] == 0 || passthrough_geom) {
                const float z = depth[q];
                out_vertex4[q * 4 + 0] = ((float)x - cam.cx) * z * inv_fx; out_vertex4[q * 4 + 1] = ((float)y - cam.cy) * z * inv_fy;
                out_vertex4[q * 4 + 2] = z; out_vertex4[q * 4 + 3] = 1;
            } else
                memcpy(out_vertex4 + q * 4, pred_vertex4 + q * 4, 16);
            /* normal: forward differences on integer pixel coordinates (geometry.glsl:39-58) */
            if (pred_normal4[q * 4 + 2] == 0 || passthrough_geom) {
                const float z = depth[q];
                const orc_f3 p = orc_f3_make(((float)x - cam.cx) * z * inv_fx, ((float)y - cam.cy) * z * inv_fy, z);
                const float zx = depth[(size_t)y * cols + imin(x + 1, cols - 1)], zy = depth[(size_t)imin(y + 1, rows - 1) * cols + x];
                const orc_f3 vx = orc_f3_make(((float)(x + 1) - cam.cx) * zx * inv_fx, ((float)y - cam.cy) * zx * inv_fy, zx);
                const orc_f3 vy = orc_f3_make(((float)x - cam.cx) * zy * inv_fx, ((float)(y + 1) - cam.cy) * zy * inv_fy, zy);
                const orc_f3 nn = orc_f3_normalized(orc_f3_cross(orc_f3_sub(vx, p), orc_f3_sub(vy, p)));
                out_normal4[q * 4 + 0] = nn.x; out_normal4[q * 4 + 1] = nn.y; out_normal4[q * 4 + 2] = nn.z; out_normal4[q * 4 + 3] = 1;
            } else
                memcpy(out_normal4 + q * 4, pred_normal4 + q * 4, 16);
            /* image */
            const uint8_t *e = pred_image + q * 4;
            if (((int)e[0] + (int)e[1] + (int)e[2]) == 0 || passthrough_rgb) memcpy(out_image + q * 4, rgba + q * 4, 4);
            else memcpy(out_image + q * 4, e, 4);
        }
}

/* CoFusion::requiresFillIn (CoFusion.cpp:547-565): NEAREST 20x down-sample, count fully non-zero pixels */
int orc_requires_fill_in(const uint8_t *pred_image, int cols, int rows, float ratio)
{
    const int dw = cols / 20, dh = rows / 20;
    int sum = 0;
    for (int j = 0; j < dh; j++)
        for (int i = 0; i < dw; i++) {
            const int sx = nearest_texel(((float)i + 0.5f) / (float)dw, cols), sy = nearest_texel(((float)j + 0.5f) / (float)dh, rows);
            const uint8_t *p = pred_image + ((size_t)sy * cols + sx) * 4;
            sum += (p[0] > 0 && p[1] > 0 && p[2] > 0);
        }
    return (float)sum / (float)(dh * dw) < ratio;
}

/* ============================ fusion =============================================
 * Model::fuse (Model.cpp:408-563): data association (data.vert:78-211) then update (update.vert:38-111).
 * new_unstable receives every emitted vertex with updateId == 2 in column-major pixel order (merge
 * stubs, which clean() always drops because colour.w == -1, are not materialised). */
static float angle_between(orc_f3 a, orc_f3 b)
{
    return orc_acosf(orc_f3_dot(a, b) / (orc_f3_norm(a) * orc_f3_norm(b)));
}

void orc_fuse(const float *surfels_in, int count, const uint32_t *index, const float *vertConf4, const float *normRad4,
              const uint8_t *rgba, const float *depth_raw, const float *depth_filt, const uint8_t *mask, const float pose[16],
              orc_cam cam, int cols, int rows, int time, float weighting, int maskID, float maxDepth, float *surfels_out,
              float *new_unstable, int *n_new)
{
    const float inv_fx = (float)(1.0 / (double)cam.fx), inv_fy = (float)(1.0 / (double)cam.fy);
    const float scale = 1.0f; /* ModelProjection::FACTOR */
    int *owner = malloc(sizeof(int) * (size_t)imax(count, 1));
    float *records = malloc(sizeof(float) * 12 * (size_t)cols * rows);
    for (int i = 0; i < count; i++) owner[i] = -1;
    int nn = 0;
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < rows; j++) { /* column-major draw order, Model.cpp:166-170 */
            const float tcx = tex_coord(i, cols), tcy = tex_coord(j, rows);
            const float x = tcx * (float)cols, y = tcy * (float)rows;
            const orc_f3 vPosLocal = get_vertex(depth_raw, cols, rows, i, j, x, y, cam, inv_fx, inv_fy);
            if (!(((int)x % 2 == time % 2) && ((int)y % 2 == time % 2))) continue;
            if ((int)mask[j * cols + i] != maskID) continue;
            /* checkNeighbours on the RAW depth (data.vert:56-74) */
            if (depth_raw[j * cols + iclamp(i - 1, 0, cols - 1)] == 0 || depth_raw[iclamp(j - 1, 0, rows - 1) * cols + i] == 0 ||
                depth_raw[j * cols + iclamp(i + 1, 0, cols - 1)] == 0 || depth_raw[iclamp(j + 1, 0, rows - 1) * cols + i] == 0)
                continue;
            if (!(vPosLocal.z > 0 && vPosLocal.z <= maxDepth)) continue;

            const orc_f3 vPos = xform_point(pose, vPosLocal);
            const orc_f3 vPos_f = get_vertex(depth_filt, cols, rows, i, j, x, y, cam, inv_fx, inv_fy);
            const uint8_t *c = rgba + ((size_t)j * cols + i) * 4;
            const orc_f3 vNormLocal = get_normal_central(vPos_f, depth_filt, cols, rows, i, j, x, y, cam, inv_fx, inv_fy);
            const orc_f3 nG = xform_dir(pose, vNormLocal);
            const float radius = get_radius(vPos_f.z, vNormLocal.z, inv_fx, inv_fy);
            const float conf = confidence(x, y, cam.cx, cam.cy, weighting);

            const float indexXStep = (1.0f / ((float)cols * scale)) * 0.5f, indexYStep = (1.0f / ((float)rows * scale)) * 0.5f;
            float bestDist = 1000;
            const float windowMultiplier = 2;
            const float xl = (x - cam.cx) * inv_fx, yl = (y - cam.cy) * inv_fy;
            const float lambda = sqrtf(xl * xl + yl * yl + 1);
            const orc_f3 ray = orc_f3_make(xl, yl, 1);
            uint32_t best = 0; int operation = 0;
            for (float ii = tcx - (scale * indexXStep * windowMultiplier); ii < tcx + (scale * indexXStep * windowMultiplier); ii += indexXStep)
                for (float jj = tcy - (scale * indexYStep * windowMultiplier); jj < tcy + (scale * indexYStep * windowMultiplier); jj += indexYStep) {
                    const uint32_t current = index[(size_t)nearest_texel(jj, rows) * cols + nearest_texel(ii, cols)];
                    if (current > 0U) {
                        const f4 vertConf = tex4_linear(vertConf4, cols, rows, ii, jj);
                        const float zdiff = (vertConf.v[2] - vPosLocal.z);
                        if (fabsf(zdiff * lambda) < 0.05f) {
                            const float dist = orc_f3_norm(orc_f3_cross(ray, orc_f3_make(vertConf.v[0], vertConf.v[1], vertConf.v[2])));
                            const f4 normRad = tex4_linear(normRad4, cols, rows, ii, jj);
                            if (dist < bestDist && (fabsf(normRad.v[2]) < 0.75f ||
                                                    fabsf(angle_between(orc_f3_make(normRad.v[0], normRad.v[1], normRad.v[2]), vNormLocal)) < 0.5f)) {
                                operation = 1; bestDist = dist; best = current;
                            }
                        }
                    }
                }
            float rec[12] = {vPos.x, vPos.y, vPos.z, conf,
                             (float)(((int)c[0] << 16) + ((int)c[1] << 8) + (int)c[2]), 0, (float)time, 0,
                             nG.x, nG.y, nG.z, radius};
            if (operation == 1) {
                rec[7] = -1;
                const int rank = i * rows + j;
                if (owner[best] < 0) { owner[best] = rank; memcpy(records + (size_t)rank * 12, rec, sizeof(rec)); } /* first wins */
            } else {
                rec[7] = -2;
                memcpy(new_unstable + (size_t)nn * 12, rec, sizeof(rec));
                nn++;
            }
        }
    *n_new = nn;
    /* update.vert:38-111 */
    for (int id = 0; id < count; id++) {
        const float *s = surfels_in + (size_t)id * 12;
        float *o = surfels_out + (size_t)id * 12;
        if (owner[id] < 0) { memcpy(o, s, 48); continue; }
        const float *r = records + (size_t)owner[id] * 12;
        const float c_k = s[3], a = r[3];
        if (r[11] < (1.0f + 0.5f) * s[11]) {
            for (int k = 0; k < 3; k++) o[k] = ((c_k * s[k]) + (a * r[k])) / (c_k + a);
            o[3] = c_k + a;
            float oc[3], nc[3];
            decode_color(s[4], oc); decode_color(r[4], nc);
            o[4] = encode_color(((c_k * oc[0]) + (a * nc[0])) / (c_k + a), ((c_k * oc[1]) + (a * nc[1])) / (c_k + a),
                                ((c_k * oc[2]) + (a * nc[2])) / (c_k + a));
            o[5] = s[5]; o[6] = s[6]; o[7] = (float)time;
            float nr[4];
            for (int k = 0; k < 4; k++) nr[k] = ((c_k * s[8 + k]) + (a * r[8 + k])) / (c_k + a);
            const orc_f3 nn3 = orc_f3_normalized(orc_f3_make(nr[0], nr[1], nr[2]));
            o[8] = nn3.x; o[9] = nn3.y; o[10] = nn3.z; o[11] = nr[3];
        } else {
            memcpy(o, s, 48);
            o[3] = c_k + a;
            o[7] = (float)time;
        }
    }
    free(records); free(owner);
}

/* ============================ clean ==============================================
 * Model::clean (Model.cpp:565-697) + copy_unstable.vert:53-149 (deformation-graph branch is dead:
 * nodes == 0) + copy_unstable.geom: ordered stream compaction of old surfels then appended ones. */
static int clean_one(float *s /* in/out 12 */, const float t_inv[16], orc_cam cam, int cols, int rows, int time, float confThreshold,
                     float outlierCoeff, int timeDelta, int maskID, const uint32_t *index, const float *vertConf4,
                     const float *colorTime4, const float *depth_filt, const uint8_t *mask)
{
    const float scale = 1.0f;
    int test = 1;
    const orc_f3 localPos = xform_point(t_inv, orc_f3_make(s[0], s[1], s[2]));
    const float x = ((cam.fx * localPos.x) / localPos.z) + cam.cx, y = ((cam.fy * localPos.y) / localPos.z) + cam.cy;
    const orc_f3 localNorm = orc_f3_normalized(xform_dir(t_inv, orc_f3_make(s[8], s[9], s[10])));
    const float x_n = x / (float)cols, y_n = y / (float)rows;
    const float stepX = 1.0f / (float)cols, stepY = 1.0f / (float)rows;
    const float indexXStep = stepX * 0.5f / scale, indexYStep = stepY * 0.5f / scale;
    const float windowMultiplier = 2;
    int count = 0, zCount = 0, violationCount = 0;
    float avgViolation = 0;
    if ((float)time - s[7] < (float)timeDelta && localPos.z > 0 && x > 0 && y > 0 && x < (float)cols && y < (float)rows) {
        for (float i = x_n - (scale * indexXStep * windowMultiplier); i < x_n + (scale * indexXStep * windowMultiplier); i += indexXStep)
            for (float j = y_n - (scale * indexYStep * windowMultiplier); j < y_n + (scale * indexYStep * windowMultiplier); j += indexYStep) {
                const uint32_t current = index[(size_t)nearest_texel(j, rows) * cols + nearest_texel(i, cols)];
                if (current > 0U) {
                    const f4 vertConf = tex4_linear(vertConf4, cols, rows, i, j);
                    const f4 colorTime = tex4_linear(colorTime4, cols, rows, i, j);
                    const float dx = vertConf.v[0] - localPos.x, dy = vertConf.v[1] - localPos.y;
                    if (colorTime.v[2] < s[6] && vertConf.v[3] > confThreshold && vertConf.v[2] > localPos.z &&
                        vertConf.v[2] - localPos.z < 0.01f && sqrtf(dx * dx + dy * dy) < s[11] * 1.4f)
                        count++;
                    if (colorTime.v[3] == (float)time && vertConf.v[3] > confThreshold && vertConf.v[2] > localPos.z &&
                        vertConf.v[2] - localPos.z > 0.01f && fabsf(localNorm.z) > 0.85f)
                        zCount++;
                }
            }
        for (float i = x_n - stepX; i <= x_n + stepX; i += stepX)
            for (float j = y_n - stepY; j <= y_n + stepY; j += stepY) {
                const float d = depth_filt[(size_t)nearest_texel(j, rows) * cols + nearest_texel(i, cols)] - localPos.z;
                if (d > 0.03f) { violationCount++; avgViolation += d; }
            }
    }
    if (count > 8 || zCount > 4) test = 0;
    if (s[7] == -2) s[7] = (float)time;
    if ((s[7] == -1 || (((float)time - s[7]) > 20 && s[3] < confThreshold))) test = 0;
    if (s[7] > 0 && (float)time - s[7] > (float)timeDelta) test = 1;
    if (violationCount > 0) {
        avgViolation /= (float)violationCount;
        s[3] *= 1.0f / (1 + outlierCoeff * avgViolation);
        const int mx = nearest_texel(x_n, cols), my = nearest_texel(y_n, rows);
        const int maskValue = mask[(size_t)my * cols + mx];
        const float wDepth = depth_filt[(size_t)my * cols + mx];
        if (maskValue != maskID && (wDepth > localPos.z - 0.05f && wDepth < localPos.z + 0.05f))
            s[3] *= (0.5f + 0.5f * (1 - outlierCoeff / 10.0f));
    }
    return test;
}

int orc_clean(const float *surfels_in, int count, const float *new_unstable, int n_new, const uint32_t *index, const float *vertConf4,
              const float *colorTime4, const float *depth_filt, const uint8_t *mask, const float pose[16], orc_cam cam, int cols,
              int rows, int time, float confThreshold, float outlierCoeff, int timeDelta, int maskID, float *surfels_out)
{
    float t_inv[16];
    inv44f(pose, t_inv);
    int n = 0;
    for (int pass = 0; pass < 2; pass++) {
        const float *src = pass ? new_unstable : surfels_in;
        const int m = pass ? n_new : count;
        for (int k = 0; k < m; k++) {
            float s[12];
            memcpy(s, src + (size_t)k * 12, 48);
            if (clean_one(s, t_inv, cam, cols, rows, time, confThreshold, outlierCoeff, timeDelta, maskID, index, vertConf4, colorTime4,
                          depth_filt, mask)) {
                memcpy(surfels_out + (size_t)n * 12, s, 48);
                n++;
            }
        }
    }
    return n;
}

/* ============================ fusion weight ======================================
 * Model::computeFusionWeight (Model.cpp:391-406) + Model::rodrigues2 (Model.cpp:817-865).  The
 * JacobiSVD re-orthonormalisation (:818-819) is pinned to the identity: the inputs are products of
 * rotations and already orthonormal to f32 rounding. */
float orc_fusion_weight(const float pose[16], const float lastPose[16], float weightMultiplier)
{
    float pinv[16], diff[16];
    inv44f(pose, pinv);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += pinv[i * 4 + k] * lastPose[k * 4 + j];
            diff[i * 4 + j] = s;
        }
    const float tn = sqrtf(diff[3] * diff[3] + diff[7] * diff[7] + diff[11] * diff[11]);
    const float *R = diff; /* R(r,c) = diff[r*4+c] */
    double rx = R[2 * 4 + 1] - R[1 * 4 + 2], ry = R[0 * 4 + 2] - R[2 * 4 + 0], rz = R[1 * 4 + 0] - R[0 * 4 + 1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (double)((R[0] + R[5] + R[10]) - 1.0f) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0) rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5; rx = sqrt(t > 0.0 ? t : 0.0);
            t = (R[5] + 1) * 0.5; ry = sqrt(t > 0.0 ? t : 0.0) * (R[0 * 4 + 1] < 0 ? -1.0 : 1.0);
            t = (R[10] + 1) * 0.5; rz = sqrt(t > 0.0 ? t : 0.0) * (R[0 * 4 + 2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[1 * 4 + 2] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    const float rv[3] = {(float)rx, (float)ry, (float)rz};
    const float rn = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    float weighting = tn > rn ? tn : rn;
    const float largest = 0.01f, minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    const float w = 1.0f - (weighting / largest);
    weighting = (w > minWeight ? w : minWeight) * weightMultiplier;
    return weighting;
}
