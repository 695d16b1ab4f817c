/*
 * orc_track.c -- CPU ORACLE for the dense tracking half of the hot path.
 * TEST INFRASTRUCTURE ONLY (see orc.h).  Kernels PINNED against the reference's own reduce.cu / cudafuncs.cu run on the CPU
 * (oracle/ref_shim, tests/test_cpu_refpin.py); the host Gauss-Newton loop PINNED against the reference's own RGBDOdometry class compiled
 * with a stand-in for Eigen whose rounding conventions are stated, not verified (poses within 5e-6, counts identical; ref_odo_v1.npz).
 *
 * Restates, function by function:
 *   Core/Cuda/cudafuncs.cu   map preparation kernels
 *   Core/Cuda/reduce.cu      icpStep / computeRgbResidual / rgbStep / so3Step
 *   Core/Utils/RGBDOdometry.cpp  the Gauss-Newton host loop
 * Per-pixel arithmetic is IEEE f32 in the reference's operation order (the
 * reference itself is built with --prec-div=false --ftz, Core/CMakeLists.txt:90,
 * so bitwise equality with a CUDA build was never defined).  The normal-equation
 * sums are accumulated exactly in fixed point (orc_math.h: orc_fix_prod), which is
 * the order-independent statement the HIP kernels are held to bit-for-bit; the
 * reference's own f32 tree order is available as orc_icp_step_f32tree to bound the
 * reassociation spread.
 */
#include "orc.h"
#include "orc_math.h"

#include <float.h>
#include <stdlib.h>

/* OpenMP only where a loop covers a full-size image: on the small pyramid levels (and in the small-image tests) a parallel region
 * costs more than the loop, and on a box whose cgroup grants fewer cores than it shows it costs a lot more */
#define ORC_OMP_MIN_PIXELS 65536

static inline orc_cam cam_level(orc_cam c, int level)
{ /* CameraModel::operator(), types.cuh:94-98 */
    int div = 1 << level;
    orc_cam r = {c.fx / div, c.fy / div, c.cx / div, c.cy / div};
    return r;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ============================ map preparation ================================= */

/* computeVmapKernel, cudafuncs.cu:109-134.  The mask test is commented out in the
 * reference (:119); invalid pixels write NaN to the x plane ONLY (:131). */
void orc_create_vmap(const float *depth, int cols, int rows, orc_cam intr, float depth_cutoff, float *vmap)
{
    const float fx_inv = 1.f / intr.fx, fy_inv = 1.f / intr.fy;
#pragma omp parallel for schedule(static) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int v = 0; v < rows; v++)
        for (int u = 0; u < cols; u++) {
            float z = depth[v * cols + u];
            if (z != 0 && z < depth_cutoff) {
                vmap[v * cols + u] = z * (u - intr.cx) * fx_inv;
                vmap[(v + rows) * cols + u] = z * (v - intr.cy) * fy_inv;
                vmap[(v + 2 * rows) * cols + u] = z;
            } else {
                vmap[v * cols + u] = orc_qnan();
            }
        }
}

/* computeNmapKernel, cudafuncs.cu:152-189 */
void orc_create_nmap(const float *vmap, int cols, int rows, float *nmap)
{
#pragma omp parallel for schedule(static) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int v = 0; v < rows; v++)
        for (int u = 0; u < cols; u++) {
            if (u == cols - 1 || v == rows - 1) { nmap[v * cols + u] = orc_qnan(); continue; }
            float x00 = vmap[v * cols + u], x01 = vmap[v * cols + u + 1], x10 = vmap[(v + 1) * cols + u];
            if (!isnan(x00) && !isnan(x01) && !isnan(x10)) {
                orc_f3 v00 = {x00, vmap[(v + rows) * cols + u], vmap[(v + 2 * rows) * cols + u]};
                orc_f3 v01 = {x01, vmap[(v + rows) * cols + u + 1], vmap[(v + 2 * rows) * cols + u + 1]};
                orc_f3 v10 = {x10, vmap[(v + 1 + rows) * cols + u], vmap[(v + 1 + 2 * rows) * cols + u]};
                orc_f3 r = orc_f3_normalized(orc_f3_cross(orc_f3_sub(v01, v00), orc_f3_sub(v10, v00)));
                nmap[v * cols + u] = r.x;
                nmap[(v + rows) * cols + u] = r.y;
                nmap[(v + 2 * rows) * cols + u] = r.z;
            } else
                nmap[v * cols + u] = orc_qnan();
        }
}

/* copyMapsKernel, cudafuncs.cu:271-311: RGBA32F -> planar, z==0 -> NaN (all 3 planes) */
void orc_copy_maps(const float *v4, const float *n4, int cols, int rows, float *vmap, float *nmap)
{
#pragma omp parallel for schedule(static) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const float *vs = v4 + (y * cols + x) * 4, *ns = n4 + (y * cols + x) * 4;
            orc_f3 vd = {orc_qnan(), orc_qnan(), orc_qnan()}, nd = vd;
            if (!(vs[2] == 0)) { vd = orc_f3_make(vs[0], vs[1], vs[2]); nd = orc_f3_make(ns[0], ns[1], ns[2]); }
            vmap[y * cols + x] = vd.x; vmap[(y + rows) * cols + x] = vd.y; vmap[(y + 2 * rows) * cols + x] = vd.z;
            nmap[y * cols + x] = nd.x; nmap[(y + rows) * cols + x] = nd.y; nmap[(y + 2 * rows) * cols + x] = nd.z;
        }
}

/* resizeMapKernel<normalize>, cudafuncs.cu:366-417: 2x2 mean, NaN if any x is NaN
 * (x plane only is written in that case). */
void orc_resize_map(const float *in, int in_cols, int in_rows, float *out, int normalize)
{
    const int dcols = in_cols / 2, drows = in_rows / 2, srows = in_rows;
#pragma omp parallel for schedule(static) if (dcols * drows >= ORC_OMP_MIN_PIXELS)
    for (int y = 0; y < drows; y++)
        for (int x = 0; x < dcols; x++) {
            int xs = x * 2, ys = y * 2;
            float x00 = in[ys * in_cols + xs], x01 = in[ys * in_cols + xs + 1];
            float x10 = in[(ys + 1) * in_cols + xs], x11 = in[(ys + 1) * in_cols + xs + 1];
            if (isnan(x00) || isnan(x01) || isnan(x10) || isnan(x11)) { out[y * dcols + x] = orc_qnan(); continue; }
            orc_f3 n;
            n.x = (x00 + x01 + x10 + x11) / 4;
            n.y = (in[(ys + srows) * in_cols + xs] + in[(ys + srows) * in_cols + xs + 1] +
                   in[(ys + srows + 1) * in_cols + xs] + in[(ys + srows + 1) * in_cols + xs + 1]) / 4;
            n.z = (in[(ys + 2 * srows) * in_cols + xs] + in[(ys + 2 * srows) * in_cols + xs + 1] +
                   in[(ys + 2 * srows + 1) * in_cols + xs] + in[(ys + 2 * srows + 1) * in_cols + xs + 1]) / 4;
            if (normalize) n = orc_f3_normalized(n);
            out[y * dcols + x] = n.x; out[(y + drows) * dcols + x] = n.y; out[(y + 2 * drows) * dcols + x] = n.z;
        }
}

/* tranformMapsKernel, cudafuncs.cu:207-249 (in place; y/z planes untouched when x is NaN) */
void orc_transform_maps(float *vmap, float *nmap, int cols, int rows, const float R[9], const float t[3])
{
    orc_m33 Rm; memcpy(Rm.m, R, sizeof(Rm.m));
    const orc_f3 tv = {t[0], t[1], t[2]};
#pragma omp parallel for schedule(static) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float vx = vmap[y * cols + x];
            float outx = orc_qnan();
            if (!isnan(vx)) {
                orc_f3 vs = {vx, vmap[(y + rows) * cols + x], vmap[(y + 2 * rows) * cols + x]};
                orc_f3 vd = orc_f3_add(orc_m33_mul(&Rm, vs), tv);
                vmap[(y + rows) * cols + x] = vd.y; vmap[(y + 2 * rows) * cols + x] = vd.z; outx = vd.x;
            }
            vmap[y * cols + x] = outx;
            float nx = nmap[y * cols + x];
            outx = orc_qnan();
            if (!isnan(nx)) {
                orc_f3 ns = {nx, nmap[(y + rows) * cols + x], nmap[(y + 2 * rows) * cols + x]};
                orc_f3 nd = orc_m33_mul(&Rm, ns);
                nmap[(y + rows) * cols + x] = nd.y; nmap[(y + 2 * rows) * cols + x] = nd.z; outx = nd.x;
            }
            nmap[y * cols + x] = outx;
        }
}

/* verticesToDepthKernel, cudafuncs.cu:602-613 */
void orc_vertices_to_depth(const float *v4, int cols, int rows, float cutoff, float *depth)
{
#pragma omp parallel for schedule(static) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int i = 0; i < cols * rows; i++) {
        float z = v4[i * 4 + 2];
        depth[i] = (z > cutoff || z <= 0) ? orc_qnan() : z;
    }
}

static const float kGauss25[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1};

/* pyrDownKernelGaussF, cudafuncs.cu:333-364.  Window excludes the last source
 * row/col (min(.., rows-1)); weights are indexed from the clamped END of the
 * window; `count` is an int accumulating float weights (exact: they are integers). */
void orc_pyrdown_gauss_f32(const float *src, int src_cols, int src_rows, float *dst)
{
    const int dcols = src_cols / 2, drows = src_rows / 2, D = 5;
#pragma omp parallel for schedule(static) if (dcols * drows >= ORC_OMP_MIN_PIXELS)
    for (int y = 0; y < drows; y++)
        for (int x = 0; x < dcols; x++) {
            int tx = imin(2 * x - D / 2 + D, src_cols - 1);
            int ty = imin(2 * y - D / 2 + D, src_rows - 1);
            float sum = 0; int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    float s = src[cy * src_cols + cx];
                    if (!isnan(s)) {
                        float w = kGauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += s * w;
                        count += (int)w;
                    }
                }
            dst[y * dcols + x] = (float)(sum / (float)count);
        }
}

/* pyrDownKernelIntensityGauss, cudafuncs.cu:534-564 (skips zeros; float->u8 truncation) */
void orc_pyrdown_gauss_u8(const uint8_t *src, int src_cols, int src_rows, uint8_t *dst)
{
    const int dcols = src_cols / 2, drows = src_rows / 2, D = 5;
#pragma omp parallel for schedule(static) if (dcols * drows >= ORC_OMP_MIN_PIXELS)
    for (int y = 0; y < drows; y++)
        for (int x = 0; x < dcols; x++) {
            int tx = imin(2 * x - D / 2 + D, src_cols - 1);
            int ty = imin(2 * y - D / 2 + D, src_rows - 1);
            float sum = 0; int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    uint8_t s = src[cy * src_cols + cx];
                    if (s > 0) {
                        float w = kGauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += (float)s * w;
                        count += (int)w;
                    }
                }
            /* count==0 -> 0/0 = NaN -> CUDA float->uchar conversion gives 0 */
            float q = sum / (float)count;
            dst[y * dcols + x] = (q != q) ? 0 : (uint8_t)(int)q;
        }
}

/* bgr2IntensityKernel, cudafuncs.cu:626-639.  The texel is RGB ordered (upload
 * CoFusion.cpp:179), so the weights land as .114 R + .299 G + .587 B. */
void orc_rgba_to_intensity(const uint8_t *rgba, int cols, int rows, uint8_t *dst)
{
#pragma omp parallel for schedule(static) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int i = 0; i < cols * rows; i++) {
        int value = (int)((float)rgba[i * 4 + 0] * 0.114f + (float)rgba[i * 4 + 1] * 0.299f + (float)rgba[i * 4 + 2] * 0.587f);
        dst[i] = (uint8_t)value;
    }
}

/* applyKernel, cudafuncs.cu:658-683 + coefficients :691-697.  kernelIndex counts
 * DOWN from 8 over the clamped neighbourhood, so borders use a shifted subset. */
void orc_sobel(const uint8_t *src, int cols, int rows, int16_t *dx, int16_t *dy)
{
    static const float gsx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    static const float gsy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
#pragma omp parallel for schedule(static) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float dxv = 0, dyv = 0; int k = 8;
            for (int j = imax(y - 1, 0); j <= imin(y + 1, rows - 1); j++)
                for (int i = imax(x - 1, 0); i <= imin(x + 1, cols - 1); i++) {
                    dxv += (float)src[j * cols + i] * gsx[k];
                    dyv += (float)src[j * cols + i] * gsy[k];
                    --k;
                }
            dx[y * cols + x] = (int16_t)(int)dxv;   /* float -> short: truncation */
            dy[y * cols + x] = (int16_t)(int)dyv;
        }
}

/* projectPointsKernel, cudafuncs.cu:718-736 */
void orc_project_cloud(const float *depth, int cols, int rows, orc_cam il, float *cloud3)
{
    const float invFx = 1.0f / il.fx, invFy = 1.0f / il.fy;
#pragma omp parallel for schedule(static) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float z = depth[y * cols + x];
            cloud3[(y * cols + x) * 3 + 0] = (float)((x - il.cx) * z * invFx);
            cloud3[(y * cols + x) * 3 + 1] = (float)((y - il.cy) * z * invFy);
            cloud3[(y * cols + x) * 3 + 2] = z;
        }
}

/* Model::generateCUDATextures, Model.cpp:341-343 (depth half; the mask pyramid is
 * dead because createVMap ignores the mask, cudafuncs.cu:119) */
void orc_depth_pyramid(const float *depth_filtered, int cols, int rows, float *l1, float *l2)
{
    orc_pyrdown_gauss_f32(depth_filtered, cols, rows, l1);
    orc_pyrdown_gauss_f32(l1, cols / 2, rows / 2, l2);
}

/* ================================ ICP ========================================= */

typedef struct {
    orc_m33 Rcurr, Rprev_inv; orc_f3 tcurr, tprev; orc_cam intr;
    const float *vc, *nc, *vp, *np; float distThres, angleThres; int cols, rows;
} icp_ctx;

/* ICPReduction::search + getProducts, reduce.cu:283-394.  Returns found; row[7]
 * is zero when not found.  err: value for the error surface (reduce.cu:301,325). */
static int icp_row(const icp_ctx *c, int x, int y, float row[7], float *err)
{
    const int cols = c->cols, rows = c->rows;
    for (int i = 0; i < 7; i++) row[i] = 0;
    *err = 0.0f;
    orc_f3 vcurr = {c->vc[y * cols + x], c->vc[(y + rows) * cols + x], c->vc[(y + 2 * rows) * cols + x]};
    orc_f3 vcurr_g = orc_f3_add(orc_m33_mul(&c->Rcurr, vcurr), c->tcurr);
    orc_f3 vcurr_cp = orc_m33_mul(&c->Rprev_inv, orc_f3_sub(vcurr_g, c->tprev));
    int ux = orc_f2i_rn(vcurr_cp.x * c->intr.fx / vcurr_cp.z + c->intr.cx);
    int uy = orc_f2i_rn(vcurr_cp.y * c->intr.fy / vcurr_cp.z + c->intr.cy);
    if (ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcurr_cp.z < 0) return 0;
    orc_f3 vprev_g = {c->vp[uy * cols + ux], c->vp[(uy + rows) * cols + ux], c->vp[(uy + 2 * rows) * cols + ux]};
    orc_f3 ncurr = {c->nc[y * cols + x], c->nc[(y + rows) * cols + x], c->nc[(y + 2 * rows) * cols + x]};
    orc_f3 ncurr_g = orc_m33_mul(&c->Rcurr, ncurr);
    orc_f3 nprev_g = {c->np[uy * cols + ux], c->np[(uy + rows) * cols + ux], c->np[(uy + 2 * rows) * cols + ux]};
    float dist = orc_f3_norm(orc_f3_sub(vprev_g, vcurr_g));
    float sine = orc_f3_norm(orc_f3_cross(ncurr_g, nprev_g));
    *err = isfinite(dist) ? dist : 0.0f;
    int found = (sine < c->angleThres && dist <= c->distThres && !isnan(ncurr.x) && !isnan(nprev_g.x));
    if (found) {
        orc_f3 s_cp = orc_m33_mul(&c->Rprev_inv, orc_f3_sub(vcurr_g, c->tprev));
        orc_f3 d_cp = orc_m33_mul(&c->Rprev_inv, orc_f3_sub(vprev_g, c->tprev));
        orc_f3 n_cp = orc_m33_mul(&c->Rprev_inv, nprev_g);
        orc_f3 cr = orc_f3_cross(s_cp, n_cp);
        row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
        row[3] = cr.x; row[4] = cr.y; row[5] = cr.z;
        row[6] = orc_f3_dot(n_cp, orc_f3_sub(s_cp, d_cp));
    }
    return found;
}

static void icp_ctx_fill(icp_ctx *c, const float Rcurr[9], const float tcurr[3], const float *vc, const float *nc,
                         const float Rprev_inv[9], const float tprev[3], orc_cam intr, const float *vp, const float *np,
                         float dist_thres, float angle_thres, int cols, int rows)
{
    memcpy(c->Rcurr.m, Rcurr, 36); memcpy(c->Rprev_inv.m, Rprev_inv, 36);
    c->tcurr = orc_f3_make(tcurr[0], tcurr[1], tcurr[2]); c->tprev = orc_f3_make(tprev[0], tprev[1], tprev[2]);
    c->intr = intr; c->vc = vc; c->nc = nc; c->vp = vp; c->np = np;
    c->distThres = dist_thres; c->angleThres = angle_thres; c->cols = cols; c->rows = rows;
}

static int g_icp_arith = ORC_ICP_ARITH_PRODUCT;
void orc_set_icp_arith(int mode)
{
    g_icp_arith = mode == ORC_ICP_ARITH_GRAM ? ORC_ICP_ARITH_GRAM : mode == ORC_ICP_ARITH_REFERENCE ? ORC_ICP_ARITH_REFERENCE : ORC_ICP_ARITH_PRODUCT;
}
int orc_get_icp_arith(void) { return g_icp_arith; }

/* ORC_ICP_ARITH_GRAM: the same 29 words as the Gram matrix of the quantised rows (wrapping 64-bit sums like the product form) */
static void se3_accumulate_gram(const float row[7], int found, int64_t sums[ORC_SE3_WORDS])
{
    if (!found) return;
    int64_t q[7];
    for (int i = 0; i < 7; i++) q[i] = orc_gram_quant(row[i], i);
    int k = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 7; j++) sums[k++] += q[i] * q[j];
    sums[27] += q[6] * q[6];
    sums[28] += 1;
}

void orc_icp_sums_to_host(const int64_t sums[ORC_SE3_WORDS], float A[36], float b[6], float residual[2])
{
    if (g_icp_arith != ORC_ICP_ARITH_GRAM) { orc_se3_sums_to_host(sums, ORC_FIX_ICP, A, b, residual); return; }
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            float value = (float)orc_fix_to_double(sums[shift++], orc_gram_bits[i] + orc_gram_bits[j]);
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    residual[0] = (float)orc_fix_to_double(sums[27], 2 * orc_gram_bits[6]);
    residual[1] = (float)sums[28];
}

static void se3_accumulate(const float row[7], int found, int F, int64_t sums[ORC_SE3_WORDS])
{
    if (!found) return; /* row is all zero: contributes nothing */
    int k = 0;
    for (int i = 0; i < 6; i++)
        for (int j = i; j < 7; j++) sums[k++] += orc_fix_prod(row[i], row[j], F);
    sums[27] += orc_fix_prod(row[6], row[6], F);
    sums[28] += 1;
}

void orc_icp_step(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                  const float Rprev_inv[9], const float tprev[3], orc_cam intr, const float *vmap_g_prev,
                  const float *nmap_g_prev, float dist_thres, float angle_thres, int cols, int rows,
                  int64_t sums[ORC_SE3_WORDS], float *err_surface)
{
    icp_ctx c;
    icp_ctx_fill(&c, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev,
                 dist_thres, angle_thres, cols, rows);
    memset(sums, 0, sizeof(int64_t) * ORC_SE3_WORDS);
    const int gram = g_icp_arith == ORC_ICP_ARITH_GRAM;
    /* integer sums: the OpenMP reduction is exact and order independent */
#pragma omp parallel for schedule(static) reduction(+ : sums[:ORC_SE3_WORDS]) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float row[7], err;
            int found = icp_row(&c, x, y, row, &err);
            if (err_surface) err_surface[y * cols + x] = err;
            if (gram) se3_accumulate_gram(row, found, sums);
            else se3_accumulate(row, found, ORC_FIX_ICP, sums);
        }
}

/* The reference's own order: thread t of block b sums pixels b*T+t + k*T*B in f32,
 * then warp(32) shuffle-down tree, block tree over warps, then the second-stage
 * reduceSum<<<1,MAX_THREADS(=512 on the host pass)>>> (reduce.cu:90-185, 396-417, 476). */
static void tree29(float (*vals)[29], int n /* n<=32 lanes */)
{ /* shuffle-down with width 32: lanes >= n read 0-extended (values of missing lanes are 0) */
    for (int offset = 16; offset > 0; offset /= 2)
        for (int l = 0; l < 32; l++)
            for (int k = 0; k < 29; k++) {
                float other = (l + offset < 32 && l + offset < n) ? vals[l + offset][k] : 0.0f;
                /* __shfl_down past the warp returns the caller's own value; the reference's
                 * result only uses lane 0, whose partners are always in range. */
                if (l + offset < 32) vals[l][k] += other;
            }
}

static void block_reduce29(float (*thread_vals)[29], int threads, float out[29])
{
    float warp_tot[32][29];
    int nwarps = threads / 32;
    memset(warp_tot, 0, sizeof(warp_tot));
    for (int w = 0; w < nwarps; w++) {
        float lanes[32][29];
        memcpy(lanes, thread_vals + w * 32, sizeof(lanes));
        tree29(lanes, 32);
        memcpy(warp_tot[w], lanes[0], sizeof(float) * 29);
    }
    tree29(warp_tot, 32);
    memcpy(out, warp_tot[0], sizeof(float) * 29);
}

/* generic two-stage f32 reduction in the reference's order over a per-pixel producer of the K summed values */
typedef void (*orc_vals_fn)(const void *ctx, int i, float vals[29]);
static void f32tree_reduce(orc_vals_fn fn, const void *ctx, int N, int K, int threads, int blocks, float *out)
{
    float(*block_out)[29] = calloc((size_t)blocks, sizeof(float[29]));
    float(*tv)[29] = calloc((size_t)threads, sizeof(float[29]));
    for (int b = 0; b < blocks; b++) {
        memset(tv, 0, sizeof(float[29]) * (size_t)threads);
        for (int t = 0; t < threads; t++)
            for (int i = b * threads + t; i < N; i += threads * blocks) {
                float v[29];
                fn(ctx, i, v);
                for (int k = 0; k < K; k++) tv[t][k] += v[k];
            }
        block_reduce29(tv, threads, block_out[b]);
    }
    /* second stage: reduceSum<<<1, MAX_THREADS>>> = 512 threads grid-stride over `blocks` partials */
    const int T2 = 512;
    float(*tv2)[29] = calloc((size_t)T2, sizeof(float[29]));
    for (int t = 0; t < T2; t++)
        for (int i = t; i < blocks; i += T2)
            for (int k = 0; k < K; k++) tv2[t][k] += block_out[i][k];
    float tot[29];
    block_reduce29(tv2, T2, tot);
    memcpy(out, tot, sizeof(float) * (size_t)K);
    free(tv2); free(tv); free(block_out);
}

static void se3_products(const float row[7], int found, float v[29])
{ /* the JtJJtrSE3 initialiser lists of reduce.cu:352-388 / 563-599 */
    int k = 0;
    for (int a = 0; a < 6; a++)
        for (int j = a; j < 7; j++) v[k++] = row[a] * row[j];
    v[27] = row[6] * row[6];
    v[28] = (float)found;
}

typedef struct { icp_ctx c; } icp_tree_ctx;
static void icp_vals(const void *ctx, int i, float v[29])
{
    const icp_ctx *c = (const icp_ctx *)ctx;
    int y = i / c->cols, x = i - y * c->cols;
    float row[7], err;
    int found = icp_row(c, x, y, row, &err);
    se3_products(row, found, v);
}

void orc_icp_step_f32tree(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                          const float Rprev_inv[9], const float tprev[3], orc_cam intr, const float *vmap_g_prev,
                          const float *nmap_g_prev, float dist_thres, float angle_thres, int cols, int rows,
                          int threads, int blocks, float out29[29])
{
    icp_ctx c;
    icp_ctx_fill(&c, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev,
                 dist_thres, angle_thres, cols, rows);
    f32tree_reduce(icp_vals, &c, cols * rows, 29, threads, blocks, out29);
}

/* host unpacking of the 29 sums, reduce.cu:481-498 */
void orc_se3_sums_to_host(const int64_t sums[ORC_SE3_WORDS], int F, float A[36], float b[6], float residual[2])
{
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            float value = (float)orc_fix_to_double(sums[shift++], F);
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    residual[0] = (float)orc_fix_to_double(sums[27], F);
    residual[1] = (float)sums[28];
}

/* ============================== RGB residual ================================== */

/* RGBResidual::getProducts, reduce.cu:785-865 (MASK_RGB_RESIDUAL is never defined) */
void orc_rgb_residual(float min_scale, const int16_t *dIdx, const int16_t *dIdy, const float *last_depth,
                      const float *next_depth, const uint8_t *last_image, const uint8_t *next_image,
                      orc_dataterm *corres, float max_depth_delta, const float kt[3], const float krkinv[9],
                      int cols, int rows, int *sigma_sum, int *count)
{
    int cnt = 0, sig = 0;
#pragma omp parallel for schedule(static) reduction(+ : cnt, sig) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int k = 0; k < cols * rows; k++) {
        int i = k / cols, j0 = k - i * cols;
        orc_dataterm c; memset(&c, 0, sizeof(c));
        if (j0 < cols - 5 && i < rows - 1) {
            int valid = 1;
            for (int u = imax(i - 2, 0); u < imin(i + 2, rows); u++)
                for (int v = imax(j0 - 2, 0); v < imin(j0 + 2, cols); v++) valid = valid && (next_image[u * cols + v] > 0);
            if (valid) {
                int valx = dIdx[i * cols + j0], valy = dIdy[i * cols + j0];
                float mTwo = (float)((valx * valx) + (valy * valy));
                if (mTwo >= min_scale) {
                    int y = i, x = j0;
                    float d1 = next_depth[y * cols + x];
                    if (!isnan(d1)) {
                        float transformed_d1 = (float)(d1 * (krkinv[6] * x + krkinv[7] * y + krkinv[8]) + kt[2]);
                        int u0 = orc_f2i_rn((d1 * (krkinv[0] * x + krkinv[1] * y + krkinv[2]) + kt[0]) / transformed_d1);
                        int v0 = orc_f2i_rn((d1 * (krkinv[3] * x + krkinv[4] * y + krkinv[5]) + kt[1]) / transformed_d1);
                        if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
                            float d0 = last_depth[v0 * cols + u0];
                            if (d0 > 0 && fabsf(transformed_d1 - d0) <= max_depth_delta && last_image[v0 * cols + u0] != 0) {
                                c.zero_x = (int16_t)u0; c.zero_y = (int16_t)v0; c.one_x = (int16_t)x; c.one_y = (int16_t)y;
                                c.diff = (float)next_image[y * cols + x] - (float)last_image[v0 * cols + u0];
                                c.valid = 1;
                                cnt += 1;
                                sig += (int)(c.diff * c.diff);
                            }
                        }
                    }
                }
            }
        }
        corres[k] = c; /* flat index, pitch ignored: reduce.cu:862 */
    }
    *count = cnt; *sigma_sum = sig;
}

/* RGBReduction::getProducts, reduce.cu:521-604 */
typedef struct {
    const orc_dataterm *corres; float sigma; const float *cloud3; float fx, fy; const int16_t *dIdx, *dIdy; float sobel_scale; int cols;
} rgb_ctx;
/* RGBReduction::getProducts, reduce.cu:521-604: one Jacobian row, 0 when the DataTerm is not valid */
static int rgb_row(const rgb_ctx *r, int i, float row[7])
{
    const orc_dataterm *c = &r->corres[i];
    for (int k = 0; k < 7; k++) row[k] = 0.f;
    if (!c->valid) return 0;
    float w = r->sigma + fabsf(c->diff);
    w = w > FLT_EPSILON ? 1.0f / w : 1.0f;
    if (r->sigma == -1) w = 1;
    row[6] = -w * c->diff;
    const float *cp = r->cloud3 + (c->zero_y * r->cols + c->zero_x) * 3;
    float invz = 1.0f / cp[2]; /* (float)(1.0/z) == 1.0f/z, both correctly rounded */
    float dI_dx_val = w * r->sobel_scale * (float)r->dIdx[c->one_y * r->cols + c->one_x];
    float dI_dy_val = w * r->sobel_scale * (float)r->dIdy[c->one_y * r->cols + c->one_x];
    float v0 = dI_dx_val * r->fx * invz;
    float v1 = dI_dy_val * r->fy * invz;
    float v2 = -(v0 * cp[0] + v1 * cp[1]) * invz;
    row[0] = v0; row[1] = v1; row[2] = v2;
    row[3] = -cp[2] * v1 + cp[1] * v2;
    row[4] = cp[2] * v0 - cp[0] * v2;
    row[5] = -cp[1] * v0 + cp[0] * v1;
    return 1;
}

void orc_rgb_step(const orc_dataterm *corres, float sigma, const float *cloud3, float fx, float fy,
                  const int16_t *dIdx, const int16_t *dIdy, float sobel_scale, int cols, int rows,
                  int64_t sums[ORC_SE3_WORDS])
{
    const rgb_ctx r = {corres, sigma, cloud3, fx, fy, dIdx, dIdy, sobel_scale, cols};
    memset(sums, 0, sizeof(int64_t) * ORC_SE3_WORDS);
#pragma omp parallel for schedule(static) reduction(+ : sums[:ORC_SE3_WORDS]) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int i = 0; i < cols * rows; i++) {
        float row[7];
        if (rgb_row(&r, i, row)) se3_accumulate(row, 1, orc_rgb_fix_bits(sigma), sums);
    }
}

static void rgb_vals(const void *ctx, int i, float v[29])
{
    float row[7];
    int found = rgb_row((const rgb_ctx *)ctx, i, row);
    se3_products(row, found, v);
}
/* the same sums in the reference's f32 order (rgbKernel<<<blocks, threads>>> + reduceSum, reduce.cu:606-670) */
void orc_rgb_step_f32tree(const orc_dataterm *corres, float sigma, const float *cloud3, float fx, float fy, const int16_t *dIdx,
                          const int16_t *dIdy, float sobel_scale, int cols, int rows, int threads, int blocks, float out29[29])
{
    const rgb_ctx r = {corres, sigma, cloud3, fx, fy, dIdx, dIdy, sobel_scale, cols};
    f32tree_reduce(rgb_vals, &r, cols * rows, 29, threads, blocks, out29);
}

int orc_rgb_fix_bits_of(float sigma) { return orc_rgb_fix_bits(sigma); }

/* ================================== SO3 ======================================= */

static inline void so3_gradient(const uint8_t *img, int cols, int x, int y, float *gx, float *gy)
{ /* SO3Reduction::getGradient, reduce.cu:989-1005 */
    float actu = (float)img[y * cols + x];
    float back = (float)img[y * cols + x - 1], fore = (float)img[y * cols + x + 1];
    *gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = (float)img[(y - 1) * cols + x]; fore = (float)img[(y + 1) * cols + x];
    *gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}

typedef struct { const uint8_t *last_image, *next_image; orc_m33 B, Ki; const float *krlr; int cols, rows; } so3_ctx;
/* SO3Reduction::getProducts, reduce.cu:1007-1090 */
static int so3_row(const so3_ctx *c, int k, float row[4])
{
    const int cols = c->cols, rows = c->rows;
    int y = k / cols, x = k - y * cols;
    row[0] = row[1] = row[2] = row[3] = 0.f;
    orc_f3 unwarped = {(float)x, (float)y, 1.0f};
    orc_f3 warped = orc_m33_mul(&c->B, unwarped);
    int wx = orc_f2i_rn(warped.x / warped.z), wy = orc_f2i_rn(warped.y / warped.z);
    if (!(wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 && x < cols - 1 && y >= 1 && y < rows - 1)) return 0;
    float gnx, gny, glx, gly;
    so3_gradient(c->next_image, cols, wx, wy, &gnx, &gny);
    so3_gradient(c->last_image, cols, x, y, &glx, &gly);
    float gx = (gnx + glx) / 2.0f, gy = (gny + gly) / 2.0f;
    orc_f3 point = orc_m33_mul(&c->Ki, unwarped);
    float z2 = point.z * point.z;
    const float *krlr = c->krlr;
    float a = krlr[0], b = krlr[1], cc = krlr[2], d = krlr[3], e = krlr[4], f = krlr[5], g = krlr[6], h = krlr[7], i = krlr[8];
    orc_f3 left = {((point.z * (d * gy + a * gx)) - (gy * g * y) - (gx * g * x)) / z2,
                   ((point.z * (e * gy + b * gx)) - (gy * h * y) - (gx * h * x)) / z2,
                   ((point.z * (f * gy + cc * gx)) - (gy * i * y) - (gx * i * x)) / z2};
    orc_f3 jac = orc_f3_cross(left, point);
    row[0] = jac.x; row[1] = jac.y; row[2] = jac.z;
    row[3] = -((float)c->next_image[wy * cols + wx] - (float)c->last_image[y * cols + x]);
    return 1;
}

void orc_so3_step(const uint8_t *last_image, const uint8_t *next_image, const float image_basis[9],
                  const float kinv[9], const float krlr[9], int cols, int rows, int64_t sums[ORC_SO3_WORDS])
{
    so3_ctx c = {last_image, next_image, {{0}}, {{0}}, krlr, cols, rows};
    memcpy(c.B.m, image_basis, 36); memcpy(c.Ki.m, kinv, 36);
    memset(sums, 0, sizeof(int64_t) * ORC_SO3_WORDS);
#pragma omp parallel for schedule(static) reduction(+ : sums[:ORC_SO3_WORDS]) if (cols * rows >= ORC_OMP_MIN_PIXELS)
    for (int k = 0; k < cols * rows; k++) {
        float row[4];
        if (!so3_row(&c, k, row)) continue;
        int s = 0;
        for (int p = 0; p < 3; p++)
            for (int q = p; q < 4; q++) sums[s++] += orc_fix_prod(row[p], row[q], ORC_FIX_SO3);
        sums[9] += orc_fix_prod(row[3], row[3], ORC_FIX_SO3);
        sums[10] += 1;
    }
}

static void so3_vals(const void *ctx, int k, float v[29])
{
    float row[4];
    int found = so3_row((const so3_ctx *)ctx, k, row);
    int s = 0;
    for (int p = 0; p < 3; p++)
        for (int q = p; q < 4; q++) v[s++] = row[p] * row[q];
    v[9] = row[3] * row[3];
    v[10] = (float)found;
}
/* the same sums in the reference's f32 order (so3Kernel<<<blocks, threads>>> + reduceSum, reduce.cu:1092-1156) */
void orc_so3_step_f32tree(const uint8_t *last_image, const uint8_t *next_image, const float image_basis[9], const float kinv[9],
                          const float krlr[9], int cols, int rows, int threads, int blocks, float out11[11])
{
    so3_ctx c = {last_image, next_image, {{0}}, {{0}}, krlr, cols, rows};
    memcpy(c.B.m, image_basis, 36); memcpy(c.Ki.m, kinv, 36);
    f32tree_reduce(so3_vals, &c, cols * rows, 11, threads, blocks, out11);
}

void orc_so3_sums_to_host(const int64_t sums[ORC_SO3_WORDS], int F, float A[9], float b[3], float residual[2])
{ /* reduce.cu:1158-1175 */
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            float value = (float)orc_fix_to_double(sums[shift++], F);
            if (j == 3) b[i] = value;
            else A[j * 3 + i] = A[i * 3 + j] = value;
        }
    residual[0] = (float)orc_fix_to_double(sums[9], F);
    residual[1] = (float)sums[10];
}

/* ============================ RGBDOdometry ==================================== */

struct orc_odometry {
    int width, height; orc_cam intr;
    float *vmaps_tmp, *nmaps_tmp;                       /* RGBA32F copies, RGBDOdometry.h:78-79 */
    float *vmaps_g_prev[ORC_NUM_PYRS], *nmaps_g_prev[ORC_NUM_PYRS];
    float *vmaps_curr[ORC_NUM_PYRS], *nmaps_curr[ORC_NUM_PYRS];
    float *lastDepth[ORC_NUM_PYRS], *nextDepth[ORC_NUM_PYRS];
    uint8_t *lastImage[ORC_NUM_PYRS], *nextImage[ORC_NUM_PYRS], *lastNextImage[ORC_NUM_PYRS];
    int16_t *dIdx[ORC_NUM_PYRS], *dIdy[ORC_NUM_PYRS];
    float *cloud[ORC_NUM_PYRS];
    orc_dataterm *corres[ORC_NUM_PYRS];
    float distThres, angleThres, sobelScale, maxDepthDeltaRGB, maxDepthRGB;
    float minGrad[ORC_NUM_PYRS];
};

orc_odometry *orc_odom_create(int width, int height, float cx, float cy, float fx, float fy)
{ /* RGBDOdometry ctor, RGBDOdometry.cpp:21-106; defaults RGBDOdometry.h:35-36 */
    orc_odometry *o = calloc(1, sizeof(*o));
    o->width = width; o->height = height;
    o->intr.fx = fx; o->intr.fy = fy; o->intr.cx = cx; o->intr.cy = cy;
    o->distThres = 0.10f;
    o->angleThres = (float)sin(20.f * 3.14159254f / 180.f);
    o->sobelScale = (float)(1.0 / pow(2.0, 3));
    o->maxDepthDeltaRGB = 0.07f; o->maxDepthRGB = 6.0f;
    o->minGrad[0] = 5; o->minGrad[1] = 3; o->minGrad[2] = 1;
    size_t n0 = (size_t)width * height;
    o->vmaps_tmp = calloc(n0 * 4, sizeof(float)); o->nmaps_tmp = calloc(n0 * 4, sizeof(float));
    for (int i = 0; i < ORC_NUM_PYRS; i++) {
        size_t n = (size_t)(width >> i) * (height >> i);
        o->vmaps_g_prev[i] = calloc(n * 3, 4); o->nmaps_g_prev[i] = calloc(n * 3, 4);
        o->vmaps_curr[i] = calloc(n * 3, 4); o->nmaps_curr[i] = calloc(n * 3, 4);
        o->lastDepth[i] = calloc(n, 4); o->nextDepth[i] = calloc(n, 4);
        o->lastImage[i] = calloc(n, 1); o->nextImage[i] = calloc(n, 1); o->lastNextImage[i] = calloc(n, 1);
        o->dIdx[i] = calloc(n, 2); o->dIdy[i] = calloc(n, 2);
        o->cloud[i] = calloc(n * 3, 4); o->corres[i] = calloc(n, sizeof(orc_dataterm));
    }
    return o;
}

void orc_odom_destroy(orc_odometry *o)
{
    if (!o) return;
    free(o->vmaps_tmp); free(o->nmaps_tmp);
    for (int i = 0; i < ORC_NUM_PYRS; i++) {
        free(o->vmaps_g_prev[i]); free(o->nmaps_g_prev[i]); free(o->vmaps_curr[i]); free(o->nmaps_curr[i]);
        free(o->lastDepth[i]); free(o->nextDepth[i]); free(o->lastImage[i]); free(o->nextImage[i]);
        free(o->lastNextImage[i]); free(o->dIdx[i]); free(o->dIdy[i]); free(o->cloud[i]); free(o->corres[i]);
    }
    free(o);
}

const void *orc_odom_buffer(const orc_odometry *o, int which, int level)
{
    switch (which) {
        case 0: return o->vmaps_curr[level]; case 1: return o->nmaps_curr[level];
        case 2: return o->vmaps_g_prev[level]; case 3: return o->nmaps_g_prev[level];
        case 4: return o->lastDepth[level]; case 5: return o->nextDepth[level];
        case 6: return o->lastImage[level]; case 7: return o->nextImage[level]; case 8: return o->lastNextImage[level];
        case 9: return o->dIdx[level]; case 10: return o->dIdy[level];
        case 11: return o->cloud[level]; case 12: return o->corres[level];
        default: return 0;
    }
}

/* RGBDOdometry::initICPModel, RGBDOdometry.cpp:143-175 */
void orc_odom_init_icp_model(orc_odometry *o, const float *pred_v4, const float *pred_n4, const float pose[16])
{
    size_t n0 = (size_t)o->width * o->height;
    memcpy(o->vmaps_tmp, pred_v4, n0 * 16); memcpy(o->nmaps_tmp, pred_n4, n0 * 16);
    orc_copy_maps(o->vmaps_tmp, o->nmaps_tmp, o->width, o->height, o->vmaps_g_prev[0], o->nmaps_g_prev[0]);
    for (int i = 1; i < ORC_NUM_PYRS; ++i) {
        orc_resize_map(o->vmaps_g_prev[i - 1], o->width >> (i - 1), o->height >> (i - 1), o->vmaps_g_prev[i], 0);
        orc_resize_map(o->nmaps_g_prev[i - 1], o->width >> (i - 1), o->height >> (i - 1), o->nmaps_g_prev[i], 1);
    }
    float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    float t[3] = {pose[3], pose[7], pose[11]};
    for (int i = 0; i < ORC_NUM_PYRS; ++i)
        orc_transform_maps(o->vmaps_g_prev[i], o->nmaps_g_prev[i], o->width >> i, o->height >> i, R, t);
}

/* RGBDOdometry::populateRGBDData, RGBDOdometry.cpp:177-194 (mask pyramid is dead data) */
static void populate_rgbd(orc_odometry *o, const uint8_t *rgba, float **depths, uint8_t **images)
{
    orc_vertices_to_depth(o->vmaps_tmp, o->width, o->height, o->maxDepthRGB, depths[0]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; i++) orc_pyrdown_gauss_f32(depths[i], o->width >> i, o->height >> i, depths[i + 1]);
    orc_rgba_to_intensity(rgba, o->width, o->height, images[0]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; i++) orc_pyrdown_gauss_u8(images[i], o->width >> i, o->height >> i, images[i + 1]);
}
/* initRGBModel / initRGB: both read vmaps_tmp, which holds the MODEL prediction
 * (RGBDOdometry.cpp:196-204) -> nextDepth == lastDepth in frame-to-model tracking. */
void orc_odom_init_rgb_model(orc_odometry *o, const uint8_t *pred_rgba) { populate_rgbd(o, pred_rgba, o->lastDepth, o->lastImage); }
void orc_odom_init_rgb(orc_odometry *o, const uint8_t *rgba) { populate_rgbd(o, rgba, o->nextDepth, o->nextImage); }

/* RGBDOdometry::initICP(depthPyramid,..), RGBDOdometry.cpp:110-118 */
void orc_odom_init_icp(orc_odometry *o, const float *const depth_pyr[ORC_NUM_PYRS], float depth_cutoff)
{
    for (int i = 0; i < ORC_NUM_PYRS; ++i) {
        orc_create_vmap(depth_pyr[i], o->width >> i, o->height >> i, cam_level(o->intr, i), depth_cutoff, o->vmaps_curr[i]);
        orc_create_nmap(o->vmaps_curr[i], o->width >> i, o->height >> i, o->nmaps_curr[i]);
    }
}

/* RGBDOdometry::initFirstRGB, RGBDOdometry.cpp:206-215 */
void orc_odom_init_first_rgb(orc_odometry *o, const uint8_t *rgba)
{
    orc_rgba_to_intensity(rgba, o->width, o->height, o->lastNextImage[0]);
    for (int i = 0; i + 1 < ORC_NUM_PYRS; i++)
        orc_pyrdown_gauss_u8(o->lastNextImage[i], o->width >> i, o->height >> i, o->lastNextImage[i + 1]);
}

static void k_matrix(orc_cam c, double K[9])
{
    memset(K, 0, sizeof(double) * 9);
    K[0] = c.fx; K[4] = c.fy; K[2] = c.cx; K[5] = c.cy; K[8] = 1;
}


/* ======================= ORC_ICP_ARITH_REFERENCE: the reference's own order =======================
 * The third rounding specification: every reduction of the Gauss-Newton loop is the reference's launch-shape
 * dependent f32 tree (thread-strided partials, 32-lane shuffle-down tree, block tree, second-stage reduceSum: reduce.cu:90-185,
 * 396-417, 475-499) at the launch shapes of GPUConfig.h:51-58 -- the constructor's defaults, which every board that is not in its
 * table of NVIDIA names gets -- and the host algebra follows the operation ORDER of the classes the reference's text instantiates as
 * oracle/ref_shim/eigen_fixed states it (left-looking pivoted LDL^T, cofactor inverses expanded along the first column / first row,
 * products accumulated left to right, isometry composition) with the C library's own cos / sin (OdometryProvider.h:48-49).  Pinned
 * BIT FOR BIT against RGBDOdometry::getIncrementalTransformation compiled from /root/reference (tests/test_cpu_refpin.py:
 * test_reference_order_gn_loop_is_the_reference_class_bit_for_bit); the HIP path has the same mode (cf_set_icp_arith 2). */
#define REF_ICP_THREADS 128
#define REF_ICP_BLOCKS 112
#define REF_RGB_THREADS 128
#define REF_RGB_BLOCKS 112
#define REF_SO3_THREADS 160
#define REF_SO3_BLOCKS 64

/* Matrix<T,3,3>::inverse() of the stand-in: cofactors with cyclic indices, determinant along the first COLUMN */
#define EIG_INV33(T, NAME)                                                                                         \
    static void NAME(const T m[9], T o[9])                                                                         \
    {                                                                                                              \
        T cof[3][3];                                                                                               \
        for (int i = 0; i < 3; i++)                                                                                \
            for (int j = 0; j < 3; j++) {                                                                          \
                const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;                  \
                cof[i][j] = m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];                     \
            }                                                                                                      \
        const T det = (cof[0][0] * m[0] + cof[1][0] * m[3]) + cof[2][0] * m[6];                                    \
        const T invdet = (T)1 / det;                                                                               \
        for (int r = 0; r < 3; r++)                                                                                \
            for (int c = 0; c < 3; c++) o[r * 3 + c] = cof[c][r] * invdet;                                         \
    }
EIG_INV33(double, eig_inv33d)
EIG_INV33(float, eig_inv33f)

/* Matrix<double,4,4>::inverse() of the stand-in: cofactors of 3x3 minors, determinant along the first ROW */
static double eig_minor3(const double m[16], int r, int c)
{
    int ri[3], ci[3];
    for (int k = 0, t = 0; k < 4; k++) if (k != r) ri[t++] = k;
    for (int k = 0, t = 0; k < 4; k++) if (k != c) ci[t++] = k;
#define M_(a, b) m[(a) * 4 + (b)]
    return M_(ri[0], ci[0]) * (M_(ri[1], ci[1]) * M_(ri[2], ci[2]) - M_(ri[1], ci[2]) * M_(ri[2], ci[1])) -
           M_(ri[0], ci[1]) * (M_(ri[1], ci[0]) * M_(ri[2], ci[2]) - M_(ri[1], ci[2]) * M_(ri[2], ci[0])) +
           M_(ri[0], ci[2]) * (M_(ri[1], ci[0]) * M_(ri[2], ci[1]) - M_(ri[1], ci[1]) * M_(ri[2], ci[0]));
#undef M_
}
static void eig_inv44d(const double m[16], double o[16])
{
    double cofm[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) cofm[r][c] = ((r + c) & 1) ? -eig_minor3(m, r, c) : eig_minor3(m, r, c);
    const double det = ((m[0] * cofm[0][0] + m[1] * cofm[0][1]) + m[2] * cofm[0][2]) + m[3] * cofm[0][3];
    const double invdet = 1.0 / det;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) o[r * 4 + c] = cofm[c][r] * invdet;
}

/* Matrix::ldlt().solve() of the stand-in: unblocked LEFT-looking LDL^T on the lower triangle with diagonal pivoting (first maximum
 * wins), solve = P, L^-1, D^-1 (|d| <= 1 / max -> 0), L^-T, P^T */
#define EIG_LDLT_SOLVE(T, NAME, TMAX)                                                                              \
    static void NAME(int N, const T *Ain, const T *b, T *x)                                                        \
    {                                                                                                              \
        T a[36], y[6], temp[6];                                                                                    \
        int tr[6];                                                                                                 \
        for (int i = 0; i < N * N; i++) a[i] = Ain[i];                                                             \
        for (int k = 0; k < N; k++) {                                                                              \
            int big = k;                                                                                           \
            T best = a[k * N + k] < 0 ? -a[k * N + k] : a[k * N + k];                                              \
            for (int i = k + 1; i < N; i++) { const T v = a[i * N + i] < 0 ? -a[i * N + i] : a[i * N + i]; if (v > best) { best = v; big = i; } } \
            tr[k] = big;                                                                                           \
            if (big != k) {                                                                                        \
                for (int j = 0; j < k; j++) { const T t = a[k * N + j]; a[k * N + j] = a[big * N + j]; a[big * N + j] = t; } \
                for (int i = big + 1; i < N; i++) { const T t = a[i * N + k]; a[i * N + k] = a[i * N + big]; a[i * N + big] = t; } \
                { const T t = a[k * N + k]; a[k * N + k] = a[big * N + big]; a[big * N + big] = t; }               \
                for (int i = k + 1; i < big; i++) { const T t = a[i * N + k]; a[i * N + k] = a[big * N + i]; a[big * N + i] = t; } \
            }                                                                                                      \
            if (k > 0) {                                                                                           \
                for (int j = 0; j < k; j++) temp[j] = a[j * N + j] * a[k * N + j];                                 \
                { T s = a[k * N + 0] * temp[0]; for (int j = 1; j < k; j++) s = s + a[k * N + j] * temp[j]; a[k * N + k] = a[k * N + k] - s; } \
                for (int i = k + 1; i < N; i++) { T s = a[i * N + 0] * temp[0]; for (int j = 1; j < k; j++) s = s + a[i * N + j] * temp[j]; a[i * N + k] = a[i * N + k] - s; } \
            }                                                                                                      \
            const T akk = a[k * N + k];                                                                            \
            if ((akk < 0 ? -akk : akk) > (T)0) for (int i = k + 1; i < N; i++) a[i * N + k] = a[i * N + k] / akk;  \
        }                                                                                                          \
        for (int i = 0; i < N; i++) y[i] = b[i];                                                                   \
        for (int k = 0; k < N; k++) { const T t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }                           \
        for (int i = 1; i < N; i++) { T s = a[i * N + 0] * y[0]; for (int j = 1; j < i; j++) s = s + a[i * N + j] * y[j]; y[i] = y[i] - s; } \
        const T tol = (T)1 / (T)TMAX;                                                                              \
        for (int i = 0; i < N; i++) { const T d = a[i * N + i]; y[i] = ((d < 0 ? -d : d) > tol) ? y[i] / d : (T)0; } \
        for (int i = N - 2; i >= 0; i--) { T s = a[(i + 1) * N + i] * y[i + 1]; for (int j = i + 2; j < N; j++) s = s + a[j * N + i] * y[j]; y[i] = y[i] - s; } \
        for (int k = N - 1; k >= 0; k--) { const T t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }                      \
        for (int i = 0; i < N; i++) x[i] = y[i];                                                                   \
    }
EIG_LDLT_SOLVE(double, eig_ldlt_solve_d, DBL_MAX)
EIG_LDLT_SOLVE(float, eig_ldlt_solve_f, FLT_MAX)

/* OdometryProvider::rodrigues as written (OdometryProvider.h:32-67), the C library's cos / sin */
static void ref_rodrigues(const double src[3], double R[9])
{
    double rx = src[0], ry = src[1], rz = src[2];
    const double theta = sqrt((src[0] * src[0] + src[1] * src[1]) + src[2] * src[2]);
    for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (theta >= DBL_EPSILON) {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
        rx *= itheta; ry *= itheta; rz *= itheta;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    }
}

/* reduce.cu:481-498 / 1158-1175: the 29 / 11 f32 totals -> A, b, residual */
static void ref_unpack29(const float h[29], float A[36], float b[6], float residual[2])
{
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float value = h[shift++];
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    if (residual) { residual[0] = h[27]; residual[1] = h[28]; }
}

static void odom_track_reference_order(orc_odometry *o, float trans[3], float rot[9], const orc_track_opts *opts,
                                       float *icp_err_surface, orc_track_stats *st)
{
    const int rgbOnly = opts->rgb_only;
    const float icpWeight = opts->icp_weight;
    const int icp = !rgbOnly && icpWeight > 0;
    const int rgb = rgbOnly || icpWeight < 100;
    orc_track_stats local; if (!st) st = &local;
    memset(st, 0, sizeof(*st));
    float Rprev[9], tprev[3], Rcurr[9], tcurr[3];
    memcpy(Rprev, rot, 36); memcpy(tprev, trans, 12); memcpy(Rcurr, rot, 36); memcpy(tcurr, trans, 12);
    if (rgb)
        for (int i = 0; i < ORC_NUM_PYRS; i++) orc_sobel(o->nextImage[i], o->width >> i, o->height >> i, o->dIdx[i], o->dIdy[i]);

    double resultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (opts->so3) { /* RGBDOdometry.cpp:239-310 */
        const int L = 2, cols = o->width >> L, rows = o->height >> L;
        float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double K[9], Kinv[9];
        k_matrix(cam_level(o->intr, L), K);
        float lastError = FLT_MAX / 2, lastCount = FLT_MAX / 2;
        double lastResultR[9]; memcpy(lastResultR, resultR, sizeof(resultR));
        for (int it = 0; it < 10; it++) {
            double KR[9], H[9];
            eig_inv33d(K, Kinv);
            orc_mul33d(K, resultR, KR); orc_mul33d(KR, Kinv, H);   /* (K * resultR) * K.inverse() */
            float basis[9], kinvf[9], krlr[9];
            for (int k = 0; k < 9; k++) { basis[k] = (float)H[k]; kinvf[k] = (float)Kinv[k]; krlr[k] = (float)KR[k]; }
            float out11[11], jtj[9], jtr[3], residual[2];
            orc_so3_step_f32tree(o->lastNextImage[L], o->nextImage[L], basis, kinvf, krlr, cols, rows, REF_SO3_THREADS, REF_SO3_BLOCKS, out11);
            int shift = 0;
            for (int i = 0; i < 3; ++i)
                for (int j = i; j < 4; ++j) {
                    const float value = out11[shift++];
                    if (j == 3) jtr[i] = value; else jtj[j * 3 + i] = jtj[i * 3 + j] = value;
                }
            residual[0] = out11[9]; residual[1] = out11[10];
            st->so3_iterations = it + 1;
            st->last_so3_error = sqrtf(residual[0]) / residual[1];
            st->last_so3_count = residual[1];
            if (st->last_so3_error < lastError && (double)fabsf(lastError - st->last_so3_count) < 0.001) break;
            else if ((double)st->last_so3_error > (double)lastError + 0.001) {
                st->last_so3_error = lastError; st->last_so3_count = lastCount;
                memcpy(resultR, lastResultR, sizeof(resultR));
                break;
            }
            lastError = st->last_so3_error; lastCount = st->last_so3_count;
            memcpy(lastResultR, resultR, sizeof(resultR));
            float delta[3];
            eig_ldlt_solve_f(3, jtj, jtr, delta);
            double dd[3] = {delta[0], delta[1], delta[2]}, rotUpdate[9];
            ref_rodrigues(dd, rotUpdate);
            float ru[9], nr[9];
            for (int k = 0; k < 9; k++) ru[k] = (float)rotUpdate[k];
            orc_mul33f(ru, R_lr, nr);
            memcpy(R_lr, nr, sizeof(nr));
            for (int k = 0; k < 9; k++) resultR[k] = R_lr[k];
        }
    }

    int iterations[ORC_NUM_PYRS];
    iterations[0] = opts->fast_odom ? 3 : 10;
    iterations[1] = opts->pyramid ? 5 : 0;
    iterations[2] = opts->pyramid ? 4 : 0;
    float Rprev_inv[9];
    eig_inv33f(Rprev, Rprev_inv);
    double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (opts->so3)
        for (int x = 0; x < 3; x++)
            for (int y = 0; y < 3; y++) resultRt[x * 4 + y] = resultR[x * 3 + y];
    float residual[2] = {0, 0};
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

    for (int i = ORC_NUM_PYRS - 1; i >= 0; i--) {
        const int cols = o->width >> i, rows = o->height >> i;
        const orc_cam il = cam_level(o->intr, i);
        if (rgb) orc_project_cloud(o->lastDepth[i], cols, rows, il, o->cloud[i]);
        double K[9], Kinv[9];
        k_matrix(il, K);
        st->last_rgb_error = FLT_MAX;
        for (int j = 0; j < iterations[i]; j++) {
            double Rt[16];
            eig_inv44d(resultRt, Rt);
            double R[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
            double tmp[9], KRK[9];
            eig_inv33d(K, Kinv);
            orc_mul33d(K, R, tmp); orc_mul33d(tmp, Kinv, KRK);
            float krkInv[9];
            for (int k = 0; k < 9; k++) krkInv[k] = (float)KRK[k];
            const double tv[3] = {Rt[3], Rt[7], Rt[11]};
            float kt[3];
            for (int r = 0; r < 3; r++) { double s = K[r * 3 + 0] * tv[0]; s = s + K[r * 3 + 1] * tv[1]; s = s + K[r * 3 + 2] * tv[2]; kt[r] = (float)s; }

            int sigma = 0, rgbSize = 0;
            if (rgb) {
                const float minScale = (float)(pow(o->minGrad[i], 2.0) / pow(o->sobelScale, 2.0));
                orc_rgb_residual(minScale, o->dIdx[i], o->dIdy[i], o->lastDepth[i], o->nextDepth[i], o->lastImage[i],
                                 o->nextImage[i], o->corres[i], o->maxDepthDeltaRGB, kt, krkInv, cols, rows, &sigma, &rgbSize);
            }
            const float tmpError = (float)(sqrt((double)sigma) / rgbSize);
            float sigmaVal = (tmpError == 0) ? 1 : (float)rgbSize;
            if (rgbOnly && tmpError > st->last_rgb_error) break;
            st->last_rgb_error = tmpError; st->last_rgb_count = (float)rgbSize;
            if (rgbOnly) sigmaVal = -1;

            float A_icp[36], b_icp[6], A_rgbd[36], b_rgbd[6];
            memset(A_icp, 0, sizeof(A_icp)); memset(b_icp, 0, sizeof(b_icp));
            memset(A_rgbd, 0, sizeof(A_rgbd)); memset(b_rgbd, 0, sizeof(b_rgbd));
            if (icp) {
                float out29[29];
                orc_icp_step_f32tree(Rcurr, tcurr, o->vmaps_curr[i], o->nmaps_curr[i], Rprev_inv, tprev, il, o->vmaps_g_prev[i],
                                     o->nmaps_g_prev[i], o->distThres, o->angleThres, cols, rows, REF_ICP_THREADS, REF_ICP_BLOCKS, out29);
                ref_unpack29(out29, A_icp, b_icp, residual);
                if (i == 0 && j == iterations[i] - 1 && icp_err_surface) {   /* the optional output of the same launch (reduce.cu:303-331) */
                    icp_ctx c;
                    icp_ctx_fill(&c, Rcurr, tcurr, o->vmaps_curr[i], o->nmaps_curr[i], Rprev_inv, tprev, il, o->vmaps_g_prev[i], o->nmaps_g_prev[i],
                                 o->distThres, o->angleThres, cols, rows);
                    for (int y = 0; y < rows; y++)
                        for (int x = 0; x < cols; x++) { float row[7], err; icp_row(&c, x, y, row, &err); icp_err_surface[y * cols + x] = err; }
                }
            }
            st->last_icp_error = sqrtf(residual[0]) / residual[1];
            st->last_icp_count = residual[1];
            if (rgb) {
                float out29[29];
                orc_rgb_step_f32tree(o->corres[i], sigmaVal, o->cloud[i], il.fx, il.fy, o->dIdx[i], o->dIdy[i], o->sobelScale, cols, rows,
                                     REF_RGB_THREADS, REF_RGB_BLOCKS, out29);
                ref_unpack29(out29, A_rgbd, b_rgbd, 0);
            }
            double lastA[36], lastb[6], result[6];
            if (icp && rgb) {
                const double w = icpWeight;
                for (int k = 0; k < 36; k++) lastA[k] = (double)A_rgbd[k] + (w * w) * (double)A_icp[k];
                for (int k = 0; k < 6; k++) lastb[k] = (double)b_rgbd[k] + w * (double)b_icp[k];
            } else if (icp) {
                for (int k = 0; k < 36; k++) lastA[k] = A_icp[k];
                for (int k = 0; k < 6; k++) lastb[k] = b_icp[k];
            } else {
                for (int k = 0; k < 36; k++) lastA[k] = A_rgbd[k];
                for (int k = 0; k < 6; k++) lastb[k] = b_rgbd[k];
            }
            eig_ldlt_solve_d(6, lastA, lastb, result);
            memcpy(st->lastA, lastA, sizeof(lastA)); memcpy(st->lastb, lastb, sizeof(lastb));

            /* OdometryProvider::computeUpdateSE3 (OdometryProvider.h:69-89) */
            double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Rr[9], nrt[16];
            const double rvec[3] = {result[3], result[4], result[5]};
            ref_rodrigues(rvec, Rr);
            for (int r = 0; r < 3; r++) { upd[r * 4 + 0] = Rr[r * 3 + 0]; upd[r * 4 + 1] = Rr[r * 3 + 1]; upd[r * 4 + 2] = Rr[r * 3 + 2]; upd[r * 4 + 3] = result[r]; }
            orc_mul44d(upd, resultRt, nrt);
            memcpy(resultRt, nrt, sizeof(nrt));
            /* rgbOdom.setIdentity(); rgbOdom.rotate(rotation.cast<float>()): linear = Identity * R, a product like any other */
            float Rf[9], Ro[9], to[3];
            for (int r = 0; r < 3; r++) { Rf[r * 3 + 0] = (float)resultRt[r * 4 + 0]; Rf[r * 3 + 1] = (float)resultRt[r * 4 + 1]; Rf[r * 3 + 2] = (float)resultRt[r * 4 + 2]; to[r] = (float)resultRt[r * 4 + 3]; }
            orc_mul33f(ident, Rf, Ro);
            /* currentT.setIdentity(); currentT.rotate(Rprev); translation = tprev; currentT = currentT * rgbOdom.inverse() (RGBDOdometry.cpp:452-460) */
            float Rp[9];
            orc_mul33f(ident, Rprev, Rp);
            const float Rinv[9] = {Ro[0], Ro[3], Ro[6], Ro[1], Ro[4], Ro[7], Ro[2], Ro[5], Ro[8]};
            float tinv[3];
            for (int r = 0; r < 3; r++) { float s = Rinv[r * 3 + 0] * to[0]; s = s + Rinv[r * 3 + 1] * to[1]; s = s + Rinv[r * 3 + 2] * to[2]; tinv[r] = -s; }
            orc_mul33f(Rp, Rinv, Rcurr);
            for (int r = 0; r < 3; r++) { float s = Rp[r * 3 + 0] * tinv[0]; s = s + Rp[r * 3 + 1] * tinv[1]; s = s + Rp[r * 3 + 2] * tinv[2]; tcurr[r] = s + tprev[r]; }
        }
    }
    if (rgb) {
        const float d[3] = {tcurr[0] - tprev[0], tcurr[1] - tprev[1], tcurr[2] - tprev[2]};
        if ((double)sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) > 0.3) { memcpy(Rcurr, Rprev, 36); memcpy(tcurr, tprev, 12); }
    }
    if (opts->so3)
        for (int i = 0; i < ORC_NUM_PYRS; i++) { uint8_t *t = o->lastNextImage[i]; o->lastNextImage[i] = o->nextImage[i]; o->nextImage[i] = t; }
    memcpy(trans, tcurr, 12); memcpy(rot, Rcurr, 36);
}

/* RGBDOdometry::getIncrementalTransformation, RGBDOdometry.cpp:217-477 */
void orc_odom_get_incremental_transformation(orc_odometry *o, float trans[3], float rot[9], const orc_track_opts *opts,
                                             float *icp_err_surface, orc_track_stats *st)
{
    if (g_icp_arith == ORC_ICP_ARITH_REFERENCE) { odom_track_reference_order(o, trans, rot, opts, icp_err_surface, st); return; }
    const int rgbOnly = opts->rgb_only;
    const float icpWeight = opts->icp_weight;
    const int icp = !rgbOnly && icpWeight > 0;
    const int rgb = rgbOnly || icpWeight < 100;
    orc_track_stats local; if (!st) st = &local;
    memset(st, 0, sizeof(*st));

    float Rprev[9], tprev[3], Rcurr[9], tcurr[3];
    memcpy(Rprev, rot, 36); memcpy(tprev, trans, 12); memcpy(Rcurr, rot, 36); memcpy(tcurr, trans, 12);

    if (rgb)
        for (int i = 0; i < ORC_NUM_PYRS; i++) orc_sobel(o->nextImage[i], o->width >> i, o->height >> i, o->dIdx[i], o->dIdy[i]);

    double resultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

    if (opts->so3) { /* :239-310 */
        const int L = 2, cols = o->width >> L, rows = o->height >> L;
        float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double K[9], Kinv[9];
        k_matrix(cam_level(o->intr, L), K);
        orc_inv33d(K, Kinv);
        float lastError = FLT_MAX / 2, lastCount = FLT_MAX / 2;
        double lastResultR[9]; memcpy(lastResultR, resultR, sizeof(resultR));
        for (int it = 0; it < 10; it++) {
            double tmp[9], H[9], KR[9];
            orc_mul33d(K, resultR, tmp); orc_mul33d(tmp, Kinv, H);
            memcpy(KR, tmp, sizeof(tmp));
            float basis[9], kinvf[9], krlr[9];
            for (int k = 0; k < 9; k++) { basis[k] = (float)H[k]; kinvf[k] = (float)Kinv[k]; krlr[k] = (float)KR[k]; }
            int64_t sums[ORC_SO3_WORDS];
            float jtj[9], jtr[3], residual[2];
            orc_so3_step(o->lastNextImage[L], o->nextImage[L], basis, kinvf, krlr, cols, rows, sums);
            orc_so3_sums_to_host(sums, ORC_FIX_SO3, jtj, jtr, residual);
            st->so3_iterations = it + 1;
            st->last_so3_error = sqrtf(residual[0]) / residual[1];
            st->last_so3_count = residual[1];
            /* "Converged": compares lastError with lastSO3Count (sic), :285 */
            if (st->last_so3_error < lastError && (double)fabsf(lastError - st->last_so3_count) < 0.001) break;
            else if ((double)st->last_so3_error > (double)lastError + 0.001) {
                st->last_so3_error = lastError; st->last_so3_count = lastCount;
                memcpy(resultR, lastResultR, sizeof(resultR));
                break;
            }
            lastError = st->last_so3_error; lastCount = st->last_so3_count;
            memcpy(lastResultR, resultR, sizeof(resultR));
            float delta[3];
            orc_ldlt_f(3, jtj, jtr, delta);
            double dd[3] = {delta[0], delta[1], delta[2]}, rotUpdate[9];
            orc_rodrigues(dd, rotUpdate);
            float ru[9], nr[9];
            for (int k = 0; k < 9; k++) ru[k] = (float)rotUpdate[k];
            orc_mul33f(ru, R_lr, nr);
            memcpy(R_lr, nr, sizeof(nr));
            for (int k = 0; k < 9; k++) resultR[k] = R_lr[k];
        }
    }

    int iterations[ORC_NUM_PYRS];
    iterations[0] = opts->fast_odom ? 3 : 10;
    iterations[1] = opts->pyramid ? 5 : 0;
    iterations[2] = opts->pyramid ? 4 : 0;

    float Rprev_inv[9];
    orc_inv33f(Rprev, Rprev_inv);

    double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (opts->so3)
        for (int x = 0; x < 3; x++)
            for (int y = 0; y < 3; y++) resultRt[x * 4 + y] = resultR[x * 3 + y];

    float residual[2] = {0, 0}; /* reference leaves this uninitialised when !icp (:401) */

    for (int i = ORC_NUM_PYRS - 1; i >= 0; i--) {
        const int cols = o->width >> i, rows = o->height >> i;
        const orc_cam il = cam_level(o->intr, i);
        if (rgb) orc_project_cloud(o->lastDepth[i], cols, rows, il, o->cloud[i]);
        double K[9], Kinv[9];
        k_matrix(il, K);
        orc_inv33d(K, Kinv);
        st->last_rgb_error = FLT_MAX;

        for (int j = 0; j < iterations[i]; j++) {
            double Rt[16];
            orc_inv44_affine_d(resultRt, Rt);
            double R[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
            double tmp[9], KRK[9];
            orc_mul33d(K, R, tmp); orc_mul33d(tmp, Kinv, KRK);
            float krkInv[9];
            for (int k = 0; k < 9; k++) krkInv[k] = (float)KRK[k];
            double tv[3] = {Rt[3], Rt[7], Rt[11]};
            float kt[3];
            for (int r = 0; r < 3; r++) kt[r] = (float)(K[r * 3 + 0] * tv[0] + K[r * 3 + 1] * tv[1] + K[r * 3 + 2] * tv[2]);

            int sigma = 0, rgbSize = 0;
            if (rgb) {
                float minScale = (float)(pow(o->minGrad[i], 2.0) / pow(o->sobelScale, 2.0));
                orc_rgb_residual(minScale, o->dIdx[i], o->dIdy[i], o->lastDepth[i], o->nextDepth[i], o->lastImage[i],
                                 o->nextImage[i], o->corres[i], o->maxDepthDeltaRGB, kt, krkInv, cols, rows, &sigma, &rgbSize);
            }
            float tmpError = (float)(sqrt((double)sigma) / rgbSize); /* sqrt(int)->double, / int, -> float */
            float sigmaVal = (tmpError == 0) ? 1 : (float)rgbSize;   /* (sic) the COUNT, :374 */
            if (rgbOnly && tmpError > st->last_rgb_error) break;
            st->last_rgb_error = tmpError; st->last_rgb_count = (float)rgbSize;
            if (rgbOnly) sigmaVal = -1;

            float A_icp[36], b_icp[6], A_rgbd[36], b_rgbd[6];
            memset(A_icp, 0, sizeof(A_icp)); memset(b_icp, 0, sizeof(b_icp));
            memset(A_rgbd, 0, sizeof(A_rgbd)); memset(b_rgbd, 0, sizeof(b_rgbd));
            if (icp) {
                int64_t sums[ORC_SE3_WORDS];
                orc_icp_step(Rcurr, tcurr, o->vmaps_curr[i], o->nmaps_curr[i], Rprev_inv, tprev, il, o->vmaps_g_prev[i],
                             o->nmaps_g_prev[i], o->distThres, o->angleThres, cols, rows, sums,
                             (i == 0 && j == iterations[i] - 1) ? icp_err_surface : 0);
                orc_icp_sums_to_host(sums, A_icp, b_icp, residual);
            }
            st->last_icp_error = sqrtf(residual[0]) / residual[1];
            st->last_icp_count = residual[1];
            if (rgb) {
                int64_t sums[ORC_SE3_WORDS]; float dummy[2];
                orc_rgb_step(o->corres[i], sigmaVal, o->cloud[i], il.fx, il.fy, o->dIdx[i], o->dIdy[i], o->sobelScale,
                             cols, rows, sums);
                orc_se3_sums_to_host(sums, orc_rgb_fix_bits(sigmaVal), A_rgbd, b_rgbd, dummy);
            }
            double lastA[36], lastb[6], result[6];
            if (icp && rgb) {
                double w = icpWeight;
                for (int k = 0; k < 36; k++) lastA[k] = (double)A_rgbd[k] + w * w * (double)A_icp[k];
                for (int k = 0; k < 6; k++) lastb[k] = (double)b_rgbd[k] + w * (double)b_icp[k];
            } else if (icp) {
                for (int k = 0; k < 36; k++) lastA[k] = A_icp[k];
                for (int k = 0; k < 6; k++) lastb[k] = b_icp[k];
            } else {
                for (int k = 0; k < 36; k++) lastA[k] = A_rgbd[k];
                for (int k = 0; k < 6; k++) lastb[k] = b_rgbd[k];
            }
            orc_ldlt_d(6, lastA, lastb, result);
            memcpy(st->lastA, lastA, sizeof(lastA)); memcpy(st->lastb, lastb, sizeof(lastb));

            /* OdometryProvider::computeUpdateSE3, OdometryProvider.h:69-89 */
            double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Rr[9], nrt[16];
            double rvec[3] = {result[3], result[4], result[5]};
            orc_rodrigues(rvec, Rr);
            for (int r = 0; r < 3; r++) { upd[r * 4 + 0] = Rr[r * 3 + 0]; upd[r * 4 + 1] = Rr[r * 3 + 1]; upd[r * 4 + 2] = Rr[r * 3 + 2]; upd[r * 4 + 3] = result[r]; }
            orc_mul44d(upd, resultRt, nrt);
            memcpy(resultRt, nrt, sizeof(nrt));
            /* rgbOdom (Isometry3f) = float(resultRt); currentT = [Rprev|tprev] * rgbOdom^-1 with the
             * isometry inverse [R^T | -R^T t], RGBDOdometry.cpp:452-460 */
            float Ro[9], to[3];
            for (int r = 0; r < 3; r++) { Ro[r * 3 + 0] = (float)resultRt[r * 4 + 0]; Ro[r * 3 + 1] = (float)resultRt[r * 4 + 1]; Ro[r * 3 + 2] = (float)resultRt[r * 4 + 2]; to[r] = (float)resultRt[r * 4 + 3]; }
            float Rinv[9] = {Ro[0], Ro[3], Ro[6], Ro[1], Ro[4], Ro[7], Ro[2], Ro[5], Ro[8]};
            float tinv[3];
            for (int r = 0; r < 3; r++) tinv[r] = -(Rinv[r * 3 + 0] * to[0] + Rinv[r * 3 + 1] * to[1] + Rinv[r * 3 + 2] * to[2]);
            orc_mul33f(Rprev, Rinv, Rcurr);
            for (int r = 0; r < 3; r++) tcurr[r] = (Rprev[r * 3 + 0] * tinv[0] + Rprev[r * 3 + 1] * tinv[1] + Rprev[r * 3 + 2] * tinv[2]) + tprev[r];
        }
    }

    if (rgb) { /* divergence guard :464-467 */
        float d[3] = {tcurr[0] - tprev[0], tcurr[1] - tprev[1], tcurr[2] - tprev[2]};
        if ((double)sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > 0.3) { memcpy(Rcurr, Rprev, 36); memcpy(tcurr, tprev, 12); }
    }
    if (opts->so3) /* :469-473 */
        for (int i = 0; i < ORC_NUM_PYRS; i++) { uint8_t *t = o->lastNextImage[i]; o->lastNextImage[i] = o->nextImage[i]; o->nextImage[i] = t; }
    memcpy(trans, tcurr, 12); memcpy(rot, Rcurr, 36);
}

/* ------------------------------------------------------------- covariance (-rl) ---- */
/* RGBDOdometry::getCovariance, RGBDOdometry.cpp:479: `lastA.cast<double>().lu().inverse()`.  Eigen's lu() is PartialPivLU: for every
 * column k the row with the largest |entry| at or below the diagonal becomes the pivot row (first maximum wins), the column below
 * the pivot is divided by it, the trailing block gets the rank-1 update; inverse() solves P A X = I column by column (unit-lower
 * forward substitution, upper back substitution).  Eigen is not in the tree: the ORDER of the additions inside its triangular
 * solves is not restated (it only matters in the last bits; the caller compares the diagonal with 1e-4, CoFusion.cpp:301-338).
 * A zero pivot is divided by as it stands, like Eigen does: the result then holds inf / NaN, and `NaN > 1e-4` is false. */
void orc_covariance(const double lastA[36], double cov[36])
{
    double lu[36];
    int perm[6];
    memcpy(lu, lastA, sizeof(lu));
    for (int i = 0; i < 6; i++) perm[i] = i;
    for (int k = 0; k < 6; k++) {
        int piv = k; double big = fabs(lu[k * 6 + k]);
        for (int r = k + 1; r < 6; r++) { const double a = fabs(lu[r * 6 + k]); if (a > big) { big = a; piv = r; } }
        if (piv != k) {
            for (int c = 0; c < 6; c++) { const double t = lu[k * 6 + c]; lu[k * 6 + c] = lu[piv * 6 + c]; lu[piv * 6 + c] = t; }
            const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        if (big != 0.0)   /* (Eigen skips the division of an all-zero column and records the singularity) */
            for (int r = k + 1; r < 6; r++) lu[r * 6 + k] /= lu[k * 6 + k];
        for (int r = k + 1; r < 6; r++)
            for (int c = k + 1; c < 6; c++) lu[r * 6 + c] -= lu[r * 6 + k] * lu[k * 6 + c];
    }
    for (int j = 0; j < 6; j++) {
        double x[6];
        for (int i = 0; i < 6; i++) x[i] = (perm[i] == j) ? 1.0 : 0.0;   /* P e_j */
        for (int i = 0; i < 6; i++)
            for (int c = 0; c < i; c++) x[i] -= lu[i * 6 + c] * x[c];
        for (int i = 5; i >= 0; i--) {
            for (int c = i + 1; c < 6; c++) x[i] -= lu[i * 6 + c] * x[c];
            x[i] /= lu[i * 6 + i];
        }
        for (int i = 0; i < 6; i++) cov[i * 6 + j] = x[i];
    }
}
