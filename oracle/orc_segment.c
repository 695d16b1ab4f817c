/*
 * orc_segment.c -- CPU ORACLE for the motion-segmentation stage.  TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Restates Core/Segmentation/Segmentation.cpp:59-706 (GT-mask branch and the CRF branch),
 * Core/Segmentation/Slic.{h,cpp} and Core/Segmentation/ConnectedLabels.hpp:50-172.
 *
 * PARITY UNPINNED, and for two pieces not even restatable from the tree: the reference calls the
 * third-party libraries gSLICr (carlren/gSLICr, cloned at HEAD by Scripts/install.sh:85; call sites
 * Slic.cpp:33-46,73-75) and densecrf (martinruenz/densecrf fork, install.sh:84; call sites
 * Segmentation.cpp:221,436-437,452,462-470), neither of which is vendored.  They are replaced by their
 * published algorithms, stated here:
 *   SLIC  = gSLICr's engine as published (carlren/gSLICr: gSLICr_Lib/engines/gSLICr_seg_engine.cpp `Perform_Segmentation`,
 *         gSLICr_seg_engine_GPU.cu constructor, gSLICr_seg_engine_shared.h), under the settings of Slic.cpp:33-43 (spixel_size 16,
 *         coh_weight 0.6, no_iters 5, RGB, GIVEN_SIZE, no connectivity enforcement).  Restated statement by statement:
 *           - init_cluster_centers_shared: centre k of grid cell (cx, cy) at pixel (cx*16 + 8, cy*16 + 8), colour of that pixel;
 *           - schedule of Perform_Segmentation: init; assign; 5 x {update; assign}  -- SIX association passes, the labels are
 *             those of the last one;
 *           - find_center_association_shared: the pixel's own grid cell (x/16, y/16) and its 3x3 neighbours, i (dy) outer, j (dx)
 *             inner, strict `<` (first minimum wins), distance compute_slic_distance =
 *                 sqrtf(dcolor * max_color_dist + coh_weight * dxy * max_xy_dist)
 *             with the constructor's normalisers max_color_dist = (5.0f / (1.7321f * 255))^2 (RGB case) and
 *             max_xy_dist = (1.0f / (1.4142f * spixel_size))^2, all in f32;
 *           - update_cluster_center + finalize_reduction_result_shared: centre and colour = f32 sum / (float)no_pixels over the
 *             pixels carrying the label (their sums are integers < 2^24, so gSLICr's f32 block reduction is exact and equals the
 *             integer sums used here); a cluster without pixels is left at centre (0,0), colour 0 (the reset value).
 *         What is NOT restated because it cannot be known from here: nvcc's contraction of `a*b + c` into FMA inside
 *         compute_slic_distance (this file, like every f32 expression of the oracle, is compiled without contraction), and image
 *         sizes that are not multiples of 16 (gSLICr rounds the grid up, Slic.cpp:28-30 rounds it down and would index past its
 *         tables; the pixel's cell is clamped to the last one here).  Until round 4 this stand-in used D = dRGB^2/20^2 +
 *         0.6 dxy^2/16^2 and 5 association passes -- a colour : space ratio ten times gSLICr's.
 *   CRF   (Kraehenbuehl & Koltun 2011, as used through DenseCRF2D): mean-field with an EXACT evaluation of
 *         the two Gaussian kernels over the 1200 nodes (densecrf approximates them on a permutohedral
 *         lattice), symmetric normalisation D^-1/2 K D^-1/2, Potts compatibility.
 * Per-superpixel means are formed from exact fixed-point (Q32) sums so that they do not depend on the
 * summation order (the reference sums f32 sequentially on the CPU, Slic.h:63-76).
 */
#include "orc.h"
#include "orc_math.h"

#include <float.h>
#include <stdlib.h>

#define SPIX 16

/* ------------------------------------------------------------------------- SLIC ---- */
void orc_slic(const uint8_t *rgba, int cols, int rows, int32_t *labels)
{
    const int gx = cols / SPIX, gy = rows / SPIX, K = gx * gy, iters = 5;
    float *c = malloc(sizeof(float) * 5 * (size_t)K); /* x y r g b */
    long long *sum = malloc(sizeof(long long) * 6 * (size_t)K);
    for (int cy = 0; cy < gy; cy++)
        for (int cx = 0; cx < gx; cx++) {
            const int px = cx * SPIX + SPIX / 2, py = cy * SPIX + SPIX / 2;
            const uint8_t *p = rgba + ((size_t)py * cols + px) * 4;
            float *k = c + (size_t)(cy * gx + cx) * 5;
            k[0] = (float)px; k[1] = (float)py; k[2] = (float)p[0]; k[3] = (float)p[1]; k[4] = (float)p[2];
        }
    /* seg_engine_GPU constructor: normalising factors, squared (f32 throughout) */
    float max_color_dist = 5.0f / (1.7321f * 255), max_xy_dist = 1.0f / (1.4142f * SPIX);
    max_color_dist *= max_color_dist; max_xy_dist *= max_xy_dist;
    const float weight = 0.6f; /* Slic.cpp:37 coh_weight */
    for (int it = 0; it <= iters; it++) {   /* Perform_Segmentation: assign; no_iters x {update; assign} */
        if (it > 0)
            for (int k = 0; k < K; k++) {   /* finalize_reduction_result_shared */
                const long long *s = sum + (size_t)k * 6;
                float *cc = c + (size_t)k * 5;
                for (int q = 0; q < 5; q++) cc[q] = s[5] ? (float)s[q] / (float)s[5] : 0.0f;
            }
        memset(sum, 0, sizeof(long long) * 6 * (size_t)K);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) {
                const uint8_t *p = rgba + ((size_t)y * cols + x) * 4;
                int cx0 = x / SPIX, cy0 = y / SPIX;
                if (cx0 >= gx) cx0 = gx - 1;
                if (cy0 >= gy) cy0 = gy - 1;
                float best = 999999.9999f; int bl = cy0 * gx + cx0;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        const int cx = cx0 + dx, cy = cy0 + dy;
                        if (cx < 0 || cy < 0 || cx >= gx || cy >= gy) continue;
                        const float *k = c + (size_t)(cy * gx + cx) * 5;
                        const float dr = (float)p[0] - k[2], dg = (float)p[1] - k[3], db = (float)p[2] - k[4];
                        const float ex = (float)x - k[0], ey = (float)y - k[1];
                        const float dcolor = dr * dr + dg * dg + db * db, dxy = ex * ex + ey * ey;
                        const float d = sqrtf(dcolor * max_color_dist + weight * dxy * max_xy_dist);
                        if (d < best) { best = d; bl = cy * gx + cx; }
                    }
                labels[(size_t)y * cols + x] = bl;
                long long *s = sum + (size_t)bl * 6;
                s[0] += x; s[1] += y; s[2] += p[0]; s[3] += p[1]; s[4] += p[2]; s[5] += 1;
            }
    }
    free(c); free(sum);
}

/* --------------------------------------------------- per-superpixel accumulation ---- */
/* Q32 sum of finite values (non-finite values contribute nothing; the reference zeroes non-finite
 * confidences afterwards, Segmentation.cpp:194-198).  Values are clamped to +-2^20 first. */
static inline int64_t q32(float v)
{
    if (!isfinite(v)) return 0;
    const float lim = 1048576.0f;
    const float c = fminf(fmaxf(v, -lim), lim);
    return (int64_t)llrint(ldexp((double)c, 32));
}

/* Slic::mapToHigh(index) + resampleEmptyIndex (Slic.h:192-206): note index / spixelY (sic) */
static int resample_empty_index(int index, int gx, int gy, int cols, int rows, const int32_t *labels)
{
    int x = (int)((index % gx) * SPIX + SPIX * 0.5), y = (int)((index / gy) * SPIX + SPIX * 0.5);
    if (y >= rows) y = rows - 1;
    if (x >= cols) x = cols - 1;
    return labels[(size_t)y * cols + x];
}

/* Slic::downsample<float> (Slic.h:48-83): in-place normalisation incl. the empty-superpixel fallback */
static void finish_mean(const int64_t *sumq, const unsigned *cnt_own, const unsigned *spixel_counts, int gx, int gy, int cols, int rows,
                        const int32_t *labels, float *out)
{
    const int K = gx * gy;
    for (int k = 0; k < K; k++) out[k] = (float)ldexp((double)sumq[k], -32);
    for (int k = 0; k < K; k++) {
        int cnt = (int)cnt_own[k], read = k;
        if (cnt == 0) { read = resample_empty_index(k, gx, gy, cols, rows, labels); cnt = (int)spixel_counts[read]; }
        out[k] = out[read] / (float)cnt;
    }
}

/* ------------------------------------------------------------- connected labels ---- */
typedef struct { unsigned char label; int top, right, bottom, left; int size; } comp_data;

/* ConnectedLabels.hpp:50-172 (4-connectivity union-find, roots renumbered in id order) */
static int connected_labels(const uint8_t *in, int cols, int rows, int *comp, comp_data **stats_out)
{
    int *roots = malloc(sizeof(int) * (size_t)cols * rows);
    int nroots = 0;
#define NEWC() (roots[nroots] = nroots, nroots++)
#define FINDROOT(i, r) do { int _i = (i); while (_i != roots[_i]) _i = roots[_i]; (r) = _i; } while (0)
    comp[0] = NEWC();
    for (int c = 1; c < cols; c++) comp[c] = (in[c] == in[c - 1]) ? comp[c - 1] : NEWC();
    for (int r = 1; r < rows; r++) {
        const uint8_t *row = in + (size_t)r * cols, *last = in + (size_t)(r - 1) * cols;
        int *cr = comp + (size_t)r * cols, *lc = comp + (size_t)(r - 1) * cols;
        cr[0] = (row[0] == last[0]) ? lc[0] : NEWC();
        for (int c = 1; c < cols; c++) {
            if (row[c] == row[c - 1]) {
                const int cLeft = cr[c - 1], cTop = lc[c];
                if (row[c] == last[c] && cLeft != cTop) {
                    int r1, r2;
                    FINDROOT(cTop, r1); FINDROOT(cLeft, r2);
                    if (r1 < r2) { roots[r2] = r1; cr[c] = r1; } else { roots[r1] = r2; cr[c] = r2; }
                } else cr[c] = cLeft;
            } else if (row[c] == last[c]) cr[c] = lc[c];
            else cr[c] = NEWC();
        }
    }
    int *mapping = malloc(sizeof(int) * (size_t)nroots);
    int rootCnt = 0;
    for (int id = 0; id < nroots; id++) {
        int root; FINDROOT(id, root);
        if (root == id) mapping[root] = rootCnt++;
        else roots[id] = root;
    }
    for (int id = 0; id < nroots; id++) roots[id] = mapping[roots[id]];
    comp_data *st = calloc((size_t)rootCnt, sizeof(comp_data));
    for (int i = 0; i < rootCnt; i++) { st[i].top = 2147483647; st[i].left = 2147483647; }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const int cc = roots[comp[(size_t)y * cols + x]];
            comp[(size_t)y * cols + x] = cc;
            comp_data *d = &st[cc];
            d->size++; d->label = in[(size_t)y * cols + x];
            if (y < d->top) d->top = y;
            if (y > d->bottom) d->bottom = y;
            if (x < d->left) d->left = x;
            if (x > d->right) d->right = x;
        }
    free(mapping); free(roots);
    *stats_out = st;
    return rootCnt;
#undef NEWC
#undef FINDROOT
}

/* public view of connected_labels for the reference pin (tests/test_cpu_refpin.py): stats rows {label, top, right, bottom, left, size} */
int orc_connected_labels(const uint8_t *in, int cols, int rows, int *comp, int *stats6, int max_stats)
{
    comp_data *st = NULL;
    const int n = connected_labels(in, cols, rows, comp, &st);
    for (int i = 0; i < n && i < max_stats; i++) {
        int *o = stats6 + (size_t)i * 6;
        o[0] = st[i].label; o[1] = st[i].top; o[2] = st[i].right; o[3] = st[i].bottom; o[4] = st[i].left; o[5] = st[i].size;
    }
    free(st);
    return n;
}

/* ------------------------------------------------------------------- dense CRF ---- */
/* expAndNormalize: column-wise softmax with max subtraction */
static void exp_and_normalize(const float *in, float *out, int L, int n)
{
    for (int i = 0; i < n; i++) {
        float mx = in[(size_t)i * L];
        for (int l = 1; l < L; l++) if (in[(size_t)i * L + l] > mx) mx = in[(size_t)i * L + l];
        float s = 0;
        for (int l = 0; l < L; l++) { const float e = orc_expf(in[(size_t)i * L + l] - mx); out[(size_t)i * L + l] = e; s += e; }
        for (int l = 0; l < L; l++) out[(size_t)i * L + l] = out[(size_t)i * L + l] / s;
    }
}

/* Summation order over the n nodes (normalisation and message passing): CRF_CHUNKS contiguous chunks of
 * ceil(n / CRF_CHUNKS) indices, index order inside a chunk, chunk totals added in chunk order.  (A fixed, blocked
 * order: the GPU evaluates the chunks in parallel and reproduces it bit for bit.) */
#define CRF_CHUNKS 16

/* Exact Gaussian kernel with symmetric normalisation: Kn[i][j] = n_i * exp(-0.5 |f_i - f_j|^2) * n_j,
 * n_i = 1/sqrt(sum_j exp(..) + 1e-20). */
static void crf_kernel(const float *feat, int D, int n, float *Kn)
{
    float *norm = malloc(sizeof(float) * (size_t)n);
    const int len = (n + CRF_CHUNKS - 1) / CRF_CHUNKS;
    for (int i = 0; i < n; i++) {
        float s = 0;
        for (int c = 0; c < CRF_CHUNKS; c++) {
            float p = 0;
            for (int j = c * len; j < n && j < (c + 1) * len; j++) {
                float d2 = 0;
                for (int d = 0; d < D; d++) { const float t = feat[(size_t)i * D + d] - feat[(size_t)j * D + d]; d2 += t * t; }
                const float k = orc_expf(-0.5f * d2);
                Kn[(size_t)i * n + j] = k;
                p += k;
            }
            s += p;
        }
        norm[i] = 1.0f / sqrtf(s + 1e-20f);
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Kn[(size_t)i * n + j] = norm[i] * Kn[(size_t)i * n + j] * norm[j];
    free(norm);
}

/* The three operations of the mean-field loop as separate entry points: the reference pin (oracle/ref_shim/include/densecrf.h)
 * lets the reference's OWN inference loop (Segmentation.cpp:452-470) call them in place of the absent densecrf library. */
void orc_crf_kernel(const float *feat, int D, int n, float *Kn) { crf_kernel(feat, D, n, Kn); }
void orc_crf_exp_and_normalize(const float *in, float *out, int L, int n) { exp_and_normalize(in, out, L, n); }
/* out[i][l] = -w * sum_j Kn[i][j] Q[j][l]  (PottsCompatibility applied to the filtered marginals), blocked summation order */
void orc_crf_apply(const float *Kn, int n, int L, float w, const float *Q, float *out)
{
    const int len = (n + CRF_CHUNKS - 1) / CRF_CHUNKS;
    for (int i = 0; i < n; i++)
        for (int l = 0; l < L; l++) {
            float a = 0;
            for (int c = 0; c < CRF_CHUNKS; c++) {
                float pa = 0;
                for (int j = c * len; j < n && j < (c + 1) * len; j++) pa += Kn[(size_t)i * n + j] * Q[(size_t)j * L + l];
                a += pa;
            }
            out[(size_t)i * L + l] = -w * a;
        }
}

void orc_crf_meanfield(const float *unary /* [n][L] */, int L, int n, const float *feat_smooth /* [n][2] */,
                       const float *feat_app /* [n][6] */, float w_smooth, float w_app, int iterations, float *Q /* [n][L] */)
{
    float *K1 = malloc(sizeof(float) * (size_t)n * n), *K2 = malloc(sizeof(float) * (size_t)n * n);
    float *tmp = malloc(sizeof(float) * (size_t)n * L), *Qn = malloc(sizeof(float) * (size_t)n * L);
    const int len = (n + CRF_CHUNKS - 1) / CRF_CHUNKS;
    crf_kernel(feat_smooth, 2, n, K1);
    crf_kernel(feat_app, 6, n, K2);
    for (size_t i = 0; i < (size_t)n * L; i++) tmp[i] = -unary[i];
    exp_and_normalize(tmp, Q, L, n);
    for (int it = 0; it < iterations; it++) {
        for (int i = 0; i < n; i++)
            for (int l = 0; l < L; l++) {
                float a = 0, b = 0;
                for (int c = 0; c < CRF_CHUNKS; c++) {
                    float pa = 0, pb = 0;
                    for (int j = c * len; j < n && j < (c + 1) * len; j++) {
                        pa += K1[(size_t)i * n + j] * Q[(size_t)j * L + l];
                        pb += K2[(size_t)i * n + j] * Q[(size_t)j * L + l];
                    }
                    a += pa; b += pb;
                }
                /* tmp1 = -unary; tmp1 -= (-w K Q) for each potential (Segmentation.cpp:462-469) */
                tmp[(size_t)i * L + l] = (-unary[(size_t)i * L + l] - (-w_smooth * a)) - (-w_app * b);
            }
        exp_and_normalize(tmp, Qn, L, n);
        memcpy(Q, Qn, sizeof(float) * (size_t)n * L);
    }
    free(K1); free(K2); free(tmp); free(Qn);
}

/* ----------------------------------------------------- performSegmentationCRF ---- */
/* Segmentation.cpp:124-706.  models[0] is the background.  icp_err[m] is the model's ICP error surface
 * [rows*cols], vertconf4[m] its splat vertexConf texture [rows*cols*4] (channel 3 = confidence). */
int orc_segment_crf(const orc_seg_params *P, int cols, int rows, const uint8_t *rgba, const float *depth, int n_models,
                    const unsigned *model_ids, const float *const *icp_err, const float *const *vertconf4, unsigned nextModelID,
                    int allowNew, uint8_t *full_seg, orc_seg_model *out_models, int *n_out, int *hasNewLabel, float *depthRange_out,
                    int32_t *labels_out, uint8_t *low_map_out)
{
    const int gx = cols / SPIX, gy = rows / SPIX, K = gx * gy;
    const size_t N = (size_t)cols * rows;
    const int numLabels = allowNew ? n_models + 1 : n_models;
    const float MAX_DEPTH = 100;
    int32_t *labels = labels_out ? labels_out : malloc(sizeof(int32_t) * N);
    orc_slic(rgba, cols, rows, labels);

    unsigned *spc = calloc((size_t)K, sizeof(unsigned)), *dcnt = calloc((size_t)K, sizeof(unsigned));
    int64_t *dsum = calloc((size_t)K, sizeof(int64_t));
    for (size_t i = 0; i < N; i++) {
        spc[labels[i]]++;
        if (depth[i] > 0.02f) { dsum[labels[i]] += q32(depth[i]); dcnt[labels[i]]++; }
    }
    float *lowDepth = malloc(sizeof(float) * (size_t)K);
    finish_mean(dsum, dcnt, spc, gx, gy, cols, rows, labels, lowDepth); /* downsampleThresholded, Slic.h:86-120 */

    float depthMin = FLT_MAX, depthMax = 0;
    for (int i = 0; i < K; i++) {
        const float d = lowDepth[i];
        if (d > MAX_DEPTH || d < 0 || !isfinite(d)) continue;
        if (depthMax < d) depthMax = d;
        if (depthMin > d) depthMin = d;
    }
    const float depthRange = depthMax - depthMin;
    *depthRange_out = depthRange;

    float **lowICP = malloc(sizeof(float *) * (size_t)(n_models + 1)), **lowConf = malloc(sizeof(float *) * (size_t)(n_models + 1));
    int64_t *acc = malloc(sizeof(int64_t) * (size_t)K);
    int modelIdToIndex[256];
    for (int i = 0; i < 256; i++) modelIdToIndex[i] = 0;
    for (int m = 0; m < n_models; m++) {
        lowICP[m] = malloc(sizeof(float) * (size_t)K); lowConf[m] = malloc(sizeof(float) * (size_t)K);
        memset(acc, 0, sizeof(int64_t) * (size_t)K);
        for (size_t i = 0; i < N; i++) acc[labels[i]] += q32(icp_err[m][i]);
        finish_mean(acc, spc, spc, gx, gy, cols, rows, labels, lowICP[m]);
        memset(acc, 0, sizeof(int64_t) * (size_t)K);
        for (size_t i = 0; i < N; i++) acc[labels[i]] += q32(vertconf4[m][i * 4 + 3]);
        finish_mean(acc, spc, spc, gx, gy, cols, rows, labels, lowConf[m]);
        out_models[m].id = model_ids[m]; out_models[m].superPixelCount = 0; out_models[m].avgConfidence = 0;
        out_models[m].depthMean = 0; out_models[m].depthStd = 0;
        out_models[m].top = 65535; out_models[m].left = 65535; out_models[m].right = 0; out_models[m].bottom = 0;
        modelIdToIndex[model_ids[m] & 255] = m;
        float avg = 0;
        for (int j = 0; j < K; j++) {
            float *cf = &lowConf[m][j];
            if (!isfinite(*cf)) { *cf = 0; continue; }
            avg += *cf;
        }
        out_models[m].avgConfidence = avg / (float)K;
    }
    int n_md = n_models;
    if (allowNew) {
        modelIdToIndex[nextModelID & 255] = n_models;
        orc_seg_model *nm = &out_models[n_models];
        nm->id = nextModelID; nm->superPixelCount = 0; nm->avgConfidence = 0; nm->depthMean = 0; nm->depthStd = 0;
        nm->top = 65535; nm->left = 65535; nm->right = 0; nm->bottom = 0;
        n_md++;
    }

    /* unaries, Segmentation.cpp:237-298 */
    const int L = numLabels;
    float *unary = malloc(sizeof(float) * (size_t)K * L);
    for (int k = 0; k < K; k++) {
        if ((double)lowConf[0][k] < 0.3) lowICP[0][k] = (float)((double)depthRange * 0.01);
        for (int i = 1; i < n_models; i++)
            if ((double)lowConf[i][k] <= 0.4) lowICP[i][k] = depthRange * P->unaryKError;
        float lowestError = lowICP[0][k] / depthRange;
        for (int i = 0; i < n_models; i++) {
            float error = lowICP[i][k];
            error /= depthRange;
            if (error < lowestError) lowestError = error;
            unary[(size_t)k * L + i] = P->unaryWeightError * error;
        }
        if (allowNew) unary[(size_t)k * L + n_models] = fmaxf(P->unaryThresholdNew - P->unaryWeightError * lowestError, 0.01f);
    }
    /* pairwise features, Segmentation.cpp:436-452 (the colour features index the FULL-res image with the
     * LOW-res index: :445-447) */
    float *f1 = malloc(sizeof(float) * (size_t)K * 2), *f2 = malloc(sizeof(float) * (size_t)K * 6);
    for (int j = 0; j < gy; j++)
        for (int i = 0; i < gx; i++) {
            const int index = j * gx + i;
            f1[index * 2 + 0] = (float)i / 2.0f; f1[index * 2 + 1] = (float)j / 2.0f; /* addPairwiseGaussian(2, 2) */
            f2[index * 6 + 0] = (float)i * P->scaleFeaturesPos;
            f2[index * 6 + 1] = (float)j * P->scaleFeaturesPos;
            f2[index * 6 + 2] = (float)rgba[(size_t)index * 4 + 0] * P->scaleFeaturesRGB;
            f2[index * 6 + 3] = (float)rgba[(size_t)index * 4 + 1] * P->scaleFeaturesRGB;
            f2[index * 6 + 4] = (float)rgba[(size_t)index * 4 + 2] * P->scaleFeaturesRGB;
            f2[index * 6 + 5] = fminf(lowDepth[index] * P->scaleFeaturesDepth, 100.0f);
        }
    for (size_t i = 0; i < (size_t)K * L; i++) if (unary[i] <= 1e-5f) unary[i] = 1e-5f;
    float *Q = malloc(sizeof(float) * (size_t)K * L);
    orc_crf_meanfield(unary, L, K, f1, f2, P->weightSmoothness, P->weightAppearance, P->crfIterations, Q);
    uint8_t *map = malloc((size_t)K);
    for (int i = 0; i < K; i++) {
        int m = 0; float best = Q[(size_t)i * L];
        for (int l = 1; l < L; l++) if (Q[(size_t)i * L + l] > best) { best = Q[(size_t)i * L + l]; m = l; }
        map[i] = (uint8_t)out_models[m].id;
    }

    /* component analysis, Segmentation.cpp:483-563 */
    int *comp = malloc(sizeof(int) * (size_t)K);
    comp_data *cc;
    const int ncc = connected_labels(map, gx, gy, comp, &cc);
    /* onlyKeepLargest: per label (except the smallest label value, std::next(begin)) keep the largest
     * component; ties keep the earlier one (:496-517) */
    {
        int minLabel = 256;
        for (int i = 0; i < ncc; i++) if (cc[i].label < minLabel) minLabel = cc[i].label;
        for (int lab = 0; lab < 256; lab++) {
            if (lab == minLabel) continue;
            int keep = -1;
            for (int i = 0; i < ncc; i++) {
                if (cc[i].label != lab) continue;
                if (keep < 0) { keep = i; continue; }
                if (cc[keep].size < cc[i].size) { cc[keep].label = 255; keep = i; } else cc[i].label = 255;
            }
        }
    }
    /* labelToComponents was built BEFORE the relabelling above and lists are pruned of removed entries */
    if (allowNew) { /* :521-530 */
        const int minSize = (int)((float)K * P->minRelSizeNew), maxSize = (int)((float)K * P->maxRelSizeNew);
        for (int i = 0; i < ncc; i++)
            if (cc[i].label == (nextModelID & 255) && (cc[i].size < minSize || cc[i].size > maxSize)) cc[i].label = 255;
    }
    for (int m = 0; m < n_md; m++) { /* bounding boxes over the components still listed for the label (:532-547) */
        orc_seg_model *md = &out_models[m];
        for (int i = 0; i < ncc; i++) {
            if (cc[i].label != (md->id & 255)) continue;
            if (cc[i].left < md->left) md->left = cc[i].left;
            if (cc[i].top < md->top) md->top = cc[i].top;
            if (cc[i].right > md->right) md->right = cc[i].right;
            if (cc[i].bottom > md->bottom) md->bottom = cc[i].bottom;
        }
        /* Slic::mapToHigh with the unsigned short members (Segmentation.h:44-47) */
        md->left = (unsigned short)(int)(md->left * SPIX + SPIX * 0.5); md->top = (unsigned short)(int)(md->top * SPIX + SPIX * 0.5);
        md->right = (unsigned short)(int)(md->right * SPIX + SPIX * 0.5); md->bottom = (unsigned short)(int)(md->bottom * SPIX + SPIX * 0.5);
    }
    {
        const unsigned borderSize = 20, fullHeight = (unsigned)rows, fullWidth = (unsigned)cols; /* :549-563 */
        for (int m = 0; m < n_md; m++) {
            orc_seg_model *md = &out_models[m];
            if (md->id == 0) continue;
            const unsigned top = (unsigned)md->top, bottom = (unsigned)md->bottom, left = (unsigned)md->left, right = (unsigned)md->right;
            if ((top < borderSize && bottom < borderSize) || (left < borderSize && right < borderSize) ||
                (top > fullHeight - borderSize && bottom > fullHeight - borderSize) ||
                (left > fullWidth - borderSize && right > fullWidth - borderSize))
                for (int i = 0; i < ncc; i++) if (cc[i].label == (md->id & 255)) cc[i].label = 255;
        }
    }
    for (int i = 0; i < K; i++) map[i] = cc[comp[i]].label;

    /* depth statistics with one trimming pass, Segmentation.cpp:570-621 */
    {
        float *sumsDepth = calloc((size_t)n_md, sizeof(float)), *sumsDev = calloc((size_t)n_md, sizeof(float));
        unsigned *cnts = calloc((size_t)n_md, sizeof(unsigned));
        for (int i = 0; i < K; i++) { if (map[i] == 255) continue; const int ix = modelIdToIndex[map[i]]; sumsDepth[ix] += lowDepth[i]; cnts[ix]++; }
        for (int m = 0; m < n_md; m++) out_models[m].depthMean = cnts[m] ? sumsDepth[m] / (float)cnts[m] : 0;
        for (int i = 0; i < K; i++) { if (map[i] == 255) continue; const int ix = modelIdToIndex[map[i]]; sumsDev[ix] += fabsf(out_models[ix].depthMean - lowDepth[i]); }
        for (int m = 0; m < n_md; m++) out_models[m].depthStd = cnts[m] ? sumsDev[m] / (float)cnts[m] : 0;
        for (int i = 0; i < K; i++) {
            if (map[i] == 255) continue;
            const int ix = modelIdToIndex[map[i]];
            if (ix != 0) {
                const float d = lowDepth[i];
                if ((double)d > 1.1 * (double)out_models[ix].depthStd + (double)out_models[ix].depthMean) {
                    sumsDepth[ix] -= d; sumsDev[ix] -= fabsf(out_models[ix].depthMean - d); cnts[ix]--;
                }
            }
        }
        for (int m = 0; m < n_md; m++) {
            out_models[m].depthMean = cnts[m] ? sumsDepth[m] / (float)cnts[m] : 0;
            out_models[m].depthStd = cnts[m] ? sumsDev[m] / (float)cnts[m] : 0;
        }
        free(sumsDepth); free(sumsDev); free(cnts);
    }
    for (int k = 0; k < K; k++) { if (map[k] == 255) continue; out_models[modelIdToIndex[map[k]]].superPixelCount++; }
    *hasNewLabel = 0;
    if (allowNew) {
        if (out_models[n_md - 1].superPixelCount > 0) *hasNewLabel = 1;
        else n_md--;
    }
    *n_out = n_md;
    for (size_t i = 0; i < N; i++) full_seg[i] = map[labels[i]]; /* Slic::upsample, Slic.h:127-139 */
    if (low_map_out) memcpy(low_map_out, map, (size_t)K);

    for (int m = 0; m < n_models; m++) { free(lowICP[m]); free(lowConf[m]); }
    free(lowICP); free(lowConf); free(acc); free(spc); free(dcnt); free(dsum); free(lowDepth); free(unary); free(f1); free(f2); free(Q);
    free(map); free(comp); free(cc);
    if (!labels_out) free(labels);
    return 0;
}

/* ------------------------------------------------ GT-mask branch (Segmentation.cpp:59-119) ---- */
/* `mapping` is the function-static table of the reference (persisting across frames): 256 bytes owned by the caller. */
int orc_segment_gt(const uint8_t *gt_mask, const float *depth, int cols, int rows, int n_models, const unsigned *model_ids,
                   unsigned nextModelID, int allowNew, uint8_t *mapping, uint8_t *full_seg, orc_seg_model *out_models, int *n_out,
                   int *hasNewLabel)
{
    const size_t N = (size_t)cols * rows;
    unsigned outIds[256];
    int modelIdToIndex[256];
    memset(outIds, 0, sizeof(outIds));
    for (int i = 0; i < 256; i++) modelIdToIndex[i] = 0;
    for (int m = 0; m < n_models; m++) modelIdToIndex[model_ids[m] & 255] = m;
    modelIdToIndex[nextModelID & 255] = n_models;
    *hasNewLabel = 0;
    memset(full_seg, 0, N);
    for (size_t i = 0; i < N; i++) {
        const uint8_t vIn = gt_mask[i];
        if (vIn) {
            if (mapping[vIn] != 0) { full_seg[i] = mapping[vIn]; outIds[full_seg[i]]++; }
            else if (allowNew && !*hasNewLabel) { full_seg[i] = (uint8_t)nextModelID; mapping[vIn] = (uint8_t)nextModelID; *hasNewLabel = 1; outIds[full_seg[i]]++; }
        } else outIds[0]++;
    }
    int n_md = 0;
    for (int m = 0; m < n_models; m++) {
        orc_seg_model *md = &out_models[n_md++];
        memset(md, 0, sizeof(*md));
        md->id = model_ids[m]; md->superPixelCount = outIds[model_ids[m] & 255] / (16 * 16); md->avgConfidence = 0.4f;
        md->top = 65535; md->left = 65535;
    }
    if (*hasNewLabel) {
        orc_seg_model *md = &out_models[n_md++];
        memset(md, 0, sizeof(*md));
        md->id = nextModelID; md->avgConfidence = 0.4f; md->top = 65535; md->left = 65535;
        const float c = (float)(outIds[nextModelID & 255] / (16 * 16));
        md->superPixelCount = (unsigned)(c > 1.0f ? c : 1.0f);
    }
    unsigned *cnts = calloc((size_t)n_md + 1, sizeof(unsigned));
    for (size_t i = 0; i < N; i++) { const int ix = modelIdToIndex[full_seg[i]]; out_models[ix].depthMean += depth[i]; cnts[ix]++; }
    for (int m = 0; m < n_md; m++) out_models[m].depthMean /= cnts[m] ? (float)cnts[m] : 1.0f;
    for (size_t i = 0; i < N; i++) { const int ix = modelIdToIndex[full_seg[i]]; out_models[ix].depthStd += fabsf(out_models[ix].depthMean - depth[i]); }
    for (int m = 0; m < n_md; m++) out_models[m].depthStd /= cnts[m] ? (float)cnts[m] : 1.0f;
    free(cnts);
    *n_out = n_md;
    return 0;
}
