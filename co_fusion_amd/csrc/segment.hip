// segment.hip -- GPU side of the motion segmentation (Core/Segmentation/*): SLIC superpixels, the
// per-superpixel accumulations that replace the reference's full-resolution texture downloads
// (Segmentation.cpp:184-188: 6.1 MB of glGetTexImage per model per frame), the dense-CRF mean field over
// the 40x30 superpixel grid, and the label up-sampling.  Plus its C-ABI (cf_seg_*).
//
// gSLICr and densecrf are third-party and not vendored by the reference (Scripts/install.sh:84-85); they are
// replaced by their published algorithms exactly as stated in oracle/orc_segment.c (same arithmetic, same
// summation order => bit-identical to the oracle):
//   * SLIC: one 16x16-pixel workgroup per grid cell; a pixel can only join one of the 3x3 neighbouring
//     clusters, so each workgroup privatises 9 x 6 integer accumulators in LDS and issues 54 global atomics.
//   * accumulation: same tiling; exact Q32 fixed-point sums (order independent).
//   * CRF: the two 1200x1200 Gaussian kernels are evaluated exactly (no permutohedral lattice), stored
//     transposed so that the sequential-in-j mean-field sums read coalesced rows.
#include <string.h>

#include <string>
#include <vector>

#include "cf_host.h"
#include "cf_surfel_device.h"

using namespace cf;

namespace cf {

constexpr int kSpix = 16;
constexpr int kMaxL = 16;  // labels incl. the "new model" label

// ---------------------------------------------------------------------------------- SLIC ----
__global__ void slic_init_kernel(const uchar4* __restrict__ rgba, int cols, int gx, int K, float* __restrict__ centres)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int cx = k % gx, cy = k / gx;
    const int px = cx * kSpix + kSpix / 2, py = cy * kSpix + kSpix / 2;
    const uchar4 p = rgba[py * cols + px];
    float* c = centres + k * 5;
    c[0] = (float)px; c[1] = (float)py; c[2] = (float)p.x; c[3] = (float)p.y; c[4] = (float)p.z;
}

// one 16x16 workgroup per grid cell
__global__ void __launch_bounds__(256) slic_assign_kernel(const uchar4* __restrict__ rgba, int cols, int rows, int gx, int gy,
                                                          const float* __restrict__ centres, int* __restrict__ labels,
                                                          unsigned long long* __restrict__ sums /* [K][6] */)
{
    __shared__ float s_c[9][5];
    __shared__ int s_lab[9];
    __shared__ unsigned s_acc[9][6];
    const int cx0 = blockIdx.x, cy0 = blockIdx.y;
    const int t = threadIdx.x;
    if (t < 9) {
        const int dx = t % 3 - 1, dy = t / 3 - 1;
        const int cx = cx0 + dx, cy = cy0 + dy;
        const bool ok = !(cx < 0 || cy < 0 || cx >= gx || cy >= gy);
        s_lab[t] = ok ? cy * gx + cx : -1;
        for (int q = 0; q < 5; q++) s_c[t][q] = ok ? centres[(cy * gx + cx) * 5 + q] : 0.f;
    }
    if (t < 54) s_acc[t / 6][t % 6] = 0;
    __syncthreads();
    const int x = cx0 * kSpix + (t & 15), y = cy0 * kSpix + (t >> 4);
    const uchar4 p = rgba[y * cols + x];
    const float inv_color = 1.0f / (20.0f * 20.0f), inv_xy = 0.6f / ((float)kSpix * (float)kSpix);
    float best = 3.402823466e+38F; int bi = 4;
#pragma unroll
    for (int n = 0; n < 9; n++) {  // dy-major, dx-minor: same scan order as the oracle
        if (s_lab[n] < 0) continue;
        const float dr = (float)p.x - s_c[n][2], dg = (float)p.y - s_c[n][3], db = (float)p.z - s_c[n][4];
        const float ex = (float)x - s_c[n][0], ey = (float)y - s_c[n][1];
        const float d = (dr * dr + dg * dg + db * db) * inv_color + (ex * ex + ey * ey) * inv_xy;
        if (d < best) { best = d; bi = n; }
    }
    labels[y * cols + x] = s_lab[bi];
    atomicAdd(&s_acc[bi][0], (unsigned)x); atomicAdd(&s_acc[bi][1], (unsigned)y); atomicAdd(&s_acc[bi][2], (unsigned)p.x);
    atomicAdd(&s_acc[bi][3], (unsigned)p.y); atomicAdd(&s_acc[bi][4], (unsigned)p.z); atomicAdd(&s_acc[bi][5], 1u);
    __syncthreads();
    if (t < 54) {
        const int n = t / 6, q = t % 6;
        if (s_lab[n] >= 0 && s_acc[n][q]) atomicAdd(&sums[(size_t)s_lab[n] * 6 + q], (unsigned long long)s_acc[n][q]);
    }
}

__global__ void slic_update_kernel(unsigned long long* __restrict__ sums, int K, float* __restrict__ centres)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    unsigned long long* s = sums + (size_t)k * 6;
    if (s[5] != 0)
        for (int q = 0; q < 5; q++) centres[k * 5 + q] = (float)(long long)s[q] / (float)(long long)s[5];
    for (int q = 0; q < 6; q++) s[q] = 0;
}

// -------------------------------------------------------------- per-superpixel sums ----
__device__ __forceinline__ long long q32(float v)
{
    if (!is_finite(v)) return 0;
    const float c = fminf(fmaxf(v, -1048576.0f), 1048576.0f);
    return __double2ll_rn((double)c * 4294967296.0);
}

struct AccArgs {
    const int* labels; const float* depth;
    const float* icp[kMaxL]; const float4* vconf[kMaxL];
    int n_models, cols, rows, gx, gy;
    unsigned* spix_count;            // [K]
    unsigned* depth_count;           // [K]
    unsigned long long* depth_sum;   // [K]
    unsigned long long* icp_sum;     // [n][K]
    unsigned long long* conf_sum;    // [n][K]
};

__global__ void __launch_bounds__(256) seg_accumulate_kernel(const AccArgs a)
{
    __shared__ int s_lab[9];
    __shared__ unsigned s_cnt[9], s_dcnt[9];
    __shared__ unsigned long long s_dsum[9];
    __shared__ unsigned long long s_icp[kMaxL][9], s_conf[kMaxL][9];
    const int cx0 = blockIdx.x, cy0 = blockIdx.y, t = threadIdx.x;
    const int K = a.gx * a.gy;
    if (t < 9) {
        const int dx = t % 3 - 1, dy = t / 3 - 1, cx = cx0 + dx, cy = cy0 + dy;
        s_lab[t] = (cx < 0 || cy < 0 || cx >= a.gx || cy >= a.gy) ? -1 : cy * a.gx + cx;
        s_cnt[t] = 0; s_dcnt[t] = 0; s_dsum[t] = 0;
    }
    for (int k = t; k < kMaxL * 9; k += 256) { s_icp[k / 9][k % 9] = 0; s_conf[k / 9][k % 9] = 0; }
    __syncthreads();
    const int x = cx0 * kSpix + (t & 15), y = cy0 * kSpix + (t >> 4);
    const int q = y * a.cols + x;
    const int lab = a.labels[q];
    int slot = 4;
#pragma unroll
    for (int n = 0; n < 9; n++) if (s_lab[n] == lab) slot = n;
    atomicAdd(&s_cnt[slot], 1u);
    const float d = a.depth[q];
    if (d > 0.02f) { atomicAdd(&s_dcnt[slot], 1u); atomicAdd(&s_dsum[slot], (unsigned long long)q32(d)); }
    for (int m = 0; m < a.n_models; m++) {
        atomicAdd(&s_icp[m][slot], (unsigned long long)q32(a.icp[m][q]));
        atomicAdd(&s_conf[m][slot], (unsigned long long)q32(a.vconf[m][q].w));
    }
    __syncthreads();
    if (t < 9 && s_lab[t] >= 0) {
        const int L = s_lab[t];
        if (s_cnt[t]) atomicAdd(&a.spix_count[L], s_cnt[t]);
        if (s_dcnt[t]) { atomicAdd(&a.depth_count[L], s_dcnt[t]); atomicAdd(&a.depth_sum[L], s_dsum[t]); }
        for (int m = 0; m < a.n_models; m++) {
            if (s_icp[m][t]) atomicAdd(&a.icp_sum[(size_t)m * K + L], s_icp[m][t]);
            if (s_conf[m][t]) atomicAdd(&a.conf_sum[(size_t)m * K + L], s_conf[m][t]);
        }
    }
}

// labels at the "empty superpixel" resample coordinates (Slic.h:192-206; index / spixelY is the reference's)
__global__ void seg_resample_kernel(const int* __restrict__ labels, int cols, int rows, int gx, int gy, int* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= gx * gy) return;
    int x = (int)((k % gx) * kSpix + kSpix * 0.5), y = (int)((k / gy) * kSpix + kSpix * 0.5);
    if (y >= rows) y = rows - 1;
    if (x >= cols) x = cols - 1;
    out[k] = labels[y * cols + x];
}

__global__ void __launch_bounds__(256) seg_upsample_kernel(const int* __restrict__ labels, const unsigned char* __restrict__ low_map, int N,
                                                           unsigned char* __restrict__ full)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) full[i] = low_map[labels[i]];
}

// ------------------------------------------------------------------------------- dense CRF ----
template <int D>
__global__ void __launch_bounds__(256) crf_raw_kernel(const float* __restrict__ feat, int n, float* __restrict__ raw)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    float d2 = 0;
#pragma unroll
    for (int d = 0; d < D; d++) { const float t = feat[i * D + d] - feat[j * D + d]; d2 += t * t; }
    raw[idx] = det_expf(-0.5f * d2);
}
// All sums over the n nodes (normalisation and message passing) run in kCrfChunks contiguous chunks of
// ceil(n / kCrfChunks) indices: sequential inside a chunk, chunk totals added in chunk order (the oracle states
// the same order).  A thread that walks all n nodes alone made the 1200-node mean field 61 % of a multi-object
// frame (10 x 192 us + 2 x 294 us, measured); with the chunked order one wave covers 64 nodes x one chunk and
// (n/64) x 16 workgroups spread over the whole device.
constexpr int kCrfChunks = 16;
// partial[c][i] = sum over chunk c of raw[i][j]  (raw is bitwise symmetric: read column-wise, coalesced)
__global__ void __launch_bounds__(64) crf_norm_partial_kernel(const float* __restrict__ raw, int n, float* __restrict__ partial)
{
    const int i = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y;
    if (i >= n) return;
    const int len = (n + kCrfChunks - 1) / kCrfChunks, j0 = c * len, j1 = min(n, j0 + len);
    float s = 0;
    for (int j = j0; j < j1; j++) s += raw[j * n + i];
    partial[c * n + i] = s;
}
// norm_i = 1/sqrt(sum_c partial[c][i] + 1e-20)
__global__ void __launch_bounds__(256) crf_norm_kernel(const float* __restrict__ partial, int n, float* __restrict__ norm)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0;
    for (int c = 0; c < kCrfChunks; c++) s += partial[c * n + i];
    norm[i] = 1.0f / sqrtf(s + 1e-20f);
}
// Kt[j][i] = (norm_i * raw[i][j]) * norm_j   (value of the symmetric-normalised kernel K[i][j], stored transposed)
__global__ void __launch_bounds__(256) crf_scale_kernel(const float* __restrict__ raw, const float* __restrict__ norm, int n,
                                                        float* __restrict__ Kt)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int j = idx / n, i = idx - j * n;
    Kt[idx] = norm[i] * raw[i * n + j] * norm[j];
}
// expAndNormalize of -unary
__global__ void __launch_bounds__(256) crf_init_kernel(const float* __restrict__ unary, int L, int n, float* __restrict__ Q)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float mx = -unary[i * L];
    for (int l = 1; l < L; l++) if (-unary[i * L + l] > mx) mx = -unary[i * L + l];
    float e[kMaxL], s = 0;
    for (int l = 0; l < L; l++) { e[l] = det_expf(-unary[i * L + l] - mx); s += e[l]; }
    for (int l = 0; l < L; l++) Q[i * L + l] = e[l] / s;
}
// one mean-field step, part 1: chunk partials of K1*Q and K2*Q for every label; partial[((c*n + i)*2 + which)*kMaxL + l].
// LT > 0: the label count as a compile-time constant (no predication); the chunk is walked five nodes at a time so
// that the 10 kernel-matrix loads of a group are in flight together (the sums stay in node order).
template <int LT>
__global__ void __launch_bounds__(64) crf_message_kernel(int L, int n, const float* __restrict__ K1t, const float* __restrict__ K2t,
                                                         const float* __restrict__ Q, float* __restrict__ partial)
{
    constexpr int LL = LT > 0 ? LT : kMaxL;
    const int i = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y;
    if (i >= n) return;
    const int len = (n + kCrfChunks - 1) / kCrfChunks, j0 = c * len, j1 = min(n, j0 + len);
    float a[LL], b[LL];
#pragma unroll
    for (int l = 0; l < LL; l++) { a[l] = 0; b[l] = 0; }
    int j = j0;
    for (; j + 5 <= j1; j += 5) {
        float k1[5], k2[5];
#pragma unroll
        for (int u = 0; u < 5; u++) { k1[u] = K1t[(j + u) * n + i]; k2[u] = K2t[(j + u) * n + i]; }
#pragma unroll
        for (int u = 0; u < 5; u++)
#pragma unroll
            for (int l = 0; l < LL; l++)
                if (LT > 0 || l < L) {
                    const float q = Q[(j + u) * L + l];
                    a[l] += k1[u] * q;
                    b[l] += k2[u] * q;
                }
    }
    for (; j < j1; j++) {
        const float k1 = K1t[j * n + i], k2 = K2t[j * n + i];
#pragma unroll
        for (int l = 0; l < LL; l++)
            if (LT > 0 || l < L) {
                const float q = Q[j * L + l];
                a[l] += k1 * q;
                b[l] += k2 * q;
            }
    }
    float* out = partial + ((size_t)(c * n + i) * 2) * kMaxL;
#pragma unroll
    for (int l = 0; l < LL; l++)
        if (LT > 0 || l < L) { out[l] = a[l]; out[kMaxL + l] = b[l]; }
}
static void launch_crf_message(hipStream_t st, dim3 grid, int L, int n, const float* K1t, const float* K2t, const float* Q, float* partial)
{
    switch (L) {
        case 1: crf_message_kernel<1><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
        case 2: crf_message_kernel<2><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
        case 3: crf_message_kernel<3><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
        case 4: crf_message_kernel<4><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
        case 5: crf_message_kernel<5><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
        case 6: crf_message_kernel<6><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
        case 7: crf_message_kernel<7><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
        case 8: crf_message_kernel<8><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
        default: crf_message_kernel<0><<<grid, 64, 0, st>>>(L, n, K1t, K2t, Q, partial); break;
    }
}
// part 2: chunk totals in chunk order, unary, softmax over the labels.  Thread (node g, label l): 16 nodes x 16 label
// slots per workgroup; the 32 chunk partials of a (node, label) are loaded independently and summed in chunk order,
// the softmax runs over the node's LDS row exactly like expAndNormalize.
__global__ void __launch_bounds__(256) crf_update_kernel(const float* __restrict__ unary, int L, int n, const float* __restrict__ partial,
                                                         float w_smooth, float w_app, float* __restrict__ Qn)
{
    __shared__ float s_t[16][kMaxL];
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + g;
    float tmp = 0;
    if (i < n && l < L) {
        float pa[kCrfChunks], pb[kCrfChunks];
#pragma unroll
        for (int c = 0; c < kCrfChunks; c++) {
            const float* in = partial + ((size_t)(c * n + i) * 2) * kMaxL;
            pa[c] = in[l]; pb[c] = in[kMaxL + l];
        }
        float a = 0, b = 0;
#pragma unroll
        for (int c = 0; c < kCrfChunks; c++) { a += pa[c]; b += pb[c]; }
        tmp = (-unary[i * L + l] - (-w_smooth * a)) - (-w_app * b);
        s_t[g][l] = tmp;
    }
    __syncthreads();
    if (i < n && l < L) {
        float mx = s_t[g][0];
        for (int k = 1; k < L; k++) if (s_t[g][k] > mx) mx = s_t[g][k];
        float sum = 0;
        for (int k = 0; k < L; k++) sum += det_expf(s_t[g][k] - mx);
        Qn[i * L + l] = det_expf(tmp - mx) / sum;
    }
}

}  // namespace cf

// ===================================================================================== C-ABI ====
#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->set_error(std::string(#call) + ": " + hipGetErrorString(e_));               \
            return CF_EHIP;                                                                    \
        }                                                                                      \
    } while (0)
#define LAUNCHCHK(ctx) HIPCHK(ctx, hipGetLastError())

struct cf_segmenter {
    cf_ctx* ctx = nullptr;
    int gx = 0, gy = 0, K = 0;
    int* labels = nullptr;
    float* centres = nullptr;
    unsigned long long* slic_sums = nullptr;
    unsigned* spix_count = nullptr; unsigned* depth_count = nullptr;
    unsigned long long *depth_sum = nullptr, *icp_sum = nullptr, *conf_sum = nullptr;
    int* resample = nullptr;
    unsigned char* low_map = nullptr;
    float *feat1 = nullptr, *feat2 = nullptr, *raw = nullptr, *norm = nullptr, *K1t = nullptr, *K2t = nullptr;
    float* partial = nullptr;            // chunk partial sums [kCrfChunks][K][2][kMaxL]
    std::vector<float> smooth_cache;     // host copy of the smoothness features K1t was built from
    float *unary = nullptr, *Q0 = nullptr, *Q1 = nullptr;
};

template <typename T>
static int seg_malloc(cf_ctx* ctx, T** p, size_t count)
{
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    HIPCHK(ctx, hipMemsetAsync(*p, 0, count * sizeof(T), ctx->stream));
    return CF_OK;
}

extern "C" {

int cf_seg_create(cf_ctx* ctx, cf_segmenter** out)
{
    if (!ctx || !out) return CF_EINVAL;
    if ((ctx->cfg.width % kSpix) || (ctx->cfg.height % kSpix)) { ctx->set_error("segmentation needs width/height multiples of 16"); return CF_EINVAL; }
    cf_segmenter* s = new cf_segmenter();
    s->ctx = ctx; s->gx = ctx->cfg.width / kSpix; s->gy = ctx->cfg.height / kSpix; s->K = s->gx * s->gy;
    *out = s;
    const size_t N = (size_t)ctx->cfg.width * ctx->cfg.height, K = (size_t)s->K;
    if (int r = seg_malloc(ctx, &s->labels, N)) return r;
    if (int r = seg_malloc(ctx, &s->centres, K * 5)) return r;
    if (int r = seg_malloc(ctx, &s->slic_sums, K * 6)) return r;
    if (int r = seg_malloc(ctx, &s->spix_count, K)) return r;
    if (int r = seg_malloc(ctx, &s->depth_count, K)) return r;
    if (int r = seg_malloc(ctx, &s->depth_sum, K)) return r;
    if (int r = seg_malloc(ctx, &s->icp_sum, K * kMaxL)) return r;
    if (int r = seg_malloc(ctx, &s->conf_sum, K * kMaxL)) return r;
    if (int r = seg_malloc(ctx, &s->resample, K)) return r;
    if (int r = seg_malloc(ctx, &s->low_map, K)) return r;
    if (int r = seg_malloc(ctx, &s->feat1, K * 2)) return r;
    if (int r = seg_malloc(ctx, &s->feat2, K * 6)) return r;
    if (int r = seg_malloc(ctx, &s->raw, K * K)) return r;
    if (int r = seg_malloc(ctx, &s->norm, K)) return r;
    if (int r = seg_malloc(ctx, &s->K1t, K * K)) return r;
    if (int r = seg_malloc(ctx, &s->K2t, K * K)) return r;
    if (int r = seg_malloc(ctx, &s->partial, (size_t)kCrfChunks * K * 2 * kMaxL)) return r;
    if (int r = seg_malloc(ctx, &s->unary, K * kMaxL)) return r;
    if (int r = seg_malloc(ctx, &s->Q0, K * kMaxL)) return r;
    if (int r = seg_malloc(ctx, &s->Q1, K * kMaxL)) return r;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

void cf_seg_destroy(cf_segmenter* s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    void* ptrs[] = {s->labels, s->centres, s->slic_sums, s->spix_count, s->depth_count, s->depth_sum, s->icp_sum, s->conf_sum, s->resample,
                    s->low_map, s->feat1, s->feat2, s->raw, s->norm, s->K1t, s->K2t, s->partial, s->unary, s->Q0, s->Q1};
    for (void* p : ptrs) (void)hipFree(p);
    delete s;
}

// Slic::setInputImage + processFrame (Slic.cpp:48-81): labels stay on the device
int cf_seg_slic(cf_segmenter* s, const uint8_t* rgba)
{
    if (!s || !rgba) return CF_EINVAL;
    cf_ctx* ctx = s->ctx; hipStream_t st = ctx->stream;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const uchar4* img = reinterpret_cast<const uchar4*>(rgba);
    slic_init_kernel<<<(s->K + 255) / 256, 256, 0, st>>>(img, W, s->gx, s->K, s->centres);
    HIPCHK(ctx, hipMemsetAsync(s->slic_sums, 0, sizeof(unsigned long long) * 6 * s->K, st));
    for (int it = 0; it < 5; it++) {
        slic_assign_kernel<<<dim3(s->gx, s->gy), 256, 0, st>>>(img, W, H, s->gx, s->gy, s->centres, s->labels, s->slic_sums);
        slic_update_kernel<<<(s->K + 255) / 256, 256, 0, st>>>(s->slic_sums, s->K, s->centres);
    }
    LAUNCHCHK(ctx);
    return CF_OK;
}

// Slic::downsample* sums (Slic.h:48-120): exact Q32 sums per superpixel; results copied to the host arrays
// (synchronous).  icp_err[m]: ICP error surface f32 [H*W]; vertconf4[m]: splat vertexConf f32x4 [H*W].
int cf_seg_accumulate(cf_segmenter* s, const float* depth, int n_models, const float* const* icp_err, const float* const* vertconf4,
                      uint32_t* spix_count_host, uint32_t* depth_count_host, int64_t* depth_sum_host, int64_t* icp_sum_host,
                      int64_t* conf_sum_host, int32_t* resample_labels_host)
{
    if (!s || !depth || n_models < 0 || n_models > kMaxL) return CF_EINVAL;
    cf_ctx* ctx = s->ctx; hipStream_t st = ctx->stream;
    const size_t K = (size_t)s->K;
    HIPCHK(ctx, hipMemsetAsync(s->spix_count, 0, sizeof(unsigned) * K, st));
    HIPCHK(ctx, hipMemsetAsync(s->depth_count, 0, sizeof(unsigned) * K, st));
    HIPCHK(ctx, hipMemsetAsync(s->depth_sum, 0, sizeof(unsigned long long) * K, st));
    HIPCHK(ctx, hipMemsetAsync(s->icp_sum, 0, sizeof(unsigned long long) * K * kMaxL, st));
    HIPCHK(ctx, hipMemsetAsync(s->conf_sum, 0, sizeof(unsigned long long) * K * kMaxL, st));
    AccArgs a;
    memset(&a, 0, sizeof(a));
    a.labels = s->labels; a.depth = depth; a.n_models = n_models; a.cols = ctx->cfg.width; a.rows = ctx->cfg.height; a.gx = s->gx; a.gy = s->gy;
    for (int m = 0; m < n_models; m++) { a.icp[m] = icp_err[m]; a.vconf[m] = reinterpret_cast<const float4*>(vertconf4[m]); }
    a.spix_count = s->spix_count; a.depth_count = s->depth_count; a.depth_sum = s->depth_sum; a.icp_sum = s->icp_sum; a.conf_sum = s->conf_sum;
    seg_accumulate_kernel<<<dim3(s->gx, s->gy), 256, 0, st>>>(a);
    seg_resample_kernel<<<(s->K + 255) / 256, 256, 0, st>>>(s->labels, ctx->cfg.width, ctx->cfg.height, s->gx, s->gy, s->resample);
    LAUNCHCHK(ctx);
    HIPCHK(ctx, hipMemcpyAsync(spix_count_host, s->spix_count, sizeof(unsigned) * K, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(depth_count_host, s->depth_count, sizeof(unsigned) * K, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(depth_sum_host, s->depth_sum, sizeof(long long) * K, hipMemcpyDeviceToHost, st));
    if (n_models) {
        HIPCHK(ctx, hipMemcpyAsync(icp_sum_host, s->icp_sum, sizeof(long long) * K * n_models, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(conf_sum_host, s->conf_sum, sizeof(long long) * K * n_models, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(ctx, hipMemcpyAsync(resample_labels_host, s->resample, sizeof(int) * K, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return CF_OK;
}

// DenseCRF2D inference as used by Segmentation.cpp:436-480 (exact kernels, see the file header).
// unary [K*L] row-per-node, feat_smooth [K*2], feat_app [K*6] host in; Q [K*L] host out (synchronous).
int cf_seg_crf(cf_segmenter* s, const float* unary_host, int L, const float* feat_smooth_host, const float* feat_app_host,
               float w_smooth, float w_app, int iterations, float* Q_host)
{
    if (!s || !unary_host || !Q_host || L <= 0 || L > kMaxL) return CF_EINVAL;
    cf_ctx* ctx = s->ctx; hipStream_t st = ctx->stream;
    const int n = s->K;
    HIPCHK(ctx, hipMemcpyAsync(s->unary, unary_host, sizeof(float) * n * L, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(s->feat2, feat_app_host, sizeof(float) * n * 6, hipMemcpyHostToDevice, st));
    const int g2 = (n * n + 255) / 256, g1 = (n + 255) / 256;
    const dim3 gc((n + 63) / 64, kCrfChunks);
    // the smoothness kernel only depends on the superpixel grid: rebuilt only when its features change
    const bool same_smooth = s->smooth_cache.size() == (size_t)n * 2 && memcmp(s->smooth_cache.data(), feat_smooth_host, sizeof(float) * n * 2) == 0;
    if (!same_smooth) {
        HIPCHK(ctx, hipMemcpyAsync(s->feat1, feat_smooth_host, sizeof(float) * n * 2, hipMemcpyHostToDevice, st));
        crf_raw_kernel<2><<<g2, 256, 0, st>>>(s->feat1, n, s->raw);
        crf_norm_partial_kernel<<<gc, 64, 0, st>>>(s->raw, n, s->partial);
        crf_norm_kernel<<<g1, 256, 0, st>>>(s->partial, n, s->norm);
        crf_scale_kernel<<<g2, 256, 0, st>>>(s->raw, s->norm, n, s->K1t);
        s->smooth_cache.assign(feat_smooth_host, feat_smooth_host + (size_t)n * 2);
    }
    crf_raw_kernel<6><<<g2, 256, 0, st>>>(s->feat2, n, s->raw);
    crf_norm_partial_kernel<<<gc, 64, 0, st>>>(s->raw, n, s->partial);
    crf_norm_kernel<<<g1, 256, 0, st>>>(s->partial, n, s->norm);
    crf_scale_kernel<<<g2, 256, 0, st>>>(s->raw, s->norm, n, s->K2t);
    crf_init_kernel<<<g1, 256, 0, st>>>(s->unary, L, n, s->Q0);
    float *q = s->Q0, *qn = s->Q1;
    for (int it = 0; it < iterations; it++) {
        launch_crf_message(st, gc, L, n, s->K1t, s->K2t, q, s->partial);
        crf_update_kernel<<<(n + 15) / 16, 256, 0, st>>>(s->unary, L, n, s->partial, w_smooth, w_app, qn);
        float* t = q; q = qn; qn = t;
    }
    LAUNCHCHK(ctx);
    HIPCHK(ctx, hipMemcpyAsync(Q_host, q, sizeof(float) * n * L, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return CF_OK;
}

// Slic::upsample<unsigned char> (Slic.h:127-139): full-resolution label mask on the device
int cf_seg_upsample(cf_segmenter* s, const uint8_t* low_map_host, uint8_t* full_dev)
{
    if (!s || !low_map_host || !full_dev) return CF_EINVAL;
    cf_ctx* ctx = s->ctx; hipStream_t st = ctx->stream;
    const int N = ctx->cfg.width * ctx->cfg.height;
    HIPCHK(ctx, hipMemcpyAsync(s->low_map, low_map_host, (size_t)s->K, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));  // low_map_host may be a caller stack/heap buffer
    seg_upsample_kernel<<<(N + 255) / 256, 256, 0, st>>>(s->labels, s->low_map, N, full_dev);
    LAUNCHCHK(ctx);
    return CF_OK;
}

int cf_seg_labels(cf_segmenter* s, void** dptr, uint64_t* bytes)
{
    if (!s || !dptr) return CF_EINVAL;
    *dptr = s->labels;
    if (bytes) *bytes = (uint64_t)s->ctx->cfg.width * s->ctx->cfg.height * 4;
    return CF_OK;
}

}  // extern "C"
