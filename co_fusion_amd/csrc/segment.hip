// segment.hip -- GPU side of the motion segmentation (Core/Segmentation/*): SLIC superpixels, the
// per-superpixel accumulations that replace the reference's full-resolution texture downloads
// (Segmentation.cpp:184-188: 6.1 MB of glGetTexImage per model per frame), the dense-CRF mean field over
// the 40x30 superpixel grid, and the label up-sampling.  Plus its C-ABI (cf_seg_*).
//
// gSLICr and densecrf are third-party and not vendored by the reference (Scripts/install.sh:84-85); they are
// replaced by their published algorithms exactly as stated in oracle/orc_segment.c (same arithmetic, same
// summation order => bit-identical to the oracle):
//   * SLIC (gSLICr's published engine: metric, normalisers, schedule -- see the oracle's header): one 16x16-pixel workgroup per grid cell; a pixel can only join one of the 3x3 neighbouring
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
constexpr int kMaxL = 256;  // label capacity of the kernels' static tables (labels incl. the "new model" label): model ids are 8 bits and 255
                           // marks a rejected superpixel, the reference's own limit (CoFusion.cpp:631-634, Segmentation.cpp).  A segmenter's
                           // buffers are sized for cf_segmenter::Lcap = the context's max_models (cf_config), 16 by default.
constexpr int kAccTile = 16;  // models per pass of the accumulation kernel (their pointers travel in the kernel arguments up to this many)

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
    // gSLICr's normalisers (seg_engine_GPU constructor, RGB case) and coherence weight (Slic.cpp:37), as in oracle/orc_segment.c
    float max_color_dist = 5.0f / (1.7321f * 255), max_xy_dist = 1.0f / (1.4142f * kSpix);
    max_color_dist *= max_color_dist; max_xy_dist *= max_xy_dist;
    const float weight = 0.6f;
    float best = 999999.9999f; int bi = 4;
#pragma unroll
    for (int n = 0; n < 9; n++) {  // dy-major, dx-minor: same scan order as the oracle
        if (s_lab[n] < 0) continue;
        const float dr = (float)p.x - s_c[n][2], dg = (float)p.y - s_c[n][3], db = (float)p.z - s_c[n][4];
        const float ex = (float)x - s_c[n][0], ey = (float)y - s_c[n][1];
        const float dcolor = dr * dr + dg * dg + db * db, dxy = ex * ex + ey * ey;
        const float d = sqrtf(dcolor * max_color_dist + weight * dxy * max_xy_dist);  // compute_slic_distance
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
    // finalize_reduction_result_shared: a cluster without pixels stays at its reset value (centre (0,0), colour 0)
    for (int q = 0; q < 5; q++) centres[k * 5 + q] = s[5] ? (float)(long long)s[q] / (float)(long long)s[5] : 0.f;
    for (int q = 0; q < 6; q++) s[q] = 0;
}

// -------------------------------------------------------------- per-superpixel sums ----
__device__ __forceinline__ long long q32(float v)
{
    if (!is_finite(v)) return 0;
    const float c = fminf(fmaxf(v, -1048576.0f), 1048576.0f);
    return __double2ll_rn((double)c * 4294967296.0);
}

// The kernels of the segmentation chain take the arguments of up to kSegBatch segmenters (the sequences of a lock-step group, all of
// one image size) in the kernel-argument segment and pick theirs with the grid's last dimension: one chain of launches for the group
// instead of one per sequence.  A single segmenter is a batch of one.
constexpr int kSegBatch = 8;
template <class A, int N = kSegBatch> struct SegBatch { A m[N]; };

struct AccArgs {
    const int* labels; const float* depth;
    const float* icp[kAccTile]; const float4* vconf[kAccTile];   // the first kAccTile models' images (kernel arguments: no pointer chasing)
    const float* const* icp_dev; const float4* const* vconf_dev;  // all n_models of them in device memory when there are more
    int n_models, cols, rows, gx, gy;
    int* resample;                   // nullable: [K] labels at the resample coordinates, written by the launch's extra grid row
    unsigned* spix_count;            // [K]
    unsigned* depth_count;           // [K]
    unsigned long long* depth_sum;   // [K]
    unsigned long long* icp_sum;     // [n][K]
    unsigned long long* conf_sum;    // [n][K]
};

__global__ void __launch_bounds__(256) seg_accumulate_kernel(const SegBatch<AccArgs> B)
{
    const AccArgs& a = B.m[blockIdx.z];
    __shared__ int s_lab[9];
    __shared__ unsigned s_cnt[9], s_dcnt[9];
    __shared__ unsigned long long s_dsum[9];
    __shared__ unsigned long long s_icp[kAccTile][9], s_conf[kAccTile][9];
    const int cx0 = blockIdx.x, cy0 = blockIdx.y, t = threadIdx.x;
    if (cy0 == a.gy) {  // the extra grid row: labels at the "empty superpixel" resample coordinates (Slic.h:192-206; index / spixelY is the reference's)
        const int k = cx0 * 256 + t;
        if (k >= a.gx * a.gy) return;
        int x = (int)((k % a.gx) * kSpix + kSpix * 0.5), y = (int)((k / a.gy) * kSpix + kSpix * 0.5);
        if (y >= a.rows) y = a.rows - 1;
        if (x >= a.cols) x = a.cols - 1;
        a.resample[k] = a.labels[y * a.cols + x];
        return;
    }
    const int K = a.gx * a.gy;
    if (t < 9) {
        const int dx = t % 3 - 1, dy = t / 3 - 1, cx = cx0 + dx, cy = cy0 + dy;
        s_lab[t] = (cx < 0 || cy < 0 || cx >= a.gx || cy >= a.gy) ? -1 : cy * a.gx + cx;
        s_cnt[t] = 0; s_dcnt[t] = 0; s_dsum[t] = 0;
    }
    for (int k = t; k < kAccTile * 9; k += 256) { s_icp[k / 9][k % 9] = 0; s_conf[k / 9][k % 9] = 0; }
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
    // the models in tiles of kAccTile (one pass for up to 16 models: what a frame normally has)
    for (int m0 = 0; m0 < a.n_models; m0 += kAccTile) {
        const int nm = min(kAccTile, a.n_models - m0);
        if (m0 > 0) {
            __syncthreads();
            for (int k = t; k < kAccTile * 9; k += 256) { s_icp[k / 9][k % 9] = 0; s_conf[k / 9][k % 9] = 0; }
            __syncthreads();
        }
        for (int m = 0; m < nm; m++) {
            const float* icp = a.n_models <= kAccTile ? a.icp[m] : a.icp_dev[m0 + m];
            const float4* vc = a.n_models <= kAccTile ? a.vconf[m] : a.vconf_dev[m0 + m];
            atomicAdd(&s_icp[m][slot], (unsigned long long)q32(icp[q]));
            atomicAdd(&s_conf[m][slot], (unsigned long long)q32(vc[q].w));
        }
        __syncthreads();
        if (t < 9 && s_lab[t] >= 0) {
            const int L = s_lab[t];
            for (int m = 0; m < nm; m++) {
                if (s_icp[m][t]) atomicAdd(&a.icp_sum[(size_t)(m0 + m) * K + L], s_icp[m][t]);
                if (s_conf[m][t]) atomicAdd(&a.conf_sum[(size_t)(m0 + m) * K + L], s_conf[m][t]);
            }
        }
    }
    __syncthreads();
    if (t < 9 && s_lab[t] >= 0) {
        const int L = s_lab[t];
        if (s_cnt[t]) atomicAdd(&a.spix_count[L], s_cnt[t]);
        if (s_dcnt[t]) { atomicAdd(&a.depth_count[L], s_dcnt[t]); atomicAdd(&a.depth_sum[L], s_dsum[t]); }
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

struct UpsampleArgs { const int* labels; const unsigned char* low_map; unsigned char* full; };
__global__ void __launch_bounds__(256) seg_upsample_kernel(const SegBatch<UpsampleArgs> B, int N)
{
    const UpsampleArgs& a = B.m[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) a.full[i] = a.low_map[a.labels[i]];
}
static void launch_upsample(hipStream_t st, const int* labels, const unsigned char* low_map, int N, unsigned char* full)
{
    SegBatch<UpsampleArgs> B{};
    B.m[0] = UpsampleArgs{labels, low_map, full};
    seg_upsample_kernel<<<dim3((N + 255) / 256, 1), 256, 0, st>>>(B, N);
}

// ------------------------------------------------------------------------------- dense CRF ----
// The symmetric-normalised Gaussian kernel K[i][j] = norm_i * exp(-|f_i - f_j|^2 / 2) * norm_j of a feature set, stored transposed.
// All sums over the n nodes (normalisation and message passing) run in kCrfChunks contiguous chunks of ceil(n / kCrfChunks) indices:
// sequential inside a chunk, chunk totals added in chunk order (the oracle states the same order).  A thread that walks all n nodes
// alone made the 1200-node mean field 61 % of a multi-object frame (10 x 192 us + 2 x 294 us, measured); with the chunked order one
// wave covers 64 nodes x one chunk.
// Until round 4 the build was four launches (raw matrix, chunk partials, norm, scale: 5.9 + 4.7 + 4.6 + 9.6 us and three boundaries for
// 1200 nodes, re-reading the 5.8 MB raw matrix twice, once transposed).  Now two: the exponentials are cheap, so both passes recompute
// them from the features (the SAME expression: raw[i][j] and raw[j][i] agree bit for bit, (a - b)^2 == (b - a)^2) and no raw matrix exists.
constexpr int kCrfChunks = 16;
template <int D>
__device__ __forceinline__ float crf_raw(const float* fi, const float* fj)
{
    float d2 = 0;
#pragma unroll
    for (int d = 0; d < D; d++) { const float t = fi[d] - fj[d]; d2 += t * t; }
    return det_expf(-0.5f * d2);
}
// One segmenter's buffers of the mean field (a batch entry of every CRF launch)
struct CrfSeq {
    const float* feat; float* norm; float* Kt;      // kernel-matrix build: features in, normalisation scratch, matrix out
    const float* K1t; const float* K2t;             // mean field: smoothness and appearance kernels
    const float* unary; float* Q0; float* Q1; float* partial;
    int L;
};
struct CrfBatch { CrfSeq m[kSegBatch]; };
// norm_i = 1/sqrt(sum_c (sum over chunk c of raw[i][j]) + 1e-20).  A workgroup owns R nodes: its 1024 threads fill the R rows of the raw
// matrix in LDS (the exponentials, fully parallel), then one thread per (row, chunk) adds its chunk in node order and one per row the
// chunk totals in chunk order.  (One lane per (node, chunk) evaluating its 75 exponentials one after the other took 21.9 us.)
template <int D>
__global__ void __launch_bounds__(1024) crf_rownorm_kernel(const CrfBatch B, int n, int R)
{
    const float* __restrict__ feat = B.m[blockIdx.y].feat; float* __restrict__ norm = B.m[blockIdx.y].norm;
    extern __shared__ float s_raw[];  // [R][n]
    __shared__ float s_fi[8 * D];
    __shared__ float s_part[8][kCrfChunks];
    const int tid = threadIdx.x, i0 = blockIdx.x * R;
    if (tid < R * D && i0 * D + tid < n * D) s_fi[tid] = feat[i0 * D + tid];
    __syncthreads();
    for (int e = tid; e < R * n; e += 1024) {
        const int il = e / n, j = e - il * n;
        float fj[D];
#pragma unroll
        for (int d = 0; d < D; d++) fj[d] = feat[j * D + d];
        s_raw[e] = (i0 + il < n) ? crf_raw<D>(s_fi + il * D, fj) : 0.f;
    }
    __syncthreads();
    if (tid < R * kCrfChunks) {
        const int il = tid / kCrfChunks, c = tid - il * kCrfChunks;
        const int len = (n + kCrfChunks - 1) / kCrfChunks, j0 = c * len, j1 = min(n, j0 + len);
        const float* row = s_raw + il * n;
        float sum = 0;
        for (int j = j0; j < j1; j++) sum += row[j];
        s_part[il][c] = sum;
    }
    __syncthreads();
    if (tid < R && i0 + tid < n) {
        float t = 0;
        for (int k = 0; k < kCrfChunks; k++) t += s_part[tid][k];
        norm[i0 + tid] = 1.0f / sqrtf(t + 1e-20f);
    }
}
// expAndNormalize of -unary
__device__ __forceinline__ void crf_init_node(const float* __restrict__ unary, int L, int i, float* __restrict__ Q)
{
    float mx = -unary[i * L];
    for (int l = 1; l < L; l++) if (-unary[i * L + l] > mx) mx = -unary[i * L + l];
    float s = 0;
    for (int l = 0; l < L; l++) s += det_expf(-unary[i * L + l] - mx);
    for (int l = 0; l < L; l++) Q[i * L + l] = det_expf(-unary[i * L + l] - mx) / s;   // (the same expression: the same bits as the summand)
}
// Kt[j][i] = (norm_i * raw[i][j]) * norm_j.  Workgroups beyond the matrix's (g2 of them) run expAndNormalize of -unary for the mean
// field's first marginals (with_init): independent work that was a 4.7 us launch of its own.
template <int D>
__global__ void __launch_bounds__(256) crf_kernel_matrix_kernel(const CrfBatch B, int n, int g2, int with_init)
{
    const CrfSeq& m = B.m[blockIdx.y];
    if ((int)blockIdx.x >= g2) {
        const int i = ((int)blockIdx.x - g2) * 256 + threadIdx.x;
        if (with_init && i < n) crf_init_node(m.unary, m.L, i, m.Q0);
        return;
    }
    const float* __restrict__ feat = m.feat; const float* __restrict__ norm = m.norm; float* __restrict__ Kt = m.Kt;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int j = idx / n, i = idx - j * n;
    float fi[D], fj[D];
#pragma unroll
    for (int d = 0; d < D; d++) { fi[d] = feat[i * D + d]; fj[d] = feat[j * D + d]; }
    Kt[idx] = norm[i] * crf_raw<D>(fi, fj) * norm[j];
}
// the kernel matrices of S feature sets (+ the first marginals beside them)
template <int D>
static void launch_crf_kernel_matrix(hipStream_t st, const CrfBatch& B, int S, int n, bool with_init)
{
    const int g2 = (n * n + 255) / 256, g1 = (n + 255) / 256;
    int R = (int)(48u * 1024u / (sizeof(float) * (size_t)n));  // rows of the raw matrix per workgroup: what 48 KB of LDS hold, at most 8
    R = R > 8 ? 8 : (R < 1 ? 1 : R);                          // (n <= 12288 nodes; 640x480 has 1200, 1280x960 4800)
    crf_rownorm_kernel<D><<<dim3((n + R - 1) / R, S), 1024, sizeof(float) * (size_t)R * n, st>>>(B, n, R);
    crf_kernel_matrix_kernel<D><<<dim3(g2 + (with_init ? g1 : 0), S), 256, 0, st>>>(B, n, g2, with_init ? 1 : 0);
}
// one mean-field step, part 1: chunk partials of K1*Q and K2*Q; partial[((c*n + i)*2 + which)*L + l].
// One lane per (node i, chunk c, label l) -- grid (n/64, chunks x labels, batch entries): the sums inside a chunk are sequential
// by definition, so the only parallelism is across nodes, chunks and labels, and with one lane per (node, chunk) only ~300 waves existed
// for 1024 SIMDs.  The chunk is walked 25 nodes at a time so that the kernel-matrix loads of a group are in flight together (the sums
// stay in node order).  flip: the marginals are read from Q1 (odd steps) / Q0 (even steps).
// Measured and dropped in round 5 (both bit-identical, DESIGN-NOTES R5): message + update as ONE launch -- a 1024-thread workgroup owning
// 8 nodes, wave = chunk, lane = (node, label), chunk sums through LDS: 13.8 us per step against 7.6 + 4.9 -- and four nodes per lane
// with 16-byte kernel-matrix loads: 8.2 against 7.6 us.  The step is three dependent rounds of loads behind a launch, not load issue.
// A third fusion (the workgroup's columns of both kernel matrices and all marginals staged in 106 KB of LDS with 16-byte loads, chunk sums
// out of LDS, update in place): 13.9 us per step, 768 against 781 frames/s (profiles/r5an_*).  Two launches it stays.
// Late in round 6, on top of the lean addresses below (bit-identical all): the whole 75-node chunk in one flight of loads (159 VGPRs, three
// waves per SIMD: message 7.2 + update 4.1 us against 6.65 + 4.64, the same sum) and two / three labels per lane, so that a (node block,
// chunk)'s kernel-matrix tiles leave the L2 once per two / three labels instead of once per label (839 / 811 against 843 frames/s on one
// box): the step waits neither for round trips nor for L2 bandwidth any more.
__global__ void __launch_bounds__(64) crf_message_kernel(const CrfBatch B, int n, int flip)
{
    const CrfSeq& m = B.m[blockIdx.z];
    const int L = m.L, l = blockIdx.y / kCrfChunks, c = blockIdx.y % kCrfChunks;  // grid.y = chunks x the batch's largest label count
    if (l >= L) return;
    const float* __restrict__ K1t = m.K1t; const float* __restrict__ K2t = m.K2t;
    const float* __restrict__ Q = flip ? m.Q1 : m.Q0; float* __restrict__ partial = m.partial;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int len = (n + kCrfChunks - 1) / kCrfChunks, j0 = c * len, j1 = min(n, j0 + len);
    float a = 0, b = 0;
    int j = j0;
    // the chunk is a chain of dependent additions but its loads are independent: 25 nodes' worth in flight at a time (a 75-node chunk
    // is three memory round trips instead of fifteen).
    // Addresses (late in round 6): `K1t[(j + u) * n + i]` made every load form a 64-bit address on the vector unit -- 65 v_lshl_add_u64 and
    // 93 v_add_u32 for 50 loads, ~1 300 instructions per wave and two waves per SIMD: the kernel was waiting for instruction issue as much
    // as for memory.  A uniform row pointer plus the lane's 32-bit node offset leaves one 64-bit addition per load (the uniform part is
    // formed on the scalar unit): 743 -> 488 instructions, 7.6 -> 6.5 us.
    const unsigned ib = (unsigned)i * 4u;
    auto at = [ib](const float* row) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + ib); };
    const float* r1 = K1t + (size_t)j0 * n;   // row j of the transposed kernels (uniform)
    const float* r2 = K2t + (size_t)j0 * n;
    const float* qp = Q + (size_t)j0 * L + l;  // (uniform: scalar loads)
    for (; j + 25 <= j1; j += 25, r1 += (size_t)25 * n, r2 += (size_t)25 * n, qp += (size_t)25 * L) {
        float k1[25], k2[25], q[25];
#pragma unroll
        for (int u = 0; u < 25; u++) { k1[u] = at(r1 + (size_t)u * n); k2[u] = at(r2 + (size_t)u * n); q[u] = qp[(size_t)u * L]; }
#pragma unroll
        for (int u = 0; u < 25; u++) { a += k1[u] * q[u]; b += k2[u] * q[u]; }
    }
    for (; j + 5 <= j1; j += 5, r1 += (size_t)5 * n, r2 += (size_t)5 * n, qp += (size_t)5 * L) {
        float k1[5], k2[5], q[5];
#pragma unroll
        for (int u = 0; u < 5; u++) { k1[u] = at(r1 + (size_t)u * n); k2[u] = at(r2 + (size_t)u * n); q[u] = qp[(size_t)u * L]; }
#pragma unroll
        for (int u = 0; u < 5; u++) { a += k1[u] * q[u]; b += k2[u] * q[u]; }
    }
    for (; j < j1; j++, r1 += n, r2 += n, qp += L) {
        const float q = qp[0];
        a += at(r1) * q;
        b += at(r2) * q;
    }
    float* out = partial + ((size_t)(c * n + i) * 2) * L;
    out[l] = a; out[L + l] = b;
}
// part 2: chunk totals in chunk order, unary, softmax over the labels.  Thread (node g, label l): 256 / LS nodes x LS label slots per
// workgroup (LS = 16 for up to 16 labels, a power of two up to 256 beyond); the chunk partials of a (node, label) are loaded
// independently and summed in chunk order, the softmax runs over the node's LDS row exactly like expAndNormalize.
template <int LS>
__global__ void __launch_bounds__(256) crf_update_kernel(const CrfBatch B, int n, float w_smooth, float w_app, int flip)
{
    const CrfSeq& m = B.m[blockIdx.y];
    const int L = m.L;
    const float* __restrict__ unary = m.unary; const float* __restrict__ partial = m.partial; float* __restrict__ Qn = flip ? m.Q0 : m.Q1;
    constexpr int G = 256 / LS;
    __shared__ float s_t[G][LS];
    const int g = threadIdx.x / LS, l = threadIdx.x % LS;
    const int i = blockIdx.x * G + g;
    float tmp = 0;
    if (i < n && l < L) {
        float pa[kCrfChunks], pb[kCrfChunks];
#pragma unroll
        for (int c = 0; c < kCrfChunks; c++) {
            const float* in = partial + ((size_t)(c * n + i) * 2) * L;
            pa[c] = in[l]; pb[c] = in[L + l];
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
// `iterations` mean-field steps of S batch entries from Q0; returns 1 when the last marginals are in Q1
static int launch_mean_field(hipStream_t st, CrfBatch& B, int S, int n, int iterations, float w_smooth, float w_app)
{
    int Lmax = 0;
    for (int e = 0; e < S; e++) Lmax = B.m[e].L > Lmax ? B.m[e].L : Lmax;
    const dim3 gm((n + 63) / 64, kCrfChunks * Lmax, S);
    int flip = 0;
    for (int it = 0; it < iterations; it++) {
        crf_message_kernel<<<gm, 64, 0, st>>>(B, n, flip);
        if (Lmax <= 16) crf_update_kernel<16><<<dim3((n + 15) / 16, S), 256, 0, st>>>(B, n, w_smooth, w_app, flip);
        else if (Lmax <= 32) crf_update_kernel<32><<<dim3((n + 7) / 8, S), 256, 0, st>>>(B, n, w_smooth, w_app, flip);
        else if (Lmax <= 64) crf_update_kernel<64><<<dim3((n + 3) / 4, S), 256, 0, st>>>(B, n, w_smooth, w_app, flip);
        else if (Lmax <= 128) crf_update_kernel<128><<<dim3((n + 1) / 2, S), 256, 0, st>>>(B, n, w_smooth, w_app, flip);
        else crf_update_kernel<256><<<dim3(n, S), 256, 0, st>>>(B, n, w_smooth, w_app, flip);
        flip ^= 1;
    }
    return flip;
}


// --------------------------------------------------- device-side unaries and post-processing ----
// Everything Segmentation::performSegmentationCRF does around SLIC and the mean field (Segmentation.cpp:160-300, 475-646), on the
// device: the host no longer reads the sums back to build the unaries, uploads them, reads the marginals back for the component
// analysis and uploads the label map (four host waits per multi-object frame); only the decisions come back.  Sequential f32 sums of
// the reference (average confidence, depth statistics) stay sequential -- one lane per model walks the K superpixels in index order --
// so the results are those of the host code (and of the oracle) bit for bit.
constexpr float kSegMaxDepth = 100.f;  // Segmentation::MAX_DEPTH

struct SegUnaryArgs {
    int K, gx, gy, n_models, L, allow_new;
    float unaryWeightError, unaryKError, unaryThresholdNew, scaleFeaturesRGB, scaleFeaturesDepth, scaleFeaturesPos;
    unsigned* spix_count; unsigned* depth_count;
    unsigned long long* depth_sum; unsigned long long* icp_sum; unsigned long long* conf_sum;   // [K], [n][K], [n][K]; zeroed on exit
    const int* resample;
    const uchar4* rgba;              // the CRF colour features read the first K pixels of the full-resolution image (sic, :445-447)
    float* raw;                      // scratch [(1 + 2n)][K]
    float* low;                      // [(1 + 2n)][K]: lowDepth, lowICP[m], lowConf[m]
    float* unary; float* feat2;      // [K][L], [K][6]
    float* avg_conf;                 // [n]
    float* depth_range;              // [1]
};

// Sequential f32 sum init + t[0] + t[1] + ... + t[n-1] (this order) by ONE wave.  term(j) -> the j-th term, 0.0f for "skip" (x + 0.0f == x
// for every x these sums can reach -- a running sum that starts at +0.0f is never -0.0f --, so skipping an element and adding zero agree).
// Returns the sum in every lane.
// Rounds 3-6 moved ONE term per step to the adder (v_readlane, an LDS broadcast, a lane shift): 11-14 ns per addition whichever way, because
// every step pays a cross-lane operation on top of the addition.  Late in round 6 the terms are BLOCKED instead: lane l owns kSeqBlock
// consecutive terms of a super-block of 64 x kSeqBlock; in "phase" l every lane adds its own block to the running sum -- sixteen dependent
// plain v_add_f32 from registers -- and the value lane l arrives at (the only one that started from the true prefix and added the right
// terms) is read back as the running sum of phase l + 1.  One cross-lane operation per sixteen additions: 8 ns per addition (what a
// dependent v_add_f32 of a lone wave costs here), 13.4 -> 9.8 us for the average confidences of 1 200 superpixels.  A lane whose block
// holds only zeros has no phase at all.  The additions and their order are exactly those of the serial loop.
constexpr int kSeqBlock = 16;
// the phases of one super-block: lane l holds its kSeqBlock consecutive terms in t[], `any` = one of them is not zero
__device__ __forceinline__ float seq_block_phases(float sum, const float (&t)[kSeqBlock], bool any)
{
    unsigned long long nz = __ballot(any);
    while (nz) {   // (uniform)
        const int ph = __builtin_ctzll(nz);
        nz &= nz - 1;
        float x = sum;
#pragma unroll
        for (int c = 0; c < kSeqBlock; c++) x = x + t[c];
        sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), ph));
    }
    return sum;
}
template <class F>
__device__ __forceinline__ float wave_sequential_sum(float init, int n, int lane, F term)
{
    float sum = init;
    for (int base = 0; base < n; base += 64 * kSeqBlock) {
        float t[kSeqBlock];
        bool any = false;
#pragma unroll
        for (int c = 0; c < kSeqBlock; c++) {
            const int j = base + lane * kSeqBlock + c;
            t[c] = j < n ? term(j) : 0.f;
            any = any || (t[c] != 0.f);
        }
        sum = seq_block_phases(sum, t, any);
    }
    return sum;
}
// What a dependent addition really costs (tools/microbench/dep_chain.hip, late in round 6): 1.70 ns -- four cycles at 2.35 GHz, the same
// with the rest of the chip busy or idle, cold or warm; v_add_f64 / v_fma_f64 1.97 ns; `s_nop 1` + v_add_f32_dpp wave_shr:1 5.1 ns; two
// interleaved chains on one wave 3.4 ns per pair (a lone wave issues one VALU instruction per four cycles whatever it depends on).  The
// "8 ns per addition" above was the whole pass divided by its terms: at K = 1 200 most of it were the two flights of sixteen strided
// 4-byte loads per lane in front of each super-block's phases and a second walk over the array to zero the non-finite entries.  This
// flavour -- n a multiple of kSeqBlock, p 16-byte aligned -- fetches a lane's block as four 16-byte loads, has the NEXT super-block's
// loads in flight during the phases of the current one, and takes the non-finite entries out on the way (zero in the sum, zero stored
// back: what the caller's second walk did).  Same additions, same order.
__device__ __forceinline__ float wave_sequential_sum_finite16(float* __restrict__ p, int n, int lane)
{
    const int nch = n / kSeqBlock;
    float sum = 0.f;
    float4 cur[4], nxt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) cur[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < nch) {
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = reinterpret_cast<const float4*>(p + (size_t)lane * kSeqBlock)[q];
    }
    for (int j0 = 0; j0 < nch; j0 += 64) {
        const int jn = j0 + 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jn < nch) {
#pragma unroll
            for (int q = 0; q < 4; q++) nxt[q] = reinterpret_cast<const float4*>(p + (size_t)jn * kSeqBlock)[q];
        }
        float t[kSeqBlock];
        bool any = false, bad = false;
#pragma unroll
        for (int q = 0; q < 4; q++) { t[q * 4] = cur[q].x; t[q * 4 + 1] = cur[q].y; t[q * 4 + 2] = cur[q].z; t[q * 4 + 3] = cur[q].w; }
#pragma unroll
        for (int c = 0; c < kSeqBlock; c++) {
            if (!is_finite(t[c])) { t[c] = 0.f; bad = true; }
            any = any || (t[c] != 0.f);
        }
        if (bad) {   // (rare)
            float4* o = reinterpret_cast<float4*>(p + (size_t)(j0 + lane) * kSeqBlock);
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] = make_float4(t[q * 4], t[q * 4 + 1], t[q * 4 + 2], t[q * 4 + 3]);
        }
        sum = seq_block_phases(sum, t, any);
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
    }
    return sum;
}

// seg_post_kernel's sums: the terms are predicates over arrays in LDS.  Rounds 5-6 moved them to the adder one by one with a lane shift
// (`s_nop 1` + `v_add_f32_dpp wave_shr:1`: 5.1 ns per term by the micro-benchmark, 63 per 64 terms), because the blocked chain read its
// sixteen terms per lane with a lane stride of sixteen words -- bank conflicts in three arrays, 25.1 against 21.8 us.  Now the arrays are
// LAID OUT for the blocked chain: the depths padded by four words per sixteen (entry k at k + 4 * (k >> 4): lane l's block starts at
// word 20 l, four conflict-free 16-byte reads), the model entry of every superpixel as 16 bits (lane l's sixteen are 32 consecutive bytes).
// `term(mine, depth, out[NCH])` forms the NCH chains' terms of one superpixel; a lane whose block holds only zeros has no phase.
constexpr int kSegDepthPad(int k) { return k + 4 * (k >> 4); }
template <int NCH, class F>
__device__ __forceinline__ void lds_blocked_sums(float (&sum)[NCH], int nch, int lane, const float* s_depth, const unsigned short* s_entry, unsigned entry, F term)
{
    for (int j0 = 0; j0 < nch; j0 += 64) {
        const int j = j0 + lane;
        float t[NCH][kSeqBlock];
        bool any = false;
#pragma unroll
        for (int h = 0; h < NCH; h++)
#pragma unroll
            for (int c = 0; c < kSeqBlock; c++) t[h][c] = 0.f;
        if (j < nch) {
            const float4* dq = reinterpret_cast<const float4*>(s_depth + 20 * j);
            const uint4* eq = reinterpret_cast<const uint4*>(s_entry + 16 * j);
            const float4 d0 = dq[0], d1 = dq[1], d2 = dq[2], d3 = dq[3];
            const uint4 e0 = eq[0], e1 = eq[1];
            const float d[kSeqBlock] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x, d3.y, d3.z, d3.w};
            const unsigned w[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
            for (int c = 0; c < kSeqBlock; c++) {
                const bool mine = ((w[c >> 1] >> ((c & 1) * 16)) & 0xffffu) == entry;
                float v[NCH];
                term(mine, d[c], v);
#pragma unroll
                for (int h = 0; h < NCH; h++) { t[h][c] = v[h]; any = any || (v[h] != 0.f); }
            }
        }
        unsigned long long nz = __ballot(any);
        while (nz) {   // (uniform)
            const int ph = __builtin_ctzll(nz);
            nz &= nz - 1;
            float x[NCH];
#pragma unroll
            for (int h = 0; h < NCH; h++) x[h] = sum[h];
#pragma unroll
            for (int c = 0; c < kSeqBlock; c++)
#pragma unroll
                for (int h = 0; h < NCH; h++) x[h] = x[h] + t[h][c];
#pragma unroll
            for (int h = 0; h < NCH; h++) sum[h] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[h]), ph));
        }
    }
}

#ifdef CF_ABLATE
// diagnostics build (CF_SEG_TRACE=<inference>): phase stamps of segmenter 0's two single-workgroup kernels, 100 MHz constant clock
__device__ unsigned long long g_seg_trace[2][16];
#define GSTAMP(which, k) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) g_seg_trace[which][k] = wall_clock64(); } while (0)
#else
#define GSTAMP(which, k) do {} while (0)
#endif
// Inclusive scan of one int per thread over the workgroup (<= 1024 threads): a wave-level scan (six shuffle steps), the waves' totals
// through LDS, every thread adds the totals of the waves in front of it -- two barriers instead of the 2 x log2(T) of the
// Hillis-Steele loop these kernels used until round 6 (twenty with sixteen waves, a few hundred ns each).  Returns the inclusive prefix;
// *total = the sum over the workgroup.  s_wave: >= 16 ints of LDS, free before and after.
__device__ __forceinline__ int block_scan_inclusive(int v, int* s_wave, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x + 63) >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    __syncthreads();   // (s_wave may still be read from a previous scan)
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < nw; w++) { const int t = s_wave[w]; all += t; if (w < wave) before += t; }
    *total = all;
    return incl + before;
}

// Slic::downsample<float> normalisation incl. the empty-superpixel fallback (Slic.h:63-76, 192-206) evaluated in place and in index
// order by the reference: an empty superpixel k reads entry `read`, which has ALREADY been divided when read < k and is still the
// raw sum when read > k.  Non-empty entries do not depend on anything else (phase 1, parallel); the rare empty ones are replayed in
// index order by one lane per array (phase 2) from a list built with an ordered scan.
__global__ void __launch_bounds__(1024) seg_unary_kernel(const SegBatch<SegUnaryArgs> B)
{
    const SegUnaryArgs a = B.m[blockIdx.x];  // (by value: the fields are loaded into scalar registers once, ahead of the phases)
    const int K = a.K, n = a.n_models, A = 1 + 2 * n, L = a.L;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float s_min[16], s_max[16];
    __shared__ float s_range;
    __shared__ int s_scan[16];
    __shared__ int s_nempty[2];
    int* empties = reinterpret_cast<int*>(a.raw + (size_t)A * K);  // scratch behind the raw sums: [2][K] (depth-empty, pixel-empty)
    GSTAMP(0, 0);
    // A: raw sums as f32, phase 1 of the normalisation
    // (eight entries per lane in flight -- sixteen, one round at five models, measured slower late in round 6: 9.6 against 7.2 us --: this workgroup is alone on the GPU, a loop of dependent round trips to HBM -- 13 of them at five
    // models -- was a third of the kernel)
    for (int base = 0; base < A * K; base += 8 * T) {
        unsigned long long sv[8]; int cv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * T + tid;
            sv[u] = 0; cv[u] = 0;
            if (idx < A * K) {
                const int arr = idx / K, k = idx - arr * K;
                const unsigned long long* sums = arr == 0 ? a.depth_sum : (arr <= n ? a.icp_sum + (size_t)(arr - 1) * K : a.conf_sum + (size_t)(arr - 1 - n) * K);
                sv[u] = sums[k];
                cv[u] = (int)(arr == 0 ? a.depth_count[k] : a.spix_count[k]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * T + tid;
            if (idx < A * K) {
                const float raw = (float)((double)(long long)sv[u] * 2.3283064365386963e-10 /* 2^-32 */);
                a.raw[idx] = raw;
                a.low[idx] = cv[u] != 0 ? raw / (float)cv[u] : raw;
            }
        }
    }
    GSTAMP(0, 1);   // raw sums -> f32, phase 1
    // ordered lists of the empty superpixels (which == 0: no depth sample, which == 1: no pixel at all): both counts ride through ONE
    // scan, sixteen bits each (K <= 4800)
    const int per = (K + T - 1) / T;
    {
        int c0 = 0, c1 = 0;
        for (int k = tid * per; k < min(K, (tid + 1) * per); k++) { c0 += a.depth_count[k] == 0; c1 += a.spix_count[k] == 0; }
        int total = 0;
        const int incl = block_scan_inclusive(c0 | (c1 << 16), s_scan, &total);
        int pos0 = (incl & 0xffff) - c0, pos1 = (incl >> 16) - c1;
        for (int k = tid * per; k < min(K, (tid + 1) * per); k++) {
            if (a.depth_count[k] == 0) empties[pos0++] = k;
            if (a.spix_count[k] == 0) empties[K + pos1++] = k;
        }
        if (tid == 0) { s_nempty[0] = total & 0xffff; s_nempty[1] = total >> 16; }
        __syncthreads();
    }
    GSTAMP(0, 2);   // ordered lists
    // phase 2: empty superpixels in index order, one lane per array
    if (tid < A) {
        float* low = a.low + (size_t)tid * K;
        const float* raw = a.raw + (size_t)tid * K;
        const int which = tid == 0 ? 0 : 1, ne = s_nempty[which];
        for (int e = 0; e < ne; e++) {
            const int k = empties[which * K + e];
            const int read = a.resample[k];
            const int cnt = (int)a.spix_count[read];
            const float base = read < k ? low[read] : raw[read];
            low[k] = base / (float)cnt;
        }
    }
    __syncthreads();
    GSTAMP(0, 3);   // empty superpixels replayed
    // depth range over the valid low-resolution depths (Segmentation.cpp:165-176) BESIDE the average confidence per model (a sequential
    // f32 sum in index order, :193-203, one WAVE per model; non-finite entries count as zero and are zeroed in place): with fewer models
    // than waves, waves [0, n) take the sums and waves [n, T / 64) the range -- neither reads what the other writes
    const int nw = T >> 6;
    const bool beside = n < nw;
    auto average_confidence = [&](int m) {
        float* conf = a.low + (size_t)(1 + n + m) * K;
        float avg;
        if ((K % kSeqBlock) == 0 && (reinterpret_cast<size_t>(conf) & 15) == 0) avg = wave_sequential_sum_finite16(conf, K, lane);
        else {
            avg = wave_sequential_sum(0.f, K, lane, [&](int j) { const float c = conf[j]; return is_finite(c) ? c : 0.f; });
            for (int j = lane; j < K; j += 64) if (!is_finite(conf[j])) conf[j] = 0;
        }
        if (lane == 0) a.avg_conf[m] = avg / (float)K;
    };
    {
        float mn = 3.402823466e+38f, mx = 0.f;
        if (beside && wave < n) average_confidence(wave);
        else {
            const int first = beside ? n * 64 : 0;
            for (int k = tid - first; k < K; k += T - first) {
                const float d = a.low[k];
                if (d > kSegMaxDepth || d < 0 || !is_finite(d)) continue;
                if (mx < d) mx = d;
                if (mn > d) mn = d;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float omn = __shfl_xor(mn, o, 64), omx = __shfl_xor(mx, o, 64);
                if (omn < mn) mn = omn;
                if (mx < omx) mx = omx;
            }
        }
        if (lane == 0) { s_min[wave] = mn; s_max[wave] = mx; }   // (the waves with the sums file the neutral elements)
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < nw; w++) { if (s_min[w] < mn) mn = s_min[w]; if (mx < s_max[w]) mx = s_max[w]; }
            s_range = mx - mn;
            a.depth_range[0] = s_range;
        }
    }
    GSTAMP(0, 4);   // depth range (+ the average confidences beside it)
    if (!beside)
        for (int m = wave; m < n; m += nw) average_confidence(m);
    __syncthreads();
    GSTAMP(0, 5);   // average confidences
    const float depthRange = s_range;
    // unaries (:237-298, 458-460) and the appearance features (:441-450), one lane per superpixel -- and per lane TWO superpixels (k and
    // k + T: K = 1200 against 1024 lanes was two rounds) with every input of both in one flight of loads: the confidences and errors of
    // eight models at a time instead of one dependent round trip per model (8.8 -> 4.5 us, late in round 6).  The stores go to elements only
    // this lane reads.
    {
        float* const icp = a.low + (size_t)K;           // [n][K]
        const float* const conf = a.low + (size_t)(1 + n) * K;
        const float fill0 = (float)((double)depthRange * 0.01), fillN = depthRange * a.unaryKError;
        const int nn = n > 0 ? n : 1;   // (model 0's rule and the first lowest error are formed whatever n is)
        for (int k0 = tid; k0 < K; k0 += 2 * T) {
            const int kk[2] = {k0, k0 + T};
            const bool ok[2] = {true, k0 + T < K};
            const int kc[2] = {k0, ok[1] ? k0 + T : k0};
            uchar4 px[2]; float lowd[2], lowest[2] = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 2; r++) { px[r] = a.rgba[kc[r]]; lowd[r] = a.low[kc[r]]; }
            for (int i0 = 0; i0 < nn; i0 += 8) {
                float cf[8][2], ic[8][2];
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const size_t at = (size_t)min(i0 + u, nn - 1) * K + kc[r];
                        cf[u][r] = conf[at]; ic[u][r] = icp[at];
                    }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u;
                    if (i < nn) {   // (uniform)
#pragma unroll
                        for (int r = 0; r < 2; r++)
                            if (ok[r]) {
                                float e = ic[u][r];
                                const bool weak = i == 0 ? (double)cf[u][r] < 0.3 : (double)cf[u][r] <= 0.4;
                                if (weak) { e = i == 0 ? fill0 : fillN; icp[(size_t)i * K + kk[r]] = e; }
                                if (i == 0) lowest[r] = e / depthRange;
                                if (i < n) {
                                    const float error = e / depthRange;
                                    if (error < lowest[r]) lowest[r] = error;
                                    float un = a.unaryWeightError * error;
                                    if (un <= 1e-5f) un = 1e-5f;
                                    a.unary[(size_t)kk[r] * L + i] = un;
                                }
                            }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (ok[r]) {
                    const int k = kk[r];
                    if (a.allow_new) {
                        float un = fmaxf(a.unaryThresholdNew - a.unaryWeightError * lowest[r], 0.01f);
                        if (un <= 1e-5f) un = 1e-5f;
                        a.unary[(size_t)k * L + n] = un;
                    }
                    const int i = k % a.gx, j = k / a.gx;
                    float* f = a.feat2 + (size_t)k * 6;
                    f[0] = (float)i * a.scaleFeaturesPos; f[1] = (float)j * a.scaleFeaturesPos;
                    f[2] = (float)px[r].x * a.scaleFeaturesRGB; f[3] = (float)px[r].y * a.scaleFeaturesRGB; f[4] = (float)px[r].z * a.scaleFeaturesRGB;
                    f[5] = fminf(lowd[r] * a.scaleFeaturesDepth, 100.0f);
                }
        }
    }
    __syncthreads();
    GSTAMP(0, 6);   // unaries + features
    // leave the accumulators clean for the next frame
    for (int k = tid; k < K; k += T) { a.spix_count[k] = 0; a.depth_count[k] = 0; a.depth_sum[k] = 0; }
    for (int idx = tid; idx < n * K; idx += T) { a.icp_sum[idx] = 0; a.conf_sum[idx] = 0; }
    GSTAMP(0, 7);
}

// smoothness features of the superpixel grid: addPairwiseGaussian(2, 2) (:437)
__global__ void seg_feat1_kernel(int gx, int K, float* __restrict__ feat1)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    feat1[k * 2 + 0] = (float)(k % gx) / 2.0f; feat1[k * 2 + 1] = (float)(k / gx) / 2.0f;
}

template <int CAP>
struct SegPostArgsT {
    int K, gx, gy, n_models, L, allow_new, width, height;
    unsigned next_id;
    float minRelSizeNew, maxRelSizeNew;
    unsigned ids[CAP];               // model ids in list order (+ the new label's id)
    const float* Q;                  // [K][L] marginals
    const float* low_depth;          // [K]
    const float* avg_conf;           // [n]
    const float* depth_range;
    int* parent; int* comp;          // scratch [K]
    int* cc;                         // scratch [6][K]: label, size, top, right, bottom, left per component
    unsigned char* low_map;          // [K] out
    cf_seg_result* result;           // device copy of the result
    cf_seg_result* result_host;      // pinned: the kernel publishes the decisions itself (no copy command behind it on the stream)
    unsigned* low_map_host;          // pinned, [ceil(K / 4)] words
};

// arg-max labels -> connected components (ConnectedLabels.hpp:50-172: 4-connectivity, components numbered by their first pixel in
// raster order) -> largest-component / size / border gates -> bounding boxes, depth statistics, super-pixel counts (:475-646).
// One workgroup: the label image has K = 1200 superpixels (4800 at 1280x960, the largest supported); labels, union-find parents and
// component numbers live in LDS, the sequential sums of the statistics run one wave per model (wave_sequential_sum).
constexpr int kSegMaxK = 4800;
constexpr int kPoseWords = 18;   // cf_seg_publish_poses: 16 pose words + ICP error + ICP inlier count, one 64-bit slot per f32 bit pattern
constexpr int kCcLds = 256;   // components whose statistics fit in LDS (a frame has tens)
using SegPostArgs = SegPostArgsT<kMaxL + 1>;   // one segmenter with up to 256 labels ...
using SegPostArgs16 = SegPostArgsT<17>;        // ... or kSegBatch of them with up to 16 models + a new label each
template <int CAP, int N>
__global__ void __launch_bounds__(1024) seg_post_kernel(const SegBatch<SegPostArgsT<CAP>, N> B)
{
    const SegPostArgsT<CAP>& a = B.m[blockIdx.x];
    const int K = a.K, gx = a.gx, L = a.L, tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int n_md = a.n_models + (a.allow_new ? 1 : 0);
    __shared__ int s_changed, s_min_label;
    __shared__ int s_scan[16];
    __shared__ int s_id2idx[256];
    __shared__ int s_box[kMaxL + 1][4];   // top, right, bottom, left per model entry (full-resolution pixels after mapToHigh)
    __shared__ unsigned s_spc[kMaxL + 1];
    __shared__ unsigned s_best[256];
    __shared__ int s_reject[kMaxL + 1];
    __shared__ __attribute__((aligned(4))) unsigned char map[kSegMaxK];
    __shared__ __attribute__((aligned(16))) int s_pc[2 * kSegMaxK];   // union-find parents | component numbers; later the statistics' arrays
    int* const parent = s_pc;
    int* const comp = s_pc + kSegMaxK;
    __shared__ int s_cc[6 * kCcLds];
    if (tid == 0) s_min_label = 256;
    GSTAMP(1, 0);
    // 1. label with the highest marginal (first maximum), as model id
    // (two superpixels per lane with eight marginals of each in one flight of loads instead of two rounds: 3.0 us of this phase either
    // way, late in round 6; the three sweeps of the component loop below are the other 8.7 us)
    for (int k = tid; k < K; k += T) {
        int m = 0; float best = a.Q[(size_t)k * L];
        for (int l = 1; l < L; l++) { const float q = a.Q[(size_t)k * L + l]; if (q > best) { best = q; m = l; } }
        map[k] = (unsigned char)a.ids[m];
        parent[k] = k;
    }
    if (tid < 256) s_id2idx[tid] = 0;
    __syncthreads();
    if (tid < a.n_models) s_id2idx[a.ids[tid] & 255] = tid;
    __syncthreads();
    if (tid == 0 && a.allow_new) s_id2idx[a.next_id & 255] = a.n_models;
    GSTAMP(1, 8);   // (arg-max alone)
    // 2. connected components: min-label propagation over the 4-neighbourhood + pointer jumping until nothing changes; the root of a
    //    component is its smallest index = its first pixel in raster order
#ifdef CF_ABLATE
    int cc_sweep = 0;
#endif
    for (;;) {
        __syncthreads();
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (int k = tid; k < K; k += T) {
            const int x = k % gx, y = k / gx;
            const unsigned char v = map[k];
            const int own = parent[k];
            int p = own;
            if (x > 0 && map[k - 1] == v) p = min(p, parent[k - 1]);
            if (x + 1 < gx && map[k + 1] == v) p = min(p, parent[k + 1]);
            if (y > 0 && map[k - gx] == v) p = min(p, parent[k - gx]);
            if (y + 1 < a.gy && map[k + gx] == v) p = min(p, parent[k + gx]);
            if (p < own) { atomicMin(&parent[own], p); atomicMin(&parent[k], p); s_changed = 1; }
        }
        __syncthreads();
#ifdef CF_ABLATE
        if (tid == 0 && blockIdx.x == 0 && cc_sweep < 3) g_seg_trace[1][10 + 2 * cc_sweep] = wall_clock64();   // (hooks of this sweep done)
#endif
        if (!s_changed) break;   // (nothing hooked: every entry is still the root the last sweep's walk left -- or itself, in the first sweep)
        // walks to the roots.  After the first sweep's hooks a superpixel's chain runs up its column and along a row -- up to 70 hops of one
        // dependent LDS read each, 4.5 of this loop's 8.7 us (per-sweep stamps, late in round 6).  Every step of a walk is now WRITTEN to the
        // walker's own entry: the walkers that pass through it later jump where it has got to, so the lanes double each other's strides
        // (pointer jumping without its barriers).  Racy and monotone: during this pass nothing hooks, an entry only ever moves to an ancestor,
        // and a walk ends at an entry that is its own parent -- the same roots.
        for (int k = tid; k < K; k += T) {
            int p = parent[k];
            for (;;) {
                const int q = parent[p];
                if (q == p) break;
                parent[k] = q;
                p = q;
            }
            parent[k] = p;
        }
#ifdef CF_ABLATE
        __syncthreads();
        if (tid == 0 && blockIdx.x == 0) { g_seg_trace[1][9]++; if (cc_sweep < 3) g_seg_trace[1][11 + 2 * cc_sweep] = wall_clock64(); }   // (sweeps of the component loop; walks done)
        cc_sweep++;
#endif
    }   // (the barrier at the top of the next sweep stands between this sweep's walks and its hooks)
    GSTAMP(1, 1);   // arg-max + connected components
    // 3. number the roots in index order (exclusive scan of the root flags); a root also files its label under its number
    const int per = (K + T - 1) / T;
    int cnt3 = 0;
    for (int k = tid * per; k < min(K, (tid + 1) * per); k++) cnt3 += parent[k] == k;
    int ncc = 0;
    const int scan3 = block_scan_inclusive(cnt3, s_scan, &ncc);
    // per-component label, size, top, right, bottom, left: in LDS unless the label image is unusually fragmented
    int* const ccb = ncc <= kCcLds ? s_cc : a.cc;
    const int ccs = ncc <= kCcLds ? kCcLds : K;
    int *c_label = ccb, *c_size = ccb + ccs, *c_top = ccb + 2 * ccs, *c_right = ccb + 3 * ccs, *c_bottom = ccb + 4 * ccs, *c_left = ccb + 5 * ccs;
    {
        int base = scan3 - cnt3;
        for (int k = tid * per; k < min(K, (tid + 1) * per); k++)
            if (parent[k] == k) { comp[k] = base; c_label[base] = map[k]; atomicMin(&s_min_label, (int)map[k]); base++; }
    }
    for (int i = tid; i < ncc; i += T) { c_size[i] = 0; c_top[i] = 2147483647; c_right[i] = 0; c_bottom[i] = 0; c_left[i] = 2147483647; }
    __syncthreads();
    for (int k = tid; k < K; k += T) if (parent[k] != k) comp[k] = comp[parent[k]];  // roots wrote their own entry; read-only for them
    __syncthreads();
    GSTAMP(1, 2);   // roots numbered
    // 4. component statistics.  A wave first combines the lanes that belong to the same component (usually one or two per wave), so
    //    that one lane per (wave, component) touches the shared counters: a thousand atomics on the background's five words otherwise
    //    queue up behind each other
    for (int kb = 0; kb < K; kb += T) {
        const int k = kb + tid;
        const bool in = k < K;
        const int c = in ? comp[k] : -1, x = in ? k % gx : 0, y = in ? k / gx : 0;
        unsigned long long todo = __ballot(in);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int c0 = __builtin_amdgcn_readlane(c, leader);
            const bool member = in && c == c0;
            const unsigned long long grp = __ballot(member);
            int ymin = member ? y : 2147483647, ymax = member ? y : 0, xmin = member ? x : 2147483647, xmax = member ? x : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ymin = min(ymin, __shfl_xor(ymin, o, 64)); ymax = max(ymax, __shfl_xor(ymax, o, 64));
                xmin = min(xmin, __shfl_xor(xmin, o, 64)); xmax = max(xmax, __shfl_xor(xmax, o, 64));
            }
            if (lane == leader) {
                atomicAdd(&c_size[c0], (int)__popcll(grp));
                atomicMin(&c_top[c0], ymin); atomicMax(&c_bottom[c0], ymax); atomicMin(&c_left[c0], xmin); atomicMax(&c_right[c0], xmax);
            }
            todo &= ~grp;
        }
    }
    __threadfence_block();
    __syncthreads();
    GSTAMP(1, 3);   // component statistics
    // 5. onlyKeepLargest (:496-517): every label but the smallest keeps its largest component, the earlier one on ties -- the
    //    sequential rule "replace the kept component only by a strictly larger one" picks exactly the maximum of (size, -index)
    if (tid < 256) s_best[tid] = 0;
    if (tid < kMaxL + 1) { s_box[tid][0] = 65535; s_box[tid][1] = 0; s_box[tid][2] = 0; s_box[tid][3] = 65535; s_reject[tid] = 0; }
    __syncthreads();
    const int minLabel = s_min_label;
    for (int i = tid; i < ncc; i += T) {
        const int lab = c_label[i];
        if (lab != minLabel && lab != 255) atomicMax(&s_best[lab], ((unsigned)c_size[i] << 16) | (unsigned)(65535 - i));
    }
    __syncthreads();
    // 6. ... and a new label must have a plausible size (:521-530)
    {
        const int minSize = (int)((float)K * a.minRelSizeNew), maxSize = (int)((float)K * a.maxRelSizeNew);
        for (int i = tid; i < ncc; i += T) {
            int lab = c_label[i];
            if (lab != minLabel && lab != 255 && (int)(65535u - (s_best[lab] & 0xffffu)) != i) lab = 255;
            if (a.allow_new && lab == (int)a.next_id && (c_size[i] < minSize || c_size[i] > maxSize)) lab = 255;
            c_label[i] = lab;
            // 7. bounding boxes over the surviving components of every model entry (:532-547)
            if (lab != 255) {
                const int e = s_id2idx[lab];
                if ((int)(a.ids[e] & 255u) == lab) {
                    atomicMin(&s_box[e][0], c_top[i]); atomicMax(&s_box[e][1], c_right[i]); atomicMax(&s_box[e][2], c_bottom[i]); atomicMin(&s_box[e][3], c_left[i]);
                }
            }
        }
    }
    __syncthreads();
    // Slic::mapToHigh, then 8. labels whose box lies inside the border strip are rejected (:549-563)
    if (tid < n_md) {
        const int top = (int)(unsigned short)(int)(s_box[tid][0] * kSpix + kSpix * 0.5), right = (int)(unsigned short)(int)(s_box[tid][1] * kSpix + kSpix * 0.5);
        const int bottom = (int)(unsigned short)(int)(s_box[tid][2] * kSpix + kSpix * 0.5), left = (int)(unsigned short)(int)(s_box[tid][3] * kSpix + kSpix * 0.5);
        s_box[tid][0] = top; s_box[tid][1] = right; s_box[tid][2] = bottom; s_box[tid][3] = left;
        if (a.ids[tid] != 0) {
            const unsigned borderSize = 20, fullHeight = (unsigned)a.height, fullWidth = (unsigned)a.width;
            const unsigned t = (unsigned)top, r = (unsigned)right, bo = (unsigned)bottom, l = (unsigned)left;
            if ((t < borderSize && bo < borderSize) || (l < borderSize && r < borderSize) ||
                (t > fullHeight - borderSize && bo > fullHeight - borderSize) || (l > fullWidth - borderSize && r > fullWidth - borderSize))
                s_reject[tid] = 1;
        }
    }
    __syncthreads();
    for (int i = tid; i < ncc; i += T) {
        const int lab = c_label[i];
        if (lab == 255) continue;
        const int e = s_id2idx[lab];
        if ((int)(a.ids[e] & 255u) == lab && s_reject[e]) c_label[i] = 255;
    }
    __threadfence_block();
    __syncthreads();
    GSTAMP(1, 4);   // gates
    // 9. final low-resolution label map
    for (int k = tid; k < K; k += T) { const unsigned char v = (unsigned char)c_label[comp[k]]; map[k] = v; a.low_map[k] = v; }
    __syncthreads();
    // (parents and component numbers are dead: their storage holds the low-resolution depths and every superpixel's model entry in the
    // layout of lds_blocked_sums; the tail up to a multiple of sixteen belongs to nobody)
    float* const s_depth = reinterpret_cast<float*>(s_pc);
    unsigned short* const s_entry = reinterpret_cast<unsigned short*>(s_pc + kSegDepthPad(kSegMaxK));
    const int nch = (K + kSeqBlock - 1) / kSeqBlock;
    for (int k = tid; k < nch * kSeqBlock; k += T) {
        const unsigned char v = k < K ? map[k] : (unsigned char)255;
        s_entry[k] = v == 255 ? (unsigned short)0xffff : (unsigned short)s_id2idx[v];
        s_depth[kSegDepthPad(k)] = k < K ? a.low_depth[k] : 0.f;
    }
    __syncthreads();
    GSTAMP(1, 5);   // label map
    // 10. depth statistics with one trimming pass (:570-621) and super-pixel counts (:624-627): sequential f32 sums in index order,
    //     one wave per model entry
    for (int ix = wave; ix < n_md; ix += (T >> 6)) {
        auto mine_at = [&](int i) { return s_entry[i] == (unsigned short)ix; };
        unsigned cnt = 0;
        for (int i = lane; i < K; i += 64) cnt += mine_at(i) ? 1u : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        const unsigned spc = cnt;
        float s1[1] = {0.f};
        lds_blocked_sums<1>(s1, nch, lane, s_depth, s_entry, (unsigned)ix, [&](bool mine, float d, float (&v)[1]) { v[0] = mine ? d : 0.f; });
        float sumDepth = s1[0];
        float mean = cnt ? sumDepth / (float)cnt : 0;
        s1[0] = 0.f;
        lds_blocked_sums<1>(s1, nch, lane, s_depth, s_entry, (unsigned)ix, [&](bool mine, float d, float (&v)[1]) { v[0] = mine ? fabsf(mean - d) : 0.f; });
        float sumDev = s1[0];
        float dev = cnt ? sumDev / (float)cnt : 0;
        if (ix != 0) {
            // trimming pass: elements beyond mean + 1.1 dev are taken out of the running sums, in index order (x - d == x + (-d))
            const double limit = 1.1 * (double)dev + (double)mean;
            unsigned out = 0;
            for (int i = lane; i < K; i += 64) out += (mine_at(i) && (double)s_depth[kSegDepthPad(i)] > limit) ? 1u : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) out += __shfl_xor(out, o, 64);
            if (out) {
                float s2[2] = {sumDepth, sumDev};
                lds_blocked_sums<2>(s2, nch, lane, s_depth, s_entry, (unsigned)ix, [&](bool mine, float d, float (&v)[2]) {
                    const bool trimmed = mine && (double)d > limit;
                    v[0] = trimmed ? -d : 0.f; v[1] = trimmed ? -fabsf(mean - d) : 0.f;
                });
                sumDepth = s2[0]; sumDev = s2[1];
            }
            cnt -= out;
        }
        mean = cnt ? sumDepth / (float)cnt : 0;
        dev = cnt ? sumDev / (float)cnt : 0;
        if (lane == 0) {
            cf_seg_model& o = a.result->model[ix];
            o.id = a.ids[ix]; o.superPixelCount = spc; o.avgConfidence = ix < a.n_models ? a.avg_conf[ix] : 0.f;
            o.depthMean = mean; o.depthStd = dev;
            o.top = s_box[ix][0]; o.right = s_box[ix][1]; o.bottom = s_box[ix][2]; o.left = s_box[ix][3];
            s_spc[ix] = spc;
        }
    }
    __syncthreads();
    GSTAMP(1, 6);   // depth statistics
    if (tid == 0) {
        int has_new = 0, n_out = n_md;
        if (a.allow_new) { if (s_spc[n_md - 1] > 0) has_new = 1; else n_out = n_md - 1; }
        a.result->has_new_label = has_new; a.result->n_models = n_out; a.result->depth_range = a.depth_range[0];
    }
    // publish: decisions and the low-resolution map into pinned host memory (what the frame's one host wait collects)
    __threadfence_block();
    __syncthreads();
    if (a.result_host) {
        const unsigned* src = reinterpret_cast<const unsigned*>(a.result);
        unsigned* dst = reinterpret_cast<unsigned*>(a.result_host);
        const int words = (int)((offsetof(cf_seg_result, model) + sizeof(cf_seg_model) * (size_t)n_md) / 4);   // header + the rows in use
        for (int k = tid; k < words; k += T) dst[k] = src[k];
    }
    if (a.low_map_host)
        for (int k = tid; k < (K + 3) / 4; k += T) a.low_map_host[k] = reinterpret_cast<const unsigned*>(map)[k];
    GSTAMP(1, 7);
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

// cf_seg_publish_poses: model m's tracked pose (T = [Rcurr | tcurr], row-major 4x4) and ICP statistics as f32 bit patterns, one 64-bit
// slot each, behind the segmentation sums -- zeros for models this process does not own, so that the caller's SUM all-reduce of the
// block leaves every model's pose on every rank (exact: one contributor per word)
struct PosePublishArgs { const OdomDev* st[kMaxL]; int n; };
__global__ void __launch_bounds__(64) pose_publish_kernel(const PosePublishArgs a, long long* __restrict__ tail)
{
    const int m = blockIdx.x, w = threadIdx.x;
    if (w >= kPoseWords) return;
    long long v = 0;
    const OdomDev* st = m < a.n ? a.st[m] : nullptr;
    if (st) {
        float f;
        if (w < 12) { const int r = w >> 2, c = w & 3; f = c < 3 ? st->Rcurr[r * 3 + c] : st->tcurr[r]; }
        else if (w < 16) f = w == 15 ? 1.f : 0.f;
        else f = w == 16 ? st->stats.last_icp_error : st->stats.last_icp_count;
        v = (long long)__float_as_uint(f);
    }
    tail[m * kPoseWords + w] = v;
}

struct cf_segmenter {
    cf_ctx* ctx = nullptr;
    int gx = 0, gy = 0, K = 0;
    int Lcap = 16;                       // label capacity of the buffers = max(16, the context's max_models) (a new label needs a free model slot)
    const void** d_acc_ptrs = nullptr;   // [2][Lcap] device copies of the models' ICP-error / vertex-confidence image pointers (> kAccTile models)
    const void** h_acc_ptrs = nullptr;   // pinned staging of the same
    int* labels = nullptr;
    float* centres = nullptr;
    unsigned long long* slic_sums = nullptr;
    unsigned* spix_count = nullptr; unsigned* depth_count = nullptr;
    unsigned long long *depth_sum = nullptr, *icp_sum = nullptr, *conf_sum = nullptr;
    int* resample = nullptr;
    unsigned char* low_map = nullptr;
    float *feat1 = nullptr, *feat2 = nullptr, *norm = nullptr, *K1t = nullptr, *K2t = nullptr;
    float* partial = nullptr;            // chunk partial sums [kCrfChunks][K][2][Lcap]
    std::vector<float> smooth_cache;     // host copy of the smoothness features K1t was built from
    float *unary = nullptr, *Q0 = nullptr, *Q1 = nullptr;
    // device-side unaries / post-processing (cf_seg_sums / cf_seg_infer / cf_seg_fetch)
    float *raw_mean = nullptr, *low_mean = nullptr;   // [(1 + 2 Lcap)][K]
    float *avg_conf = nullptr, *depth_range = nullptr;
    int *parent = nullptr, *comp = nullptr, *cc = nullptr;
    cf_seg_result* d_result = nullptr;
    cf_seg_result* h_result = nullptr;   // pinned
    unsigned char* h_low_map = nullptr;  // pinned [K]
    long long* h_pose_tail = nullptr;    // pinned [Lcap][kPoseWords]: the tail of the sums block after the caller's all-reduce
    bool poses_published = false;
    bool grid_kernel_built = false;      // K1t holds the kernel of the grid's own smoothness features (seg_feat1_kernel)
};

template <typename T>
static int seg_malloc(cf_ctx* ctx, T** p, size_t count)
{
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    HIPCHK(ctx, hipMemsetAsync(*p, 0, count * sizeof(T), ctx->stream));
    return CF_OK;
}

// the models' image pointers of the accumulation launch: kernel arguments for up to kAccTile models, a device table beyond
static int acc_pointers(cf_segmenter* s, AccArgs& a, int n_models, const float* const* icp_err, const float* const* vertconf4)
{
    if (n_models <= kAccTile) {
        for (int m = 0; m < n_models; m++) { a.icp[m] = icp_err[m]; a.vconf[m] = reinterpret_cast<const float4*>(vertconf4[m]); }
        return CF_OK;
    }
    cf_ctx* ctx = s->ctx;
    for (int m = 0; m < n_models; m++) { s->h_acc_ptrs[m] = icp_err[m]; s->h_acc_ptrs[s->Lcap + m] = vertconf4[m]; }
    HIPCHK(ctx, hipMemcpyAsync(s->d_acc_ptrs, s->h_acc_ptrs, sizeof(void*) * 2 * (size_t)s->Lcap, hipMemcpyHostToDevice, ctx->stream));
    a.icp_dev = reinterpret_cast<const float* const*>(s->d_acc_ptrs);
    a.vconf_dev = reinterpret_cast<const float4* const*>(s->d_acc_ptrs + s->Lcap);
    return CF_OK;
}

extern "C" {

int cf_seg_create(cf_ctx* ctx, cf_segmenter** out)
{
    if (!ctx || !out) return CF_EINVAL;
    if ((ctx->cfg.width % kSpix) || (ctx->cfg.height % kSpix)) { ctx->set_error("segmentation needs width/height multiples of 16"); return CF_EINVAL; }
    if ((ctx->cfg.width / kSpix) * (ctx->cfg.height / kSpix) > kSegMaxK) { ctx->set_error("segmentation supports at most 4800 superpixels (1280x960)"); return CF_EINVAL; }
    cf_segmenter* s = new cf_segmenter();
    s->ctx = ctx; s->gx = ctx->cfg.width / kSpix; s->gy = ctx->cfg.height / kSpix; s->K = s->gx * s->gy;
    *out = s;
    s->Lcap = ctx->cfg.max_models < 16 ? 16 : (ctx->cfg.max_models > kMaxL - 1 ? kMaxL - 1 : ctx->cfg.max_models);   // at least 16 (as before round 4); ids 0..254, 255 = rejected
    const size_t N = (size_t)ctx->cfg.width * ctx->cfg.height, K = (size_t)s->K, Lc = (size_t)s->Lcap;
    if (int r = seg_malloc(ctx, &s->d_acc_ptrs, 2 * Lc)) return r;
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&s->h_acc_ptrs), sizeof(void*) * 2 * Lc));
    if (int r = seg_malloc(ctx, &s->labels, N)) return r;
    if (int r = seg_malloc(ctx, &s->centres, K * 5)) return r;
    if (int r = seg_malloc(ctx, &s->slic_sums, K * 6)) return r;
    if (int r = seg_malloc(ctx, &s->spix_count, K)) return r;
    if (int r = seg_malloc(ctx, &s->depth_count, K)) return r;
    if (int r = seg_malloc(ctx, &s->depth_sum, K)) return r;
    // [icp | conf | pose tail] in one block: one collective of a model-parallel caller covers all of it (cf_seg_publish_poses)
    if (int r = seg_malloc(ctx, &s->icp_sum, 2 * K * Lc + Lc * kPoseWords)) return r;
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&s->h_pose_tail), sizeof(long long) * Lc * kPoseWords));
    s->conf_sum = s->icp_sum + K * Lc;
    if (int r = seg_malloc(ctx, &s->resample, K)) return r;
    if (int r = seg_malloc(ctx, &s->low_map, K)) return r;
    if (int r = seg_malloc(ctx, &s->feat1, K * 2)) return r;
    if (int r = seg_malloc(ctx, &s->feat2, K * 6)) return r;
    if (int r = seg_malloc(ctx, &s->norm, K)) return r;
    if (int r = seg_malloc(ctx, &s->K1t, K * K)) return r;
    if (int r = seg_malloc(ctx, &s->K2t, K * K)) return r;
    if (int r = seg_malloc(ctx, &s->partial, (size_t)kCrfChunks * K * 2 * Lc)) return r;
    if (int r = seg_malloc(ctx, &s->unary, K * Lc)) return r;
    if (int r = seg_malloc(ctx, &s->Q0, K * Lc)) return r;
    if (int r = seg_malloc(ctx, &s->Q1, K * Lc)) return r;
    if (int r = seg_malloc(ctx, &s->raw_mean, K * (3 + 2 * Lc))) return r;  // raw sums + the two lists of empty superpixels
    if (int r = seg_malloc(ctx, &s->low_mean, K * (1 + 2 * Lc))) return r;
    if (int r = seg_malloc(ctx, &s->avg_conf, Lc)) return r;
    if (int r = seg_malloc(ctx, &s->depth_range, (size_t)1)) return r;
    if (int r = seg_malloc(ctx, &s->parent, K)) return r;
    if (int r = seg_malloc(ctx, &s->comp, K)) return r;
    if (int r = seg_malloc(ctx, &s->cc, 6 * K)) return r;
    if (int r = seg_malloc(ctx, &s->d_result, (size_t)1)) return r;
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&s->h_result), sizeof(cf_seg_result), hipHostMallocCoherent));  // seg_post_kernel stores into it
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&s->h_low_map), ((size_t)K + 3) / 4 * 4, hipHostMallocCoherent));  // written as 32-bit words by seg_post_kernel
    memset(s->h_result, 0, sizeof(cf_seg_result));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

void cf_seg_destroy(cf_segmenter* s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    void* ptrs[] = {s->labels, s->centres, s->slic_sums, s->spix_count, s->depth_count, s->depth_sum, s->icp_sum, s->resample,
                    s->low_map, s->feat1, s->feat2, s->norm, s->K1t, s->K2t, s->partial, s->unary, s->Q0, s->Q1,
                    s->raw_mean, s->low_mean, s->avg_conf, s->depth_range, s->parent, s->comp, s->cc, s->d_result, (void*)s->d_acc_ptrs};
    for (void* p : ptrs) (void)hipFree(p);
    if (s->h_result) (void)hipHostFree(s->h_result);
    if (s->h_low_map) (void)hipHostFree(s->h_low_map);
    if (s->h_pose_tail) (void)hipHostFree(s->h_pose_tail);
    if (s->h_acc_ptrs) (void)hipHostFree(s->h_acc_ptrs);
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
    // gSLICr's Perform_Segmentation: assign; no_iters (5, Slic.cpp:38) x {update; assign} -- the labels are the sixth pass's
    for (int it = 0; it <= 5; it++) {
        if (it > 0) slic_update_kernel<<<(s->K + 255) / 256, 256, 0, st>>>(s->slic_sums, s->K, s->centres);
        slic_assign_kernel<<<dim3(s->gx, s->gy), 256, 0, st>>>(img, W, H, s->gx, s->gy, s->centres, s->labels, s->slic_sums);
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
    if (!s || !depth || n_models < 0 || n_models > s->Lcap) return CF_EINVAL;
    cf_ctx* ctx = s->ctx; hipStream_t st = ctx->stream;
    const size_t K = (size_t)s->K;
    HIPCHK(ctx, hipMemsetAsync(s->spix_count, 0, sizeof(unsigned) * K, st));
    HIPCHK(ctx, hipMemsetAsync(s->depth_count, 0, sizeof(unsigned) * K, st));
    HIPCHK(ctx, hipMemsetAsync(s->depth_sum, 0, sizeof(unsigned long long) * K, st));
    HIPCHK(ctx, hipMemsetAsync(s->icp_sum, 0, sizeof(unsigned long long) * K * (size_t)s->Lcap, st));
    HIPCHK(ctx, hipMemsetAsync(s->conf_sum, 0, sizeof(unsigned long long) * K * (size_t)s->Lcap, st));
    AccArgs a;
    memset(&a, 0, sizeof(a));
    a.labels = s->labels; a.depth = depth; a.n_models = n_models; a.cols = ctx->cfg.width; a.rows = ctx->cfg.height; a.gx = s->gx; a.gy = s->gy;
    if (int r = acc_pointers(s, a, n_models, icp_err, vertconf4)) return r;
    a.spix_count = s->spix_count; a.depth_count = s->depth_count; a.depth_sum = s->depth_sum; a.icp_sum = s->icp_sum; a.conf_sum = s->conf_sum;
    {
        SegBatch<AccArgs> B;
        memset(&B, 0, sizeof(B));
        B.m[0] = a;
        seg_accumulate_kernel<<<dim3(s->gx, s->gy, 1), 256, 0, st>>>(B);
    }
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

static CrfSeq crf_seq(cf_segmenter* s, int L)
{
    CrfSeq m{};
    m.feat = s->feat2; m.norm = s->norm; m.Kt = s->K2t; m.K1t = s->K1t; m.K2t = s->K2t;
    m.unary = s->unary; m.Q0 = s->Q0; m.Q1 = s->Q1; m.partial = s->partial; m.L = L;
    return m;
}
// the smoothness kernel K1t from feat1 (make_features: the grid's own features first -- seg_feat1_kernel)
static void build_grid_kernel(cf_segmenter* s, bool make_features)
{
    hipStream_t st = s->ctx->stream;
    const int n = s->K;
    if (make_features) seg_feat1_kernel<<<(n + 255) / 256, 256, 0, st>>>(s->gx, n, s->feat1);
    CrfBatch B;
    memset(&B, 0, sizeof(B));
    B.m[0] = crf_seq(s, 0);
    B.m[0].feat = s->feat1; B.m[0].Kt = s->K1t;
    launch_crf_kernel_matrix<2>(st, B, 1, n, false);
}

// DenseCRF2D inference as used by Segmentation.cpp:436-480 (exact kernels, see the file header).
// unary [K*L] row-per-node, feat_smooth [K*2], feat_app [K*6] host in; Q [K*L] host out (synchronous).
int cf_seg_crf(cf_segmenter* s, const float* unary_host, int L, const float* feat_smooth_host, const float* feat_app_host,
               float w_smooth, float w_app, int iterations, float* Q_host)
{
    if (!s || !unary_host || !Q_host || L <= 0 || L > s->Lcap) return CF_EINVAL;
    cf_ctx* ctx = s->ctx; hipStream_t st = ctx->stream;
    const int n = s->K;
    HIPCHK(ctx, hipMemcpyAsync(s->unary, unary_host, sizeof(float) * n * L, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(s->feat2, feat_app_host, sizeof(float) * n * 6, hipMemcpyHostToDevice, st));
    // the smoothness kernel only depends on the superpixel grid: rebuilt only when its features change
    const bool same_smooth = s->smooth_cache.size() == (size_t)n * 2 && memcmp(s->smooth_cache.data(), feat_smooth_host, sizeof(float) * n * 2) == 0;
    if (!same_smooth) {
        HIPCHK(ctx, hipMemcpyAsync(s->feat1, feat_smooth_host, sizeof(float) * n * 2, hipMemcpyHostToDevice, st));
        build_grid_kernel(s, false);
        s->smooth_cache.assign(feat_smooth_host, feat_smooth_host + (size_t)n * 2);
        s->grid_kernel_built = false;
    }
    CrfBatch B;
    memset(&B, 0, sizeof(B));
    B.m[0] = crf_seq(s, L);
    launch_crf_kernel_matrix<6>(st, B, 1, n, true);
    const float* q = launch_mean_field(st, B, 1, n, iterations, w_smooth, w_app) ? s->Q1 : s->Q0;
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
    launch_upsample(st, s->labels, s->low_map, N, full_dev);
    LAUNCHCHK(ctx);
    return CF_OK;
}

// ---- device-resident flavour: sums -> [collective] -> unaries -> mean field -> post-processing -> mask, no host wait ----
// One entry of a batched segmentation: what cf_seg_sums + cf_seg_infer take for one segmenter
struct SegJob {
    cf_segmenter* s; const float* depth; int n_models; const float* const* icp_err; const float* const* vertconf4;
    const uint8_t* rgba; const uint32_t* model_ids; uint32_t next_model_id; int allow_new; uint8_t* full_dev;
};
// the sums of S <= kSegBatch segmenters of one image size in ONE launch (+ the resample labels in its extra grid row)
static int enqueue_accumulate(cf_ctx* ctx, const SegJob* jobs, int S)
{
    hipStream_t st = ctx->stream;
    cf_segmenter* s0 = jobs[0].s;
    SegBatch<AccArgs> B;
    memset(&B, 0, sizeof(B));
    const bool ride = s0->gy <= 256;  // (gx workgroups of 256 threads cover the K = gx * gy resample points)
    for (int e = 0; e < S; e++) {
        cf_segmenter* s = jobs[e].s;
        AccArgs& a = B.m[e];
        a.labels = s->labels; a.depth = jobs[e].depth; a.n_models = jobs[e].n_models; a.cols = ctx->cfg.width; a.rows = ctx->cfg.height; a.gx = s->gx; a.gy = s->gy;
        if (int r = acc_pointers(s, a, jobs[e].n_models, jobs[e].icp_err, jobs[e].vertconf4)) return r;
        a.spix_count = s->spix_count; a.depth_count = s->depth_count; a.depth_sum = s->depth_sum; a.icp_sum = s->icp_sum; a.conf_sum = s->conf_sum;
        a.resample = ride ? s->resample : nullptr;
    }
    seg_accumulate_kernel<<<dim3(s0->gx, s0->gy + (ride ? 1 : 0), S), 256, 0, st>>>(B);
    if (!ride)
        for (int e = 0; e < S; e++)
            seg_resample_kernel<<<(jobs[e].s->K + 255) / 256, 256, 0, st>>>(jobs[e].s->labels, ctx->cfg.width, ctx->cfg.height, s0->gx, s0->gy, jobs[e].s->resample);
    LAUNCHCHK(ctx);
    return CF_OK;
}

// Slic::downsample* sums of the frame and of every model (Slic.h:48-120), left on the device.  *sums_dev (nullable) receives the
// device address of the per-model sums -- int64 [2][max_models][K]: ICP-error sums of model m at [0][m][.], confidence sums at [1][m][.] --
// which a model-parallel caller SUM-all-reduces in place over the ranks before cf_seg_infer (owners contribute, everybody else
// passes zero images).  The accumulators are expected zero on entry; cf_seg_infer leaves them zero again.
int cf_seg_sums(cf_segmenter* s, const float* depth, int n_models, const float* const* icp_err, const float* const* vertconf4,
                int64_t** sums_dev, uint64_t* sums_words)
{
    if (!s || !depth || n_models <= 0 || n_models > s->Lcap || !icp_err || !vertconf4) return CF_EINVAL;
    SegJob job{};
    job.s = s; job.depth = depth; job.n_models = n_models; job.icp_err = icp_err; job.vertconf4 = vertconf4;
    if (int r = enqueue_accumulate(s->ctx, &job, 1)) return r;
    if (sums_dev) *sums_dev = reinterpret_cast<int64_t*>(s->icp_sum);
    if (sums_words) *sums_words = 2ull * (uint64_t)s->Lcap * (uint64_t)s->K + (uint64_t)s->Lcap * kPoseWords;  // the pose tail rides along (zeros unless published)
    return CF_OK;
}

// Model-parallel callers: put the poses the trackers hold on THIS process behind the sums (trackers[m] == NULL: model m is tracked
// elsewhere), between cf_seg_sums and the all-reduce; after cf_seg_infer + cf_seg_fetch, cf_seg_fetch_poses hands out all of them.
// Replaces a separate (blocking) pose exchange per frame.
int cf_seg_publish_poses(cf_segmenter* s, int n_models, cf_odom* const* trackers)
{
    if (!s || n_models <= 0 || n_models > s->Lcap || !trackers) return CF_EINVAL;
    PosePublishArgs a{};
    a.n = n_models;
    for (int m = 0; m < n_models; m++) a.st[m] = trackers[m] ? trackers[m]->d_state : nullptr;
    pose_publish_kernel<<<s->Lcap, 64, 0, s->ctx->stream>>>(a, reinterpret_cast<long long*>(s->icp_sum) + 2 * (size_t)s->Lcap * s->K);
    LAUNCHCHK(s->ctx);
    s->poses_published = true;
    return CF_OK;
}
// words: [n_models][18] (16 pose words row-major, ICP error, ICP inlier count) as written by the owners; valid after cf_seg_fetch
int cf_seg_fetch_poses(cf_segmenter* s, int n_models, int64_t* words_host)
{
    if (!s || n_models <= 0 || n_models > s->Lcap || !words_host) return CF_EINVAL;
    if (!s->poses_published) { s->ctx->set_error("cf_seg_fetch_poses: no poses were published for this inference"); return CF_ESTATE; }
    HIPCHK(s->ctx, hipStreamSynchronize(s->ctx->stream));
    memcpy(words_host, s->h_pose_tail, sizeof(int64_t) * (size_t)n_models * kPoseWords);
    s->poses_published = false;
    return CF_OK;
}

}  // extern "C"
// Everything after the sums (Segmentation.cpp:160-706) for S <= kSegBatch segmenters of one image size in one chain of launches: unaries,
// kernel matrices, mean-field steps, arg-max / connected components / gates / statistics, up-sampling into full_dev.  CAP: the capacity
// of the id table in the post-processing arguments (17 for a batch, 257 for one segmenter with many labels).
template <int CAP, int N>
static int enqueue_infer(cf_ctx* ctx, const cf_seg_params* P, const SegJob* jobs, int S)
{
    hipStream_t st = ctx->stream;
    const int n = jobs[0].s->K;
    SegBatch<SegUnaryArgs> U;
    CrfBatch C;
    SegBatch<SegPostArgsT<CAP>, N> PB;
    SegBatch<UpsampleArgs> UP;
    memset(&U, 0, sizeof(U)); memset(&C, 0, sizeof(C)); memset(&PB, 0, sizeof(PB)); memset(&UP, 0, sizeof(UP));
    for (int e = 0; e < S; e++) {
        cf_segmenter* s = jobs[e].s;
        const int n_models = jobs[e].n_models, L = n_models + (jobs[e].allow_new ? 1 : 0);
        SegUnaryArgs& u = U.m[e];
        u.K = n; u.gx = s->gx; u.gy = s->gy; u.n_models = n_models; u.L = L; u.allow_new = jobs[e].allow_new ? 1 : 0;
        u.unaryWeightError = P->unaryWeightError; u.unaryKError = P->unaryKError; u.unaryThresholdNew = P->unaryThresholdNew;
        u.scaleFeaturesRGB = P->scaleFeaturesRGB; u.scaleFeaturesDepth = P->scaleFeaturesDepth; u.scaleFeaturesPos = P->scaleFeaturesPos;
        u.spix_count = s->spix_count; u.depth_count = s->depth_count; u.depth_sum = s->depth_sum; u.icp_sum = s->icp_sum; u.conf_sum = s->conf_sum;
        u.resample = s->resample; u.rgba = reinterpret_cast<const uchar4*>(jobs[e].rgba);
        u.raw = s->raw_mean; u.low = s->low_mean; u.unary = s->unary; u.feat2 = s->feat2; u.avg_conf = s->avg_conf; u.depth_range = s->depth_range;
        C.m[e] = crf_seq(s, L);
    }
    seg_unary_kernel<<<S, 1024, 0, st>>>(U);
    for (int e = 0; e < S; e++) {
        cf_segmenter* s = jobs[e].s;
        if (!s->grid_kernel_built) {  // the smoothness kernel only depends on the superpixel grid: built once per segmenter
            build_grid_kernel(s, true);
            s->grid_kernel_built = true;
            s->smooth_cache.clear();
        }
    }
    launch_crf_kernel_matrix<6>(st, C, S, n, true);
    const int flip = launch_mean_field(st, C, S, n, P->crfIterations, P->weightSmoothness, P->weightAppearance);
    for (int e = 0; e < S; e++) {
        cf_segmenter* s = jobs[e].s;
        const int n_models = jobs[e].n_models, L = n_models + (jobs[e].allow_new ? 1 : 0);
        SegPostArgsT<CAP>& p = PB.m[e];
        p.K = n; p.gx = s->gx; p.gy = s->gy; p.n_models = n_models; p.L = L; p.allow_new = jobs[e].allow_new ? 1 : 0;
        p.width = ctx->cfg.width; p.height = ctx->cfg.height; p.next_id = jobs[e].next_model_id;
        p.minRelSizeNew = P->minRelSizeNew; p.maxRelSizeNew = P->maxRelSizeNew;
        for (int m = 0; m < n_models; m++) p.ids[m] = jobs[e].model_ids[m];
        if (jobs[e].allow_new) p.ids[n_models] = jobs[e].next_model_id;
        p.Q = flip ? s->Q1 : s->Q0; p.low_depth = s->low_mean; p.avg_conf = s->avg_conf; p.depth_range = s->depth_range;
        p.parent = s->parent; p.comp = s->comp; p.cc = s->cc; p.low_map = s->low_map; p.result = s->d_result;
        p.result_host = s->h_result; p.low_map_host = reinterpret_cast<unsigned*>(s->h_low_map);
        UP.m[e] = UpsampleArgs{s->labels, s->low_map, jobs[e].full_dev};
    }
    seg_post_kernel<CAP, N><<<S, 1024, 0, st>>>(PB);
    const int Npx = ctx->cfg.width * ctx->cfg.height;
    seg_upsample_kernel<<<dim3((Npx + 255) / 256, S), 256, 0, st>>>(UP, Npx);
    LAUNCHCHK(ctx);
#ifdef CF_ABLATE
    {
        static const int trace_call = getenv("CF_SEG_TRACE") ? atoi(getenv("CF_SEG_TRACE")) : -1;
        static int seen = 0;
        if (trace_call >= 0 && seen++ == trace_call) {
            HIPCHK(ctx, hipStreamSynchronize(st));
            unsigned long long h[2][16];
            HIPCHK(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seg_trace), sizeof(h)));
            const char* un[] = {"raw sums", "ordered lists", "empty superpixels", "depth range", "average confidence", "unaries + features", "zeroing"};
            const char* pn[] = {"arg-max + components", "roots numbered", "component statistics", "gates", "label map", "depth statistics", "publish"};
            for (int k = 0; k < 7; k++) fprintf(stderr, "[seg trace] unary %-22s %6lld ns\n", un[k], (long long)(h[0][k + 1] - h[0][k]) * 10);
            for (int k = 0; k < 7; k++) fprintf(stderr, "[seg trace] post  %-22s %6lld ns\n", pn[k], (long long)(h[1][k + 1] - h[1][k]) * 10);
            fprintf(stderr, "[seg trace] post  arg-max alone %6lld ns; component sweeps since the start: %lld\n", (long long)(h[1][8] - h[1][0]) * 10, (long long)h[1][9]);
            for (int k = 0; k < 3; k++)
                fprintf(stderr, "[seg trace] post  component sweep %d: hooks done at %6lld ns, walks done at %6lld ns (from the kernel's first stamp; stale if the sweep did not run)\n", k,
                        (long long)(h[1][10 + 2 * k] - h[1][0]) * 10, (long long)(h[1][11 + 2 * k] - h[1][0]) * 10);
        }
    }
#endif
    for (int e = 0; e < S; e++) {
        cf_segmenter* s = jobs[e].s;
        if (s->poses_published) {  // the tail now holds what the caller's all-reduce made of it; the next frame starts from zeros again
            long long* tail = reinterpret_cast<long long*>(s->icp_sum) + 2 * (size_t)s->Lcap * s->K;
            HIPCHK(ctx, hipMemcpyAsync(s->h_pose_tail, tail, sizeof(long long) * (size_t)s->Lcap * kPoseWords, hipMemcpyDeviceToHost, st));
        }
    }
    return CF_OK;
}

extern "C" {
// Only enqueues; the decisions arrive with cf_seg_fetch.
int cf_seg_infer(cf_segmenter* s, const cf_seg_params* P, const uint8_t* rgba, int n_models, const uint32_t* model_ids, uint32_t next_model_id,
                 int allow_new, uint8_t* full_dev)
{
    if (!s || !P || !rgba || !model_ids || !full_dev || n_models <= 0) return CF_EINVAL;
    const int L = n_models + (allow_new ? 1 : 0);
    if (L > s->Lcap) { s->ctx->set_error("segmentation: more labels than the context's max_models (" + std::to_string(s->Lcap) + ")"); return CF_EINVAL; }
    SegJob job{};
    job.s = s; job.n_models = n_models; job.rgba = rgba; job.model_ids = model_ids; job.next_model_id = next_model_id; job.allow_new = allow_new; job.full_dev = full_dev;
    if (L <= 16) return enqueue_infer<17, kSegBatch>(s->ctx, P, &job, 1);
    return enqueue_infer<kMaxL + 1, 1>(s->ctx, P, &job, 1);
}

// cf_seg_sums + cf_seg_infer of several segmenters of ONE context (the sequences of a lock-step group) through shared launches: the
// chain of ~30 launch-floor kernels is issued once per kSegBatch segmenters instead of once per segmenter.  Per segmenter the results
// are those of the two single calls, bit for bit; the decisions arrive with each segmenter's cf_seg_fetch.  No collective can sit between
// the sums and the inference here (single-process callers).
int cf_seg_run_batch(cf_ctx* ctx, const cf_seg_params* P, const cf_seg_job* jobs_in, int n_jobs)
{
    if (!ctx || !P || !jobs_in || n_jobs <= 0) return CF_EINVAL;
    std::vector<SegJob> jobs((size_t)n_jobs);
    bool batchable = true;
    for (int e = 0; e < n_jobs; e++) {
        const cf_seg_job& j = jobs_in[e];
        if (!j.seg || j.seg->ctx != ctx || !j.depth || !j.icp_err || !j.vertconf4 || !j.rgba || !j.model_ids || !j.full_dev || j.n_models <= 0) return CF_EINVAL;
        const int L = j.n_models + (j.allow_new ? 1 : 0);
        if (L > j.seg->Lcap) { ctx->set_error("segmentation: more labels than the context's max_models (" + std::to_string(j.seg->Lcap) + ")"); return CF_EINVAL; }
        for (int k = 0; k < e; k++) if (jobs_in[k].seg == j.seg) return CF_EINVAL;
        batchable = batchable && L <= 16 && j.seg->K == jobs_in[0].seg->K && j.seg->gx == jobs_in[0].seg->gx;
        jobs[e] = SegJob{j.seg, j.depth, j.n_models, j.icp_err, j.vertconf4, j.rgba, j.model_ids, j.next_model_id, j.allow_new, j.full_dev};
    }
    if (!batchable) {  // (a sequence with more than 16 labels: one chain per segmenter)
        for (int e = 0; e < n_jobs; e++) {
            if (int r = enqueue_accumulate(ctx, &jobs[e], 1)) return r;
            if (int r = cf_seg_infer(jobs[e].s, P, jobs[e].rgba, jobs[e].n_models, jobs[e].model_ids, jobs[e].next_model_id, jobs[e].allow_new, jobs[e].full_dev)) return r;
        }
        return CF_OK;
    }
    for (int base = 0; base < n_jobs; base += kSegBatch) {
        const int S = n_jobs - base < kSegBatch ? n_jobs - base : kSegBatch;
        if (int r = enqueue_accumulate(ctx, &jobs[base], S)) return r;
        if (int r = enqueue_infer<17, kSegBatch>(ctx, P, &jobs[base], S)) return r;
    }
    return CF_OK;
}

// waits for the stream and hands out the decisions of the last cf_seg_infer (low_map_host: nullable [K])
int cf_seg_fetch(cf_segmenter* s, cf_seg_result* out, uint8_t* low_map_host)
{
    if (!s || !out) return CF_EINVAL;
    if (int r = cf_wait_stream(s->ctx)) return r;
    *out = *s->h_result;
    if (low_map_host) memcpy(low_map_host, s->h_low_map, (size_t)s->K);
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
