// track_ref.hip -- the REFERENCE-ORDER tracker: cf_set_icp_arith(ctx, CF_ICP_ARITH_REFERENCE).
//
// The default tracker (track_reduce.hip) sums exact integers, so its results do not depend on the launch shape; the reference's do --
// every reduction of RGBDOdometry::getIncrementalTransformation is an f32 tree whose shape follows the launch configuration
// (Core/Cuda/reduce.cu:90-185: 32-lane shuffle-down tree, block tree, second-stage reduceSum; :396-417 / :606-626 / :1092-1112:
// thread-strided partial sums) -- and the poses that come out of the two differ in the seventh digit.  Enough to flip association and
// confidence thresholds of the fusion a few times per frame, which is why "surfel counts exactly" only held against the oracle.
//
// This file is the third rounding specification: the SAME per-pixel arithmetic, reduced in the reference's own order at the launch
// shapes of Core/Utils/GPUConfig.h:51-58 (the constructor's defaults: what every board that is not in its table of NVIDIA names runs
// with), and the reference's host loop around it (RGBDOdometry.cpp:217-477: kernel, second-stage kernel, device synchronisation, a
// 116-byte read-back, host solve) with the algebra of gn_ref_host.h.  Its results equal those of the reference's RGBDOdometry class
// compiled from /root/reference bit for bit (tests/test_refpin_gpu.py, tests/golden/ref_odo_v1.npz / ref_traj_v1.npz), whole
// trajectories and surfel counts included.  It is a parity mode: ~60 host round trips per tracker and frame, like the reference.
//
// A CUDA warp is 32 lanes, a CDNA4 wave 64: the reference's warp-level tree is run on the two 32-lane halves of a wave with
// width-32 shuffles (__shfl_down(v, o, 32): a lane whose partner would be in the other half reads itself, exactly what a warp's
// out-of-range __shfl_down returns), the block tree over blockDim / 32 "warps" as written.
#include <vector>

#include "cf_device.h"
#include "cf_host.h"
#include "gn_ref_host.h"

namespace cf {

// launch shapes: GPUConfig.h:51-58 (threads, blocks), second stage reduce.cu:476 / 667 / 1153 with MAX_THREADS of the host pass
// (cudafuncs.cuh:55-59: __CUDA_ARCH__ is not defined there)
constexpr int kRefIcpThreads = 128, kRefIcpBlocks = 112;
constexpr int kRefRgbThreads = 128, kRefRgbBlocks = 112;
constexpr int kRefResThreads = 256, kRefResBlocks = 336;
constexpr int kRefSo3Threads = 160, kRefSo3Blocks = 64;
constexpr int kRefStage2Threads = 512;
constexpr int kRefStride = 32;   // floats per partial (29 / 11 used)

// warpReduceSum / blockReduceSum (reduce.cu:90-165, 187-240): every field of the struct goes through the same tree
template <int K>
__device__ __forceinline__ void ref_warp_reduce(float (&v)[K])
{
    for (int offset = 16; offset > 0; offset /= 2)
#pragma unroll
        for (int k = 0; k < K; k++) v[k] += __shfl_down(v[k], offset, 32);
}
template <int K>
__device__ __forceinline__ void ref_block_reduce(float (&v)[K])
{
    __shared__ float shared[32][K];
    const int lane = threadIdx.x % 32, wid = threadIdx.x / 32;
    ref_warp_reduce<K>(v);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < K; k++) shared[wid][k] = v[k];
    __syncthreads();
    // "ensure we only grab a value from shared memory if that warp existed"
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = ((int)threadIdx.x < (int)blockDim.x / 32) ? shared[lane][k] : 0.f;
    if (wid == 0) ref_warp_reduce<K>(v);
}
// reduceSum<<<1, MAX_THREADS>>> (reduce.cu:166-185)
template <int K>
__global__ void __launch_bounds__(kRefStage2Threads) ref_reduce_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int N)
{
    float sum[K];
#pragma unroll
    for (int k = 0; k < K; k++) sum[k] = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += blockDim.x * gridDim.x)
#pragma unroll
        for (int k = 0; k < K; k++) sum[k] += in[(size_t)i * kRefStride + k];
    ref_block_reduce<K>(sum);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; k++) out[blockIdx.x * kRefStride + k] = sum[k];
}

// the JtJJtrSE3 initialiser lists of reduce.cu:352-388 / 563-599
__device__ __forceinline__ void ref_se3_products(const float (&row)[7], bool found, float (&v)[29])
{
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int j = a; j < 7; j++) v[k++] = row[a] * row[j];
    v[27] = row[6] * row[6];
    v[28] = (float)found;
}

// ---- ICPReduction (reduce.cu:257-425) ---------------------------------------------------------------------------------------------
struct RefIcpArgs {
    m33 Rcurr, Rprev_inv; f3 tcurr, tprev; cf_cam intr;
    const float* vc; const float* nc; const float* vp; const float* np;
    float distThres, angleThres; int cols, rows;
    float* err;   // nullable: outErrorSurface
};
__device__ __forceinline__ bool ref_icp_row(const RefIcpArgs& a, int i, float (&row)[7])
{
    const int cols = a.cols, rows = a.rows, N = cols * rows;
    const int y = i / cols, x = i - y * cols;
    (void)x; (void)rows;
#pragma unroll
    for (int k = 0; k < 7; k++) row[k] = 0.f;
    const f3 vcurr = {a.vc[i], a.vc[i + N], a.vc[i + 2 * N]};
    const f3 vcurr_g = mul(a.Rcurr, vcurr) + a.tcurr;
    const f3 vcurr_cp = mul(a.Rprev_inv, vcurr_g - a.tprev);
    const int ux = f2i_rn(vcurr_cp.x * a.intr.fx / vcurr_cp.z + a.intr.cx);
    const int uy = f2i_rn(vcurr_cp.y * a.intr.fy / vcurr_cp.z + a.intr.cy);
    if (ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcurr_cp.z < 0) {
        if (a.err) a.err[i] = 0.0f;
        return false;
    }
    const int g = uy * cols + ux;
    const f3 vprev_g = {a.vp[g], a.vp[g + N], a.vp[g + 2 * N]};
    const f3 ncurr = {a.nc[i], a.nc[i + N], a.nc[i + 2 * N]};
    const f3 ncurr_g = mul(a.Rcurr, ncurr);
    const f3 nprev_g = {a.np[g], a.np[g + N], a.np[g + 2 * N]};
    const float dist = norm(vprev_g - vcurr_g);
    const float sine = norm(cross(ncurr_g, nprev_g));
    if (a.err) a.err[i] = is_finite(dist) ? dist : 0.0f;
    const bool found = sine < a.angleThres && dist <= a.distThres && !is_nan(ncurr.x) && !is_nan(nprev_g.x);
    if (found) {
        const f3 s_cp = mul(a.Rprev_inv, vcurr_g - a.tprev);
        const f3 d_cp = mul(a.Rprev_inv, vprev_g - a.tprev);
        const f3 n_cp = mul(a.Rprev_inv, nprev_g);
        const f3 cr = cross(s_cp, n_cp);
        row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
        row[3] = cr.x; row[4] = cr.y; row[5] = cr.z;
        row[6] = dot(n_cp, s_cp - d_cp);
    }
    return found;
}
__global__ void __launch_bounds__(kRefIcpThreads) ref_icp_kernel(const RefIcpArgs a, float* __restrict__ part)
{
    float sum[29];
#pragma unroll
    for (int k = 0; k < 29; k++) sum[k] = 0.f;
    const int N = a.cols * a.rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += blockDim.x * gridDim.x) {
        float row[7], v[29];
        const bool found = ref_icp_row(a, i, row);
        ref_se3_products(row, found, v);
#pragma unroll
        for (int k = 0; k < 29; k++) sum[k] += v[k];
    }
    ref_block_reduce<29>(sum);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 29; k++) part[blockIdx.x * kRefStride + k] = sum[k];
}

// ---- RGBResidual (reduce.cu:748-971): one DataTerm per pixel; the two sums are integers (count, sum of int(diff^2)) ---------------------
struct RefResArgs {
    float minScale; const int16_t* dIdx; const int16_t* dIdy; const float* lastDepth; const float* nextDepth;
    const uint8_t* lastImage; const uint8_t* nextImage; cf_dataterm* corres; float maxDepthDelta;
    float kt[3], krk[9]; int cols, rows;
    int* sums;   // [0] count, [1] sigma
};
__global__ void __launch_bounds__(kRefResThreads) ref_rgb_residual_kernel(const RefResArgs a)
{
    const int cols = a.cols, rows = a.rows, N = cols * rows;
    int cnt = 0, sig = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += blockDim.x * gridDim.x) {
        const int i = k / cols, j0 = k - i * cols;
        cf_dataterm c; c.zero_x = c.zero_y = c.one_x = c.one_y = 0; c.diff = 0.f; c.valid = 0;
        if (j0 < cols - 5 && i < rows - 1) {
            bool valid = true;
            for (int u = max(i - 2, 0); u < min(i + 2, rows); u++)
                for (int v = max(j0 - 2, 0); v < min(j0 + 2, cols); v++) valid = valid && (a.nextImage[u * cols + v] > 0);
            if (valid) {
                const int valx = a.dIdx[k], valy = a.dIdy[k];
                const float mTwo = (float)((valx * valx) + (valy * valy));
                if (mTwo >= a.minScale) {
                    const int y = i, x = j0;
                    const float d1 = a.nextDepth[k];
                    if (!is_nan(d1)) {
                        const float transformed_d1 = (float)(d1 * (a.krk[6] * x + a.krk[7] * y + a.krk[8]) + a.kt[2]);
                        const int u0 = f2i_rn((d1 * (a.krk[0] * x + a.krk[1] * y + a.krk[2]) + a.kt[0]) / transformed_d1);
                        const int v0 = f2i_rn((d1 * (a.krk[3] * x + a.krk[4] * y + a.krk[5]) + a.kt[1]) / transformed_d1);
                        if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
                            const float d0 = a.lastDepth[v0 * cols + u0];
                            const uint8_t li = a.lastImage[v0 * cols + u0];
                            if (d0 > 0 && fabsf(transformed_d1 - d0) <= a.maxDepthDelta && li != 0) {
                                c.zero_x = (int16_t)u0; c.zero_y = (int16_t)v0; c.one_x = (int16_t)x; c.one_y = (int16_t)y;
                                c.diff = (float)a.nextImage[k] - (float)li;
                                c.valid = 1;
                                cnt += 1; sig += (int)(c.diff * c.diff);
                            }
                        }
                    }
                }
            }
        }
        *reinterpret_cast<int4*>(&a.corres[k]) = *reinterpret_cast<const int4*>(&c);   // flat index, reduce.cu:862
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); sig += __shfl_xor(sig, o, 64); }
    if ((threadIdx.x & 63) == 0) { if (cnt) atomicAdd(&a.sums[0], cnt); if (sig) atomicAdd(&a.sums[1], sig); }
}

// ---- RGBReduction (reduce.cu:503-687) --------------------------------------------------------------------------------------------------
struct RefRgbArgs {
    const cf_dataterm* corres; float sigma; const float* cloud; float fx, fy; const int16_t* dIdx; const int16_t* dIdy;
    float sobelScale; int cols, rows;
};
__device__ __forceinline__ bool ref_rgb_row(const RefRgbArgs& a, int i, float (&row)[7])
{
#pragma unroll
    for (int k = 0; k < 7; k++) row[k] = 0.f;
    const cf_dataterm c = a.corres[i];
    if (!c.valid) return false;
    float w = a.sigma + fabsf(c.diff);
    w = w > 1.19209290E-07F ? 1.0f / w : 1.0f;
    if (a.sigma == -1) w = 1;
    row[6] = -w * c.diff;
    const float* cp = a.cloud + (size_t)(c.zero_y * a.cols + c.zero_x) * 3;
    const float px = cp[0], py = cp[1], pz = cp[2];
    const float invz = 1.0f / pz;
    const float dI_dx_val = w * a.sobelScale * (float)a.dIdx[c.one_y * a.cols + c.one_x];
    const float dI_dy_val = w * a.sobelScale * (float)a.dIdy[c.one_y * a.cols + c.one_x];
    const float v0 = dI_dx_val * a.fx * invz;
    const float v1 = dI_dy_val * a.fy * invz;
    const float v2 = -(v0 * px + v1 * py) * invz;
    row[0] = v0; row[1] = v1; row[2] = v2;
    row[3] = -pz * v1 + py * v2;
    row[4] = pz * v0 - px * v2;
    row[5] = -py * v0 + px * v1;
    return true;
}
__global__ void __launch_bounds__(kRefRgbThreads) ref_rgb_step_kernel(const RefRgbArgs a, float* __restrict__ part)
{
    float sum[29];
#pragma unroll
    for (int k = 0; k < 29; k++) sum[k] = 0.f;
    const int N = a.cols * a.rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += blockDim.x * gridDim.x) {
        float row[7], v[29];
        const bool found = ref_rgb_row(a, i, row);
        ref_se3_products(row, found, v);
#pragma unroll
        for (int k = 0; k < 29; k++) sum[k] += v[k];
    }
    ref_block_reduce<29>(sum);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 29; k++) part[blockIdx.x * kRefStride + k] = sum[k];
}

// ---- SO3Reduction (reduce.cu:973-1176) --------------------------------------------------------------------------------------------------
struct RefSo3Args { const uint8_t* lastImage; const uint8_t* nextImage; m33 B, Ki; float krlr[9]; int cols, rows; };
__device__ __forceinline__ void ref_so3_gradient(const uint8_t* __restrict__ img, int cols, int x, int y, float& gx, float& gy)
{
    const float actu = (float)img[y * cols + x];
    float back = (float)img[y * cols + x - 1], fore = (float)img[y * cols + x + 1];
    gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = (float)img[(y - 1) * cols + x]; fore = (float)img[(y + 1) * cols + x];
    gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}
__device__ __forceinline__ bool ref_so3_row(const RefSo3Args& c, int k, float (&row)[4])
{
    const int cols = c.cols, rows = c.rows;
    const int y = k / cols, x = k - y * cols;
    row[0] = row[1] = row[2] = row[3] = 0.f;
    const f3 unwarped = {(float)x, (float)y, 1.0f};
    const f3 warped = mul(c.B, unwarped);
    const int wx = f2i_rn(warped.x / warped.z), wy = f2i_rn(warped.y / warped.z);
    if (!(wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 && x < cols - 1 && y >= 1 && y < rows - 1)) return false;
    float gnx, gny, glx, gly;
    ref_so3_gradient(c.nextImage, cols, wx, wy, gnx, gny);
    ref_so3_gradient(c.lastImage, cols, x, y, glx, gly);
    const float gx = (gnx + glx) / 2.0f, gy = (gny + gly) / 2.0f;
    const f3 point = mul(c.Ki, unwarped);
    const float z2 = point.z * point.z;
    const float a = c.krlr[0], b = c.krlr[1], cc = c.krlr[2], d = c.krlr[3], e = c.krlr[4], f = c.krlr[5], g = c.krlr[6], h = c.krlr[7], ii = c.krlr[8];
    const f3 left = {((point.z * (d * gy + a * gx)) - (gy * g * y) - (gx * g * x)) / z2,
                     ((point.z * (e * gy + b * gx)) - (gy * h * y) - (gx * h * x)) / z2,
                     ((point.z * (f * gy + cc * gx)) - (gy * ii * y) - (gx * ii * x)) / z2};
    const f3 jac = cross(left, point);
    row[0] = jac.x; row[1] = jac.y; row[2] = jac.z;
    row[3] = -((float)c.nextImage[wy * cols + wx] - (float)c.lastImage[y * cols + x]);
    return true;
}
__global__ void __launch_bounds__(kRefSo3Threads) ref_so3_kernel(const RefSo3Args a, float* __restrict__ part)
{
    float sum[11];
#pragma unroll
    for (int k = 0; k < 11; k++) sum[k] = 0.f;
    const int N = a.cols * a.rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += blockDim.x * gridDim.x) {
        float row[4], v[11];
        const bool found = ref_so3_row(a, i, row);
        int s = 0;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p; q < 4; q++) v[s++] = row[p] * row[q];
        v[9] = row[3] * row[3];
        v[10] = (float)found;
#pragma unroll
        for (int k = 0; k < 11; k++) sum[k] += v[k];
    }
    ref_block_reduce<11>(sum);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 11; k++) part[blockIdx.x * kRefStride + k] = sum[k];
}

// ================================================================================================================================
// host: RGBDOdometry::getIncrementalTransformation (RGBDOdometry.cpp:217-477) for ONE tracker, on `s`.  The tracker's pyramids are what
// the preparation launches left in its buffers (the same ones the default tracker reads); the result goes to the tracker's pinned host
// state and its device state, where cf_odom_fetch_result / cf_models_preindex look for it.
// ================================================================================================================================
#define REFCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) { ctx->set_error(std::string(#call) + ": " + hipGetErrorString(e_)); return CF_EHIP; } \
    } while (0)

int ref_scratch(cf_ctx* ctx)
{
    if (ctx->d_ref) return CF_OK;
    // [0, 112 x 32): partials; then 3 x 32 totals (ICP, RGB, SO3) and two integers
    REFCHK(hipMalloc(reinterpret_cast<void**>(&ctx->d_ref), sizeof(float) * (size_t)(kRefIcpBlocks + 4) * kRefStride));
    REFCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_ref), sizeof(float) * 4 * kRefStride));
    return CF_OK;
}

int ref_track(cf_ctx* ctx, cf_odom* od, const float pose[16], const cf_track_opts* opts, float* err_surface)
{
    using namespace refhost;
    if (int r = ref_scratch(ctx)) return r;
    hipStream_t s = ctx->stream;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const cf_cam intr = {ctx->cfg.fx, ctx->cfg.fy, ctx->cfg.cx, ctx->cfg.cy};
    auto level_cam = [&](int l) { const int div = 1 << l; return cf_cam{intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div}; };
    float* const d_part = ctx->d_ref;
    float* const d_tot = ctx->d_ref + (size_t)kRefIcpBlocks * kRefStride;        // [3][32]
    int* const d_cnt = reinterpret_cast<int*>(d_tot + 3 * kRefStride);           // [2]
    float* const h_tot = ctx->h_ref;                                               // [3][32] + ints behind
    int* const h_cnt = reinterpret_cast<int*>(h_tot + 3 * kRefStride);

    const bool rgbOnly = opts->rgb_only != 0;
    const float icpWeight = opts->icp_weight;
    const bool icp = !rgbOnly && icpWeight > 0;
    const bool rgb = rgbOnly || icpWeight < 100;
    cf_track_stats st;
    memset(&st, 0, sizeof(st));
    st.cull_box[0] = 0; st.cull_box[1] = 0; st.cull_box[2] = W - 1; st.cull_box[3] = H - 1;

    float Rprev[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    float tprev[3] = {pose[3], pose[7], pose[11]};
    float Rcurr[9], tcurr[3];
    memcpy(Rcurr, Rprev, sizeof(Rcurr)); memcpy(tcurr, tprev, sizeof(tcurr));

    double resultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (opts->so3) {   // RGBDOdometry.cpp:239-310
        const int L = 2, cols = W >> L, rows = H >> L;
        float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double K[9], Kinv[9];
        const cf_cam il = level_cam(L);
        k_matrix(il.fx, il.fy, il.cx, il.cy, K);
        float lastError = FLT_MAX / 2, lastCount = FLT_MAX / 2;
        double lastResultR[9]; memcpy(lastResultR, resultR, sizeof(resultR));
        for (int it = 0; it < 10; it++) {
            double KR[9], Hm[9];
            inv33<double>(K, Kinv);
            mul33<double>(K, resultR, KR); mul33<double>(KR, Kinv, Hm);
            RefSo3Args a{};
            a.lastImage = od->lastNextImage[L]; a.nextImage = od->nextImage[L]; a.cols = cols; a.rows = rows;
            for (int k = 0; k < 9; k++) { a.B.m[k] = (float)Hm[k]; a.Ki.m[k] = (float)Kinv[k]; a.krlr[k] = (float)KR[k]; }
            ref_so3_kernel<<<kRefSo3Blocks, kRefSo3Threads, 0, s>>>(a, d_part);
            ref_reduce_sum_kernel<11><<<1, kRefStage2Threads, 0, s>>>(d_part, d_tot + 2 * kRefStride, kRefSo3Blocks);
            REFCHK(hipGetLastError());
            REFCHK(hipMemcpyAsync(h_tot + 2 * kRefStride, d_tot + 2 * kRefStride, sizeof(float) * 11, hipMemcpyDeviceToHost, s));
            REFCHK(hipStreamSynchronize(s));
            const float* o = h_tot + 2 * kRefStride;
            float jtj[9], jtr[3], residual[2];
            int shift = 0;
            for (int i = 0; i < 3; ++i)
                for (int j = i; j < 4; ++j) {   // reduce.cu:1161-1172
                    const float value = o[shift++];
                    if (j == 3) jtr[i] = value; else jtj[j * 3 + i] = jtj[i * 3 + j] = value;
                }
            residual[0] = o[9]; residual[1] = o[10];
            st.so3_iterations = it + 1;
            st.last_so3_error = sqrtf(residual[0]) / residual[1];
            st.last_so3_count = residual[1];
            if (st.last_so3_error < lastError && (double)fabsf(lastError - st.last_so3_count) < 0.001) break;   // (sic: error against count, :285)
            else if ((double)st.last_so3_error > (double)lastError + 0.001) {
                st.last_so3_error = lastError; st.last_so3_count = lastCount;
                memcpy(resultR, lastResultR, sizeof(resultR));
                break;
            }
            lastError = st.last_so3_error; lastCount = st.last_so3_count;
            memcpy(lastResultR, resultR, sizeof(resultR));
            float delta[3];
            ldlt_solve<float, 3>(jtj, jtr, delta, FLT_MAX);
            const double dd[3] = {delta[0], delta[1], delta[2]};
            double rotUpdate[9];
            rodrigues(dd, rotUpdate);
            float ru[9], nr[9];
            for (int k = 0; k < 9; k++) ru[k] = (float)rotUpdate[k];
            mul33<float>(ru, R_lr, nr);
            memcpy(R_lr, nr, sizeof(nr));
            for (int k = 0; k < 9; k++) resultR[k] = R_lr[k];
        }
    }

    int iterations[3];
    iterations[0] = opts->fast_odom ? 3 : 10;
    iterations[1] = opts->pyramid ? 5 : 0;
    iterations[2] = opts->pyramid ? 4 : 0;
    float Rprev_inv[9];
    inv33<float>(Rprev, Rprev_inv);
    double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (opts->so3)
        for (int x = 0; x < 3; x++)
            for (int y = 0; y < 3; y++) resultRt[x * 4 + y] = resultR[x * 3 + y];
    float residual[2] = {0, 0};
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    st.last_rgb_error = 0;

    for (int i = 2; i >= 0; i--) {
        const int cols = W >> i, rows = H >> i;
        const cf_cam il = level_cam(i);
        double K[9], Kinv[9];
        k_matrix(il.fx, il.fy, il.cx, il.cy, K);
        st.last_rgb_error = FLT_MAX;   // lastRGBError = max at the top of every level (:343)
        for (int j = 0; j < iterations[i]; j++) {
            double Rt[16];
            inv44(resultRt, Rt);
            const double R[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
            double tmp[9], KRK[9];
            inv33<double>(K, Kinv);
            mul33<double>(K, R, tmp); mul33<double>(tmp, Kinv, KRK);
            const double tv[3] = {Rt[3], Rt[7], Rt[11]};
            double Kt[3];
            mul33v<double>(K, tv, Kt);

            int sigma = 0, rgbSize = 0;
            if (rgb) {
                RefResArgs a{};
                a.minScale = (float)(pow((double)od->minGrad[i], 2.0) / pow((double)od->sobelScale, 2.0));
                a.dIdx = od->dIdx[i]; a.dIdy = od->dIdy[i]; a.lastDepth = od->lastDepth[i]; a.nextDepth = od->next_depth(i);
                a.lastImage = od->lastImage[i]; a.nextImage = od->nextImage[i]; a.corres = od->corres[i]; a.maxDepthDelta = od->maxDepthDeltaRGB;
                for (int k = 0; k < 9; k++) a.krk[k] = (float)KRK[k];
                for (int k = 0; k < 3; k++) a.kt[k] = (float)Kt[k];
                a.cols = cols; a.rows = rows; a.sums = d_cnt;
                REFCHK(hipMemsetAsync(d_cnt, 0, 2 * sizeof(int), s));
                ref_rgb_residual_kernel<<<kRefResBlocks, kRefResThreads, 0, s>>>(a);
                REFCHK(hipMemcpyAsync(h_cnt, d_cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
            }
            const bool want_err = icp && i == 0 && j == iterations[i] - 1 && err_surface;
            if (icp) {
                RefIcpArgs a{};
                memcpy(a.Rcurr.m, Rcurr, 36); memcpy(a.Rprev_inv.m, Rprev_inv, 36);
                a.tcurr = f3{tcurr[0], tcurr[1], tcurr[2]}; a.tprev = f3{tprev[0], tprev[1], tprev[2]};
                a.intr = il;
                a.vc = od->ext_vmap_curr[i] ? od->ext_vmap_curr[i] : od->vmap_curr[i];
                a.nc = od->ext_nmap_curr[i] ? od->ext_nmap_curr[i] : od->nmap_curr[i];
                a.vp = od->vmap_g_prev[i]; a.np = od->nmap_g_prev[i];
                a.distThres = od->distThres; a.angleThres = od->angleThres; a.cols = cols; a.rows = rows;
                a.err = want_err ? err_surface : nullptr;
                ref_icp_kernel<<<kRefIcpBlocks, kRefIcpThreads, 0, s>>>(a, d_part);
                ref_reduce_sum_kernel<29><<<1, kRefStage2Threads, 0, s>>>(d_part, d_tot, kRefIcpBlocks);
                REFCHK(hipMemcpyAsync(h_tot, d_tot, sizeof(float) * 29, hipMemcpyDeviceToHost, s));
            }
            REFCHK(hipGetLastError());
            REFCHK(hipStreamSynchronize(s));
            if (rgb) { rgbSize = h_cnt[0]; sigma = h_cnt[1]; }

            const float tmpError = (float)(sqrt((double)sigma) / rgbSize);
            float sigmaVal = (tmpError == 0) ? 1 : (float)rgbSize;   // (sic) the COUNT, :374
            if (rgbOnly && tmpError > st.last_rgb_error) break;
            st.last_rgb_error = tmpError; st.last_rgb_count = (float)rgbSize;
            if (rgbOnly) sigmaVal = -1;

            float A_icp[36], b_icp[6], A_rgbd[36], b_rgbd[6];
            memset(A_icp, 0, sizeof(A_icp)); memset(b_icp, 0, sizeof(b_icp));
            memset(A_rgbd, 0, sizeof(A_rgbd)); memset(b_rgbd, 0, sizeof(b_rgbd));
            if (icp) unpack29(h_tot, A_icp, b_icp, residual);
            st.last_icp_error = sqrtf(residual[0]) / residual[1];
            st.last_icp_count = residual[1];
            if (rgb) {
                RefRgbArgs a{};
                a.corres = od->corres[i]; a.sigma = sigmaVal; a.cloud = od->cloud[i]; a.fx = il.fx; a.fy = il.fy;
                a.dIdx = od->dIdx[i]; a.dIdy = od->dIdy[i]; a.sobelScale = od->sobelScale; a.cols = cols; a.rows = rows;
                ref_rgb_step_kernel<<<kRefRgbBlocks, kRefRgbThreads, 0, s>>>(a, d_part);
                ref_reduce_sum_kernel<29><<<1, kRefStage2Threads, 0, s>>>(d_part, d_tot + kRefStride, kRefRgbBlocks);
                REFCHK(hipGetLastError());
                REFCHK(hipMemcpyAsync(h_tot + kRefStride, d_tot + kRefStride, sizeof(float) * 29, hipMemcpyDeviceToHost, s));
                REFCHK(hipStreamSynchronize(s));
                unpack29(h_tot + kRefStride, A_rgbd, b_rgbd, nullptr);
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
            ldlt_solve<double, 6>(lastA, lastb, result, DBL_MAX);
            memcpy(st.lastA, lastA, sizeof(lastA)); memcpy(st.lastb, lastb, sizeof(lastb));

            // OdometryProvider::computeUpdateSE3 (OdometryProvider.h:69-89)
            double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Rr[9], nrt[16];
            const double rvec[3] = {result[3], result[4], result[5]};
            rodrigues(rvec, Rr);
            for (int r = 0; r < 3; r++) { upd[r * 4 + 0] = Rr[r * 3 + 0]; upd[r * 4 + 1] = Rr[r * 3 + 1]; upd[r * 4 + 2] = Rr[r * 3 + 2]; upd[r * 4 + 3] = result[r]; }
            mul44(upd, resultRt, nrt);
            memcpy(resultRt, nrt, sizeof(nrt));
            // rgbOdom.setIdentity(); rgbOdom.rotate(rotation.cast<float>()): linear = Identity * R -- a product like any other
            float Rf[9], Ro[9], to[3];
            for (int r = 0; r < 3; r++) { Rf[r * 3 + 0] = (float)resultRt[r * 4 + 0]; Rf[r * 3 + 1] = (float)resultRt[r * 4 + 1]; Rf[r * 3 + 2] = (float)resultRt[r * 4 + 2]; to[r] = (float)resultRt[r * 4 + 3]; }
            mul33<float>(ident, Rf, Ro);
            // currentT.setIdentity(); currentT.rotate(Rprev); translation = tprev; currentT = currentT * rgbOdom.inverse() (RGBDOdometry.cpp:452-460)
            float Rp[9];
            mul33<float>(ident, Rprev, Rp);
            const float Rinv[9] = {Ro[0], Ro[3], Ro[6], Ro[1], Ro[4], Ro[7], Ro[2], Ro[5], Ro[8]};
            float tinv[3], t2[3];
            mul33v<float>(Rinv, to, tinv);
            for (int r = 0; r < 3; r++) tinv[r] = -tinv[r];
            mul33<float>(Rp, Rinv, Rcurr);
            mul33v<float>(Rp, tinv, t2);
            for (int r = 0; r < 3; r++) tcurr[r] = t2[r] + tprev[r];
        }
    }
    if (rgb) {   // divergence guard :464-467
        const float d[3] = {tcurr[0] - tprev[0], tcurr[1] - tprev[1], tcurr[2] - tprev[2]};
        if ((double)sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) > 0.3) { memcpy(Rcurr, Rprev, 36); memcpy(tcurr, tprev, 12); }
    }
    // results where the default tracker's last solve leaves them: the pinned host state and the device state
    OdomDev* h = od->h_state;
    memcpy(h->Rprev, Rprev, 36); memcpy(h->tprev, tprev, 12); memcpy(h->Rprev_inv, Rprev_inv, 36);
    memcpy(h->Rcurr, Rcurr, 36); memcpy(h->tcurr, tcurr, 12);
    {   // the inverse of the final pose, where the default tracker's last solve leaves it (OdomDev::pose_inv: cf_models_preindex reads it)
        float pose[16] = {Rcurr[0], Rcurr[1], Rcurr[2], tcurr[0], Rcurr[3], Rcurr[4], Rcurr[5], tcurr[1], Rcurr[6], Rcurr[7], Rcurr[8], tcurr[2], 0, 0, 0, 1};
        inv44f(pose, h->pose_inv);
    }
    memcpy(h->resultRt, resultRt, sizeof(resultRt));
    h->stats = st;
    refresh_hot(h);
    REFCHK(hipMemcpyAsync(od->d_state, h, sizeof(OdomDev), hipMemcpyHostToDevice, s));
    return CF_OK;
}

// ---- the single reductions (icpStep / rgbStep / so3Step of the C-ABI under cf_set_icp_arith 2): the reference's launch pair and its
// read-back (reduce.cu:427-499, 628-687, 1114-1176) ---------------------------------------------------------------------------------
int ref_icp_step(cf_ctx* ctx, const float Rcurr[9], const float tcurr[3], const float* vc, const float* nc, const float Rprev_inv[9], const float tprev[3],
                 cf_cam intr, const float* vp, const float* np, float dist_thres, float angle_thres, int cols, int rows, float* err, float out29[29])
{
    if (int r = ref_scratch(ctx)) return r;
    hipStream_t s = ctx->stream;
    float* const d_part = ctx->d_ref;
    float* const d_tot = ctx->d_ref + (size_t)kRefIcpBlocks * kRefStride;
    RefIcpArgs a{};
    memcpy(a.Rcurr.m, Rcurr, 36); memcpy(a.Rprev_inv.m, Rprev_inv, 36);
    a.tcurr = f3{tcurr[0], tcurr[1], tcurr[2]}; a.tprev = f3{tprev[0], tprev[1], tprev[2]};
    a.intr = intr; a.vc = vc; a.nc = nc; a.vp = vp; a.np = np; a.distThres = dist_thres; a.angleThres = angle_thres; a.cols = cols; a.rows = rows; a.err = err;
    ref_icp_kernel<<<kRefIcpBlocks, kRefIcpThreads, 0, s>>>(a, d_part);
    ref_reduce_sum_kernel<29><<<1, kRefStage2Threads, 0, s>>>(d_part, d_tot, kRefIcpBlocks);
    REFCHK(hipGetLastError());
    REFCHK(hipMemcpyAsync(ctx->h_ref, d_tot, sizeof(float) * 29, hipMemcpyDeviceToHost, s));
    REFCHK(hipStreamSynchronize(s));
    memcpy(out29, ctx->h_ref, sizeof(float) * 29);
    return CF_OK;
}
int ref_rgb_step(cf_ctx* ctx, const cf_dataterm* corres, float sigma, const float* cloud3, float fx, float fy, const int16_t* dIdx, const int16_t* dIdy,
                 float sobel_scale, int cols, int rows, float out29[29])
{
    if (int r = ref_scratch(ctx)) return r;
    hipStream_t s = ctx->stream;
    float* const d_part = ctx->d_ref;
    float* const d_tot = ctx->d_ref + (size_t)kRefIcpBlocks * kRefStride;
    RefRgbArgs a{};
    a.corres = corres; a.sigma = sigma; a.cloud = cloud3; a.fx = fx; a.fy = fy; a.dIdx = dIdx; a.dIdy = dIdy; a.sobelScale = sobel_scale; a.cols = cols; a.rows = rows;
    ref_rgb_step_kernel<<<kRefRgbBlocks, kRefRgbThreads, 0, s>>>(a, d_part);
    ref_reduce_sum_kernel<29><<<1, kRefStage2Threads, 0, s>>>(d_part, d_tot, kRefRgbBlocks);
    REFCHK(hipGetLastError());
    REFCHK(hipMemcpyAsync(ctx->h_ref, d_tot, sizeof(float) * 29, hipMemcpyDeviceToHost, s));
    REFCHK(hipStreamSynchronize(s));
    memcpy(out29, ctx->h_ref, sizeof(float) * 29);
    return CF_OK;
}
int ref_so3_step(cf_ctx* ctx, const uint8_t* last_image, const uint8_t* next_image, const float basis[9], const float kinv[9], const float krlr[9],
                 int cols, int rows, float out11[11])
{
    if (int r = ref_scratch(ctx)) return r;
    hipStream_t s = ctx->stream;
    float* const d_part = ctx->d_ref;
    float* const d_tot = ctx->d_ref + (size_t)kRefIcpBlocks * kRefStride;
    RefSo3Args a{};
    a.lastImage = last_image; a.nextImage = next_image; a.cols = cols; a.rows = rows;
    for (int k = 0; k < 9; k++) { a.B.m[k] = basis[k]; a.Ki.m[k] = kinv[k]; a.krlr[k] = krlr[k]; }
    ref_so3_kernel<<<kRefSo3Blocks, kRefSo3Threads, 0, s>>>(a, d_part);
    ref_reduce_sum_kernel<11><<<1, kRefStage2Threads, 0, s>>>(d_part, d_tot, kRefSo3Blocks);
    REFCHK(hipGetLastError());
    REFCHK(hipMemcpyAsync(ctx->h_ref, d_tot, sizeof(float) * 11, hipMemcpyDeviceToHost, s));
    REFCHK(hipStreamSynchronize(s));
    memcpy(out11, ctx->h_ref, sizeof(float) * 11);
    return CF_OK;
}

}  // namespace cf
