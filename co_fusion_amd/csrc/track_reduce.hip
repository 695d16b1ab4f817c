// track_reduce.hip -- the dense-tracking reductions and the device-resident Gauss-Newton loop.
//
// MI355X-native replacement of Core/Cuda/reduce.cu (icpStep / computeRgbResidual / rgbStep /
// so3Step) and of the host loop RGBDOdometry::getIncrementalTransformation
// (Core/Utils/RGBDOdometry.cpp:217-477).
//
// Design (DESIGN.md "tracking"):
//  * The reference does, per GN iteration, 3 x (kernel -> 1-block reduceSum -> cudaDeviceSynchronize
//    -> D2H) and solves the 6x6 system on the host: <= 67 host round trips per model per frame.
//    Here the pose, the 6x6 solve (f64 LDL^T) and the SE3 update live on the device; one frame's
//    whole schedule (SO3 pre-alignment + 4/5/10 iterations) is enqueued without any host wait,
//    and all active models advance in lock-step inside the same launches (blockIdx.y = model).
//  * Reductions are wave64 butterflies over *integer* (fixed-point) partial sums followed by
//    grouped 64-bit atomics: exact, order independent, identical for every launch shape / GPU count.
//  * ICP is HBM/L2-bound streaming (48 B/pixel: 6 coalesced plane loads + 6 gathered loads);
//    blockIdx -> pixel-range mapping is XCD-aware: workgroup b runs on XCD b%8, and XCD x owns
//    the x-th horizontal band of the image, so the gathered model-map rows stay in that XCD's
//    4 MiB L2 across the 19 iterations of a frame.
#include "cf_device.h"
#include "cf_kernels.h"
#include "track_prep_dev.h"

namespace cf {

__device__ __forceinline__ cf_cam cam_level(cf_cam c, int level)
{  // CameraModel::operator(), types.cuh:94-98
    const int div = 1 << level;
    return cf_cam{c.fx / div, c.fy / div, c.cx / div, c.cy / div};
}

__device__ __forceinline__ int idiv(int n, IDiv d) { return (int)(__umulhi((unsigned)n, d.M) >> d.s); }  // n / cols, cf_kernels.h: make_idiv

__device__ __forceinline__ float clamp_row(float v, float lim) { return fminf(fmaxf(v, -lim), lim); }

// acc[k] += RNE(row_i*row_j*2^F) for the 27 upper-triangular SE3 products + residual
template <int F>
__device__ __forceinline__ void se3_accumulate(const float (&row)[7], unsigned long long (&acc)[32])
{
    constexpr float lim = (float)(1 << ((50 - F) / 2));
    constexpr float scale = (F == 32) ? 4294967296.0f : (float)(1u << (F & 31));
    double r[7], rs[6];
#pragma unroll
    for (int i = 0; i < 7; i++) r[i] = (double)clamp_row(row[i], lim);
#pragma unroll
    for (int i = 0; i < 6; i++) rs[i] = (double)(clamp_row(row[i], lim) * scale);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 7; j++)
            if (j >= i)  // static index: k(i,j) = 7i - i(i-1)/2 + (j-i)
                acc[7 * i - (i * (i - 1)) / 2 + (j - i)] += (unsigned long long)__double_as_longlong(fma(rs[i], r[j], kMagic));
    acc[27] += (unsigned long long)__double_as_longlong(fma(r[6] * (double)scale, r[6], kMagic));
}

// same with a wave-uniform run-time scale 2^F (RGB step: F follows sigma, rgb_fix_bits)
__device__ __forceinline__ void se3_accumulate_dyn(const float (&row)[7], unsigned long long (&acc)[32], float lim, float scale)
{
    double r[7], rs[6];
#pragma unroll
    for (int i = 0; i < 7; i++) r[i] = (double)clamp_row(row[i], lim);
#pragma unroll
    for (int i = 0; i < 6; i++) rs[i] = (double)(clamp_row(row[i], lim) * scale);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 7; j++)
            if (j >= i)
                acc[7 * i - (i * (i - 1)) / 2 + (j - i)] += (unsigned long long)__double_as_longlong(fma(rs[i], r[j], kMagic));
    acc[27] += (unsigned long long)__double_as_longlong(fma(r[6] * (double)scale, r[6], kMagic));
}

// ------------------------------------------------------------------------------------------------
// Product form of the ICP sums, round 5: the lanes add their products into the WORKGROUP's accumulators in LDS -- word k of lane l at
// [k][l], 64-bit LDS atomics without return, no bank conflicts -- and the butterfly runs ONCE per workgroup over what all waves (and
// all runs of a wave) left there, split over four waves (seven or eight words each).  Until then every wave ran the 32 x u64 butterfly
// on its own registers: 190 of the ~510 VALU instructions of a wave that the launch is bound by, and the 64 accumulator registers
// that set its occupancy.  Only lanes with a correspondence add; each adds the magic number's bits once per word, so the count of
// correspondences (word 28: a counter of its own, one LDS add per wave) times those bits comes off behind the butterfly.  Integer
// sums: who adds what in which order does not change a bit.
constexpr int kIcpLdsWords = 28;
__shared__ unsigned long long s_icp_acc[kIcpLdsWords][64];
__shared__ unsigned s_icp_found;
__device__ __forceinline__ void lds_add_u64(unsigned long long* p, unsigned long long v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void icp_lds_zero()   // (workgroup-uniform; the barrier behind it is the caller's)
{
    for (int i = threadIdx.x; i < kIcpLdsWords * 64; i += blockDim.x) (&s_icp_acc[0][0])[i] = 0;
    if (threadIdx.x == 0) s_icp_found = 0;
}
template <int F>
__device__ __forceinline__ void se3_accumulate_lds(const float (&row)[7], int lane)
{
    constexpr float lim = (float)(1 << ((50 - F) / 2));
    constexpr float scale = (F == 32) ? 4294967296.0f : (float)(1u << (F & 31));
    unsigned long long* col = &s_icp_acc[0][lane];
    double r[7];
#pragma unroll
    for (int i = 0; i < 7; i++) r[i] = (double)clamp_row(row[i], lim);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double rs = (double)(clamp_row(row[i], lim) * scale);
#pragma unroll
        for (int j = 0; j < 7; j++)
            if (j >= i) lds_add_u64(col + 64 * (7 * i - (i * (i - 1)) / 2 + (j - i)), (unsigned long long)__double_as_longlong(fma(rs, r[j], kMagic)));
    }
    lds_add_u64(col + 64 * 27, (unsigned long long)__double_as_longlong(fma(r[6] * (double)scale, r[6], kMagic)));
}
// ... and the workgroup's totals to its accumulator group in memory (all waves call it: the barrier is inside).  Wave g of the first
// four reduces words 8 g .. 8 g + 7 (fewer than four waves: they take turns); lanes 8 i of a wave end with word 8 g + i.
__device__ __forceinline__ void icp_lds_commit(int lane, int wave, int nwaves, unsigned long long* __restrict__ dst /* [32] of this group */)
{
    __syncthreads();
    const unsigned found = s_icp_found;
    if (found == 0) return;
    for (int g = wave; g < 4; g += nwaves) {
        unsigned long long acc[8];
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (g < 3 || k < 4) ? s_icp_acc[(g < 3 || k < 4) ? 8 * g + k : 0][lane] : 0ull;
        unsigned long long v = wave_reduce8_u64(acc, lane);
        const int w = 8 * g + (lane >> 3);
        if (w < 28) v -= (unsigned long long)found * kMagicBits;
        else v = w == 28 ? (unsigned long long)found : 0ull;
        if ((lane & 7) == 0 && v != 0) atomicAdd(&dst[w], v);
    }
}

// ------------------------------------------------------------------------------------------------
// cross-wave combine + grouped atomics.  v = wave total of word ((lane>>1)&31) (wave_reduce32_u64).
// XCD_LOCAL: the atomics are performed in the L2 of THIS XCD (workgroup scope) -- for sums that only workgroups of the same XCD add to and
// read back (rgb_step_solve_kernel).
template <int MAXW, bool XCD_LOCAL = false>
__device__ __forceinline__ void block_commit32(unsigned long long v, int lane, int wave, int nwaves,
                                               unsigned long long* __restrict__ dst /* [32] of this group */)
{
    __shared__ unsigned long long lds[MAXW][32];
    if ((lane & 1) == 0) lds[wave][lane >> 1] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        unsigned long long t = 0;
        for (int w = 0; w < nwaves; w++) t += lds[w][threadIdx.x];
        if (t != 0) {
            if constexpr (XCD_LOCAL) __hip_atomic_fetch_add(&dst[threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicAdd(&dst[threadIdx.x], t);
        }
    }
}

// ... the workgroup's 32 totals (4 waves) stored as one row: no atomic, no read-modify-write in the L2 (rgb_step_solve_kernel)
__device__ __forceinline__ void block_store32(unsigned long long v, int lane, int wave, unsigned long long* __restrict__ row /* [32] */)
{
    __shared__ unsigned long long lds[4][32];
    if ((lane & 1) == 0) lds[wave][lane >> 1] = v;
    __syncthreads();
    if (threadIdx.x < 32) row[threadIdx.x] = lds[0][threadIdx.x] + lds[1][threadIdx.x] + lds[2][threadIdx.x] + lds[3][threadIdx.x];
}

// ================================================================================================
// ICP:  ICPReduction::search + getProducts, reduce.cu:283-394
// ================================================================================================
// The two gates of the correspondence test compare square roots with constants (reduce.cu:321-325:
// sine < angleThres, dist <= distThres).  sqrtf is correctly rounded and monotonic, so each gate is decided
// exactly by comparing the radicand with a precomputed f32 bound (IcpArgs::angleSqLt / distSqLe, sqrt_gate_* below):
// no square root per pixel unless the error surface (which stores dist itself) is requested.
struct IcpProj { f3 vcurr_g; int g; int inb; int ux, uy; };

__device__ __forceinline__ IcpProj icp_project(const m33& Rcurr, const f3& tcurr, const m33& Rprev_inv, const f3& tprev, const cf_cam& intr,
                                               int cols, int rows, f3 vcurr)
{
    IcpProj o;
    o.vcurr_g = mul(Rcurr, vcurr) + tcurr;
    const f3 vcurr_cp = mul(Rprev_inv, o.vcurr_g - tprev);
    const int ux = f2i_rn(vcurr_cp.x * intr.fx / vcurr_cp.z + intr.cx);
    const int uy = f2i_rn(vcurr_cp.y * intr.fy / vcurr_cp.z + intr.cy);
    o.inb = !(ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcurr_cp.z < 0);
    o.g = o.inb ? uy * cols + ux : 0;
    o.ux = o.inb ? ux : 0; o.uy = o.inb ? uy : 0;
    return o;
}

// ------------------------------------------------------------------------------------------------
// Screen-box culling.  A pixel of the current frame finds a correspondence only if its vertex, taken into the camera the prediction was
// rendered from (vcurr_cp = Rprev^-1 (Rcurr vcurr + tcurr - tprev)), projects onto a VALID pixel of the prediction -- i.e. lies inside
// that pixel's pyramid -- and is within distThres of the model vertex there (reduce.cu:321-325), hence at a depth within distThres of it.
// All such vertices lie in one frustum piece of the prediction camera: the pixel rectangle of the valid predicted vertices (lo / hi [0..1],
// level-0 pixels; model_maps_tiled_body) between the depths lo[2] - distThres and hi[2] + distThres.  Its eight corners, taken into the
// current camera and projected, bound the pixels that can contribute (a projective map takes the convex piece into the convex hull of
// the corners' images): everything outside adds exact zeros and is skipped before it loads anything.
// Conservative: the rectangle is widened by 3 pixels (a level-l pixel of the model maps is valid only if its 2^l x 2^l level-0 sources
// are, and its pyramid overhangs them by 2^(l-1) level-0 pixels; + rounding of the per-pixel f32 projection), the depths by 1 % + 1 mm on
// top of distThres, the projected rectangle by 3 pixels, and a near plane that is not clearly in front of either camera (or anything
// not finite) gives the whole image.  Called by a whole wave; the result is valid in every lane.
// Rb / tb: pose of the prediction camera (OdomDev::box_R / box_t).
__device__ __forceinline__ void screen_box(const float* lo, const float* hi, const float* Rb, const float* tb, const float* Rcurr, const float* tcurr,
                                           cf_cam intr, float distThres, int W, int H, int lane, int (&out)[4], float (&zout)[2])
{
    const float finf = __int_as_float(0x7f800000);
    zout[0] = -finf; zout[1] = finf;
    if (!(lo[0] <= hi[0])) { out[0] = 1; out[1] = 1; out[2] = 0; out[3] = 0; return; }  // no predicted vertex: nothing can match
    const float m = distThres * 1.01f + 1e-3f;
    const float px = (lane & 1) ? hi[0] + 3.f : lo[0] - 3.f, py = (lane & 2) ? hi[1] + 3.f : lo[1] - 3.f;
    const float znear = lo[2] - m, pz = (lane & 4) ? hi[2] + m : znear;
    const float cx_ = (px - intr.cx) / intr.fx * pz, cy_ = (py - intr.cy) / intr.fy * pz;
    // into the global frame, then into the current camera: Rcurr^T (Rcurr is a rotation up to f32 rounding)
    const float dx = (Rb[0] * cx_ + Rb[1] * cy_ + Rb[2] * pz + tb[0]) - tcurr[0];
    const float dy = (Rb[3] * cx_ + Rb[4] * cy_ + Rb[5] * pz + tb[1]) - tcurr[1];
    const float dz = (Rb[6] * cx_ + Rb[7] * cy_ + Rb[8] * pz + tb[2]) - tcurr[2];
    const float xc = Rcurr[0] * dx + Rcurr[3] * dy + Rcurr[6] * dz;
    const float yc = Rcurr[1] * dx + Rcurr[4] * dy + Rcurr[7] * dz;
    const float zc = Rcurr[2] * dx + Rcurr[5] * dy + Rcurr[8] * dz;
    const float u = intr.fx * xc / zc + intr.cx, v = intr.fy * yc / zc + intr.cy;
    const bool bad = !(znear > 0.05f) || !(zc > 0.05f) || !is_finite(u) || !is_finite(v);
    float u0 = u, u1 = u, v0 = v, v1 = v, z0 = zc, z1 = zc;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        u0 = fminf(u0, __shfl_xor(u0, o, 64)); u1 = fmaxf(u1, __shfl_xor(u1, o, 64));
        v0 = fminf(v0, __shfl_xor(v0, o, 64)); v1 = fmaxf(v1, __shfl_xor(v1, o, 64));
        z0 = fminf(z0, __shfl_xor(z0, o, 64)); z1 = fmaxf(z1, __shfl_xor(z1, o, 64));
    }
    if (__any(bad)) { out[0] = 0; out[1] = 0; out[2] = W - 1; out[3] = H - 1; return; }
    // the depth (z in the current camera) of a matching vertex lies between the extreme corners: a linear map of a box
    zout[0] = z0 - (1e-3f + 1e-3f * fabsf(z0)); zout[1] = z1 + (1e-3f + 1e-3f * fabsf(z1));
    const float fw = (float)(W + 16), fh = (float)(H + 16);
    out[0] = (int)floorf(fminf(fmaxf(u0, -16.f), fw)) - 3; out[1] = (int)floorf(fminf(fmaxf(v0, -16.f), fh)) - 3;
    out[2] = (int)ceilf(fminf(fmaxf(u1, -16.f), fw)) + 3; out[3] = (int)ceilf(fminf(fmaxf(v1, -16.f), fh)) + 3;
}

template <int PPT> struct VecF;
template <> struct VecF<1> { using T = float; };
template <> struct VecF<2> { using T = float2; };
template <> struct VecF<4> { using T = float4; };

template <int PPT>
__device__ __forceinline__ void load_vec(const float* p, float (&o)[PPT])
{
    using V = typename VecF<PPT>::T;
    const V v = *reinterpret_cast<const V*>(p);
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int i = 0; i < PPT; i++) o[i] = f[i];
}

// XCD-aware logical block id: hardware block b lands on XCD b%8; give XCD x the x-th contiguous
// range of logical blocks (= a horizontal band of the image).
__device__ __forceinline__ int xcd_logical_block(int b, int nlog)
{
    const int per = (nlog + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

template <bool COMPACT> __device__ void rgb_residual_body(const RgbArgs& ra, int model, int blk, int nblk);

// One launch per Gauss-Newton iteration carries BOTH pose-dependent streaming passes, which are independent
// of each other: workgroups [0, n_icp_blocks) run the ICP reduction, the rest the RGB residual pass
// (rgb_residual_body).  At 640x480 either pass alone is launch/latency-bound (5-8 us for 8-15 MB); sharing
// a launch overlaps their ramp-up, gather latency and atomics tail.
//
// Kernel arguments by value: every pointer the kernel dereferences arrives in the kernarg segment
// (scalar loads, global-address-space vector loads, no pointer chasing through device structs).
// Only the pose/flags, which the solve updates on the device every iteration, are read from
// memory -- as scalar loads issued in parallel with the first coalesced map loads.
//
// VALU budget (the kernel is VALU-bound once several models share a launch): the lanes' products go to the workgroup's accumulators in
// LDS and ONE butterfly per workgroup reduces them (se3_accumulate_lds / icp_lds_commit; until round 5 every wave ran a 32 x u64
// butterfly on its registers); a wave whose pixels cannot produce a correspondence (projection out of view, model map empty
// there -- the common case for object models, which cover a small part of the image) leaves after the projection.
static_assert(sizeof(IcpArgs) + sizeof(RgbArgs) + 64 <= 4096, "the kernel-argument segment holds 4 KB: lower kMaxBatch");  // (rgb_slot_step_kernel takes both as well)
//
// GRAM (cf_set_icp_arith 1): the accumulation and the butterfly are replaced by the matrix cores -- the rows are rounded to integers,
// staged through LDS as signed 8-bit limbs and contracted over the wave's pixels by v_mfma_i32_32x32x32_i8 (cf_device.h: gram_*);
// dynamic LDS = kGramWaveDwords * 4 bytes per wave (gram_block_commit adds the waves' tiles and recombines the limbs).  A different rounding specification (ORC_ICP_ARITH_GRAM in the oracle).
// Timing ablations of the launch (CF_ICP_REPLAY, DESIGN-NOTES): bits 8.. of IcpArgs::flags switch parts of the kernel off.  They exist in
// a diagnostics build only (make ABLATE=1 -> -DCF_ABLATE); in the production kernel ABL() is the constant 0 and the tests vanish.
#ifdef CF_ABLATE
#define ABL(bits) ((args.flags >> 8) & (bits))
#else
#define ABL(bits) 0
#endif
extern __shared__ int gram_lds[];
// One run of pixels of one model: the 6 plane loads, projection, gather, gates, rows, accumulation and the wave butterfly.
// i0 = this lane's first pixel, in_range = the lane's pixels take part.  Returns false when the tracker has nothing to do at this level
// (workgroup-uniform: the caller leaves).  Product form: the sums are left in the workgroup's LDS accumulators (icp_lds_commit); Gram form: gram_has.
// The tracker state is written by the solve kernel of the PREVIOUS launch and only read here: through the constant address space its
// (wave-uniform) loads are scalar loads whatever the compiler can prove about the stores around them -- with the run loop in the kernel
// it fell back to per-lane vector loads of the pose, 44 more VGPRs and three waves of occupancy less.
typedef const __attribute__((address_space(4))) OdomDev* StatePtr;
// the hot state (cf_kernels.h: GnHot), one 64-byte line per s_load_dwordx16; hot_pin() keeps the loads of a clause together in front of
// ONE wait (without it the compiler sinks each load to its first use: a chain of dependent round trips to memory again)
typedef int hot16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ hot16 hot_line(StatePtr st, int line)
{
    return *(reinterpret_cast<const __attribute__((address_space(4))) hot16*>(&st->hot) + line);
}
__device__ __forceinline__ void hot_pin(const hot16& a) { asm volatile("" :: "s"(a)); }
__device__ __forceinline__ void hot_pin(const hot16& a, const hot16& b) { asm volatile("" :: "s"(a), "s"(b)); }
// (by value through __int_as_float: __builtin_bit_cast(float, l[k]) on a vector ELEMENT reads element 0 with this compiler -- ROCm 7.2 clang)
__device__ __forceinline__ float hot_f(const hot16& l, int k) { const int v = l[k]; return __int_as_float(v); }
struct IcpHot {   // lines 0 + 1
    int icp, level_done; float zlo, zhi; m33 Rcurr, Rprev_inv; f3 tcurr, tprev; int box[4];
};
__device__ __forceinline__ IcpHot icp_hot(StatePtr st)
{
    hot16 l0 = hot_line(st, 0), l1 = hot_line(st, 1);
    hot_pin(l0, l1);
    IcpHot h;
    h.icp = l0[0]; h.level_done = l0[1]; h.zlo = hot_f(l0, 2); h.zhi = hot_f(l0, 3);
#pragma unroll
    for (int k = 0; k < 9; k++) { h.Rcurr.m[k] = hot_f(l0, 4 + k); h.Rprev_inv.m[k] = hot_f(l1, k); }
    h.tcurr = f3{hot_f(l0, 13), hot_f(l0, 14), hot_f(l0, 15)};
    h.tprev = f3{hot_f(l1, 9), hot_f(l1, 10), hot_f(l1, 11)};
#pragma unroll
    for (int k = 0; k < 4; k++) h.box[k] = l1[12 + k];
    return h;
}
struct RgbHot { int rgb, rgbOnly, level_done; float krk[9], kt[3]; };   // line 2
__device__ __forceinline__ RgbHot rgb_hot(StatePtr st)
{
    hot16 l2 = hot_line(st, 2);
    hot_pin(l2);
    RgbHot h;
    h.rgb = l2[0]; h.rgbOnly = l2[1]; h.level_done = l2[2];
#pragma unroll
    for (int k = 0; k < 9; k++) h.krk[k] = hot_f(l2, 4 + k);
#pragma unroll
    for (int k = 0; k < 3; k++) h.kt[k] = hot_f(l2, 13 + k);
    return h;
}
static_assert(offsetof(GnHot, Rcurr) == 16 && offsetof(GnHot, tcurr) == 52 && offsetof(GnHot, Rprev_inv) == 64 && offsetof(GnHot, tprev) == 100 &&
              offsetof(GnHot, cull_box) == 112 && offsetof(GnHot, rgb) == 128 && offsetof(GnHot, krkInv) == 144 && offsetof(GnHot, kt) == 180, "icp_hot / rgb_hot spell the layout out");

template <int PPT, bool GRAM>
__device__ __forceinline__ bool icp_run(const IcpArgs& args, const IcpModelArgs& ma, StatePtr st, const IcpHot& pre, bool have_pre, int i0, bool in_range,
                                        bool whole, int band0, int band1, float* __restrict__ errs, int abl, int lane, int wave,
                                        unsigned long long& v, bool& gram_has, bool& done)
{
    const int cols = args.cols, rows = args.rows, N = cols * rows;
    const float* __restrict__ vc = ma.vc;
    const float* __restrict__ nc = ma.nc;
    const float* __restrict__ vp = ma.vp;
    const float* __restrict__ np = ma.np;
    float vx[PPT], vy[PPT], vz[PPT], nx[PPT], ny[PPT], nz[PPT];
#pragma unroll
    for (int p = 0; p < PPT; p++) { vx[p] = vy[p] = vz[p] = nx[p] = ny[p] = nz[p] = qnan(); }
    if (in_range) {  // the frame maps do not depend on the tracker state: issued before the state is looked at
        load_vec<PPT>(vc + i0, vx); load_vec<PPT>(vc + i0 + N, vy); load_vec<PPT>(vc + i0 + 2 * N, vz);
        load_vec<PPT>(nc + i0, nx); load_vec<PPT>(nc + i0 + N, ny); load_vec<PPT>(nc + i0 + 2 * N, nz);
    }
    IcpHot hs = pre;                              // (a caller that had to look at the box first hands the state in; otherwise its one
                                                  // scalar round trip runs beside the plane loads just issued)
    if (!have_pre) hs = icp_hot(st);
    if (!hs.icp || hs.level_done) return false;
    const m33 Rcurr = hs.Rcurr, Rprev_inv = hs.Rprev_inv;
    const f3 tcurr = hs.tcurr, tprev = hs.tprev;

    // projection + gather of the model maps
    IcpProj pr[PPT];
    f3 vprev[PPT], nprev[PPT];
    int cand = 0;
#pragma unroll
    for (int p = 0; p < PPT; p++) { pr[p].vcurr_g = f3{qnan(), qnan(), qnan()}; pr[p].g = 0; pr[p].inb = 0; pr[p].ux = pr[p].uy = 0; vprev[p] = pr[p].vcurr_g; nprev[p] = pr[p].vcurr_g; }
    if (__any(in_range))  // (a wave culled by the screen box skips the projection as well)
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        pr[p] = icp_project(Rcurr, tcurr, Rprev_inv, tprev, args.intr, cols, rows, f3{vx[p], vy[p], vz[p]});
        if (!in_range) pr[p].inb = 0;
        bool occupied = pr[p].inb != 0;
        if (occupied && ma.occ) {  // the model map is invalid (NaN) everywhere inside an empty 4x4 block: no gather needed
            const int tile = (pr[p].uy >> args.occ_shift) * args.occ_w + (pr[p].ux >> args.occ_shift);
            occupied = ma.occ[tile] != 0;
        }
        if (occupied) {
            const int g = pr[p].g;
            vprev[p] = f3{vp[g], vp[g + N], vp[g + 2 * N]};
            nprev[p] = f3{np[g], np[g + N], np[g + 2 * N]};
        }
        // necessary for a correspondence: in view, both normals valid, a finite model vertex
        if (whole && (i0 + p < band0 || i0 + p >= band1)) continue;  // outside this rank's band: error surface only
        cand |= (pr[p].inb && !is_nan(nx[p]) && !is_nan(nprev[p].x) && !is_nan(vprev[p].x)) ? 1 : 0;
    }
    const bool wave_cand = __any(cand) != 0;
    if (errs) {  // last level-0 iteration: the error surface stores dist for every pixel (0 if not finite / out of view)
#pragma unroll
        for (int p = 0; p < PPT; p++)
            if (in_range) {
                float err = 0.f;
                if (pr[p].inb) { const float dist = norm(vprev[p] - pr[p].vcurr_g); err = is_finite(dist) ? dist : 0.0f; }
                errs[i0 + p] = err;
            }
    }
    // A wave without a single candidate contributes exact zeros (all rows are zero): skip the rows, the accumulation and
    // the butterfly.  Object models cover a small part of the image, so most of their waves take this exit.
    if (wave_cand || abl) {
        float row[PPT][7];
        int fnd[PPT], any_found = 0;
#pragma unroll
        for (int p = 0; p < PPT; p++) {
#pragma unroll
            for (int k = 0; k < 7; k++) row[p][k] = 0.f;
            const f3 ncurr_g = mul(Rcurr, f3{nx[p], ny[p], nz[p]});
            const f3 dv = vprev[p] - pr[p].vcurr_g;
            const float dist2 = dot(dv, dv);
            const f3 cr0 = cross(ncurr_g, nprev[p]);
            const float sine2 = dot(cr0, cr0);
            fnd[p] = (pr[p].inb && sine2 < args.angleSqLt && dist2 <= args.distSqLe && !is_nan(nx[p]) && !is_nan(nprev[p].x)) ? 1 : 0;
            if (whole && (i0 + p < band0 || i0 + p >= band1)) fnd[p] = 0;
            any_found |= fnd[p];
            if (fnd[p]) {
                const f3 s_cp = mul(Rprev_inv, pr[p].vcurr_g - tprev);
                const f3 d_cp = mul(Rprev_inv, vprev[p] - tprev);
                const f3 n_cp = mul(Rprev_inv, nprev[p]);
                const f3 cr = cross(s_cp, n_cp);
                row[p][0] = n_cp.x; row[p][1] = n_cp.y; row[p][2] = n_cp.z;
                row[p][3] = cr.x; row[p][4] = cr.y; row[p][5] = cr.z;
                row[p][6] = dot(n_cp, s_cp - d_cp);
            }
        }
        if constexpr (GRAM) {
            if (__any(any_found) || abl != 0) {
                int* wl = gram_lds + wave * kGramWaveDwords;
                gram_v16i macc;
#pragma unroll
                for (int r = 0; r < 16; r++) macc[r] = 0;
#pragma unroll
                for (int p = 0; p < PPT; p++) {
#pragma unroll
                    for (int k = 0; k < 7; k++)
                        wl[k * kGramRowStride + lane] = (int)gram_limbs((int)rintf(clamp_row(row[p][k], kGramLim[k]) * (float)(1 << kGramBits[k])));
                    wl[7 * kGramRowStride + lane] = (int)gram_limbs(fnd[p]);
                    __builtin_amdgcn_wave_barrier();   // (one wave, LDS in program order: the reads below see every lane's dwords)
                    macc = gram_wave_mfma(wl, lane, macc);
                    __builtin_amdgcn_wave_barrier();
                }
                gram_wave_store(wl, macc, lane);   // (after the last read of the staging area; LDS runs in program order)
                gram_has = true;
            }
        } else
        if (__any(any_found) || abl != 0) {
#pragma unroll
            for (int p = 0; p < PPT; p++) {
                if (fnd[p] && !(abl & 1)) se3_accumulate_lds<kFixICP>(row[p], lane);
                const int nf = __popcll(__ballot(fnd[p] != 0));
                if (lane == 0 && nf) (void)__hip_atomic_fetch_add(&s_icp_found, (unsigned)nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (abl & 2) { done = true; return true; }
        }
    }
    return true;
}

// The ICP error surface (icpStep's optional output, reduce.cu:327-331 / RGBDOdometry.cpp:414-431: the distance between every pixel's
// vertex and the model vertex it projects onto, 0 when not finite / out of view) of the last level-0 iteration, one pixel per lane.
// Until round 4 that iteration's ICP pass wrote it for every tracker, which kept the launch from culling anything (23.8 us against
// 15.4 us for the nine iterations before it); now the culled trackers stay culled and THEIR surfaces are written by a launch of its own
// between that iteration's {ICP || residual} launch and its RGB step (icp_error_surface_kernel) -- the same pose, the expressions of
// icp_run, the same bits.  (Round 4 had these workgroups in the RGB step's launch; with the solve inside that launch --
// rgb_step_solve_kernel, cf_set_gn_mode 2 -- the pose would change under them.)  Unculled trackers write theirs in the ICP pass as before.
__device__ __forceinline__ void icp_error_surface_body(const IcpArgs& args, const IcpModelArgs& ma, int blk)
{
    float* __restrict__ errs = ma.err;
    if (!errs || !ma.cull) return;  // (an unculled tracker wrote its surface in the ICP pass it ran over the whole image anyway)
    StatePtr st = (StatePtr)ma.st;
    const int cols = args.cols, rows = args.rows, N = cols * rows;
    const int i = blk * (int)blockDim.x + (int)threadIdx.x;
    const bool in_range = i < N;
    float vx = qnan(), vy = qnan(), vz = qnan();
    if (in_range) { vx = ma.vc[i]; vy = ma.vc[i + N]; vz = ma.vc[i + 2 * N]; }
    const IcpHot hs = icp_hot(st);
    if (!hs.icp || hs.level_done) return;  // (icp_run leaves before it writes anything)
    const m33 Rcurr = hs.Rcurr, Rprev_inv = hs.Rprev_inv;
    const f3 tcurr = hs.tcurr, tprev = hs.tprev;
    if (!in_range) return;
    const IcpProj pr = icp_project(Rcurr, tcurr, Rprev_inv, tprev, args.intr, cols, rows, f3{vx, vy, vz});
    f3 vprev = {qnan(), qnan(), qnan()};
    bool occupied = pr.inb != 0;
    if (occupied && ma.occ) occupied = ma.occ[(pr.uy >> args.occ_shift) * args.occ_w + (pr.ux >> args.occ_shift)] != 0;
    if (occupied) vprev = f3{ma.vp[pr.g], ma.vp[pr.g + N], ma.vp[pr.g + 2 * N]};
    float err = 0.f;
    if (pr.inb) { const float dist = norm(vprev - pr.vcurr_g); err = is_finite(dist) ? dist : 0.0f; }
    errs[i] = err;
}

__global__ void __launch_bounds__(256) icp_error_surface_kernel(const IcpArgs args) { icp_error_surface_body(args, args.m[blockIdx.y], (int)blockIdx.x); }

// GRID.  One-dimensional, in SLOTS: a slot is the ICP reduction or the RGB residual pass of one model.  IcpArgs::slot_end holds the running
// totals of the workgroups, IcpArgs::slot_desc what every slot is; the launcher orders them longest work first (launch_icp_kernel_arith).
// Every slot starts at a multiple of 8, so hardware workgroup b and its slot-local index agree on the XCD (b % 8).
//  * A model that is not culled gets one workgroup per run of T * PPT pixels of its image (or row band), XCD x owning the x-th
//    horizontal band (xcd_logical_block).
//  * A CULLED model (IcpModelArgs::box_blocks > 0) gets box_blocks workgroups -- sized by the host from the screen box the model ended
//    the previous frame with -- whose waves are dealt the 64-pixel runs INSIDE the model's current screen box (cull_runs: the rectangle
//    in units of runs when the image width is a multiple of 64, the rows of the box otherwise); a wave walks on by the number of waves
//    when the box has more runs than the host expected.  Until round 4 a culled model had the whole image's workgroups, 80 % of
//    which read the box and left: 4 800 of the 7 500 workgroups of a five-model level-0 launch, dispatched ahead of the work that the
//    launch waits for.  Sums are integers: which wave adds which pixel does not change a bit.
template <int PPT, bool GRAM>
__device__ __forceinline__ void icp_reduce_body(const IcpArgs& args, const RgbArgs& ra, int n_icp_blocks)
{
    const int b = blockIdx.x;
    // Slot decode on the scalar unit, from ONE clause of kernel-argument loads: table entries and descriptor bytes by static index (a
    // dynamic index into the argument segment is a dependent load each), the slot's bounds picked up along the way.  Then the model's
    // argument block as one more clause (pinned: the compiler otherwise loads field by field at first use -- ten dependent scalar
    // round trips in front of a wave's first vector load, each a miss for the first wave on a CU).
    int slot0 = 0, send, desc;
    {
        int e[12], d[12];
#pragma unroll
        for (int k = 0; k < 12; k++) { e[k] = args.slot_end[k]; d[k] = args.slot_desc[k]; }
        asm volatile("" :: "s"(e[0]), "s"(e[1]), "s"(e[2]), "s"(e[3]), "s"(e[4]), "s"(e[5]), "s"(e[6]), "s"(e[7]), "s"(e[8]), "s"(e[9]), "s"(e[10]), "s"(e[11]),
                     "s"(d[0]), "s"(d[4]), "s"(d[8]), "s"(args.slots_used), "s"(ra.compact), "s"(ra.slot_px), "s"((int)blockDim.x));
        send = e[0]; desc = d[0];
#pragma unroll
        for (int k = 1; k < 12; k++) { const bool ge = b >= e[k - 1]; slot0 = ge ? e[k - 1] : slot0; send = ge ? e[k] : send; desc = ge ? d[k] : desc; }
    }
    if (args.slots_used > 12) {   // (more than six trackers: the rest of the table)
#pragma unroll
        for (int k = 12; k < kMaxSlots; k++) { const bool ge = b >= args.slot_end[k - 1]; slot0 = ge ? args.slot_end[k - 1] : slot0; send = ge ? args.slot_end[k] : send; desc = ge ? (int)args.slot_desc[k] : desc; }
    }
    const int bx = b - slot0;
    const int model = (int)(desc & 0x7fu);
    if (desc & kResidualSlot) {
        if (ABL(32) || (ABL(16) && args.m[model].cull)) return;  // timing ablations (CF_ICP_REPLAY)
        if (ra.compact) rgb_residual_body<true>(ra, model, bx, send - slot0);
        else rgb_residual_body<false>(ra, model, bx, 0);
        return;
    }
    const IcpModelArgs ma = args.m[model];
    asm volatile("" :: "s"(ma.vc), "s"(ma.nc), "s"(ma.vp), "s"(ma.np), "s"(ma.st), "s"(ma.acc), "s"(ma.err), "s"(ma.occ), "s"(ma.zr), "s"(ma.row_begin), "s"(ma.row_end),
                 "s"(ma.cull), "s"(ma.box_blocks), "s"(args.cols), "s"(args.rows), "s"(args.flags), "s"(args.occ_shift), "s"(args.occ_w), "s"(args.cdiv.M), "s"(args.cdiv.s),
                 "s"(args.row_begin), "s"(args.row_end), "s"(args.intr.fx), "s"(args.intr.fy), "s"(args.intr.cx), "s"(args.intr.cy), "s"(args.angleSqLt), "s"(args.distSqLe));
    if (ABL(256) && !ma.cull) return;  // timing ablation (CF_ICP_REPLAY): unculled models do nothing
    if constexpr (!GRAM) { icp_lds_zero(); __syncthreads(); }   // (behind the argument clause, in front of the first vector load: the waves of a workgroup start together)
    StatePtr st = (StatePtr)ma.st;
    const int cols = args.cols, rows = args.rows, N = cols * rows;
    const int T = blockDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int abl = ABL(7);  // micro-benchmark ablation bits (0 in production)
    unsigned long long v = 0;
    bool gram_has = false, done = false;

    if (ma.box_blocks > 0) {  // culled model, runs of its screen box, one pixel per lane (the launcher: product form, no error surface)
        if (ABL(8)) return;  // timing ablation: culled models do nothing
        const IcpHot hs = icp_hot(st);
        if (!hs.icp || hs.level_done) return;
        const int L = 2 - args.occ_shift;
        const int box[4] = {hs.box[0], hs.box[1], hs.box[2], hs.box[3]};
        const CullRuns cr = cull_runs(box, L, cols, rows);
        if (ABL(512)) return;  // timing ablation (CF_ICP_REPLAY): the cull test alone
        const int wpb = T >> 6, stride = ma.box_blocks * wpb;
        const float rcp = __builtin_amdgcn_rcpf((float)(cr.nrx > 0 ? cr.nrx : 1));
        const float zlo = hs.zlo, zhi = hs.zhi;
#pragma nounroll
        for (int r = bx * wpb + __builtin_amdgcn_readfirstlane(wave); r < cr.total; r += stride) {
            int start;
            bool in_range = true;
            if (cr.nrx > 0) {  // rectangle of runs: r / nrx exactly ((r + 0.5) / nrx is at least 0.5 / nrx away from an integer, r < 2^16)
                const int q = (int)(((float)r + 0.5f) * rcp);
                start = (cr.y0 + q) * cols + ((cr.x0 + (r - q * cr.nrx)) << 6);
            } else {           // runs of the box's rows; a run may straddle rows: the rectangle test of the run's first / last pixel
                start = (cr.y0 + r) << 6;
                const int bx0 = (box[0] >> L) - 1, by0 = (box[1] >> L) - 1, bx1 = (box[2] >> L) + 1, by1 = (box[3] >> L) + 1;
                const int w1 = min(start + 63, N - 1);
                const int q0 = idiv(start, args.cdiv), q1 = idiv(w1, args.cdiv);
                if (q1 < by0 || q0 > by1 || (q0 == q1 && (w1 - q0 * cols < bx0 || start - q0 * cols > bx1))) in_range = false;
            }
            // ... and by depth: the run carries the interval of its valid depths (frame_maps_kernel); if it misses the interval the
            // model's dilated box spans in this camera, no pixel of the run can match
            if (in_range && ma.zr) { const float2 zz = ma.zr[start >> 6]; if (!(zz.x <= zhi && zz.y >= zlo)) in_range = false; }
            const int i0 = start + lane;
            if (!icp_run<1, false>(args, ma, st, hs, true, i0, in_range && i0 < N, false, 0, N, nullptr, abl, lane, wave, v, gram_has, done)) return;
            if (done) return;
        }
        if constexpr (!GRAM) icp_lds_commit(lane, wave, T >> 6, ma.acc + (size_t)(bx % kGroups) * 32);
        return;
    }

    // optional row band [row_begin, row_end) (a rank's share when one model's reduction is split over GPUs): of the whole launch
    // (stand-alone band step) or of this model (split background inside the lock-step loop)
    const int rb = ma.row_end > 0 ? ma.row_begin : args.row_begin, re = ma.row_end > 0 ? ma.row_end : args.row_end;
    const int band0 = rb * cols, band1 = (re > 0 ? re : rows) * cols;
    // the error surface (last level-0 iteration) is written for the WHOLE image on every rank of a split model -- the segmentation
    // reads all of it -- while only the band's pixels enter the sums
    // (flags & 1: this launch writes the error surfaces; & 2: ... except those of culled trackers, which icp_error_surface_kernel
    // writes right behind this launch, so that their ICP pass stays culled)
    const bool err_here = (args.flags & 1) && !((args.flags & 2) && ma.cull);
    const bool whole = err_here && ma.err != nullptr && ma.row_end > 0;
    const int pix0 = whole ? 0 : band0, pix1 = whole ? N : band1;
    const int nlog = (pix1 - pix0 + T * PPT - 1) >> __builtin_ctz(T * PPT);  // (workgroup sizes are powers of two: cf_set_icp_launch)
    // (the slot is sized for the whole image; a model with a row band has fewer logical blocks, and the XCD interleave below is a
    // bijection only on the first 8 * ceil(nlog / 8) hardware blocks)
    if ((bx >> 3) >= ((nlog + 7) >> 3)) return;
    // A culled model keeps the workgroups of a few image rows only: giving every XCD a horizontal band would leave that work on
    // the XCDs whose bands the rectangle crosses.  Its workgroups are dealt round-robin instead (neighbouring pixel runs on
    // different XCDs), so what survives the culling is spread over the whole chip.
    const int lb = (ma.cull && !ABL(1024)) ? bx : xcd_logical_block(bx, nlog);  // (1024: timing ablation, bands for everybody)
    if (lb >= nlog) return;
    float* __restrict__ errs = err_here ? ma.err : nullptr;

    const int i0 = pix0 + (lb * T + threadIdx.x) * PPT;
    bool in_range = i0 < pix1;  // cols is a multiple of PPT, so the whole vector is in range
    const bool cull_here = ma.cull && !err_here;
    IcpHot hs{};
    if (cull_here) hs = icp_hot(st);
    // (Measured and dropped, round 3: issuing the plane loads BEFORE the box test, so that in-box waves would not pay the box's scalar
    // round trip in front of them: 22.3 against 21.5 us.)
    // Screen-box culling on the whole-image mapping (Gram form, several pixels per lane, stand-alone steps): a workgroup whose pixel
    // run misses the rectangle leaves after one scalar load; inside a workgroup that straddles it, the waves outside load nothing and
    // go straight to the commit.  Not on the error-surface iteration, which writes every pixel.
    if (cull_here) {
        if (ABL(8)) return;  // timing ablation: culled models do nothing
        const int L = 2 - args.occ_shift;
        const int bx0 = (hs.box[0] >> L) - 1, by0 = (hs.box[1] >> L) - 1, bx1 = (hs.box[2] >> L) + 1, by1 = (hs.box[3] >> L) + 1;
        const int p0 = pix0 + lb * T * PPT, p1 = min(p0 + T * PPT, pix1) - 1;  // first / last pixel of this workgroup
        const int r0 = idiv(p0, args.cdiv), r1 = idiv(p1, args.cdiv);
        if (r1 < by0 || r0 > by1) return;
        if (r0 == r1 && (p1 - r0 * cols < bx0 || p0 - r0 * cols > bx1)) return;
        const int w0 = p0 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) * 64 * PPT, w1 = min(w0 + 64 * PPT, pix1) - 1;
        const int q0 = idiv(w0, args.cdiv), q1 = idiv(max(w1, 0), args.cdiv);
        if (ABL(512)) return;  // timing ablation (CF_ICP_REPLAY): the cull test alone
        if (w0 > w1 || q1 < by0 || q0 > by1 || (q0 == q1 && (w1 - q0 * cols < bx0 || w0 - q0 * cols > bx1))) in_range = false;
        // ... and by depth: the run of 64 pixels this wave owns (per pixel of a lane) carries the interval of its valid depths
        // (frame_maps_kernel); if it misses the interval the model's dilated box spans in this camera, no pixel of the run can match
        if (in_range && ma.zr && w0 <= w1) {
            const float zlo = hs.zlo, zhi = hs.zhi;
            bool any = false;
#pragma unroll
            for (int p = 0; p < PPT; p++) {
                const int c = (w0 >> 6) + p;   // runs are aligned: pix0 == 0 for a culled model, w0 a multiple of 64 * PPT
                if (c * 64 <= w1) { const float2 r = ma.zr[c]; any = any || (r.x <= zhi && r.y >= zlo); }
            }
            if (!any) in_range = false;
        }
    }
    if (!icp_run<PPT, GRAM>(args, ma, st, hs, cull_here, i0, in_range, whole, band0, band1, errs, abl, lane, wave, v, gram_has, done)) return;
    if (done) return;
    if constexpr (GRAM) gram_block_commit(gram_lds, gram_has, lane, wave, T >> 6, ma.acc + (size_t)(lb % kGroups) * 32);
    else icp_lds_commit(lane, wave, T >> 6, ma.acc + (size_t)(lb % kGroups) * 32);
}

#ifdef CF_ABLATE
// diagnostics build: per-workgroup begin / end stamps of one launch (CF_ICP_TRACE, cabi.hip) -- [workgroup][4] = begin, end (100 MHz
// constant clock), XCC_ID | HW_ID << 8, 0
__device__ unsigned long long* g_icp_trace = nullptr;
#endif
// (waves_per_eu 6: the register allocator then lands on 71 VGPRs = seven waves per SIMD with the full scalar register file; asked for
// seven it caps the SGPRs at 94 and spills them through VGPR lanes -- 327 against 45 v_readlane / v_writelane in the kernel.)
template <int PPT, int LEVEL_TAG, bool GRAM>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(PPT <= 2 && !GRAM ? 6 : 1))) icp_reduce_kernel(const IcpArgs args, const RgbArgs ra, int n_icp_blocks)
{
#ifdef CF_ABLATE
    unsigned long long* const tr = g_icp_trace;
    unsigned long long t0 = 0;
    if (tr) t0 = wall_clock64();
#endif
    icp_reduce_body<PPT, GRAM>(args, ra, n_icp_blocks);
#ifdef CF_ABLATE
    if (tr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t1 = wall_clock64();
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u, hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            unsigned long long* o = tr + (size_t)blockIdx.x * 4;
            o[0] = t0; o[1] = t1; o[2] = xcc | ((unsigned long long)hwid << 8); o[3] = 0;
        }
    }
#endif
}

// Where do workgroups land?  The one-XCD meetings (SO(3) pre-alignment, cf_set_gn_mode 2) and the XCD bands rely on hardware workgroup b
// running on XCD b mod 8 (tools/microbench/xcc_map.hip measured it on the MI355X).  The probe states it for the device at hand: 64
// workgroups write their XCC_ID; true iff the first eight are all different and workgroup b repeats workgroup b mod 8's.
__global__ void __launch_bounds__(64) xcc_probe_kernel(unsigned* __restrict__ out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
}
bool probe_xcd_round_robin(hipStream_t s)
{
    unsigned* d = nullptr;
    unsigned h[64];
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(h)) != hipSuccess) return false;
    xcc_probe_kernel<<<64, 64, 0, s>>>(d);
    const bool ok = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return false;
    unsigned seen = 0;
    for (int b = 0; b < 8; b++) seen |= 1u << h[b];
    if (__builtin_popcount(seen) != 8) return false;
    for (int b = 8; b < 64; b++) if (h[b] != h[b & 7]) return false;
    return true;
}

// host side: the f32 bounds that decide "sqrtf(x) < T" and "sqrtf(x) <= T" exactly (sqrtf is correctly rounded, monotonic)
float sqrt_gate_lt(float T)
{   // smallest x with sqrtf(x) >= T  =>  sqrtf(x) < T  <=>  x < bound
    if (!(T > 0.f)) return 0.f;
    float x = T * T;
    while (sqrtf(x) >= T && x > 0.f) x = nextafterf(x, 0.f);
    while (sqrtf(x) < T) x = nextafterf(x, INFINITY);
    return x;
}
float sqrt_gate_le(float T)
{   // largest x with sqrtf(x) <= T  =>  sqrtf(x) <= T  <=>  x <= bound
    if (!(T >= 0.f)) return -1.f;
    if (std::isinf(T)) return INFINITY;  // gate disabled: every radicand passes (x <= inf)
    float x = T * T;
    while (sqrtf(x) <= T && std::isfinite(x)) x = nextafterf(x, INFINITY);
    while (sqrtf(x) > T) x = nextafterf(x, 0.f);
    return x;
}

// ================================================================================================
// RGB residual: RGBResidual::getProducts, reduce.cu:785-865
// ================================================================================================
// The iteration-invariant half of the validity test (window non-zero, gradient magnitude, d1 valid)
// is hoisted into a per-frame candidate mask; the reference re-evaluates it every iteration.
__global__ void __launch_bounds__(256) rgb_cand_kernel(const int16_t* __restrict__ dIdx, const int16_t* __restrict__ dIdy,
                                                       const float* __restrict__ next_depth,
                                                       const uint8_t* __restrict__ next_image, float min_scale, int cols,
                                                       int rows, uint8_t* __restrict__ cand)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cols * rows) return;
    const int i = k / cols, j0 = k - i * cols;
    uint8_t ok = 0;
    if (j0 < cols - 5 && i < rows - 1) {
        bool valid = true;
        for (int u = max(i - 2, 0); u < min(i + 2, rows); u++)
            for (int v = max(j0 - 2, 0); v < min(j0 + 2, cols); v++) valid = valid && (next_image[u * cols + v] > 0);
        if (valid) {
            const int valx = dIdx[k], valy = dIdy[k];
            const float mTwo = (float)((valx * valx) + (valy * valy));
            if (mTwo >= min_scale && !is_nan(next_depth[k])) ok = 1;
        }
    }
    cand[k] = ok;
}

// One candidate pixel of the residual pass: the pose-dependent half of RGBResidual::getProducts.
// Returns validity; g = flat index of the matched pixel in the last image, diff = next - last intensity.
__device__ __forceinline__ bool rgb_residual_pixel(const RgbArgs& ra, const RgbModelArgs& m, const float* __restrict__ krk,
                                                   const float* __restrict__ kt, int k, float d1, float ni, int& u0, int& v0, float& diff)
{
    const int cols = ra.cols, rows = ra.rows;
    const int y = idiv(k, ra.cdiv), x = k - y * cols;
    const float transformed_d1 = (float)(d1 * (krk[6] * x + krk[7] * y + krk[8]) + kt[2]);
    u0 = f2i_rn((d1 * (krk[0] * x + krk[1] * y + krk[2]) + kt[0]) / transformed_d1);
    v0 = f2i_rn((d1 * (krk[3] * x + krk[4] * y + krk[5]) + kt[1]) / transformed_d1);
    if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
        const float d0 = m.lastDepth[v0 * cols + u0];
        const uint8_t li = m.lastImage[v0 * cols + u0];
        if (d0 > 0 && fabsf(transformed_d1 - d0) <= ra.maxDepthDelta && li != 0) {
            diff = ni - (float)li;
            return true;
        }
    }
    return false;
}

// COMPACT == false: the reference's output format, one DataTerm record per pixel (valid or not), read back by
// rgb_step_kernel -- 16 B/pixel written and re-read although < 10 % of the records are valid.
// COMPACT == true (the device-resident Gauss-Newton loop): four pixels per thread, and the valid correspondences of a
// workgroup are packed into that workgroup's own slot of the record buffer (8 B records, slot = 4 x workgroup size; places
// inside the slot come from LDS atomics, the count goes to slot_counts[workgroup]) -- no global atomic and no extra memory
// round trip on the producer side.  The order inside a slot depends on the schedule, the sums taken over it do not.
//
// Slots of a culled tracker (round 5).  An object model's candidate mask is empty outside its prediction, and until round 5 the
// ~1200 waves per object that found nothing but an empty mask word were 4 800 of the level-0 launch's 16 700.  The preparation now
// records the first and the last 256-pixel chunk with a candidate (RgbModelArgs::res_range); this tracker's `nblk` workgroups -- sized by
// the host from what the previous call saw -- take the slots in between, walking on by nblk when there are more than expected.
// Slots outside hold no record: the RGB step applies the same test instead of reading their (stale) counts.
struct SlotRange { int first, last; };
// (the two words through the constant address space -- the preparation wrote them launches ago --, from a dummy address when the tracker
// has no range, so that the load is unconditional and can share a clause with the hot state's: rgb_hot_and_range)
__device__ __forceinline__ SlotRange residual_slot_range_from(const RgbArgs& ra, bool ranged, unsigned lo_inv, unsigned hi_p1, int n_slots)
{
    if (!ranged) return SlotRange{0, n_slots - 1};
    if (hi_p1 == 0) return SlotRange{0, -1};
    const int sh = __builtin_ctz(ra.slot_px) - 8;                     // slot_px = 4 x workgroup size: a power of two >= 256
    return SlotRange{(int)((~lo_inv) >> sh), min((int)((hi_p1 - 1u) >> sh), n_slots - 1)};
}
__device__ __forceinline__ RgbHot rgb_hot_and_range(const RgbArgs& ra, const RgbModelArgs& m, int n_slots, SlotRange& sr)
{
    StatePtr st = (StatePtr)m.st;
    typedef const __attribute__((address_space(4))) unsigned* U4;
    U4 rp = m.res_range ? (U4)m.res_range : (U4)&st->hot;
    const hot16 l2 = hot_line(st, 2);
    const unsigned lo_inv = rp[0], hi_p1 = rp[1];
    asm volatile("" :: "s"(l2), "s"(lo_inv), "s"(hi_p1));
    sr = residual_slot_range_from(ra, m.res_range != nullptr, lo_inv, hi_p1, n_slots);
    RgbHot h;
    h.rgb = l2[0]; h.rgbOnly = l2[1]; h.level_done = l2[2];
#pragma unroll
    for (int k = 0; k < 9; k++) h.krk[k] = hot_f(l2, 4 + k);
#pragma unroll
    for (int k = 0; k < 3; k++) h.kt[k] = hot_f(l2, 13 + k);
    return h;
}
__device__ __forceinline__ SlotRange residual_slot_range(const RgbArgs& ra, const RgbModelArgs& m, int n_slots)
{
    if (!m.res_range) return SlotRange{0, n_slots - 1};
    const unsigned lo_inv = m.res_range[0], hi_p1 = m.res_range[1];   // (uniform: scalar loads)
    if (hi_p1 == 0) return SlotRange{0, -1};
    const int sh = __builtin_ctz(ra.slot_px) - 8;                     // slot_px = 4 x workgroup size: a power of two >= 256
    return SlotRange{(int)((~lo_inv) >> sh), min((int)((hi_p1 - 1u) >> sh), n_slots - 1)};
}

template <bool COMPACT>
__device__ void rgb_residual_body(const RgbArgs& ra, int model, int blk, int nblk)
{
    const RgbModelArgs m = ra.m[model];   // (one clause of kernel-argument loads: see icp_reduce_body)
    asm volatile("" :: "s"(m.st), "s"(m.cand), "s"(m.nextDepth), "s"(m.lastDepth), "s"(m.lastImage), "s"(m.nextImage), "s"(m.icp_acc), "s"(m.recs), "s"(m.slot_counts),
                 "s"(m.res_range), "s"(m.no_counts), "s"(m.corres), "s"(ra.cols), "s"(ra.rows), "s"(ra.slot_px), "s"(ra.cdiv.M), "s"(ra.cdiv.s), "s"(ra.maxDepthDelta));
    StatePtr st = (StatePtr)m.st;
    const int cols = ra.cols, rows = ra.rows, N = cols * rows;
    const int T = blockDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (!COMPACT) {
        const RgbHot hs = rgb_hot(st);
        if (!hs.rgb || hs.level_done) return;
        const int k = blk * T + threadIdx.x;
        int cnt = 0, sig = 0;
        if (k < N) {
            cf_dataterm c; c.zero_x = c.zero_y = c.one_x = c.one_y = 0; c.diff = 0.f; c.valid = 0;
            if (m.cand[k]) {
                int u0, v0; float diff;
                if (rgb_residual_pixel(ra, m, hs.krk, hs.kt, k, m.nextDepth[k], (float)m.nextImage[k], u0, v0, diff)) {
                    const int y = idiv(k, ra.cdiv), x = k - y * cols;
                    c.zero_x = (int16_t)u0; c.zero_y = (int16_t)v0; c.one_x = (int16_t)x; c.one_y = (int16_t)y;
                    c.diff = diff; c.valid = 1;
                    cnt = 1; sig = (int)(diff * diff);
                }
            }
            *reinterpret_cast<int4*>(&m.corres[k]) = *reinterpret_cast<const int4*>(&c);
        }
        // block reduce (count, sigma) -> grouped atomics into words 29/30 of the ICP accumulator
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); sig += __shfl_xor(sig, o, 64); }
        __shared__ int s_cnt[16], s_sig[16];
        if (lane == 0) { s_cnt[wave] = cnt; s_sig[wave] = sig; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int c4 = 0, g4 = 0;
            for (int w = 0; w < (T >> 6); w++) { c4 += s_cnt[w]; g4 += s_sig[w]; }
            unsigned long long* dst = m.icp_acc + (size_t)(blk % kGroups) * 32;
            if (c4) atomicAdd(&dst[29], (unsigned long long)c4);
            if (g4) atomicAdd(&dst[30], (unsigned long long)(long long)g4);
        }
    } else {
        __shared__ int s_n, s_sig;
        const int n_slots = (N + T * 4 - 1) / (T * 4);
        SlotRange sr;
        const RgbHot hs = rgb_hot_and_range(ra, m, n_slots, sr);   // (one scalar clause for the state and the range)
        const bool on = hs.rgb && !hs.level_done;  // uniform
        if (!on) return;
#pragma nounroll
        for (blk += sr.first; blk <= sr.last; blk += max(nblk, 1)) {
        const int k0 = (blk * T + threadIdx.x) * 4;  // cols % 4 == 0: the four pixels share a row
        unsigned cw = 0;
        if (k0 < N) cw = *reinterpret_cast<const unsigned*>(m.cand + k0);
        if (threadIdx.x == 0) { s_n = 0; s_sig = 0; }
        __syncthreads();
        if (__any(cw != 0)) {
            int g[4], dq[4], nvalid = 0, sig = 0;
            float d1[4] = {0, 0, 0, 0}; unsigned iw = 0;
            if (cw) {
                const float4 dv = *reinterpret_cast<const float4*>(m.nextDepth + k0);
                d1[0] = dv.x; d1[1] = dv.y; d1[2] = dv.z; d1[3] = dv.w;
                iw = *reinterpret_cast<const unsigned*>(m.nextImage + k0);
            }
            // RGBResidual::getProducts (reduce.cu:785-865), the pose-dependent half, in three phases so that the gathers of a thread's four
            // pixels are in flight TOGETHER: until round 5 each pixel ran project -> gather depth -> test -> gather intensity before the
            // next one started -- eight dependent round trips per thread, 6 us per background workgroup (tools/icp_trace_summary.py).
            const int y = idiv(k0, ra.cdiv), x0 = k0 - y * cols;
            float td1[4]; int gi[4]; bool inb[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int x = x0 + p;
                td1[p] = (float)(d1[p] * (hs.krk[6] * x + hs.krk[7] * y + hs.krk[8]) + hs.kt[2]);
                const int u0 = f2i_rn((d1[p] * (hs.krk[0] * x + hs.krk[1] * y + hs.krk[2]) + hs.kt[0]) / td1[p]);
                const int v0 = f2i_rn((d1[p] * (hs.krk[3] * x + hs.krk[4] * y + hs.krk[5]) + hs.kt[1]) / td1[p]);
                inb[p] = ((cw >> (8 * p)) & 0xffu) != 0 && u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows;
                gi[p] = inb[p] ? v0 * cols + u0 : 0;
            }
            float d0[4]; unsigned li[4];
#pragma unroll
            for (int p = 0; p < 4; p++) { d0[p] = 0.f; li[p] = 0; if (inb[p]) { d0[p] = m.lastDepth[gi[p]]; li[p] = m.lastImage[gi[p]]; } }
#pragma unroll
            for (int p = 0; p < 4; p++) {
                g[p] = -1; dq[p] = 0;
                if (inb[p] && d0[p] > 0 && fabsf(td1[p] - d0[p]) <= ra.maxDepthDelta && li[p] != 0) {
                    const float diff = (float)((iw >> (8 * p)) & 0xffu) - (float)li[p];
                    g[p] = gi[p]; dq[p] = (int)diff;  // next - last intensity: an integer in [-255, 255]
                    nvalid++; sig += (int)(diff * diff);
                }
            }
            int wn = nvalid, ws = sig;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { wn += __shfl_xor(wn, o, 64); ws += __shfl_xor(ws, o, 64); }
            if (wn) {
                int incl = nvalid;  // inclusive prefix of nvalid inside the wave
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                int wbase = 0;
                if (lane == 0) { wbase = atomicAdd(&s_n, wn); atomicAdd(&s_sig, ws); }  // LDS: the wave's place inside the slot
                wbase = __shfl(wbase, 0, 64);
                uint2* __restrict__ out = m.recs + (size_t)blk * T * 4 + wbase + incl - nvalid;
#pragma unroll
                for (int p = 0; p < 4; p++)
                    if (g[p] >= 0) { *out++ = make_uint2((unsigned)(k0 + p), (unsigned)g[p] | ((unsigned)(dq[p] + 256) << 22)); }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int bn = s_n, g4 = s_sig;
            m.slot_counts[blk] = (unsigned)bn;  // every workgroup publishes its count: the list pass reads it unconditionally
            unsigned long long* dst = m.icp_acc + (size_t)(blk % kGroups) * 32;
            if (bn && !m.no_counts) atomicAdd(&dst[29], (unsigned long long)bn);
            if (g4 && !m.no_counts) atomicAdd(&dst[30], (unsigned long long)(long long)g4);
        }
        }   // (thread 0 has read s_n / s_sig before it clears them for the next slot; everybody else meets it at the barrier behind that)
    }
}

__global__ void __launch_bounds__(1024) rgb_residual_kernel(const RgbArgs ra) { rgb_residual_body<false>(ra, blockIdx.y, blockIdx.x, 0); }

// sum of word `w` over the groups (wave 0 only; result valid in all lanes of wave 0)
__device__ __forceinline__ unsigned long long group_sum(const unsigned long long* acc, int w, int lane)
{
    unsigned long long v = acc[(size_t)lane * 32 + w];  // kGroups == 64 == lanes
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_xor_u64(v, o);
    return v;
}

// wrapping 32-bit sum over the 64 lanes of a wave, the same value in every lane: DPP inside each row of 16 lanes (two quad permutations, the
// mirrors of the half row and of the row), then the four row totals through scalar registers
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1, 0, 3, 2]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2, 3, 0, 1]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);   // row_mirror
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16) + (unsigned)__builtin_amdgcn_readlane((int)v, 32) +
           (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}

// sigma handed to rgbStep (RGBDOdometry.cpp:373-385): (tmpError == 0) ? 1 : count   (sic: the COUNT)
__device__ __forceinline__ float sigma_val_from(int count, int sigma, int rgbOnly)
{
    if (rgbOnly) return -1.f;
    // tmpError = sqrt(sigma)/count is 0 iff sigma == 0 and count != 0 (0/0 is NaN, NaN != 0)
    return (sigma == 0 && count != 0) ? 1.f : (float)count;
}

// ================================================================================================
// RGB step: RGBReduction::getProducts, reduce.cu:521-604
// ================================================================================================
// Jacobian row of one valid correspondence (o = flat index in the next image, g = in the last image / point cloud)
__device__ __forceinline__ void rgb_step_row(const RgbArgs& ra, const RgbModelArgs& m, float sigma, float diff, int o, int g, float (&row)[7])
{
    const cf_cam il = ra.il;
    float w = sigma + fabsf(diff);
    w = w > 1.19209290E-07F ? 1.0f / w : 1.0f;
    if (sigma == -1) w = 1;
    row[6] = -w * diff;
    const float* cp = m.cloud + (size_t)g * 3;
    const float px = cp[0], py = cp[1], pz = cp[2];
    const float invz = 1.0f / pz;
    const float dI_dx_val = w * ra.sobelScale * (float)m.dIdx[o];
    const float dI_dy_val = w * ra.sobelScale * (float)m.dIdy[o];
    const float v0 = dI_dx_val * il.fx * invz;
    const float v1 = dI_dy_val * il.fy * invz;
    const float v2 = -(v0 * px + v1 * py) * invz;
    row[0] = v0; row[1] = v1; row[2] = v2;
    row[3] = -pz * v1 + py * v2;
    row[4] = pz * v0 - px * v2;
    row[5] = -py * v0 + px * v1;
}

__global__ void __launch_bounds__(256) rgb_step_kernel(const RgbArgs ra)
{
    const RgbModelArgs& m = ra.m[blockIdx.y];
    const RgbHot hs = rgb_hot((StatePtr)m.st);
    if (hs.rgb && !hs.level_done) {
        const int cols = ra.cols, rows = ra.rows, N = cols * rows;
        __shared__ float s_sigma;
        if (threadIdx.x < 64) {
            const long long cnt = (long long)group_sum(m.icp_acc, 29, threadIdx.x);
            const long long sg = (long long)group_sum(m.icp_acc, 30, threadIdx.x);
            if (threadIdx.x == 0) s_sigma = sigma_val_from((int)cnt, (int)sg, hs.rgbOnly);
        }
        const int i = blockIdx.x * 256 + threadIdx.x;
        int4 raw = make_int4(0, 0, 0, 0);
        if (i < N) raw = *reinterpret_cast<const int4*>(&m.corres[i]);
        __syncthreads();
        const float sigma = s_sigma;
        unsigned long long acc[32];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = 0ull - kMagicBits;
        acc[28] = acc[29] = acc[30] = acc[31] = 0;
        float row[7] = {0, 0, 0, 0, 0, 0, 0};
        int found = 0;
        if (i < N) {
            const cf_dataterm c = *reinterpret_cast<const cf_dataterm*>(&raw);
            if (c.valid) {
                found = 1;
                rgb_step_row(ra, m, sigma, c.diff, c.one_y * cols + c.one_x, c.zero_y * cols + c.zero_x, row);
            }
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned long long v = 0;
        if (__any(found)) {  // a wave without a valid correspondence adds exact zeros
            const int F = rgb_fix_bits(sigma);
            se3_accumulate_dyn(row, acc, ldexpf(1.0f, (50 - F) / 2), ldexpf(1.0f, F));
            acc[28] = (unsigned long long)found;
            v = wave_reduce32_u64(acc, lane);
        }
        block_commit32<4>(v, lane, wave, 4, m.rgb_acc + (size_t)(blockIdx.x % kGroups) * 32);
    }
}

// ================================================================================================
// SO3: SO3Reduction::getProducts, reduce.cu:1007-1090
// ================================================================================================
__device__ __forceinline__ void so3_gradient(const uint8_t* __restrict__ img, int cols, int x, int y, float& gx, float& gy)
{  // reduce.cu:989-1005
    const float actu = (float)img[y * cols + x];
    float back = (float)img[y * cols + x - 1], fore = (float)img[y * cols + x + 1];
    gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = (float)img[(y - 1) * cols + x]; fore = (float)img[(y + 1) * cols + x];
    gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}

// one (grid-strided over blockIdx.x) pass over the level-2 images; this workgroup's totals[0..10] end up in LDS
__device__ __forceinline__ void so3_pass(const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage,
                                         const m33& B, const m33& Ki, const float* __restrict__ krlr, int cols, int rows,
                                         unsigned long long (*lds)[16], unsigned long long* totals, int block, int blocks)
{
    constexpr float lim = (float)(1 << ((50 - kFixSO3) / 2));
    constexpr float scale = (float)(1 << kFixSO3);
    const int N = cols * rows, T = blockDim.x;
    unsigned long long acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0;
    const float a = krlr[0], b = krlr[1], c = krlr[2], d = krlr[3], e = krlr[4], f = krlr[5], g = krlr[6], h = krlr[7],
                ii = krlr[8];
    // (Measured and dropped, round 6: four pixels of a thread at a time -- their warps first, the 4 x 10 byte loads of the gradient stencils in
    // flight together, then the rows: so3_prealign_kernel 82.0 against 63.1 us on one box, profiles/r6u_*.  The registers of four stencils
    // cost the launch's other half, the RGB preparation workgroups, their occupancy; the pass itself is 4-7 us of a 9 us iteration.)
    for (int k = block * T + threadIdx.x; k < N; k += blocks * T) {
        const int y = k / cols, x = k - y * cols;
        const f3 unwarped = {(float)x, (float)y, 1.0f};
        const f3 warped = mul(B, unwarped);
        const int wx = f2i_rn(warped.x / warped.z), wy = f2i_rn(warped.y / warped.z);
        if (!(wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 && x < cols - 1 && y >= 1 && y < rows - 1))
            continue;
        float gnx, gny, glx, gly;
        so3_gradient(nextImage, cols, wx, wy, gnx, gny);
        so3_gradient(lastImage, cols, x, y, glx, gly);
        const float gx = (gnx + glx) / 2.0f, gy = (gny + gly) / 2.0f;
        const f3 point = mul(Ki, unwarped);
        const float z2 = point.z * point.z;
        const f3 left = {((point.z * (d * gy + a * gx)) - (gy * g * y) - (gx * g * x)) / z2,
                         ((point.z * (e * gy + b * gx)) - (gy * h * y) - (gx * h * x)) / z2,
                         ((point.z * (f * gy + c * gx)) - (gy * ii * y) - (gx * ii * x)) / z2};
        const f3 jac = cross(left, point);
        const float row[4] = {jac.x, jac.y, jac.z, -((float)nextImage[wy * cols + wx] - (float)lastImage[y * cols + x])};
        double r[4], rs[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { r[q] = (double)clamp_row(row[q], lim); rs[q] = (double)(clamp_row(row[q], lim) * scale); }
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q >= p)  // k(p,q) = 4p - p(p-1)/2 + (q-p)
                    acc[4 * p - (p * (p - 1)) / 2 + (q - p)] += (unsigned long long)__double_as_longlong(fma(rs[p], r[q], kMagic)) - kMagicBits;
        acc[9] += (unsigned long long)__double_as_longlong(fma(rs[3], r[3], kMagic)) - kMagicBits;
        acc[10] += 1;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long v = wave_reduce16_u64(acc, lane);
    if ((lane & 3) == 0) lds[wave][lane >> 2] = v;
    __syncthreads();
    if (threadIdx.x < 16) {
        unsigned long long t = 0;
        for (int w = 0; w < (T >> 6); w++) t += lds[w][threadIdx.x];
        totals[threadIdx.x] = t;
    }
    __syncthreads();
}

// reduce.cu:1158-1175 host unpack, on device
__device__ inline void so3_unpack(const unsigned long long* t, float A[9], float b[3], float residual[2])
{
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            const float value = fix_to_f32((long long)t[shift++], kFixSO3);
            if (j == 3) b[i] = value;
            else A[j * 3 + i] = A[i * 3 + j] = value;
        }
    residual[0] = fix_to_f32((long long)t[9], kFixSO3);
    residual[1] = (float)(long long)t[10];
}

__device__ inline void k_matrix(cf_cam c, double K[9])
{
    for (int i = 0; i < 9; i++) K[i] = 0;
    K[0] = c.fx; K[4] = c.fy; K[2] = c.cx; K[5] = c.cy; K[8] = 1;
}

// krkInv / kt for the next iteration (RGBDOdometry.cpp:347-358)
__device__ inline void prepare_iteration(OdomDev* od, int level)
{
    double K[9], Kinv[9], Rt[16];
    k_matrix(cam_level(od->intr, level), K);
    inv33<double>(K, Kinv);
    inv44_affine(od->resultRt, Rt);
    const double R[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
    double tmp[9], KRK[9];
    mul33<double>(K, R, tmp);
    mul33<double>(tmp, Kinv, KRK);
    for (int k = 0; k < 9; k++) od->krkInv[k] = (float)KRK[k];
    const double tv[3] = {Rt[3], Rt[7], Rt[11]};
    for (int r = 0; r < 3; r++) od->kt[r] = (float)(K[r * 3 + 0] * tv[0] + K[r * 3 + 1] * tv[1] + K[r * 3 + 2] * tv[2]);
}

// Stand-alone single SO3 step (C-ABI so3Step): one workgroup, totals to out16
__global__ void __launch_bounds__(1024) so3_step_kernel(const uint8_t* __restrict__ lastImage,
                                                        const uint8_t* __restrict__ nextImage, m33 B, m33 Ki, m33 krlr,
                                                        int cols, int rows, unsigned long long* __restrict__ out16)
{
    __shared__ unsigned long long lds[16][16];
    __shared__ unsigned long long totals[16];
    so3_pass(lastImage, nextImage, B, Ki, krlr.m, cols, rows, lds, totals, blockIdx.x, gridDim.x);
    if (threadIdx.x < 16) out16[threadIdx.x] = totals[threadIdx.x];
}

// Whole SO3 pre-alignment (RGBDOdometry.cpp:239-310) in ONE launch.  The pass over the 160x120 level is VALU-bound
// on a single CU (10.5 us per iteration, measured), so kSo3Blocks co-resident workgroups per model share it:
// each reduces its pixels, adds its 11 fixed-point totals to the iteration's slot of a global accumulator and
// meets the others at an atomic arrival counter; every workgroup then reads the totals and runs the identical 3x3
// solve + Rodrigues on its own LDS copy of the state, so nothing but integer atomics crosses workgroups and the
// data-dependent early exits stay uniform.  The last workgroup to leave re-zeroes the sync block for the next frame.
// Also seeds resultRt and the first iteration's krkInv/kt.
//
// ONE XCD PER MODEL (round 4).  Until round 4 the meeting was device-scope: atomics through the fabric, a release fence that writes the
// XCD's L2 back, polling loads that bypass it -- tools/microbench/xcd_barrier.hip measures 9.1 us for such a barrier of 32 workgroups
// (11.3 us across the chip) against 1.1 us when the workgroups share an XCD and meet in its L2 (atomics at workgroup scope execute in
// the L2, and so do the returning atomics the counters and sums are read with; nothing is written back).  The launch therefore has 8 x
// kSo3Blocks workgroups per model and keeps those whose index is (model mod 8) modulo 8: the dispatcher deals consecutive workgroups
// round-robin over the XCDs (what xcd_logical_block relies on too), so they share one.  Should that ever not hold, the arrival
// counters live in different L2s, the bounded wait below expires and raises the fault word -- cf_odom_fetch_result returns CF_ESTATE
// instead of a pose from partial sums.  The sums are integers: the bits do not depend on any of this.
// Reads that are answered by the L2 itself: a RETURNING read-modify-write (OR with 0).  A load marked sc0 is a group-scope load, which
// the CU's vector L1 may serve -- with it the workgroups spun on a stale arrival count (reproduced in tools/microbench/xcd_barrier.hip).
__device__ __forceinline__ unsigned l2_read_u32(unsigned* p)
{
    unsigned v;
    const unsigned zero = 0;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long l2_read_u64(unsigned long long* p)
{
    unsigned long long v;
    const unsigned long long zero = 0;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
    return v;
}
// The launch is one-dimensional: [gx workgroups per model of the pre-alignment | prep_bx workgroups per model of the RGB preparation
// (Sobel + candidate mask + cloud: rgb_prep_body)].  The two read the same pyramids and depend on nothing of each other; the
// pre-alignment is a latency chain on 16 workgroups per model, the preparation fills the rest of the chip meanwhile.
//
// The tracker state of the call arrives here as well: the host fills its pinned copy, every pre-alignment workgroup stages that copy into
// LDS (one coalesced read over PCIe, hidden beside the preparation workgroups) and the lead workgroup of each tracker stores it into the
// device state the rest of the schedule reads -- no copy command in front of the loop (two of them cost ~10 us on the stream per frame).
#ifdef CF_ABLATE
// diagnostics build (CF_SO3_TRACE): stamps of tracker 0's lead workgroup in the pre-alignment loop, [iteration][8]
__device__ unsigned long long g_so3_trace[12][8];
#define OSTAMP(it, k) do { if (lead && by == 0 && threadIdx.x == 0) g_so3_trace[it][k] = wall_clock64(); } while (0)
#else
#define OSTAMP(it, k) do {} while (0)
#endif
__global__ void __launch_bounds__(256) so3_prealign_kernel(const TrackerStates ts, So3Sync* __restrict__ syncs, int do_so3,
                                                           int first_level, int gx, int so3_blocks, const RgbPrepBatch prep, int prep_bx)
{
    if ((int)blockIdx.x >= so3_blocks) {
        const int r = (int)blockIdx.x - so3_blocks, m = r / prep_bx;
        rgb_prep_body(prep.m[m], r - m * prep_bx);
        return;
    }
    const int by = (int)blockIdx.x / gx, bxx = (int)blockIdx.x - by * gx;  // gx is 1 or a multiple of 8: bxx mod 8 is the XCD
    OdomDev* const god = ts.dev[by];
    So3Sync* sync = syncs + by;
    __shared__ OdomDev s_od;
    __shared__ unsigned long long lds[16][16];
    __shared__ unsigned long long totals[16];
    __shared__ float s_basis[9], s_kinv[9], s_krlr[9];
    __shared__ int s_done;
    __shared__ double s_resultR[9];
    __shared__ double s_K[9], s_Kinv[9];
    __shared__ float s_Rlr[9];
    __shared__ float s_lastError, s_lastCount;
    __shared__ double s_lastResultR[9];
    __shared__ float s_jtj[9], s_jtr[3], s_delta[3], s_fws[15];
    __shared__ int s_iws[3];
    // with the pre-alignment the launch is 8 x kSo3Blocks wide: this model's workgroups are the ones on XCD (model mod 8)
    const bool one_xcd = do_so3 && gx > 1;
    if (one_xcd && (bxx & 7) != (by & 7)) return;
    const int bx = one_xcd ? (bxx >> 3) : bxx;
    const bool lead = bx == 0;  // the workgroup that uploads the state, publishes statistics and the final state
    const unsigned G = one_xcd ? (unsigned)gx >> 3 : (unsigned)gx;
    {
        static_assert(sizeof(OdomDev) % 4 == 0, "OdomDev is staged as 32-bit words");
        constexpr int kWords = (int)(sizeof(OdomDev) / 4);
        const unsigned* __restrict__ src = reinterpret_cast<const unsigned*>(ts.host[by]);
        constexpr int kPer = (kWords + 255) / 256;   // (both loads of a thread in flight together: they cross PCIe)
        unsigned w[kPer];
#pragma unroll
        for (int q = 0; q < kPer; q++) w[q] = ((int)threadIdx.x + 256 * q < kWords) ? src[threadIdx.x + 256 * q] : 0u;
#pragma unroll
        for (int q = 0; q < kPer; q++) if ((int)threadIdx.x + 256 * q < kWords) reinterpret_cast<unsigned*>(&s_od)[threadIdx.x + 256 * q] = w[q];
        __syncthreads();
        if (lead) for (int k = threadIdx.x; k < kWords; k += 256) reinterpret_cast<unsigned*>(god)[k] = reinterpret_cast<const unsigned*>(&s_od)[k];
        __syncthreads();  // (the lead's later stores into the device state follow the upload)
    }
    const OdomDev* const od = &s_od;  // what the host passed; results go to the device state (god)
    const int L = 2, cols = od->width >> L, rows = od->height >> L;
    if (threadIdx.x == 0) {
        for (int k = 0; k < 9; k++) { s_resultR[k] = (k % 4 == 0) ? 1.0 : 0.0; s_lastResultR[k] = s_resultR[k]; s_Rlr[k] = (k % 4 == 0) ? 1.f : 0.f; }
        k_matrix(cam_level(od->intr, L), s_K);
        inv33<double>(s_K, s_Kinv);
        s_lastError = 3.402823466e+38F / 2; s_lastCount = 3.402823466e+38F / 2;
        s_done = 0;
        if (lead) { god->stats.so3_iterations = 0; god->stats.last_so3_error = 0; god->stats.last_so3_count = 0; }
    }
    __syncthreads();
    if (do_so3) {
        const uint8_t* __restrict__ lastNext = od->lastNextImage[L];
        const uint8_t* __restrict__ next = od->nextImage[L];
        OSTAMP(11, 0);
        for (int it = 0; it < 10; it++) {
            OSTAMP(it, 0);
            if (threadIdx.x == 0) {
                double tmp[9], H[9];
                mul33<double>(s_K, s_resultR, tmp);
                mul33<double>(tmp, s_Kinv, H);
                for (int k = 0; k < 9; k++) { s_basis[k] = (float)H[k]; s_kinv[k] = (float)s_Kinv[k]; s_krlr[k] = (float)tmp[k]; }
            }
            __syncthreads();
            m33 B, Ki;
            for (int k = 0; k < 9; k++) { B.m[k] = s_basis[k]; Ki.m[k] = s_kinv[k]; }
            OSTAMP(it, 1);
            so3_pass(lastNext, next, B, Ki, s_krlr, cols, rows, lds, totals, bx, (int)G);  // ends with this workgroup's totals in LDS
            OSTAMP(it, 2);
            if (G > 1) {
                if (threadIdx.x < 64) {  // wave 0: publish, arrive, wait, collect -- everything in this XCD's L2
                    unsigned long long* slot = sync->acc[it];
                    if (threadIdx.x < 11 && totals[threadIdx.x] != 0)
                        __hip_atomic_fetch_add(&slot[threadIdx.x], totals[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the L2 has taken this wave's sums before it arrives
                    if (threadIdx.x == 0) {
                        __hip_atomic_fetch_add(&sync->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const unsigned target = (unsigned)(it + 1) * G;
                        unsigned spins = 0;
                        while (l2_read_u32(&sync->arrive) < target) {
                            if (++spins > (1u << 22)) { god->stats.fault = 1; break; }  // never hang the GPU; the host reports CF_ESTATE
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (threadIdx.x < 16) totals[threadIdx.x] = l2_read_u64(&slot[threadIdx.x]);
                }
                __syncthreads();
            }
            OSTAMP(it, 3);
            if (threadIdx.x == 0) {
                float jtj[9], jtr[3], residual[2];
                so3_unpack(totals, jtj, jtr, residual);
                if (lead) god->stats.so3_iterations = it + 1;
                float err = sqrtf(residual[0]) / residual[1];
                float cnt = residual[1];
                if (err < s_lastError && (double)fabsf(s_lastError - cnt) < 0.001) {
                    s_done = 1;  // "converged" (compares error with COUNT, RGBDOdometry.cpp:285)
                } else if ((double)err > (double)s_lastError + 0.001) {
                    err = s_lastError; cnt = s_lastCount;
                    for (int k = 0; k < 9; k++) s_resultR[k] = s_lastResultR[k];
                    s_done = 1;
                } else {
                    s_lastError = err; s_lastCount = cnt;
                    for (int k = 0; k < 9; k++) s_lastResultR[k] = s_resultR[k];
                    for (int k = 0; k < 9; k++) s_jtj[k] = jtj[k];
                    for (int k = 0; k < 3; k++) s_jtr[k] = jtr[k];
                    ldlt_solve<float, 3>(s_jtj, s_jtr, s_delta, 1.17549435e-38f, s_fws, s_iws);
                    const float delta[3] = {s_delta[0], s_delta[1], s_delta[2]};
                    const double dd[3] = {delta[0], delta[1], delta[2]};
                    double rotUpdate[9];
                    rodrigues(dd, rotUpdate);
                    float ru[9], nr[9];
                    for (int k = 0; k < 9; k++) ru[k] = (float)rotUpdate[k];
                    mul33<float>(ru, s_Rlr, nr);
                    for (int k = 0; k < 9; k++) { s_Rlr[k] = nr[k]; s_resultR[k] = nr[k]; }
                }
                if (lead) { god->stats.last_so3_error = err; god->stats.last_so3_count = cnt; }
            }
            __syncthreads();
            OSTAMP(it, 4);
            if (s_done) break;
        }
        OSTAMP(11, 1);
        if (G > 1 && threadIdx.x == 0) {  // last one out resets the sync block (all workgroups are past their final read)
            if (__hip_atomic_fetch_add(&sync->depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == G - 1) {
                for (int it = 0; it < 10; it++)
                    for (int w = 0; w < 16; w++) sync->acc[it][w] = 0;
                sync->arrive = 0; sync->depart = 0;
            }
        }
    }
    if (lead && threadIdx.x < 64) {
        // latch the bounding box the model-map pass accumulated (and clear the accumulator for the next frame); first screen box
        const int lane = threadIdx.x;
        unsigned key = 0;
        if (od->cull && lane < 6) { key = od->aabb_acc[lane]; od->aabb_acc[lane] = 0; }
        const float val = lane < 3 ? fkey_inv(~key) : fkey_inv(key);
        float lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k] = __shfl(val, k, 64); hi[k] = __shfl(val, 3 + k, 64); }
        if (__shfl((int)key, 3, 64) == 0) { lo[0] = 1.f; hi[0] = 0.f; }  // never written: empty
        int ib[4] = {0, 0, od->width - 1, od->height - 1};
        float zb[2] = {-__int_as_float(0x7f800000), __int_as_float(0x7f800000)};
        if (od->cull) screen_box(lo, hi, od->box_R, od->box_t, od->Rcurr, od->tcurr, od->intr, od->distThres, od->width, od->height, lane, ib, zb);
        if (lane == 0) {
            for (int k = 0; k < 3; k++) { god->box_lo[k] = lo[k]; god->box_hi[k] = hi[k]; }
            for (int k = 0; k < 4; k++) god->stats.cull_box[k] = ib[k];
            god->cull_z[0] = zb[0]; god->cull_z[1] = zb[1];
        }
    }
    if (lead && threadIdx.x == 0) {
        for (int k = 0; k < 16; k++) god->resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
        if (do_so3)
            for (int x = 0; x < 3; x++)
                for (int y = 0; y < 3; y++) god->resultRt[x * 4 + y] = s_resultR[x * 3 + y];
        god->lastRGBError = 3.402823466e+38F;
        god->level_done = 0;
        god->residual[0] = 0; god->residual[1] = 0;
        prepare_iteration(god, first_level);
        refresh_hot(god);   // what the workgroups of the per-iteration launches read (cf_kernels.h: GnHot)
    }
}

// ================================================================================================
// per-iteration solve: sums -> A,b (f32) -> f64 combine -> LDL^T -> SE3 update -> next krkInv/kt
// (RGBDOdometry.cpp:371-461, reduce.cu:481-498, OdometryProvider.h:69-89)
// ================================================================================================
__device__ inline void se3_unpack(const unsigned long long* t, int F, float A[36], float b[6], float residual[2])
{
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float value = fix_to_f32((long long)t[shift++], F);
            if (j == 6) b[i] = value;
            else A[j * 6 + i] = A[i * 6 + j] = value;
        }
    residual[0] = fix_to_f32((long long)t[27], F);
    residual[1] = (float)(long long)t[28];
}

// one word of se3_unpack: t < 27 -> A / b entry, 27 -> sum of squared residuals, 28 -> inlier count
// F < 0: the Gram form of the ICP sums (cf_device.h: word (i, j) carries kGramBits[i] + kGramBits[j] fraction bits)
__device__ __forceinline__ void se3_unpack_word(const unsigned long long* sums, int t, int F, float* A, float* b, float* residual)
{
    if (t < 27) {
        int i = 0, rem = t;
        while (rem >= 7 - i) { rem -= 7 - i; i++; }
        const int j = i + rem;
        const int bits = F >= 0 ? F : (i < 3 ? 20 : 17) + (j < 3 ? 20 : j < 6 ? 17 : 22);
        const float value = fix_to_f32((long long)sums[t], bits);
        if (j == 6) b[i] = value;
        else A[j * 6 + i] = A[i * 6 + j] = value;
    } else if (residual) {
        if (t == 27) residual[0] = fix_to_f32((long long)sums[27], F >= 0 ? F : 44);
        else if (t == 28) residual[1] = (float)(long long)sums[28];
    }
}
static_assert(kGramBits[0] == 20 && kGramBits[2] == 20 && kGramBits[3] == 17 && kGramBits[5] == 17 && kGramBits[6] == 22, "se3_unpack_word spells the Gram scales out");

// The solve is latency-bound serial work (f64 LDL^T, Rodrigues, SE3 products) on a nearly idle GPU, so:
//  * the whole device-resident state is staged through LDS (no dependent global round trips),
//  * the parallel pieces (group totals, fixed-point -> f32 unpack, f64 combine, 4x4 / 3x3 products) are spread
//    over lanes with exactly the element expressions of the serial helpers, the 6x6 pivoted LDL^T runs across
//    one wave (ldlt_solve6_wave), and K^-1 of the next level is formed by another wave meanwhile,
//  * only Rodrigues and the 3x3 pose composition stay on one lane.
// Must be called by all 256 threads of a workgroup.
// RGB_IN_L2: the RGB sums were added by workgroup-scope atomics of this launch (rgb_step_solve_kernel): they are read where they live,
// in this XCD's L2, with agent-scope loads -- a plain load may be served by the CU's L1.
#ifdef CF_ABLATE
// diagnostics build (CF_SOLVE_TRACE): phase stamps of tracker 0's solves of one tracking call, [solve][16] on the 100 MHz constant clock
__device__ unsigned long long* g_solve_trace = nullptr;
__device__ unsigned g_solve_iter = 0;
#define SSTAMP(k) do { if (g_solve_trace && threadIdx.x == 0 && blockIdx.x == 0) g_solve_trace[(size_t)(g_solve_iter & 63u) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define SSTAMP(k) do {} while (0)
#endif
// same-wave exchange through LDS: a wave's LDS operations execute in program order, so all that is needed between a lane's store and
// another lane's load is that the compiler keeps them in that order
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// ROUND 6 (9.2 us per solve, the frame's most expensive kernel in total).  The kernel is ONE chain of dependent
// instructions on wave 0 -- ~2500 of them at 4-8 cycles each (profiles/r6c_solve_phases.txt: 1.4 us in the unpack alone, a divergent
// search loop per word, executed twice for the ICP and the RGB lanes of the same wave) -- so what counts is the number of instructions on
// that chain and the barriers that make wave 0 wait for stores:
//  * every lane of the factorisation forms ITS OWN matrix element straight from the totals (closed-form word index, one pass for both
//    systems) -- no unpacked f32 matrices in LDS, no search loop, two barriers less;
//  * everything behind the factorisation is spread over lanes with exactly the element expressions of the serial helpers (Rodrigues'
//    nine entries, the 4x4 product, the f32 pose composition, the affine inverse, K R K^-1), exchanged through LDS inside wave 0;
//  * the statistics (two square roots, two divisions) and the screen box run on idle waves beside that chain, refresh_hot is folded into
//    the write-back, and the accumulators are zeroed by the LAST instructions of the kernel (a barrier waits for outstanding stores
//    too: 16 stores per thread in front of one cost 0.7 us);
//  * five barriers instead of eleven.
// Must be called by all 256 threads of a workgroup.
// RGB_IN_L2: the RGB sums were added by workgroup-scope atomics of this launch (rgb_step_solve_kernel): they are read where they live,
// in this XCD's L2, with agent-scope loads -- a plain load may be served by the CU's L1.
template <bool RGB_IN_L2 = false>
__device__ __forceinline__ void gn_solve_body(int icp_fix, OdomDev* god, unsigned long long* icp_acc, unsigned long long* rgb_acc, int next_level,
                                              int last_of_level, OdomDev* god_host, int slot_px, int part_first = 0, int part_last = -1)
{
    __shared__ OdomDev s_od;
    __shared__ unsigned long long s_icp[32], s_rgb[32];
    __shared__ unsigned long long s_part[2][8][32];
    __shared__ double s_result[6];
    __shared__ double s_upd[16], s_K[9], s_Kinv[9], s_Rt[16], s_tmp[9];
    __shared__ int s_flags[2];   // active, stop (wave 0 decides; the statistics lane on wave 3 reads them behind a barrier)
    static_assert(sizeof(OdomDev) % 4 == 0, "OdomDev is staged as 32-bit words");
    constexpr int kWords = (int)(sizeof(OdomDev) / 4);
    constexpr int kMutableFrom = (int)(offsetof(OdomDev, Rprev) / 4);
    const int tid = threadIdx.x;
    SSTAMP(0);
    OdomDev* const od = &s_od;
    // the state's words and the accumulator words in ONE flight of loads (a rolled copy loop waits for every load before it stores to
    // LDS: two dependent round trips in front of the accumulator loads, ~1.5 us of every solve until round 5)
    constexpr int kPer = (kWords + 255) / 256;
    unsigned stw[kPer];
#pragma unroll
    for (int q = 0; q < kPer; q++) stw[q] = (tid + 256 * q < kWords) ? reinterpret_cast<const unsigned*>(god)[tid + 256 * q] : 0u;
    {   // 256 threads: word = t & 31, slice = t >> 5 (8 slices of 8 groups)
        const int w = tid & 31, sl = tid >> 5;
        unsigned long long a = 0, b = 0;
        if constexpr (RGB_IN_L2) {   // rgb_acc holds one row of 32 totals per workgroup of the step, rows part_first .. part_last (rgb_step_solve_kernel).
            // Agent-scope loads: never served by the CU's L1, answered by the L2 the rows were stored to
            for (int g = sl * (kGroups / 8); g < (sl + 1) * (kGroups / 8); g++) a += icp_acc[(size_t)g * 32 + w];
            for (int r0 = part_first + sl; r0 <= part_last; r0 += 64) {   // eight rows in flight per lane (a rolled loop waits for every load)
                unsigned long long t[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int r = r0 + 8 * k;
                    t[k] = __hip_atomic_load(&rgb_acc[(size_t)(r <= part_last ? r : part_last) * 32 + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (r > part_last) t[k] = 0;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) b += t[k];
            }
        } else
        for (int g = sl * (kGroups / 8); g < (sl + 1) * (kGroups / 8); g++) {
            a += icp_acc[(size_t)g * 32 + w];
            b += rgb_acc[(size_t)g * 32 + w];
        }
        s_part[0][sl][w] = a; s_part[1][sl][w] = b;
    }
#pragma unroll
    for (int q = 0; q < kPer; q++) if (tid + 256 * q < kWords) reinterpret_cast<unsigned*>(&s_od)[tid + 256 * q] = stw[q];
    __syncthreads();                                                                       // ---- barrier 1: state + partial sums in LDS
    SSTAMP(1);
    const bool cull = od->cull != 0;
    if (tid < 64) {
        // ============================== wave 0: the chain ==============================
        const int lane = tid;
        if (lane < 32) {
            unsigned long long a = 0, b = 0;
            for (int sl = 0; sl < 8; sl++) { a += s_part[0][sl][lane]; b += s_part[1][sl][lane]; }
            s_icp[lane] = a; s_rgb[lane] = b;
        }
        wave_lds_sync();
        SSTAMP(2);
        // uniform decisions (RGBDOdometry.cpp:371-392).  The RGB error (an f64 square root and division) only decides anything in the RGB-only
        // mode; otherwise it is a statistic, computed on another wave beside this chain
        bool active = od->level_done == 0, stop = false;
        const bool rgbOnly = od->rgbOnly != 0;
        const int rgbCount = (int)(long long)s_icp[29], rgbSigma = (int)(long long)s_icp[30];
        if (active && rgbOnly) {
            const float tmpError = (float)(sqrt((double)rgbSigma) / (double)rgbCount);
            if (tmpError > od->lastRGBError) { stop = true; active = false; }
        }
        if (lane == 0) { s_flags[0] = active ? 1 : 0; s_flags[1] = stop ? 1 : 0; }
        const bool useIcp = od->icp != 0, useRgb = od->rgb != 0;
        if (active) {
            // lane L < 36: A[L / 6][L % 6] of the combined system from word (min, max) of the two sums (reduce.cu:481-498: the host fills the
            // matrix symmetrically from the upper triangle); lanes 36..41: b[L - 36] from word (i, 6); lanes 42 / 43: the ICP residual pair
            double a = 0.0;
            {
                const int L = lane;
                const int i = L < 36 ? L / 6 : L - 36, j = L < 36 ? L - 6 * (L / 6) : 6;
                const int lo = i < j ? i : j, hi = i < j ? j : i;
                int t = 7 * lo - ((lo * (lo - 1)) >> 1) + (hi - lo);
                if (L >= 42) t = L == 42 ? 27 : 28;
                if (L >= 44) t = 0;
                const int bits_icp = icp_fix >= 0 ? icp_fix : (L == 42 ? 44 : (lo < 3 ? 20 : 17) + (hi < 3 ? 20 : hi < 6 ? 17 : 22));
                const int bits_rgb = rgb_fix_bits(sigma_val_from(rgbCount, rgbSigma, od->rgbOnly));
                const long long qi = (long long)s_icp[t], qr = (long long)s_rgb[t];
                const float vi = useIcp ? fix_to_f32(qi, bits_icp) : 0.f, vr = useRgb ? fix_to_f32(qr, bits_rgb) : 0.f;
                if (L < 42) {
                    const bool isA = L < 36;
                    if (useIcp && useRgb) {
                        const double w = od->icpWeight;
                        a = isA ? (double)vr + w * w * (double)vi : (double)vr + w * (double)vi;
                    } else a = useIcp ? (double)vi : (double)vr;
                    if (isA) od->stats.lastA[L] = a;
                    else od->stats.lastb[L - 36] = a;
                } else if (useIcp) {
                    if (L == 42) od->residual[0] = vi;
                    else if (L == 43) od->residual[1] = (float)qi;
                }
            }
            wave_lds_sync();
            SSTAMP(3);
            ldlt_solve6_wave(lane < 42 ? a : 0.0, s_result, 2.2250738585072014e-308, lane);   // (lanes 36..41: b)
            wave_lds_sync();
            SSTAMP(4);
            // computeUpdateSE3 (OdometryProvider.h:69-89).  Rodrigues (:32-67): theta, its sine / cosine and 1 / theta are the same in every
            // lane; lane q < 16 then forms ITS entry of the update matrix [R | t; 0 0 0 1]
            {
                double rx = s_result[3], ry = s_result[4], rz = s_result[5];
                const double theta = sqrt(rx * rx + ry * ry + rz * rz);
                const int q = lane & 15, r = q >> 2, c = q & 3;
                double e = (r == c) ? 1.0 : 0.0;   // identity (theta below epsilon), and the last row
                if (theta >= 2.2204460492503131e-16 && r < 3 && c < 3) {
                    double sn, cs;
                    det_sincos(theta, &sn, &cs);
                    const double c1 = 1.0 - cs;
                    const double itheta = 1.0 / theta;
                    rx *= itheta; ry *= itheta; rz *= itheta;
                    const double ra = r == 0 ? rx : (r == 1 ? ry : rz), rb = c == 0 ? rx : (c == 1 ? ry : rz);
                    const double rrt = (r <= c) ? ra * rb : rb * ra;                       // {rx rx, rx ry, rx rz, rx ry, ry ry, ry rz, rx rz, ry rz, rz rz}
                    // [r]_x = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0}
                    const int k = r * 3 + c;
                    const double rxm = k == 1 ? -rz : k == 2 ? ry : k == 3 ? rz : k == 5 ? -rx : k == 6 ? -ry : k == 7 ? rx : 0.0;
                    const double I = (r == c) ? 1.0 : 0.0;
                    e = cs * I + c1 * rrt + sn * rxm;
                }
                if (r < 3 && c == 3) e = s_result[r];
                if (lane < 16) s_upd[q] = e;
            }
            wave_lds_sync();
            SSTAMP(5);
            if (lane < 16) {  // mul44(upd, resultRt) element (i, j)
                const int i = lane >> 2, j = lane & 3;
                double sacc = s_upd[i * 4 + 0] * od->resultRt[0 * 4 + j];
                sacc = sacc + s_upd[i * 4 + 1] * od->resultRt[1 * 4 + j];
                sacc = sacc + s_upd[i * 4 + 2] * od->resultRt[2 * 4 + j];
                sacc = sacc + s_upd[i * 4 + 3] * od->resultRt[3 * 4 + j];
                s_Rt[lane] = sacc;   // (staging: every lane has read the old resultRt before anybody overwrites it)
            }
            wave_lds_sync();
            if (lane < 16) od->resultRt[lane] = s_Rt[lane];
            wave_lds_sync();
            // pose composition in f32 (RGBDOdometry.cpp:449-461): Ro / to = (float) resultRt, Rinv = Ro^T, tinv = -(Rinv to),
            // Rcurr = Rprev Rinv, tcurr = Rprev tinv + tprev -- lane q < 9: Rcurr[q], lanes 9..11: tcurr
            if (lane < 12) {
                const double* nrt = od->resultRt;
                if (lane < 9) {
                    const int i = lane / 3, j = lane - 3 * (lane / 3);
                    // Rinv[k * 3 + j] = Ro[j * 3 + k] = (float) nrt[j * 4 + k]
                    od->Rcurr[lane] = od->Rprev[i * 3 + 0] * (float)nrt[j * 4 + 0] + od->Rprev[i * 3 + 1] * (float)nrt[j * 4 + 1] + od->Rprev[i * 3 + 2] * (float)nrt[j * 4 + 2];
                } else {
                    const int r = lane - 9;
                    const float to0 = (float)nrt[0 * 4 + 3], to1 = (float)nrt[1 * 4 + 3], to2 = (float)nrt[2 * 4 + 3];
                    float tinv[3];
#pragma unroll
                    for (int k = 0; k < 3; k++)   // Rinv[k * 3 + m] = Ro[m * 3 + k]
                        tinv[k] = -((float)nrt[0 * 4 + k] * to0 + (float)nrt[1 * 4 + k] * to1 + (float)nrt[2 * 4 + k] * to2);
                    od->tcurr[r] = (od->Rprev[r * 3 + 0] * tinv[0] + od->Rprev[r * 3 + 1] * tinv[1] + od->Rprev[r * 3 + 2] * tinv[2]) + od->tprev[r];
                }
            }
        }
        SSTAMP(6);
    } else if (next_level >= 0 && tid == 64) {  // another wave: intrinsics of the next iteration's level
        double K[9], Kinv[9];
        k_matrix(cam_level(od->intr, next_level), K);
        inv33<double>(K, Kinv);
        for (int k = 0; k < 9; k++) { s_K[k] = K[k]; s_Kinv[k] = Kinv[k]; }
    }
    __syncthreads();                                                                       // ---- barrier 2: the new pose; K of the next level
    if (tid < 64) {
        const int lane = tid;
        if (next_level >= 0) {  // prepare_iteration(od, next_level): Rt = inv44_affine(resultRt), krkInv = K R K^-1, kt = K t
            // inv33 by cofactors (cf_device.h): the determinant in every lane, lane k < 9 its entry (L[p] L[q] - L[r] L[s]) / det
            const double* a = od->resultRt;
            const double L0 = a[0], L1 = a[1], L2 = a[2], L3 = a[4], L4 = a[5], L5 = a[6], L6 = a[8], L7 = a[9], L8 = a[10];
            const double c00 = L4 * L8 - L5 * L7, c01 = L5 * L6 - L3 * L8, c02 = L3 * L7 - L4 * L6;
            const double det = L0 * c00 + L1 * c01 + L2 * c02;
            const double id = 1.0 / det;
            const double Lm[9] = {L0, L1, L2, L3, L4, L5, L6, L7, L8};
            // o[k] = (L[P] * L[Q] - L[R] * L[S]) * id:  k: 0 (4,8,5,7) 1 (2,7,1,8) 2 (1,5,2,4) 3 (5,6,3,8) 4 (0,8,2,6) 5 (2,3,0,5) 6 (3,7,4,6) 7 (1,6,0,7) 8 (0,4,1,3)
            const int k9 = lane < 9 ? lane : 0;
            double p1 = 0, q1 = 0, r1 = 0, s1 = 0;
#pragma unroll
            for (int m = 0; m < 9; m++) {
                constexpr int P[9] = {4, 2, 1, 5, 0, 2, 3, 1, 0}, Q[9] = {8, 7, 5, 6, 8, 3, 7, 6, 4}, Rr[9] = {5, 1, 2, 3, 2, 0, 4, 0, 1}, S[9] = {7, 8, 4, 8, 6, 5, 6, 7, 3};
                if (k9 == m) { p1 = Lm[P[m]]; q1 = Lm[Q[m]]; r1 = Lm[Rr[m]]; s1 = Lm[S[m]]; }
            }
            const double li = (p1 * q1 - r1 * s1) * id;
            if (lane < 9) s_Rt[(lane / 3) * 4 + (lane - 3 * (lane / 3))] = li;
            wave_lds_sync();
            if (lane < 9) {  // tmp = K * R
                const int i = lane / 3, j = lane - 3 * (lane / 3);
                s_tmp[lane] = s_K[i * 3 + 0] * s_Rt[0 * 4 + j] + s_K[i * 3 + 1] * s_Rt[1 * 4 + j] + s_K[i * 3 + 2] * s_Rt[2 * 4 + j];
            } else if (lane >= 16 && lane < 19) {  // translation of the inverse: -(Li row i . t)
                const int i = lane - 16;
                s_Rt[i * 4 + 3] = -(s_Rt[i * 4 + 0] * a[3] + s_Rt[i * 4 + 1] * a[7] + s_Rt[i * 4 + 2] * a[11]);
            }
            wave_lds_sync();
            if (lane < 9) {
                const int i = lane / 3, j = lane - 3 * (lane / 3);
                od->krkInv[lane] = (float)(s_tmp[i * 3 + 0] * s_Kinv[0 * 3 + j] + s_tmp[i * 3 + 1] * s_Kinv[1 * 3 + j] + s_tmp[i * 3 + 2] * s_Kinv[2 * 3 + j]);
            } else if (lane >= 16 && lane < 19) {
                const int r = lane - 16;
                od->kt[r] = (float)(s_K[r * 3 + 0] * s_Rt[3] + s_K[r * 3 + 1] * s_Rt[7] + s_K[r * 3 + 2] * s_Rt[11]);
            }
        } else {
            if (lane == 0 && od->rgb) {  // end of the schedule: divergence guard (RGBDOdometry.cpp:464-467)
                const float d0 = od->tcurr[0] - od->tprev[0], d1 = od->tcurr[1] - od->tprev[1], d2 = od->tcurr[2] - od->tprev[2];
                if ((double)sqrtf(d0 * d0 + d1 * d1 + d2 * d2) > 0.3) {
                    for (int k = 0; k < 9; k++) od->Rcurr[k] = od->Rprev[k];
                    for (int k = 0; k < 3; k++) od->tcurr[k] = od->tprev[k];
                }
            }
            if (lane == 0) {  // ... and the inverse of the pose the call ends with, for the index pass that is enqueued before the host sees it
                float pose[16], inv[16];
                for (int r = 0; r < 3; r++) { pose[r * 4 + 0] = od->Rcurr[r * 3 + 0]; pose[r * 4 + 1] = od->Rcurr[r * 3 + 1]; pose[r * 4 + 2] = od->Rcurr[r * 3 + 2]; pose[r * 4 + 3] = od->tcurr[r]; }
                pose[12] = 0; pose[13] = 0; pose[14] = 0; pose[15] = 1;   // (the facade's Model::pose after a tracking call: CoFusion::fetchTracking)
                inv44f(pose, inv);
                for (int q = 0; q < 16; q++) od->pose_inv[q] = inv[q];
            }
            // ... and the candidate range of a culled tracker: how many record slots each level needed goes back to the host (it sizes the
            // next call's residual workgroups), the accumulator is cleared for the next call's preparation
            if (lane >= 32 && lane < 35 && od->res_range) {
                unsigned* rr = od->res_range + 2 * (lane - 32);
                const unsigned lo_inv = rr[0], hi_p1 = rr[1];
                const int sh = __builtin_ctz(slot_px) - 8;
                od->res_seen[lane - 32] = hi_p1 ? (int)(((hi_p1 - 1u) >> sh) - ((~lo_inv) >> sh)) + 1 : 0;
                rr[0] = 0; rr[1] = 0;
            }
        }
        SSTAMP(7);
    } else if (tid >= 128 && tid < 192) {
        if (next_level >= 0 && cull) {  // an idle wave: the screen box under the new pose
            int ib[4]; float zb[2];
            screen_box(od->box_lo, od->box_hi, od->box_R, od->box_t, od->Rcurr, od->tcurr, od->intr, od->distThres, od->width, od->height, tid - 128, ib, zb);
            if (tid == 128) {
                od->stats.cull_box[0] = ib[0]; od->stats.cull_box[1] = ib[1]; od->stats.cull_box[2] = ib[2]; od->stats.cull_box[3] = ib[3];
                od->cull_z[0] = zb[0]; od->cull_z[1] = zb[1];
            }
        }
    } else if (tid == 192) {
        // another idle wave: the level's flags and the statistics (RGBDOdometry.cpp:371-392, 401-402) -- in the order of the serial code:
        // stop, statistics, end of level
        const bool active = s_flags[0] != 0, stop = s_flags[1] != 0;
        if (stop) od->level_done = 1;
        if (active) {
            const int rgbCount = (int)(long long)s_icp[29];
            const float tmpError = (float)(sqrt((double)(int)(long long)s_icp[30]) / (double)rgbCount);
            od->lastRGBError = tmpError;
            od->stats.last_rgb_error = tmpError; od->stats.last_rgb_count = (float)rgbCount;
            od->stats.last_icp_error = sqrtf(od->residual[0]) / od->residual[1];
            od->stats.last_icp_count = od->residual[1];
        }
        if (last_of_level) { od->level_done = 0; od->lastRGBError = 3.402823466e+38F; }
        od->solves += 1;   // (whether or not the iteration was active: the host counts launches)
    }
    __syncthreads();                                                                       // ---- barrier 3: everything the state will hold
    {
        // write-back of the mutable words; the hot block (refresh_hot, cf_kernels.h: GnHot) is derived on the way: hot word w <- its source field
        constexpr int o_icp = (int)(offsetof(OdomDev, icp) / 4), o_ld = (int)(offsetof(OdomDev, level_done) / 4), o_cz = (int)(offsetof(OdomDev, cull_z) / 4),
                      o_Rc = (int)(offsetof(OdomDev, Rcurr) / 4), o_tc = (int)(offsetof(OdomDev, tcurr) / 4), o_Ri = (int)(offsetof(OdomDev, Rprev_inv) / 4),
                      o_tp = (int)(offsetof(OdomDev, tprev) / 4), o_cb = (int)((offsetof(OdomDev, stats) + offsetof(cf_track_stats, cull_box)) / 4),
                      o_rgb = (int)(offsetof(OdomDev, rgb) / 4), o_ro = (int)(offsetof(OdomDev, rgbOnly) / 4), o_krk = (int)(offsetof(OdomDev, krkInv) / 4),
                      o_kt = (int)(offsetof(OdomDev, kt) / 4), o_hot = (int)(offsetof(OdomDev, hot) / 4);
        const unsigned* words = reinterpret_cast<const unsigned*>(&s_od);
        OdomDev* const twin = god_host ? god_host : s_od.host_twin;
        const bool to_twin = twin && next_level < 0;   // end of the schedule: the result (pose, statistics, fault word) goes to the tracker's pinned
                                                       // host copy as well -- the frame's host wait finds it there without a copy command on the stream
        for (int k = kMutableFrom + tid; k < kWords; k += 256) {
            unsigned v = words[k];
            const int w = k - o_hot;
            if (w >= 0 && w < 48) {
                const int src = w < 1 ? o_icp : w < 2 ? o_ld : w < 4 ? o_cz + (w - 2) : w < 13 ? o_Rc + (w - 4) : w < 16 ? o_tc + (w - 13) : w < 25 ? o_Ri + (w - 16)
                              : w < 28 ? o_tp + (w - 25) : w < 32 ? o_cb + (w - 28) : w < 33 ? o_rgb : w < 34 ? o_ro : w < 35 ? o_ld : w < 36 ? -1
                              : w < 45 ? o_krk + (w - 36) : o_kt + (w - 45);
                v = src >= 0 ? words[src] : 0u;
            }
            reinterpret_cast<unsigned*>(god)[k] = v;
            if (to_twin) reinterpret_cast<unsigned*>(twin)[k] = v;
        }
    }
    // zero the accumulators for the next iteration, 16 bytes per store (their sums were read in the first flight of loads; stores issued
    // in front of a barrier would make it wait a memory round trip)
    {
        ulonglong2* const zi = reinterpret_cast<ulonglong2*>(icp_acc);
        ulonglong2* const zr = reinterpret_cast<ulonglong2*>(rgb_acc);
        for (int k = tid; k < kGroups * 16; k += 256) { zi[k] = make_ulonglong2(0, 0); zr[k] = make_ulonglong2(0, 0); }
    }
    SSTAMP(8);
#ifdef CF_ABLATE
    if (g_solve_trace && threadIdx.x == 0 && blockIdx.x == 0) g_solve_iter++;
#endif
}

__global__ void __launch_bounds__(256) gn_solve_kernel(const GnArgs args, int next_level, int last_of_level)
{
    gn_solve_body(args.icp_gram ? -1 : kFixICP, args.od[blockIdx.x], args.icp_acc[blockIdx.x], args.rgb_acc[blockIdx.x], next_level, last_of_level, args.od_host[blockIdx.x], args.slot_px);
}

// RGB step over the per-workgroup record slots the residual pass left (grid: one workgroup per slot x models).  A slot holds at
// most 4 x producer-workgroup-size records and typically < 10 % of that; thread r reads record r of its slot speculatively together
// with the slot's count, so the pass has the same two dependent memory round trips as rgb_step_kernel on a tenth of the bytes.
//
// Measured and dropped (round 2, profiles/r02b): running this pass and the solve in ONE launch -- 32 workgroups per model reduce a
// global list, fence, arrive at a counter, workgroup 0 waits and solves.  22.6 us per launch against 6.3 + 8.4 us for the two
// separate kernels plus one boundary: the device-scope release fence and the arrival wait cost more than a kernel boundary does.
__device__ __forceinline__ void rgb_slot_step_body(const RgbArgs& ra, int n_slots)
{
    const RgbModelArgs m = ra.m[blockIdx.y];
    asm volatile("" :: "s"(m.st), "s"(m.icp_acc), "s"(m.rgb_acc), "s"(m.recs), "s"(m.slot_counts), "s"(m.res_range), "s"(m.cloud), "s"(m.dIdx), "s"(m.dIdy),
                 "s"(ra.cols), "s"(ra.rows), "s"(ra.slot_px), "s"(ra.sobelScale), "s"(ra.il.fx), "s"(ra.il.fy));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t slot0 = (size_t)blockIdx.x * ra.slot_px;
    // ONE flight of loads: the slot's count, this thread's record (speculative: valid if tid < n) and -- first wave -- the two words of the
    // accumulator groups that give sigma, all pinned in front of the uniform `n == 0` exit.  Without the pin the compiler sinks the record
    // and accumulator loads below that exit: three dependent round trips (count -> records + sums -> gathers) where two will do (round 6,
    // from the ISA: the `speculative` load of round 2 had not been speculative in the binary).
    unsigned n = m.slot_counts[blockIdx.x];
    const size_t Npx = (size_t)ra.cols * ra.rows;
    uint2 rc = m.recs[slot0 + tid < Npx ? slot0 + tid : Npx - 1];   // (unconditional, address clamped: a load under a branch is waited for at its end)
    unsigned long long g_cnt = 0, g_sig = 0;
    if (tid < 64) { g_cnt = m.icp_acc[(size_t)tid * 32 + 29]; g_sig = m.icp_acc[(size_t)tid * 32 + 30]; }   // kGroups == 64 == lanes
    asm volatile("" ::: "memory");   // (the vector loads are issued in front of the scalar round trip of the tracker's hot state, not behind it)
    SlotRange sr;
    const RgbHot hs = rgb_hot_and_range(ra, m, n_slots, sr);
    asm volatile("" : "+v"(n), "+v"(rc.x), "+v"(rc.y), "+v"(g_cnt), "+v"(g_sig));
    // (a slot outside a culled tracker's candidate range was not visited by the residual pass: its count is stale, it holds nothing)
    if ((int)blockIdx.x < sr.first || (int)blockIdx.x > sr.last) n = 0;
    if (!(hs.rgb && !hs.level_done) || n == 0) return;  // uniform
    __shared__ float s_sigma;
    if (tid < 64) {
        // sigma_val_from takes the LOW 32 bits of the two totals (the reference's `int` count and sigma), and the low word of a sum is the
        // wrapping sum of the low words: a 32-bit reduction over the 64 groups -- four DPP steps inside the rows of 16 lanes, the four row
        // totals added on the scalar unit -- instead of twelve dependent LDS shuffles of 64-bit values (0.35 us of this launch)
        const unsigned c32 = wave_sum_u32((unsigned)g_cnt), s32 = wave_sum_u32((unsigned)g_sig);
        if (tid == 0) s_sigma = sigma_val_from((int)c32, (int)s32, hs.rgbOnly);
    }
    __syncthreads();
    const float sigma = s_sigma;
    const int F = rgb_fix_bits(sigma);
    const float lim = ldexpf(1.0f, (50 - F) / 2), scale = ldexpf(1.0f, F);
    unsigned long long acc[32];
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = 0;
    unsigned long long terms = 0;
    for (unsigned r = tid; r < n; r += 256) {
        if (r >= 256) rc = m.recs[slot0 + r];
        float row[7];
        rgb_step_row(ra, m, sigma, (float)((int)(rc.y >> 22) - 256), (int)rc.x, (int)(rc.y & 0x3fffffu), row);
        se3_accumulate_dyn(row, acc, lim, scale);
        terms++;
    }
    unsigned long long v = 0;
    if (__any(terms != 0)) {
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] -= terms * kMagicBits;
        acc[28] = terms;
        v = wave_reduce32_u64(acc, lane);
    }
    block_commit32<4>(v, lane, wave, 4, m.rgb_acc + (size_t)(blockIdx.x % kGroups) * 32);
}
__global__ void __launch_bounds__(256) rgb_slot_step_kernel(const RgbArgs ra, int n_slots) { rgb_slot_step_body(ra, n_slots); }
// ... and, on the last level-0 iteration, the error surfaces of the culled trackers in the SAME launch (workgroups behind the record
// slots; until late in round 6 icp_error_surface_kernel ran as a launch of its own between the {ICP || residual} launch and this one:
// 6.5 us + a launch boundary on the Gauss-Newton chain of every frame).  Both read the tracker state the solve has not touched yet.
__global__ void __launch_bounds__(256) rgb_slot_step_err_kernel(const RgbArgs ra, int n_slots, const IcpArgs e)
{
    if ((int)blockIdx.x >= n_slots) { icp_error_surface_body(e, e.m[blockIdx.y], (int)blockIdx.x - n_slots); return; }
    rgb_slot_step_body(ra, n_slots);
}

// MODE 2 (round 5): the RGB step and the solve in ONE launch.  Rounds 2-3 measured this twice with device-scope synchronisation and lost
// both times (a release fence per workgroup writes the XCD's L2 back; returning device-scope atomics cost a memory round trip each).
// What round 4's SO(3) kernel showed is that workgroups of ONE XCD can meet in its L2 for ~1.5 us: so the step workgroups of tracker m
// are placed on XCD m mod 8 (hardware workgroup b runs on XCD b mod 8: tools/microbench/xcc_map.hip), add their sums with
// workgroup-scope atomics -- performed in that L2 --, wait for them, and take a ticket there.  Nobody waits for anybody: the workgroup
// that draws the last ticket runs the solve (gn_solve_body reads the RGB sums back from the L2 with agent-scope loads), the others
// leave.  What the solve writes (state, cleared accumulators, the ticket counter) is written back at the end of the kernel like any
// other store.  One launch boundary (~2.5 us) and the solve kernel's own ramp less per Gauss-Newton iteration: 57 -> 38 launches per
// frame.  The sums are integers: which workgroup adds what, and who solves, does not change a bit.
// Grid: 8 x n_quads x ceil(n / 8) workgroups, n_quads = ceil(n_slots / 2) (two record slots per workgroup), b = 8 * (quad + n_quads * (m / 8)) + m % 8.
__global__ void __launch_bounds__(256) rgb_step_solve_kernel(const RgbArgs ra, So3Sync* __restrict__ syncs, int n_slots, int n_quads, IDiv quad_div,
                                                             int n, int icp_fix, int next_level, int last_of_level)
{
    const int b = (int)blockIdx.x;
#ifdef CF_ABLATE
    unsigned long long* const tr = g_icp_trace ? g_icp_trace + (size_t)b * 8 : nullptr;
    if (tr && threadIdx.x == 0) { tr[0] = wall_clock64(); tr[1] = tr[2] = tr[3] = tr[4] = 0; tr[5] = 0xffff; }
#define STAMP(k) do { if (tr && threadIdx.x == 0) tr[k] = wall_clock64(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
    const int q = b >> 3, hi = n_quads > 1 ? idiv(q, quad_div) : q;
    const int model = (b & 7) + 8 * hi, quad = q - hi * n_quads;
    if (model >= n) return;
    const RgbModelArgs& m = ra.m[model];
    const RgbHot hs = rgb_hot((StatePtr)m.st);
    SlotRange sr = residual_slot_range(ra, m, n_slots);
    const bool no_slot = sr.last < sr.first;   // a culled tracker without a single candidate: its first workgroup stands in (and solves)
    if (no_slot) { sr.first = 0; sr.last = 0; }
    const int qf = sr.first >> 1, ql = sr.last >> 1;
    if (quad < qf || quad > ql) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Two record slots per workgroup, two waves per slot (`quad`: the pair's index).  The tracker's workgroups share ONE XCD: a workgroup
    // per slot (round 4's shape) makes the background's 300 slots two rounds of residency there (measured: +4 us), a WAVE per slot leaves a
    // slot with 700 records to eleven dependent passes of one wave (measured: +8 us).  150 workgroups x 4 waves fit the XCD in one round.
    const int slot = quad * 2 + (wave >> 1), half = tid & 127;
    const bool slot_ok = !no_slot && slot >= sr.first && slot <= sr.last;
    const size_t slot0 = (size_t)slot * ra.slot_px;
    const unsigned nrec = slot_ok ? m.slot_counts[slot] : 0u;
    uint2 rc = make_uint2(0, 0);
    if (slot_ok && slot0 + half < (size_t)ra.cols * ra.rows) rc = m.recs[slot0 + half];   // speculative: valid if half < nrec
    if (hs.rgb && !hs.level_done) {  // uniform
        unsigned long long v = 0;
        if (nrec != 0) {   // (wave-uniform)
            const long long cnt = (long long)group_sum(m.icp_acc, 29, lane);
            const long long sg = (long long)group_sum(m.icp_acc, 30, lane);
            const float sigma = sigma_val_from((int)cnt, (int)sg, hs.rgbOnly);
            const int F = rgb_fix_bits(sigma);
            const float lim = ldexpf(1.0f, (50 - F) / 2), scale = ldexpf(1.0f, F);
            unsigned long long acc[32];
#pragma unroll
            for (int k = 0; k < 32; k++) acc[k] = 0;
            unsigned long long terms = 0;
            for (unsigned r = half; r < nrec; r += 128) {
                if (r >= 128) rc = m.recs[slot0 + r];
                float row[7];
                rgb_step_row(ra, m, sigma, (float)((int)(rc.y >> 22) - 256), (int)rc.x, (int)(rc.y & 0x3fffffu), row);
                se3_accumulate_dyn(row, acc, lim, scale);
                terms++;
            }
            if (__any(terms != 0)) {
#pragma unroll
                for (int k = 0; k < 28; k++) acc[k] -= terms * kMagicBits;
                acc[28] = terms;
                v = wave_reduce32_u64(acc, lane);
            }
        }
        STAMP(1);
        block_store32(v, lane, wave, m.rgb_acc + (size_t)quad * 32);
    } else if (tid < 32) m.rgb_acc[(size_t)quad * 32 + tid] = 0;
#ifdef CF_ABLATE
    if (tr && threadIdx.x == 0) tr[5] = (unsigned long long)model;
#endif
    // The tickets, drawn by the wave that issued the atomics, once the L2 has taken them.  Two levels: returning atomics on ONE address
    // take the L2 ~17 ns each, so a workgroup draws from the counter of its quad's residue class mod kStepSubs (each in a cache line of
    // its own), and the last of a class draws from the tracker's top counter.
    __shared__ int s_last;
    So3Sync* const sync = syncs + model;
    if (tid < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(2);
        if (tid == 0) {
            static_assert(kStepSubs == 16, "the class arithmetic below shifts by 4");
            const int j = quad & (kStepSubs - 1);
            // quads = j (mod kStepSubs) inside [qf, ql]; classes that have any
            const unsigned in_class = (unsigned)(((ql - j) >> 4) - ((qf - 1 - j) >> 4));
            const int span = ql - qf + 1;
            const unsigned classes = (unsigned)(span < kStepSubs ? span : kStepSubs);
            int last = 0;
            if (__hip_atomic_fetch_add(&sync->step_sub[j][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == in_class - 1u) {
                sync->step_sub[j][0] = 0;   // (every ticket of this class is drawn; the next launch finds the counter cleared)
                if (__hip_atomic_fetch_add(&sync->step_top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == classes - 1u) { sync->step_top = 0; last = 1; }
            }
            s_last = last;
        }
    }
    STAMP(3);
    __syncthreads();
    if (!s_last) return;
    OdomDev* const god = m.st;
    gn_solve_body<true>(icp_fix, god, m.icp_acc, m.rgb_acc, next_level, last_of_level, nullptr, ra.slot_px, qf, ql);
    STAMP(4);
#undef STAMP
}

// total of the grouped accumulator -> out[32] (stand-alone steps)
__global__ void __launch_bounds__(64) acc_total_kernel(const unsigned long long* __restrict__ acc, unsigned long long* __restrict__ out)
{
    for (int w = 0; w < 32; w++) {
        const unsigned long long v = group_sum(acc, w, threadIdx.x);
        if (threadIdx.x == 0) out[w] = v;
    }
}

// Split reductions (one tracker's image rows over several GPUs): a rank's partial sums are FOLDED before they travel -- the 64 accumulator
// groups of every split tracker summed into group 0, the other groups cleared -- so that the all-reduce of the Gauss-Newton loop carries
// the 32 words of the 6x6 system (256 bytes: north_star's "RCCL all-reduce of the 6x6 system") instead of the 16 KB of grouped partial
// sums it carried until round 5.  Everybody downstream (the RGB step's sigma, the solve) sums the groups as before
// and finds the totals in group 0 and zeros elsewhere: integer sums, the same bits.
struct FoldArgs { unsigned long long* acc[kMaxBatch]; };
__global__ void __launch_bounds__(64) acc_fold_kernel(const FoldArgs a)
{
    unsigned long long* __restrict__ acc = a.acc[blockIdx.x];
    const int lane = threadIdx.x;
    unsigned long long tot = 0;
#pragma unroll 4
    for (int w = 0; w < 32; w++) { const unsigned long long v = group_sum(acc, w, lane); if (lane == w) tot = v; }
    for (int w = 0; w < 32; w++) acc[(size_t)lane * 32 + w] = 0;   // (every load above has been consumed by a shuffle: the wave is past them)
    if (lane < 32) acc[lane] = tot;                                  // (same wave, program order: behind the zeroes of group 0)
}

// ------------------------------------------------------------------------------ launchers ----
// ev0/ev1 (nullable) receive the dispatch's own begin/end timestamps (the figures rocprofv3 reports), not the
// stream time around it.  n_res_blocks > 0 appends the RGB residual workgroups to the same launch.
#ifdef CF_ABLATE
static IcpArgs g_last_icp_args; static int g_last_n_icp_blocks = 0, g_last_grid = 0;
#endif
template <int TAG, bool GRAM>
static void launch_icp_kernel_arith(hipStream_t s, IcpLaunch cfg, const IcpArgs& args_in, const RgbArgs& ra_in, bool icp, int n_res_blocks, int n,
                                    hipEvent_t ev0, hipEvent_t ev1)
{
    IcpArgs args = args_in; args.cdiv = make_idiv(args.cols);
    RgbArgs ra = ra_in; ra.cdiv = make_idiv(ra.cols);
    const int N = ((args.row_end > 0 ? args.row_end : args.rows) - args.row_begin) * args.cols;
    const int per_block = cfg.threads * cfg.ppt;
    const int nlog = (N + per_block - 1) / per_block;
    const int full = icp ? ((nlog + 7) / 8) * 8 : 0;
    // culled models: the workgroups the caller asked for (IcpModelArgs::box_blocks, box_blocks_for), at most the whole image's.  The
    // mapping walks its runs one pixel per lane whatever the launch's pixels per lane are (those apply to unculled models), is built for
    // the product form, and not used on the error-surface iteration (which writes every pixel) nor with row bands.
    int blocks[kMaxBatch];
    for (int m = 0; m < n; m++) {
        IcpModelArgs& ma = args.m[m];
        const bool ok = icp && ma.cull && ma.box_blocks > 0 && !GRAM && (!(args.flags & 1) || (args.flags & 2)) && args.row_end == 0 && ma.row_end == 0;
        ma.box_blocks = ok ? (ma.box_blocks < full ? ((ma.box_blocks + 7) / 8) * 8 : full) : 0;
        blocks[m] = ok ? ma.box_blocks : full;
    }
    // residual workgroups: one per record slot of the level, or the caller's RgbModelArgs::res_blocks (culled trackers: the slots between
    // the first and the last candidate the previous call saw, residual_blocks_for)
    for (int m = 0; m < n && n_res_blocks > 0; m++) {
        RgbModelArgs& rm = ra.m[m];
        if (!ra.compact || !rm.res_range || rm.res_blocks <= 0 || rm.res_blocks > n_res_blocks) rm.res_blocks = n_res_blocks;
        if (!ra.compact) rm.res_range = nullptr;
    }
    // ORDER OF THE SLOTS = order of dispatch: the culled trackers' runs (the longest chain of dependent round trips: box, depth interval,
    // planes, occupancy, gather), the unculled ICP reductions, then the residual passes.  Seven orders were measured in round 5 (longest work
    // first, residual passes first, ...; a diagnostics build reads CF_ICP_ORDER): 13.8-14.0 us for this one, 14.0-14.5 us for the others --
    // the launch is a little over one round of resident workgroups and its length is the sum of everybody's residency, not its tail
    // (tools/icp_trace_summary.py).  Every slot is padded to a multiple of 8 workgroups so that the hardware workgroup id and the
    // slot-local index agree on the XCD; the padding leaves at once.
    int total = 0, slot = 0, n_icp_blocks = 0;
    auto add = [&](int m, bool residual, int count) {
        total += ((count + 7) / 8) * 8;
        args.slot_end[slot] = total; args.slot_desc[slot] = (unsigned char)(m | (residual ? kResidualSlot : 0)); slot++;
        if (!residual) n_icp_blocks += count;
    };
#ifdef CF_ABLATE
    static const int order = getenv("CF_ICP_ORDER") ? atoi(getenv("CF_ICP_ORDER")) : 0;
#else
    constexpr int order = 0;
#endif
    const bool res = n_res_blocks > 0;
    auto culled = [&](int m) { return args.m[m].box_blocks > 0 || (args.m[m].cull && ra.m[m].res_range); };
    if (order == 0) {          // rounds 3-4: culled ICP, unculled ICP, residual passes
        for (int pass = 0; pass < 2 && icp; pass++) for (int m = 0; m < n; m++) if ((args.m[m].box_blocks > 0) == (pass == 0)) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) add(m, true, ra.m[m].res_blocks);
    } else if (order == 4) {   // culled ICP, unculled residual, unculled ICP, culled residual
        for (int m = 0; m < n && icp; m++) if (args.m[m].box_blocks > 0) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) if (!culled(m)) add(m, true, ra.m[m].res_blocks);
        for (int m = 0; m < n && icp; m++) if (!(args.m[m].box_blocks > 0)) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) if (culled(m)) add(m, true, ra.m[m].res_blocks);
    } else if (order == 5) {   // culled ICP, unculled ICP, culled residual, unculled residual
        for (int pass = 0; pass < 2 && icp; pass++) for (int m = 0; m < n; m++) if ((args.m[m].box_blocks > 0) == (pass == 0)) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) if (culled(m)) add(m, true, ra.m[m].res_blocks);
        for (int m = 0; m < n && res; m++) if (!culled(m)) add(m, true, ra.m[m].res_blocks);
    } else if (order == 6) {   // unculled ICP, culled ICP, residual passes
        for (int pass = 0; pass < 2 && icp; pass++) for (int m = 0; m < n; m++) if ((args.m[m].box_blocks > 0) == (pass == 1)) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) add(m, true, ra.m[m].res_blocks);
    } else if (order == 2) {   // unculled ICP, unculled residual, culled ICP, culled residual
        for (int m = 0; m < n && icp; m++) if (!(args.m[m].box_blocks > 0)) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) if (!culled(m)) add(m, true, ra.m[m].res_blocks);
        for (int m = 0; m < n && icp; m++) if (args.m[m].box_blocks > 0) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) if (culled(m)) add(m, true, ra.m[m].res_blocks);
    } else if (order == 3) {   // unculled residual, culled ICP, unculled ICP, culled residual
        for (int m = 0; m < n && res; m++) if (!culled(m)) add(m, true, ra.m[m].res_blocks);
        for (int m = 0; m < n && icp; m++) if (args.m[m].box_blocks > 0) add(m, false, blocks[m]);
        for (int m = 0; m < n && icp; m++) if (!(args.m[m].box_blocks > 0)) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) if (culled(m)) add(m, true, ra.m[m].res_blocks);
    } else {                   // longest first
        for (int m = 0; m < n && res; m++) if (!culled(m)) add(m, true, ra.m[m].res_blocks);
        for (int m = 0; m < n && icp; m++) if (!(args.m[m].box_blocks > 0)) add(m, false, blocks[m]);
        for (int m = 0; m < n && icp; m++) if (args.m[m].box_blocks > 0) add(m, false, blocks[m]);
        for (int m = 0; m < n && res; m++) if (culled(m)) add(m, true, ra.m[m].res_blocks);
    }
    args.slots_used = slot;
    for (; slot < kMaxSlots; slot++) { args.slot_end[slot] = 0x7fffffff; args.slot_desc[slot] = 0; }
    const dim3 grid(total);
#ifdef CF_ABLATE
    g_last_icp_args = args; g_last_n_icp_blocks = n_icp_blocks; g_last_grid = (int)grid.x;
#endif
    const unsigned lds = GRAM ? (unsigned)(cfg.threads / 64) * kGramWaveDwords * sizeof(int) : 0u;
    if (!ev0 && !ev1) {  // plain launches (also what a stream capture records)
        switch (cfg.ppt) {
            case 4: icp_reduce_kernel<4, TAG, GRAM><<<grid, dim3(cfg.threads), lds, s>>>(args, ra, n_icp_blocks); break;
            case 2: icp_reduce_kernel<2, TAG, GRAM><<<grid, dim3(cfg.threads), lds, s>>>(args, ra, n_icp_blocks); break;
            default: icp_reduce_kernel<1, TAG, GRAM><<<grid, dim3(cfg.threads), lds, s>>>(args, ra, n_icp_blocks); break;
        }
        return;
    }
    switch (cfg.ppt) {
        case 4: hipExtLaunchKernelGGL((icp_reduce_kernel<4, TAG, GRAM>), grid, dim3(cfg.threads), lds, s, ev0, ev1, 0, args, ra, n_icp_blocks); break;
        case 2: hipExtLaunchKernelGGL((icp_reduce_kernel<2, TAG, GRAM>), grid, dim3(cfg.threads), lds, s, ev0, ev1, 0, args, ra, n_icp_blocks); break;
        default: hipExtLaunchKernelGGL((icp_reduce_kernel<1, TAG, GRAM>), grid, dim3(cfg.threads), lds, s, ev0, ev1, 0, args, ra, n_icp_blocks); break;
    }
}
template <int TAG>
static void launch_icp_kernel(hipStream_t s, IcpLaunch cfg, const IcpArgs& args, const RgbArgs& ra, bool icp, int n_res_blocks, int n,
                              hipEvent_t ev0, hipEvent_t ev1)
{
    if (cfg.gram) launch_icp_kernel_arith<TAG, true>(s, cfg, args, ra, icp, n_res_blocks, n, ev0, ev1);
    else launch_icp_kernel_arith<TAG, false>(s, cfg, args, ra, icp, n_res_blocks, n, ev0, ev1);
}

static void launch_icp_rgbres(hipStream_t s, IcpLaunch cfg, const IcpArgs& args, const RgbArgs& ra, bool icp, bool rgb, int n, int level,
                              hipEvent_t ev0, hipEvent_t ev1)
{
    // pixels per lane of unculled trackers, 0 = the library's choice: two at level 0 in the product form -- half the waves of the launch's
    // largest slot for the same loads in flight, 12.4 against 13.0 us with five trackers since the accumulators live in LDS (round 5; with
    // them in registers two pixels cost a wave of occupancy and lost) --, one on the small levels and in the Gram form
    if (cfg.ppt == 0) cfg.ppt = (level == 0 && !cfg.gram) ? 2 : 1;
    const int N = (icp ? args.cols * args.rows : ra.cols * ra.rows);
    const int res_per_block = cfg.threads * (ra.compact ? 4 : 1);  // compact list pass: four pixels per thread
    const int n_res_blocks = rgb ? (N + res_per_block - 1) / res_per_block : 0;
    // distinct symbols per pyramid level so that rocprofv3 --stats separates them
    // (tag = level for one model, level + 4 for lock-step batches of several models)
    if (n > 1) {
        if (level == 0) launch_icp_kernel<4>(s, cfg, args, ra, icp, n_res_blocks, n, ev0, ev1);
        else if (level == 1) launch_icp_kernel<5>(s, cfg, args, ra, icp, n_res_blocks, n, ev0, ev1);
        else launch_icp_kernel<6>(s, cfg, args, ra, icp, n_res_blocks, n, ev0, ev1);
    } else if (level == 0) launch_icp_kernel<0>(s, cfg, args, ra, icp, n_res_blocks, n, ev0, ev1);
    else if (level == 1) launch_icp_kernel<1>(s, cfg, args, ra, icp, n_res_blocks, n, ev0, ev1);
    else launch_icp_kernel<2>(s, cfg, args, ra, icp, n_res_blocks, n, ev0, ev1);
}

void launch_icp_level(hipStream_t s, IcpLaunch cfg, const IcpArgs& args, int n, int level, hipEvent_t ev0, hipEvent_t ev1)
{
    launch_icp_rgbres(s, cfg, args, RgbArgs{}, true, false, n, level, ev0, ev1);
}

// Per Gauss-Newton iteration: ONE launch for {ICP reduction || RGB residual}, then rgbStep, then the one-workgroup
// solve.  Letting rgbStep's last workgroup run the solve was measured twice and lost both times: with a device-scope
// release fence per workgroup (it writes back the XCD's L2: 53 us instead of 6 + 8 us), and with returning atomics +
// a ticket instead of the fence (correct and deterministic, but 946 instead of 1061 frames/s: every workgroup then
// waits for its atomics' round trip).  A third experiment ran the WHOLE 4/5/10 schedule as one persistent launch (256 resident
// workgroups per model, registers carrying the partial sums across a thread's pixels, two grid barriers per iteration built
// from integer atomics + an arrival counter, every workgroup repeating the solve on its own LDS copy of the state): bit-exact
// for all option sets, but 640 us per frame instead of 420 us for the 57 launches -- an in-kernel timer showed ~8 us per
// grid barrier, a chain of about five device-scope memory round trips of ~1.5 us each across the XCDs, whereas a dependent
// launch costs ~2.5 us (tools/microbench/launch_floor.hip).  On this part the kernel boundary IS the cheapest grid barrier.
// Kept as separate launches.
bool launch_gn_track(hipStream_t s, IcpLaunch cfg, const TrackerStates& states, So3Sync* so3_syncs, const GnHook* hook, const IcpArgs icp_args[3],
                     const RgbArgs rgb_args[3], int n, int width, int height, bool so3, bool pyramid, bool fast_odom, bool rgb, bool icp, int mode,
                     ProfSink* prof, OdomDev* const* h_states, const RgbPrepBatch* prep)
{
    int iterations[3];
    iterations[0] = fast_odom ? 3 : 10;
    iterations[1] = pyramid ? 5 : 0;
    iterations[2] = pyramid ? 4 : 0;
    int first_level = 2;
    while (first_level > 0 && iterations[first_level] == 0) first_level--;
    {
        const int gx = so3 ? 8 * kSo3Blocks : 1;  // (8x: one XCD per model)
        static const RgbPrepBatch none{};
        const int prep_bx = prep ? prep->m[0].L.blk_end[2] : 0;
        so3_prealign_kernel<<<gx * n + prep_bx * n, 256, 0, s>>>(states, so3_syncs, so3 ? 1 : 0, first_level, gx, gx * n, prep ? *prep : none, prep_bx);
    }
    GnArgs gn{};
    gn.icp_gram = cfg.gram;
    gn.slot_px = cfg.threads * 4;
    for (int m = 0; m < n; m++) {
        gn.od[m] = const_cast<OdomDev*>(icp_args[0].m[m].st);
        gn.icp_acc[m] = icp_args[0].m[m].acc;
        gn.rgb_acc[m] = rgb_args[0].m[m].rgb_acc;
        gn.od_host[m] = h_states ? h_states[m] : nullptr;
    }
    const bool slots = mode != 0;
    // the error surfaces of the last level-0 iteration: those of culled trackers by a launch of its own behind that iteration's
    // {ICP || residual} launch (these trackers stay culled in their ICP pass: IcpArgs::flags 3), the others by their ICP pass (flags 1)
    bool any_culled = false;
    for (int m = 0; m < n; m++) any_culled = any_culled || (icp_args[0].m[m].cull && icp_args[0].m[m].err);
    const bool err_aside = icp && any_culled;
    bool hook_failed = false;
    for (int i = 2; i >= 0; i--) {
        const int N = (width >> i) * (height >> i);
        for (int j = 0; j < iterations[i]; j++) {
            const bool last_of_level = (j == iterations[i] - 1);
            int next_level = i;
            if (last_of_level) {
                next_level = i - 1;
                while (next_level >= 0 && iterations[next_level] == 0) next_level--;
            }
            RgbArgs ra = rgb_args[i];
            ra.compact = slots ? 1 : 0;
            ra.slot_px = cfg.threads * 4;
            {
                // the roofline figure is quoted on the dominant kernel: the level-0 instantiation
                const bool timed = prof && prof->enabled && i == 0 && prof->used + 4 <= prof->capacity;
                IcpArgs a = icp_args[i];
                a.flags = (i == 0 && last_of_level) ? (err_aside ? 3 : 1) : 0;
                launch_icp_rgbres(s, cfg, a, ra, icp, rgb, n, i, timed ? prof->events[prof->used] : nullptr,
                                  timed ? prof->events[prof->used + 1] : nullptr);
                if (timed) {
                    prof->used += 2;
                    prof->bytes += (uint64_t)N * ((icp ? 24 + 24 * (uint64_t)n : 0) + (rgb ? (slots ? kRgbResidualBytesCompact : kRgbResidualBytes) * (uint64_t)n : 0));
                    prof->launches += 1;
                }
            }
            // (the culled trackers' error surfaces ride in the RGB step's launch when there is one: rgb_slot_step_err_kernel)
            const bool err_here = err_aside && i == 0 && last_of_level;
            const bool err_with_step = err_here && rgb && slots && mode != 2;
            if (err_here && !err_with_step) {
                IcpArgs e = icp_args[i]; e.cdiv = make_idiv(e.cols);
                icp_error_surface_kernel<<<dim3((N + 255) / 256, n), 256, 0, s>>>(e);
            }
            if (hook && hook->fn) {  // split reductions: the partial sums of this rank's row band become the totals on every rank
                FoldArgs fa{}; int nf = 0;
                for (int m = 0; m < n; m++) if (hook->split[m]) fa.acc[nf++] = gn.icp_acc[m];
                if (nf) acc_fold_kernel<<<nf, 64, 0, s>>>(fa);   // 64 groups -> group 0: 256 bytes per tracker cross the links, not 16 KB
                for (int m = 0; m < n; m++)
                    if (hook->split[m] && hook->fn(hook->user, 0, gn.icp_acc[m], 32, (void*)s) != 0) hook_failed = true;
            }
            if (rgb && mode == 2) {   // the RGB step's last workgroup of every tracker solves
                const int n_slots = (N + ra.slot_px - 1) / ra.slot_px, n_quads = (n_slots + 1) / 2;
                rgb_step_solve_kernel<<<8 * n_quads * ((n + 7) / 8), 256, 0, s>>>(ra, so3_syncs, n_slots, n_quads, make_idiv(n_quads > 1 ? n_quads : 2), n,
                                                                                 cfg.gram ? -1 : kFixICP, next_level, last_of_level ? 1 : 0);
                continue;
            }
            if (rgb) {
                if (slots) {
                    const int n_slots = (N + ra.slot_px - 1) / ra.slot_px;
                    if (err_with_step) {
                        IcpArgs e = icp_args[i]; e.cdiv = make_idiv(e.cols);
                        rgb_slot_step_err_kernel<<<dim3(n_slots + (N + 255) / 256, n), 256, 0, s>>>(ra, n_slots, e);
                    } else rgb_slot_step_kernel<<<dim3(n_slots, n), 256, 0, s>>>(ra, n_slots);
                }
                else rgb_step_kernel<<<dim3((N + 255) / 256, n), 256, 0, s>>>(ra);
            }
            gn_solve_kernel<<<n, 256, 0, s>>>(gn, next_level, last_of_level ? 1 : 0);
        }
    }
    return !hook_failed;
}

// diagnostics (CF_ICP_REPLAY): the level-0 {ICP || residual} launch of a batch, `reps` times back to back with the given ablation mask;
// returns the average duration in us.  The sums it leaves in the accumulators are garbage: the caller zeroes them.
float replay_icp_level0(hipStream_t s, IcpLaunch cfg, const IcpArgs& a0, const RgbArgs& r0, int n, int slots, int ablate, int reps, hipEvent_t e0, hipEvent_t e1)
{
    IcpArgs a = a0; a.flags = ablate << 8;
    RgbArgs ra = r0; ra.compact = slots ? 1 : 0; ra.slot_px = cfg.threads * 4;
    for (int i = 0; i < 3; i++) launch_icp_rgbres(s, cfg, a, ra, true, true, n, 0, nullptr, nullptr);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; i++) launch_icp_rgbres(s, cfg, a, ra, true, true, n, 0, nullptr, nullptr);
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

#ifdef CF_ABLATE
// diagnostics (CF_SO3_TRACE): the stamps of the last pre-alignment, printed
void trace_so3_dump(hipStream_t s)
{
    (void)hipStreamSynchronize(s);
    unsigned long long h[12][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_so3_trace), sizeof(h)) != hipSuccess) return;
    fprintf(stderr, "[so3 trace] loop %lld ns\n", (long long)(h[11][1] - h[11][0]) * 10);
    for (int it = 0; it < 10; it++) {
        if (!h[it][0] || h[it][4] < h[it][0]) continue;
        fprintf(stderr, "[so3 trace] it %d: basis %5lld  pass %5lld  meeting %5lld  solve %5lld ns\n", it, (long long)(h[it][1] - h[it][0]) * 10,
                (long long)(h[it][2] - h[it][1]) * 10, (long long)(h[it][3] - h[it][2]) * 10, (long long)(h[it][4] - h[it][3]) * 10);
    }
    unsigned long long z[12][8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_so3_trace), z, sizeof(z));
}
// diagnostics (CF_SOLVE_TRACE): phase stamps of tracker 0's solves of the tracking call enqueued between begin and end
static unsigned long long* g_solve_trace_dev = nullptr;
void trace_solve_begin()
{
    if (!g_solve_trace_dev && hipMalloc(reinterpret_cast<void**>(&g_solve_trace_dev), 64 * 16 * 8) != hipSuccess) return;
    (void)hipMemset(g_solve_trace_dev, 0, 64 * 16 * 8);
    const unsigned zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_solve_iter), &zero, sizeof(zero));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_solve_trace), &g_solve_trace_dev, sizeof(g_solve_trace_dev));
}
void trace_solve_end(hipStream_t s, const char* path)
{
    (void)hipStreamSynchronize(s);
    unsigned long long* none = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_solve_trace), &none, sizeof(none));
    if (!g_solve_trace_dev) return;
    std::vector<unsigned long long> h(64 * 16);
    (void)hipMemcpy(h.data(), g_solve_trace_dev, h.size() * 8, hipMemcpyDeviceToHost);
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "# solve: ns from the kernel's first stamp -- loaded | totals | unpacked | LDLT | rodrigues | pose | next-iteration | write-back\n");
    for (int it = 0; it < 64; it++) {
        const unsigned long long* o = &h[(size_t)it * 16];
        if (!o[0]) continue;
        fprintf(f, "%2d", it);
        for (int k = 1; k <= 8; k++) fprintf(f, " %6lld", o[k] ? (long long)(o[k] - o[0]) * 10 : -1ll);
        fprintf(f, "\n");
    }
    fclose(f);
}
// diagnostics (CF_STEP_TRACE): one level-0 {ICP || residual} launch + one traced rgb_step_solve_kernel launch (next_level 0: the state moves
// on by one iteration; the caller's results are garbage afterwards); lines "workgroup model begin step commit ticket solve_end" in ns
void trace_step_solve(hipStream_t s, IcpLaunch cfg, const IcpArgs& a0, const RgbArgs& r0, So3Sync* syncs, int n, const char* path)
{
    IcpArgs a = a0; a.flags = 0;
    RgbArgs ra = r0; ra.compact = 1; ra.slot_px = cfg.threads * 4;
    const int N = ra.cols * ra.rows, n_slots = (N + ra.slot_px - 1) / ra.slot_px, n_quads = (n_slots + 1) / 2, grid = 8 * n_quads * ((n + 7) / 8);
    unsigned long long* d = nullptr; unsigned long long* none = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), (size_t)grid * 64) != hipSuccess) return;
    FILE* f = fopen(path, "w");
    for (int rep = 0; rep < 3 && f; rep++) {
        (void)hipMemset(d, 0, (size_t)grid * 64);
        launch_icp_rgbres(s, cfg, a, ra, true, true, n, 0, nullptr, nullptr);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_trace), &d, sizeof(d));
        rgb_step_solve_kernel<<<grid, 256, 0, s>>>(ra, syncs, n_slots, n_quads, make_idiv(n_quads > 1 ? n_quads : 2), n, cfg.gram ? -1 : kFixICP, 0, 0);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_trace), &none, sizeof(none));
        std::vector<unsigned long long> h((size_t)grid * 8);
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < grid; b++) if (h[(size_t)b * 8] && h[(size_t)b * 8] < tmin) tmin = h[(size_t)b * 8];
        fprintf(f, "# rep %d grid %d trackers %d\n", rep, grid, n);
        for (int b = 0; b < grid; b++) {
            const unsigned long long* o = &h[(size_t)b * 8];
            if (!o[0] || o[5] == 0xffff) continue;
            auto rel = [&](unsigned long long t) { return t ? (long long)(t - tmin) * 10 : -1ll; };
            fprintf(f, "%d %d %lld %lld %lld %lld %lld\n", b, (int)o[5], rel(o[0]), rel(o[1]), rel(o[2]), rel(o[3]), rel(o[4]));
        }
    }
    if (f) fclose(f);
    (void)hipFree(d);
}
// diagnostics (CF_ICP_TRACE): the level-0 {ICP || residual} launch of a batch three times back to back, the third with per-workgroup
// stamps; writes "workgroup kind model begin_ns end_ns xcc hwid" lines (kind: 0 culled ICP, 1 unculled ICP, 2 residual) to `path`
void trace_icp_level0(hipStream_t s, IcpLaunch cfg, const IcpArgs& a0, const RgbArgs& r0, int n, int slots, const char* path)
{
    IcpArgs a = a0; a.flags = 0;
    RgbArgs ra = r0; ra.compact = slots ? 1 : 0; ra.slot_px = cfg.threads * 4;
    unsigned long long* d = nullptr; unsigned long long* none = nullptr;
    const size_t cap = 1u << 16;
    if (hipMalloc(reinterpret_cast<void**>(&d), cap * 32) != hipSuccess) return;
    (void)hipMemset(d, 0, cap * 32);
    for (int i = 0; i < 2; i++) launch_icp_rgbres(s, cfg, a, ra, true, true, n, 0, nullptr, nullptr);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_trace), &d, sizeof(d));
    launch_icp_rgbres(s, cfg, a, ra, true, true, n, 0, nullptr, nullptr);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_icp_trace), &none, sizeof(none));
    std::vector<unsigned long long> h((size_t)g_last_grid * 4);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    FILE* f = fopen(path, "w");
    if (!f) return;
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < g_last_grid; b++) if (h[(size_t)b * 4] && h[(size_t)b * 4] < tmin) tmin = h[(size_t)b * 4];
    fprintf(f, "# grid %d icp_blocks %d trackers %d\n", g_last_grid, g_last_n_icp_blocks, n);
    for (int b = 0; b < g_last_grid; b++) {
        int slot = 0; while (slot < kMaxSlots - 1 && b >= g_last_icp_args.slot_end[slot]) slot++;
        const int model = g_last_icp_args.slot_desc[slot] & 0x7f;
        const int kind = (g_last_icp_args.slot_desc[slot] & kResidualSlot) ? 2 : (g_last_icp_args.m[model].box_blocks > 0 ? 0 : 1);
        const unsigned long long* o = &h[(size_t)b * 4];
        fprintf(f, "%d %d %d %lld %lld %u %u\n", b, kind, model, (long long)(o[0] - tmin) * 10, (long long)(o[1] - tmin) * 10, (unsigned)(o[2] & 255u), (unsigned)(o[2] >> 8));
    }
    fclose(f);
}
#endif

// ---- stand-alone steps (C-ABI parity with computeRgbResidual / rgbStep) -------------------
void launch_rgb_residual(hipStream_t s, const RgbArgs& ra, int n)
{
    const int N = ra.cols * ra.rows;
    RgbArgs a = ra; a.compact = 0; a.cdiv = make_idiv(a.cols);
    rgb_residual_kernel<<<dim3((N + 255) / 256, n), 256, 0, s>>>(a);
}
void launch_rgb_step(hipStream_t s, const RgbArgs& ra, int n)
{
    const int N = ra.cols * ra.rows;
    rgb_step_kernel<<<dim3((N + 255) / 256, n), 256, 0, s>>>(ra);
}
void launch_acc_total(hipStream_t s, const unsigned long long* acc, unsigned long long* out)
{
    acc_total_kernel<<<1, 64, 0, s>>>(acc, out);
}
void launch_rgb_cand(hipStream_t s, const int16_t* dIdx, const int16_t* dIdy, const float* next_depth,
                     const uint8_t* next_image, float min_scale, int cols, int rows, uint8_t* cand)
{
    rgb_cand_kernel<<<(cols * rows + 255) / 256, 256, 0, s>>>(dIdx, dIdy, next_depth, next_image, min_scale, cols, rows, cand);
}
void launch_so3_step(hipStream_t s, const uint8_t* last_image, const uint8_t* next_image, const float basis[9],
                     const float kinv[9], const float krlr[9], int cols, int rows, unsigned long long* out16)
{
    m33 B, Ki, Kr;
    for (int i = 0; i < 9; i++) { B.m[i] = basis[i]; Ki.m[i] = kinv[i]; Kr.m[i] = krlr[i]; }
    so3_step_kernel<<<1, 1024, 0, s>>>(last_image, next_image, B, Ki, Kr, cols, rows, out16);
}

}  // namespace cf
