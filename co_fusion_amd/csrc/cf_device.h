// cf_device.h -- device-side math for the CDNA4 (gfx950) kernels.
//
// All per-element arithmetic is plain IEEE f32/f64 (+ - * / sqrt) compiled with
// -ffp-contract=off, in the operation order of the reference's CUDA kernels
// (Core/Cuda/operators.cuh:55-91), so integer decisions (gating, rounding to
// pixels, validity) are reproducible bit-for-bit against the CPU oracle.
//
// Normal-equation sums are accumulated EXACTLY: each product row_i*row_j is
// formed in f64 (exact for f32 inputs), scaled by 2^F and rounded once to an
// integer with the "magic number" trick, then added into wrapping 64-bit
// integers.  Integer sums are order independent => the result does not depend
// on launch shape, wave scheduling or GPU count.  (The reference sums f32 in a
// launch-shape dependent tree, Core/Cuda/reduce.cu:90-165.)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cf_kernels.h"

namespace cf {

constexpr int kFixICP = 32;
constexpr int kFixRGB = 32;  // for sigma >= 4096; in general rgb_fix_bits(sigma)

// Fraction bits of the RGB step's sums as a function of the weight scale sigma handed to rgbStep (the correspondence
// COUNT, or -1 / 1 for unit weights; RGBDOdometry.cpp:373-385, reduce.cu:533-540): Jacobian rows scale like 1/sigma, so
// the fixed-point window moves with it.  Integer function of sigma's bits; same spec as oracle/orc_math.h.
__host__ __device__ inline int rgb_fix_bits(float sigma)
{
    if (sigma == -1.0f || !(sigma >= 2.0f)) return 8;
    union { float f; unsigned u; } v; v.f = sigma;
    const int F = 8 + 2 * ((int)((v.u >> 23) & 255u) - 127);
    return F > 32 ? 32 : F;
}
constexpr int kFixSO3 = 12;
constexpr int kSE3Words = 32;  // 27 products, residual, inliers, 3 pad
constexpr int kSO3Words = 16;  // 9 products, residual, inliers, pad

struct f3 { float x, y, z; };
struct m33 { float m[9]; };  // row-major == reference mat33 (types.cuh:61-73)

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b)
{
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float norm(f3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ f3 normalized(f3 a)
{
    const float rn = 1.0f / sqrtf(dot(a, a));
    return f3{a.x * rn, a.y * rn, a.z * rn};
}
__device__ __forceinline__ f3 mul(const m33& m, f3 a)
{
    return f3{m.m[0] * a.x + m.m[1] * a.y + m.m[2] * a.z, m.m[3] * a.x + m.m[4] * a.y + m.m[5] * a.z,
              m.m[6] * a.x + m.m[7] * a.y + m.m[8] * a.z};
}

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fffffff); }
__device__ __forceinline__ bool is_nan(float v) { return v != v; }
__device__ __forceinline__ bool is_finite(float v) { return fabsf(v) < __int_as_float(0x7f800000); }

// order-preserving 32-bit key of a float (a < b  <=>  fkey(a) < fkey(b) for non-NaN values) and its inverse; the
// bounding boxes the culling tests use are reduced with integer atomicMax on these keys
__device__ __forceinline__ unsigned fkey(float f)
{
    const unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k)
{
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

// __float2int_rn semantics: round-half-even, saturating, NaN -> 0
// (v_cvt_i32_f32 saturates out-of-range inputs and turns NaN into 0 by itself; written as the instruction because the C++ conversion is
//  undefined out of range and the guarded form compiled to three nested exec-mask branches per call)
__device__ __forceinline__ int f2i_rn(float v)
{
    int r;
    const float t = rintf(v);
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(t));
    return r;
}

// ---- exact fixed-point accumulation --------------------------------------------
// bits(1.5*2^52 + q) = bits(1.5*2^52) + q for |q| < 2^51, so adding the raw bit
// patterns accumulates q; the n*bits(magic) offset is removed once per thread.
constexpr double kMagic = 6755399441055744.0;                 // 1.5 * 2^52
constexpr unsigned long long kMagicBits = 0x4338000000000000ull;

template <int F>
__device__ __forceinline__ unsigned long long fix_bits(double a_scaled /* a*2^F */, double b)
{
    // p = a*b*2^F is exact in f64 (48-bit product, power-of-two scale); clamp to +-2^50
    double p = a_scaled * b;
    p = fmin(fmax(p, -1125899906842624.0), 1125899906842624.0);  // NaN -> -2^50?  callers never pass NaN
    return (unsigned long long)__double_as_longlong(p + kMagic);
}

// 64-lane butterfly that reduces NV (<=32, power of two) 64-bit values per lane to one
// total per lane: after the call lane l holds the wave total of value index
// (l >> (6 - log2(NV)... see below) in acc[0].
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor((int)lo, mask, 64);
    hi = __shfl_xor((int)hi, mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// ---- wave64 transpose-reduce of 64-bit integer partial sums ---------------------------------
// Halving butterfly: at the step with lane distance M, lanes with bit M set keep the upper HALF of
// the value array, the others the lower HALF, and each adds its partner's copy of the half it
// keeps: 31 + 1 exchanges instead of 32 values x 6 shuffle steps.  CDNA4 specifics:
//   * distance 32 / 16: v_permlane32_swap / v_permlane16_swap exchange the halves of TWO registers
//     in one VALU instruction and need no selects (a' + b' is the wanted sum in every lane);
//   * distance 8 / 4 / 2 / 1: DPP moves (row_ror:8, row_half_mirror, quad_perm) -- any perfect
//     matching between the "keeps lower" and "keeps upper" lanes of a group is a valid partner.
// No LDS traffic (the ds_bpermute form of __shfl_xor made this reduction LDS-pipe bound).
// All indices are static so the accumulators stay in VGPRs.
// 64-bit adds are written on explicit 32-bit halves (v_add_co / v_addc): composing u64 values from shuffled halves made the
// compiler split every add into zero-extended partial sums (about twice the instructions).
__device__ __forceinline__ void add64(unsigned& alo, unsigned& ahi, unsigned blo, unsigned bhi)
{
    const unsigned lo = alo + blo;
    const unsigned c = lo < alo ? 1u : 0u;
    ahi = ahi + bhi + c;
    alo = lo;
}
template <int HALF, int N>
__device__ __forceinline__ void swap32_step(unsigned (&lo)[N], unsigned (&hi)[N])
{
#pragma unroll
    for (int i = 0; i < HALF; i++) {
        const auto l = __builtin_amdgcn_permlane32_swap(lo[i], lo[i + HALF], false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi[i], hi[i + HALF], false, false);
        unsigned a = l[0], b = h[0];
        add64(a, b, l[1], h[1]);
        lo[i] = a; hi[i] = b;
    }
}
template <int HALF, int N>
__device__ __forceinline__ void swap16_step(unsigned (&lo)[N], unsigned (&hi)[N])
{
#pragma unroll
    for (int i = 0; i < HALF; i++) {
        const auto l = __builtin_amdgcn_permlane16_swap(lo[i], lo[i + HALF], false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi[i], hi[i + HALF], false, false);
        unsigned a = l[0], b = h[0];
        add64(a, b, l[1], h[1]);
        lo[i] = a; hi[i] = b;
    }
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int HALF, int CTRL, int N>
__device__ __forceinline__ void dpp_step(unsigned (&lo)[N], unsigned (&hi)[N], bool upper)
{
#pragma unroll
    for (int i = 0; i < HALF; i++) {
        const unsigned slo = upper ? lo[i] : lo[i + HALF], shi = upper ? hi[i] : hi[i + HALF];  // the half this lane gives away
        unsigned klo = upper ? lo[i + HALF] : lo[i], khi = upper ? hi[i + HALF] : hi[i];        // the half it keeps
        add64(klo, khi, dpp_u32<CTRL>(slo), dpp_u32<CTRL>(shi));
        lo[i] = klo; hi[i] = khi;
    }
}
constexpr int kDppRor8 = 0x128, kDppHalfMirror = 0x141, kDppXor2 = 0x4E, kDppXor1 = 0xB1;

// Reduce acc[0..31] across the 64 lanes of a wave.  Returns, in lane l, the wave total of value
// index ((l >> 1) & 31).
__device__ __forceinline__ unsigned long long wave_reduce32_u64(const unsigned long long (&acc)[32], int lane)
{
    unsigned lo[32], hi[32];
#pragma unroll
    for (int i = 0; i < 32; i++) { lo[i] = (unsigned)acc[i]; hi[i] = (unsigned)(acc[i] >> 32); }
    swap32_step<16>(lo, hi);
    swap16_step<8>(lo, hi);
    dpp_step<4, kDppRor8>(lo, hi, (lane & 8) != 0);
    dpp_step<2, kDppHalfMirror>(lo, hi, (lane & 4) != 0);
    dpp_step<1, kDppXor2>(lo, hi, (lane & 2) != 0);
    unsigned a = lo[0], b = hi[0];
    add64(a, b, dpp_u32<kDppXor1>(lo[0]), dpp_u32<kDppXor1>(hi[0]));
    return ((unsigned long long)b << 32) | a;
}

// 16-value flavour (SO3): lane l ends with the total of value index ((l >> 2) & 15)
__device__ __forceinline__ unsigned long long wave_reduce16_u64(const unsigned long long (&acc)[16], int lane)
{
    unsigned lo[16], hi[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { lo[i] = (unsigned)acc[i]; hi[i] = (unsigned)(acc[i] >> 32); }
    swap32_step<8>(lo, hi);
    swap16_step<4>(lo, hi);
    dpp_step<2, kDppRor8>(lo, hi, (lane & 8) != 0);
    dpp_step<1, kDppHalfMirror>(lo, hi, (lane & 4) != 0);
    unsigned a = lo[0], b = hi[0];
    add64(a, b, dpp_u32<kDppXor2>(lo[0]), dpp_u32<kDppXor2>(hi[0]));
    add64(a, b, dpp_u32<kDppXor1>(a), dpp_u32<kDppXor1>(b));
    return ((unsigned long long)b << 32) | a;
}

// 8-value flavour (the ICP commit, one word group per wave): lane l ends with the total of value index ((l >> 3) & 7)
__device__ __forceinline__ unsigned long long wave_reduce8_u64(const unsigned long long (&acc)[8], int lane)
{
    unsigned lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { lo[i] = (unsigned)acc[i]; hi[i] = (unsigned)(acc[i] >> 32); }
    swap32_step<4>(lo, hi);
    swap16_step<2>(lo, hi);
    dpp_step<1, kDppRor8>(lo, hi, (lane & 8) != 0);
    unsigned a = lo[0], b = hi[0];
    add64(a, b, dpp_u32<kDppHalfMirror>(a), dpp_u32<kDppHalfMirror>(b));   // (lane i <-> 7 - i, then i ^ 2, then i ^ 1: all eight lanes of a group)
    add64(a, b, dpp_u32<kDppXor2>(a), dpp_u32<kDppXor2>(b));
    add64(a, b, dpp_u32<kDppXor1>(a), dpp_u32<kDppXor1>(b));
    return ((unsigned long long)b << 32) | a;
}

// ---- Gram form of the ICP sums (cf_set_icp_arith 1; oracle: ORC_ICP_ARITH_GRAM) -------------------------------------------
// Every row ENTRY is rounded once to a fixed-point grid: q_i = RNE(clamp(row_i, +-kGramLim[i]) * 2^kGramBits[i]), |q_i| <= 2^22;
// entry 7 is 1 for a found correspondence.  The 29 words are entries of the Gram matrix sum_pixels q q^T of that integer matrix,
// which is a dense contraction over the pixels: each q_i is split into three balanced signed 8-bit limbs (q = d0 + 2^8 d1 + 2^16 d2,
// -128 <= d <= 127), the 64 pixels of a wave are the K dimension of v_mfma_i32_32x32x32_i8 on the 32 x 64 limb matrix (row 4i + a =
// limb a of entry i, a = 3 unused) against ITSELF; the int32 tiles of a workgroup's waves are added through LDS and the 9 limb products
// of an entry pair are recombined with shifts ONCE PER WORKGROUP (gram_block_commit), then 29 64-bit atomics as in the product form.
// Exact: int8 x int8 products in int32 accumulators (at most 2^14 * 4096 pixels per workgroup), 64-bit recombination.
// (kGramBits: cf_kernels.h -- the host unpack needs it too)
constexpr float kGramLim[7] = {4.f, 4.f, 4.f, 32.f, 32.f, 32.f, 1.f};
constexpr int kGramRowStride = 68;                       // dwords per limb-matrix row in LDS (64 pixels + pad: the 8 rows start in different banks)
constexpr int kGramWaveDwords = 12 * 64;                 // one wave's staging area: 8 rows of limbs, afterwards its 12 x 64 partial products
typedef int gram_v4i __attribute__((ext_vector_type(4)));
typedef int gram_v16i __attribute__((ext_vector_type(16)));

// balanced signed limbs of q (|q| <= 0x7f7f7f): byte k of the result, read as int8, is d_k
__device__ __forceinline__ unsigned gram_limbs(int q) { return ((unsigned)q + 0x00808080u) ^ 0x00808080u; }

// word of the 32-word accumulator layout (se3_unpack) that lane `lane` of a wave holds after gram_wave_finish, or -1
__device__ __forceinline__ int gram_word_of_lane(int lane)
{
    const int i = 2 * (lane & 3) + (lane >> 5), j = (lane & 31) >> 2;
    if (i <= 5 && j >= i && j <= 6) return 7 * i - (i * (i - 1)) / 2 + (j - i);
    if (i == 6 && j == 6) return 27;
    if (i == 7 && j == 7) return 28;
    return -1;
}

// One K = 64 step: the wave's 64 pixels (8 limb dwords each, already in `wl`: wl[i * kGramRowStride + pixel]) times themselves.
// Lane l feeds limb-matrix row l & 31 (entry (l & 31) >> 2, limb l & 3) for the 16 pixels 32 s + 16 (l >> 5) + t of step s as A AND as B:
// the order of the pixels inside K does not matter for a Gram matrix, only that A and B agree.
__device__ __forceinline__ gram_v16i gram_wave_mfma(const int* wl, int lane, gram_v16i acc)
{
    const unsigned a = (unsigned)lane & 3u;
    const unsigned selLo = a | ((4u + a) << 8) | 0x0c0c0000u;          // v_perm_b32: bytes 0-3 = second source, 4-7 = first, 0x0c = zero
    const unsigned selHi = 0x00000c0cu | (a << 16) | ((4u + a) << 24);
    const int* src = wl + ((lane & 31) >> 2) * kGramRowStride + 16 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 2; s++) {
        gram_v4i x[4];
#pragma unroll
        for (int c = 0; c < 4; c++) x[c] = *reinterpret_cast<const gram_v4i*>(src + 32 * s + 4 * c);
        gram_v4i f;
#pragma unroll
        for (int c = 0; c < 4; c++)
            f[c] = (int)(__builtin_amdgcn_perm((unsigned)x[c][1], (unsigned)x[c][0], selLo) | __builtin_amdgcn_perm((unsigned)x[c][3], (unsigned)x[c][2], selHi));
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(f, f, acc, 0, 0, 0);
    }
    return acc;
}

// acc (C/D layout of the 32x32 shapes: column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5)) -> the 64-bit Gram entries.  Register
// r = 4 g + a2 of lane l is (entry i = 2 g + (l >> 5), limb a2) x (entry j = (l & 31) >> 2, limb b = l & 3): weight 2^(8 (a2 + b)),
// summed over a2 in the lane and over b across the quad.  Returns the entry of word gram_word_of_lane(lane).
__device__ __forceinline__ unsigned long long gram_wave_finish(const gram_v16i& acc, int lane)
{
    const int b8 = 8 * (lane & 3);
    unsigned rlo = 0, rhi = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        long long P = (long long)acc[4 * g] + ((long long)acc[4 * g + 1] << 8) + ((long long)acc[4 * g + 2] << 16);
        P <<= b8;
        unsigned lo = (unsigned)P, hi = (unsigned)((unsigned long long)P >> 32);
        add64(lo, hi, dpp_u32<kDppXor1>(lo), dpp_u32<kDppXor1>(hi));
        add64(lo, hi, dpp_u32<kDppXor2>(lo), dpp_u32<kDppXor2>(hi));
        if ((lane & 3) == g) { rlo = lo; rhi = hi; }
    }
    return ((unsigned long long)rhi << 32) | rlo;
}

// Workgroup commit of the Gram form: every wave that found correspondences has left the 12 used registers of its tile in its staging
// area (gram_wave_store); wave 0 adds the tiles, recombines the limbs and issues the atomics.  Called by all threads of the workgroup.
__device__ __forceinline__ void gram_wave_store(int* wl, const gram_v16i& acc, int lane)
{
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int a = 0; a < 3; a++) wl[(3 * g + a) * 64 + lane] = acc[4 * g + a];
}
__device__ __forceinline__ void gram_block_commit(const int* lds, bool has, int lane, int wave, int nwaves,
                                                  unsigned long long* __restrict__ dst /* [32] of this group */)
{
    __shared__ int s_has[16];
    if (lane == 0) s_has[wave] = has ? 1 : 0;
    __syncthreads();
    if (wave != 0) return;
    gram_v16i c;
#pragma unroll
    for (int r = 0; r < 16; r++) c[r] = 0;
    bool any = false;
    for (int w = 0; w < nwaves; w++) {
        if (!s_has[w]) continue;
        any = true;
        const int* src = lds + w * kGramWaveDwords + lane;
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int a = 0; a < 3; a++) c[4 * g + a] += src[(3 * g + a) * 64];
    }
    if (!any) return;
    const unsigned long long v = gram_wave_finish(c, lane);
    const int word = gram_word_of_lane(lane);
    if (word >= 0 && v != 0) atomicAdd(&dst[word], v);
}

// ---- deterministic f64 sin/cos (same spec as oracle/orc_math.h: orc_sincos) ------
__device__ __forceinline__ void det_sincos(double x, double* s, double* c)
{
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_1 = 1.57079632673412561417e+00;
    const double pio2_1t = 6.07710050650619224932e-11;
    double fn = rint(x * two_over_pi);
    double r = (x - fn * pio2_1) - fn * pio2_1t;
    long long n = (long long)fn;
    double r2 = r * r;
    double sp = -1.0 / 355687428096000.0;
    sp = sp * r2 + 1.0 / 1307674368000.0;
    sp = sp * r2 - 1.0 / 6227020800.0;
    sp = sp * r2 + 1.0 / 39916800.0;
    sp = sp * r2 - 1.0 / 362880.0;
    sp = sp * r2 + 1.0 / 5040.0;
    sp = sp * r2 - 1.0 / 120.0;
    sp = sp * r2 + 1.0 / 6.0;
    double sr = r - r * r2 * sp;
    double cp = 1.0 / 20922789888000.0;
    cp = cp * r2 - 1.0 / 87178291200.0;
    cp = cp * r2 + 1.0 / 479001600.0;
    cp = cp * r2 - 1.0 / 3628800.0;
    cp = cp * r2 + 1.0 / 40320.0;
    cp = cp * r2 - 1.0 / 720.0;
    cp = cp * r2 + 1.0 / 24.0;
    cp = cp * r2 - 1.0 / 2.0;
    double cr = 1.0 + r2 * cp;
    switch ((int)(n & 3)) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

// OdometryProvider::rodrigues (Core/Utils/OdometryProvider.h:32-67)
__device__ inline void rodrigues(const double w[3], double R[9])
{
    double rx = w[0], ry = w[1], rz = w[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    if (theta >= 2.2204460492503131e-16) {
        double s, c;
        det_sincos(theta, &s, &c);
        double c1 = 1.0 - c;
        double itheta = 1.0 / theta;
        rx *= itheta; ry *= itheta; rz *= itheta;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx_[k];
    }
}

template <typename T>
__device__ inline void inv33(const T a[9], T o[9])
{
    T c00 = a[4] * a[8] - a[5] * a[7];
    T c01 = a[5] * a[6] - a[3] * a[8];
    T c02 = a[3] * a[7] - a[4] * a[6];
    T det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    T id = (T)1 / det;
    o[0] = c00 * id;
    o[1] = (a[2] * a[7] - a[1] * a[8]) * id;
    o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c01 * id;
    o[4] = (a[0] * a[8] - a[2] * a[6]) * id;
    o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c02 * id;
    o[7] = (a[1] * a[6] - a[0] * a[7]) * id;
    o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

template <typename T>
__device__ inline void mul33(const T a[9], const T b[9], T o[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

__device__ inline void mul44(const double a[16], const double b[16], double o[16])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = a[i * 4 + 0] * b[0 * 4 + j];
            s = s + a[i * 4 + 1] * b[1 * 4 + j];
            s = s + a[i * 4 + 2] * b[2 * 4 + j];
            s = s + a[i * 4 + 3] * b[3 * 4 + j];
            o[i * 4 + j] = s;
        }
}

__device__ inline void inv44_affine(const double a[16], double o[16])
{
    double L[9] = {a[0], a[1], a[2], a[4], a[5], a[6], a[8], a[9], a[10]}, Li[9];
    inv33<double>(L, Li);
    for (int i = 0; i < 3; i++) {
        o[i * 4 + 0] = Li[i * 3 + 0]; o[i * 4 + 1] = Li[i * 3 + 1]; o[i * 4 + 2] = Li[i * 3 + 2];
        o[i * 4 + 3] = -(Li[i * 3 + 0] * a[3] + Li[i * 3 + 1] * a[7] + Li[i * 3 + 2] * a[11]);
    }
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}

// LDL^T with diagonal pivoting; zero pivots -> zero solution component (the
// ldlt().solve() behaviour the reference relies on, RGBDOdometry.cpp:435).
// All arrays are indexed dynamically (pivoting), so the caller passes LDS storage: private arrays
// would be demoted to scratch (global memory round trips on the single solving lane).
// ws: N*N + 2N elements, iws: N ints; Ain/b/x should live in LDS as well.
template <typename T, int N>
__device__ inline void ldlt_solve(const T* Ain, const T* b, T* x, T tiny, T* ws, int* iws)
{
    T* A = ws; T* y = ws + N * N; T* d = ws + N * N + N;
    int* perm = iws;
    for (int i = 0; i < N * N; i++) A[i] = Ain[i];
    for (int i = 0; i < N; i++) perm[i] = i;
    for (int k = 0; k < N; k++) {
        int p = k;
        T best = A[k * N + k] < 0 ? -A[k * N + k] : A[k * N + k];
        for (int i = k + 1; i < N; i++) {
            T v = A[i * N + i] < 0 ? -A[i * N + i] : A[i * N + i];
            if (v > best) { best = v; p = i; }
        }
        if (p != k) {
            for (int j = 0; j < N; j++) { T t = A[k * N + j]; A[k * N + j] = A[p * N + j]; A[p * N + j] = t; }
            for (int i = 0; i < N; i++) { T t = A[i * N + k]; A[i * N + k] = A[i * N + p]; A[i * N + p] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        T akk = A[k * N + k];
        d[k] = akk;
        T aabs = akk < 0 ? -akk : akk;
        if (aabs > tiny) {
            for (int i = k + 1; i < N; i++) A[i * N + k] = A[i * N + k] / akk;
            for (int i = k + 1; i < N; i++)
                for (int j = k + 1; j <= i; j++) {
                    A[i * N + j] = A[i * N + j] - A[i * N + k] * akk * A[j * N + k];
                    A[j * N + i] = A[i * N + j];
                }
        } else {
            for (int i = k + 1; i < N; i++) A[i * N + k] = 0;
        }
    }
    for (int i = 0; i < N; i++) {
        T s = b[perm[i]];
        for (int j = 0; j < i; j++) s = s - A[i * N + j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < N; i++) {
        T aabs = d[i] < 0 ? -d[i] : d[i];
        y[i] = (aabs > tiny) ? y[i] / d[i] : (T)0;
    }
    for (int i = N - 1; i >= 0; i--) {
        T s = y[i];
        for (int j = i + 1; j < N; j++) s = s - A[j * N + i] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < N; i++) x[perm[i]] = y[i];
}

// ---- the same 6x6 factorisation, spread over one wave -------------------------------------------------
// Lane t < 36 owns A[t/6][t%6] in a register (only the lower triangle is kept up to date); pivot search and the substitutions use
// v_readlane (uniform lane ids), the column broadcast and the symmetric row/column swap ds_bpermute.  Every
// element goes through exactly the operations of ldlt_solve<double, 6> in the same order (right-looking
// update, one update per k), so the result is bit-identical; the serial version stays for N = 3 / f32.
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bpermute_f64(double v, int byte_addr)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// must be called by all 64 lanes of a wave; lanes 0..35: A[lane / 6][lane % 6], lanes 36..41: b[lane - 36]; x: [6] (LDS, written by lane 0).
// Round 6: the right-hand side rides along as a seventh column -- L y = P b is solved INSIDE the factorisation (at step k every later
// entry takes b_i - l_ik * y_k: for a fixed i the very multiplications and subtractions of the forward substitution, in its order
// k = 0, 1, ...; the transpositions permute b with the rows), which takes the forward substitution's chain of 30 dependent operations
// off the solve; the six divisions by D run on six lanes at once.
__device__ __forceinline__ void ldlt_solve6_wave(double a, double* x, double tiny, int lane)
{
    const bool isB = lane >= 36 && lane < 42;
    const int i = (lane < 36) ? lane / 6 : (isB ? lane - 36 : 0), j = (lane < 36) ? lane % 6 : (isB ? 6 : 0);
    int perm[6] = {0, 1, 2, 3, 4, 5};
    double d[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        // Pivot: the first q >= k whose |a_qq| is the largest (the serial scan `if (v > best)` keeps the first of equal values).  The
        // entries are finite (integer sums, scaled), so that is the lowest q with |a_qq| == max: five v_max_f64 on lane values read
        // into scalar registers, then one compare + select per candidate -- the scan itself was thirteen instructions per candidate,
        // a fifth of the factorisation (round 6, from the ISA).
        int p = 5;
        {
            double dq[6];
#pragma unroll
            for (int q = k; q < 6; q++) dq[q] = __builtin_fabs(readlane_f64(a, q * 7));
            double M = dq[k];
#pragma unroll
            for (int q = k + 1; q < 6; q++) M = __builtin_fmax(M, dq[q]);
#pragma unroll
            for (int q = 5; q >= k; q--) p = (dq[q] == M) ? q : p;
        }
        p = __builtin_amdgcn_readfirstlane(p);
        if (p != k) {
            // symmetric transposition k <-> p on the LOWER triangle: element (i, j), i >= j, takes S(pi(i), pi(j)) of the symmetric matrix,
            // which lives at (max, min) of the two indices; the right-hand side swaps entries k and p
            const int si = (i == k) ? p : ((i == p) ? k : i), sj = (j == k) ? p : ((j == p) ? k : j);
            a = bpermute_f64(a, (isB ? 36 + si : (si > sj ? si : sj) * 6 + (si > sj ? sj : si)) * 4);
            const int pk = perm[k];
            int pp = pk;
#pragma unroll
            for (int q = k + 1; q < 6; q++)
                if (q == p) { pp = perm[q]; perm[q] = pk; }
            perm[k] = pp;
        }
        const double akk = readlane_f64(a, k * 7);
        d[k] = akk;
        const double aabs = akk < 0 ? -akk : akk;
        const bool pivot_ok = aabs > tiny;
        if (j == k && i > k) a = pivot_ok ? a / akk : 0.0;
        // column k of L for this lane's row and column: two shuffles (lane i*6+k, lane j*6+k; the constant 4k rides in the instruction's
        // offset field).  Until round 6 a chain of lane reads and selects, ten instructions per remaining row: 2040 -> 1840 ns for the
        // factorisation (in-kernel clocks, profiles/r6ld_*); shuffling only the first two or three steps: 1880.
        const double lik = bpermute_f64(a, (i * 6 + k) * 4);
        const double ljk = bpermute_f64(a, (j * 6 + k) * 4);   // (the right-hand side's lanes, "column 6", read a lane they do not use)
        const double yk = readlane_f64(a, 36 + k);
        if (isB) { if (i > k) a = a - lik * yk; }
        else if (pivot_ok && i > k && j > k && j <= i) a = a - lik * akk * ljk;
        // (no mirror into the upper triangle: pivot search, column reads and the back substitution only ever read the lower one, and the
        // transposition above finds its sources there)
    }
    double y[6];
#pragma unroll
    for (int r = 0; r < 6; r++) y[r] = readlane_f64(a, 36 + r);
    {   // D^-1: the six divisions are independent -- lane r takes y[r] / d[r], one division deep instead of six in a row on every lane
        double yn = y[0], dn = d[0];
#pragma unroll
        for (int r = 1; r < 6; r++) if (lane == r) { yn = y[r]; dn = d[r]; }
        const double aabs = dn < 0 ? -dn : dn;
        const double q = (aabs > tiny) ? yn / dn : 0.0;
#pragma unroll
        for (int r = 0; r < 6; r++) y[r] = readlane_f64(q, r);
    }
#pragma unroll
    for (int r = 5; r >= 0; r--) {
        double s = y[r];
#pragma unroll
        for (int c = r + 1; c < 6; c++) s = s - readlane_f64(a, c * 6 + r) * y[c];
        y[r] = s;
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 6; r++) x[perm[r]] = y[r];
    }
}

// Measured and dropped (round 2): the same factorisation with LDS as the exchange medium (matrix mirrored in LDS once per step, pivot
// search and column values as broadcast loads, each trailing lane dividing its own two column entries): 2.84 us against 2.60 us for
// this version inside the 7 us solve kernel (in-kernel clocks).  The time is the chain of dependent f64 operations -- six steps of
// {pivot compare, IEEE division, multiply, multiply, subtract} and the two substitutions --, not the data movement.

// exact conversion of a fixed-point sum to the reference's f32 host value
__device__ __forceinline__ float fix_to_f32(long long q, int F) { return (float)ldexp((double)q, -F); }

}  // namespace cf
