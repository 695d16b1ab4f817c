/*
 * orc_math.h -- small fixed-size linear algebra + deterministic scalar helpers
 * for the CPU ORACLE (test infrastructure only; see oracle/README.md).
 *
 * Everything here is plain IEEE arithmetic built from + - * / sqrt so that the
 * HIP kernels (compiled with -ffp-contract=off) can reproduce it bit for bit.
 * The oracle is compiled with -ffp-contract=off -fno-fast-math as well.
 *
 * Reference arithmetic being restated:
 *   float3 ops / mat33*float3 ........ Core/Cuda/operators.cuh:55-91
 *   Rodrigues (f64) .................. Core/Utils/OdometryProvider.h:32-67
 *   SE3 update ....................... Core/Utils/OdometryProvider.h:69-89
 *   6x6 / 3x3 ldlt().solve() ......... Eigen (third party, absent) -- restated
 *                                      as LDL^T with diagonal pivoting.
 */
#ifndef ORC_MATH_H_
#define ORC_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } orc_f3;
typedef struct { float m[9]; } orc_m33;   /* row-major, == reference mat33 (types.cuh:61-73) */

static inline orc_f3 orc_f3_make(float x, float y, float z) { orc_f3 r = {x, y, z}; return r; }
static inline orc_f3 orc_f3_sub(orc_f3 a, orc_f3 b) { return orc_f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline orc_f3 orc_f3_add(orc_f3 a, orc_f3 b) { return orc_f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline float orc_f3_dot(orc_f3 a, orc_f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline orc_f3 orc_f3_cross(orc_f3 a, orc_f3 b)
{
    return orc_f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float orc_f3_norm(orc_f3 a) { return sqrtf(orc_f3_dot(a, a)); }
/* reference uses rsqrtf (operators.cuh:82); we pin it as 1/sqrt (both IEEE). */
static inline orc_f3 orc_f3_normalized(orc_f3 a)
{
    const float rn = 1.0f / sqrtf(orc_f3_dot(a, a));
    return orc_f3_make(a.x * rn, a.y * rn, a.z * rn);
}
static inline orc_f3 orc_m33_mul(const orc_m33 *m, orc_f3 a)
{
    return orc_f3_make(m->m[0] * a.x + m->m[1] * a.y + m->m[2] * a.z,
                       m->m[3] * a.x + m->m[4] * a.y + m->m[5] * a.z,
                       m->m[6] * a.x + m->m[7] * a.y + m->m[8] * a.z);
}

/* __float2int_rn (reduce.cu:295): round-half-even, saturating, NaN -> 0 */
static inline int orc_f2i_rn(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)rintf(v);
}

static inline float orc_qnan(void)
{
    union { uint32_t u; float f; } c; c.u = 0x7fffffffu; return c.f;  /* cudafuncs.cu:131 */
}


/* ---- deterministic f32 exp / acos (shared spec with the HIP kernels) ---------------------
 * GLSL's exp()/acos() precision is implementation defined (the reference runs them inside the
 * GL driver); both sides of the parity tests use these fixed polynomial forms instead of libm so
 * that confidences and the 0.5 rad normal gate are reproducible bit for bit. */
static inline float orc_expf(float x)
{
    if (!(x > -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.0f) return INFINITY;
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
static inline float orc_acos_r(float z)
{
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    const float p = z * (pS0 + z * (pS1 + z * pS2));
    const float q = 1.0f + z * qS1;
    return p / q;
}
static inline float orc_acosf(float x)
{
    const float pio2 = 1.57079637050628662109375f, pi = 3.1415927410125732421875f;
    if (x != x || x > 1.0f || x < -1.0f) return orc_qnan();
    if (fabsf(x) < 0.5f) return pio2 - (x + x * orc_acos_r(x * x));
    if (x < 0.0f) {
        const float z = (1.0f + x) * 0.5f, s = sqrtf(z);
        return pi - 2.0f * (s + s * orc_acos_r(z));
    }
    const float z = (1.0f - x) * 0.5f, s = sqrtf(z);
    return 2.0f * (s + s * orc_acos_r(z));
}

/* ---- fixed-point accumulation of normal-equation products -------------------
 * q = RNE_to_int64(a*b*2^F), a and b clamped to +-2^((50-F)/2).
 * Sums of q are order independent, so every launch shape / GPU count / the
 * oracle produce identical bits.  (The reference sums f32 in a launch-shape
 * dependent tree, reduce.cu:90-165; see DESIGN.md "exact reductions".)
 */
static inline float orc_clamp_row(float v, int F)
{
    const float lim = (float)(1 << ((50 - F) / 2));   /* 2^9 for F=32, 2^19 for F=12 */
    return fminf(fmaxf(v, -lim), lim);
}
static inline int64_t orc_fix_prod(float a, float b, int F)
{
    /* row entries are clamped to +-2^((50-F)/2) so the scaled product stays below 2^50;
     * the product of two f32 is exact in f64 (24+24 bits), the power-of-two scale is exact,
     * and llrint rounds once (RNE) */
    double p = (double)orc_clamp_row(a, F) * (double)orc_clamp_row(b, F);
    return (int64_t)llrint(ldexp(p, F));
}
static inline double orc_fix_to_double(int64_t q, int F) { return ldexp((double)q, -F); }

/* Gram form of the ICP sums (ORC_ICP_ARITH_GRAM): row entry i -> integer q_i = RNE(clamp(row_i, +-lim_i) * 2^bits_i), |q_i| <= 2^22
 * (three signed 8-bit limbs); entry 7 is the constant 1 of a found correspondence, so q_7*q_7 counts the inliers.  Same tables
 * as cf_device.h (kGramBits / kGramLim). */
static const int orc_gram_bits[8] = {20, 20, 20, 17, 17, 17, 22, 0};
static const float orc_gram_lim[7] = {4.0f, 4.0f, 4.0f, 32.0f, 32.0f, 32.0f, 1.0f};
static inline int32_t orc_gram_quant(float v, int i)
{
    const float lim = orc_gram_lim[i];
    return (int32_t)lrintf(fminf(fmaxf(v, -lim), lim) * ldexpf(1.0f, orc_gram_bits[i]));  /* power-of-two scale: exact; one RNE */
}

/* Fraction bits of the RGB step's fixed-point sums, chosen from the weight's scale: rgbStep is handed sigma = the
 * correspondence COUNT n (RGBDOdometry.cpp:373-374), so a Jacobian row scales like 1/n; sigma = -1 (rgbOnly, w = 1,
 * reduce.cu:537-540) and sigma = 1 (zero residual) leave the rows unscaled (|row| up to ~2^18).
 *   F = 8                         for sigma == -1 or sigma < 2      (rows up to 2^21, sums up to 2^55)
 *   F = min(32, 8 + 2*floor(log2 sigma))  otherwise                 (n >= 4096 -> the Q32 used for ICP)
 * an integer function of sigma's bit pattern, shared spec with the HIP kernels (cf_device.h rgb_fix_bits). */
static inline int orc_rgb_fix_bits(float sigma)
{
    if (sigma == -1.0f || !(sigma >= 2.0f)) return 8;
    uint32_t u; memcpy(&u, &sigma, 4);
    int e = (int)((u >> 23) & 255u) - 127;
    int F = 8 + 2 * e;
    return F > 32 ? 32 : F;
}

/* ---- deterministic f64 sin/cos (Cody-Waite + Taylor), shared spec with HIP ---- */
static inline void orc_sincos(double x, double *s, double *c)
{
    /* reduce to r in [-pi/4, pi/4], quadrant n */
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_1 = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double pio2_1t = 6.07710050650619224932e-11; /* pi/2 - pio2_1 */
    double fn = rint(x * two_over_pi);
    double r = (x - fn * pio2_1) - fn * pio2_1t;
    int64_t n = (int64_t)fn;
    double r2 = r * r;
    /* Taylor to r^17 / r^16, Horner */
    double sp = -1.0 / 355687428096000.0;           /* -1/17! */
    sp = sp * r2 + 1.0 / 1307674368000.0;            /* 1/15! */
    sp = sp * r2 - 1.0 / 6227020800.0;               /* 1/13! */
    sp = sp * r2 + 1.0 / 39916800.0;                 /* 1/11! */
    sp = sp * r2 - 1.0 / 362880.0;                   /* 1/9! */
    sp = sp * r2 + 1.0 / 5040.0;
    sp = sp * r2 - 1.0 / 120.0;
    sp = sp * r2 + 1.0 / 6.0;
    double sr = r - r * r2 * sp;
    double cp = 1.0 / 20922789888000.0;              /* 1/16! */
    cp = cp * r2 - 1.0 / 87178291200.0;              /* 1/14! */
    cp = cp * r2 + 1.0 / 479001600.0;                /* 1/12! */
    cp = cp * r2 - 1.0 / 3628800.0;                  /* 1/10! */
    cp = cp * r2 + 1.0 / 40320.0;
    cp = cp * r2 - 1.0 / 720.0;
    cp = cp * r2 + 1.0 / 24.0;
    cp = cp * r2 - 1.0 / 2.0;
    double cr = 1.0 + r2 * cp;
    switch ((int)(n & 3)) {
        case 0: *s = sr;  *c = cr;  break;
        case 1: *s = cr;  *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

/* OdometryProvider::rodrigues, OdometryProvider.h:32-67 (f64, row-major 3x3) */
static inline void orc_rodrigues(const double w[3], double R[9])
{
    double rx = w[0], ry = w[1], rz = w[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    if (theta >= 2.2204460492503131e-16) {
        double s, c;
        orc_sincos(theta, &s, &c);
        double c1 = 1.0 - c;
        double itheta = 1.0 / theta;
        rx *= itheta; ry *= itheta; rz *= itheta;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx_[k];
    }
}

/* 3x3 inverse by cofactors (stands in for Eigen's fixed-size inverse) */
#define ORC_INV33(T, NAME)                                                                   \
    static inline void NAME(const T a[9], T o[9])                                            \
    {                                                                                        \
        T c00 = a[4] * a[8] - a[5] * a[7];                                                   \
        T c01 = a[5] * a[6] - a[3] * a[8];                                                   \
        T c02 = a[3] * a[7] - a[4] * a[6];                                                   \
        T det = a[0] * c00 + a[1] * c01 + a[2] * c02;                                        \
        T id = (T)1 / det;                                                                   \
        o[0] = c00 * id;                                                                     \
        o[1] = (a[2] * a[7] - a[1] * a[8]) * id;                                             \
        o[2] = (a[1] * a[5] - a[2] * a[4]) * id;                                             \
        o[3] = c01 * id;                                                                     \
        o[4] = (a[0] * a[8] - a[2] * a[6]) * id;                                             \
        o[5] = (a[2] * a[3] - a[0] * a[5]) * id;                                             \
        o[6] = c02 * id;                                                                     \
        o[7] = (a[1] * a[6] - a[0] * a[7]) * id;                                             \
        o[8] = (a[0] * a[4] - a[1] * a[3]) * id;                                             \
    }
ORC_INV33(float, orc_inv33f)
ORC_INV33(double, orc_inv33d)

static inline void orc_mul33d(const double a[9], const double b[9], double o[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}
static inline void orc_mul33f(const float a[9], const float b[9], float o[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}
static inline void orc_mul44d(const double a[16], const double b[16], double o[16])
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

/* Inverse of a 4x4 whose last row is (0,0,0,1) -- resultRt always has this form
 * (it is a product of such matrices, OdometryProvider.h:72-83).  The linear part
 * is inverted generally (cofactors), as Eigen's general inverse would. */
static inline void orc_inv44_affine_d(const double a[16], double o[16])
{
    double L[9] = {a[0], a[1], a[2], a[4], a[5], a[6], a[8], a[9], a[10]}, Li[9];
    orc_inv33d(L, Li);
    for (int i = 0; i < 3; i++) {
        o[i * 4 + 0] = Li[i * 3 + 0]; o[i * 4 + 1] = Li[i * 3 + 1]; o[i * 4 + 2] = Li[i * 3 + 2];
        o[i * 4 + 3] = -(Li[i * 3 + 0] * a[3] + Li[i * 3 + 1] * a[7] + Li[i * 3 + 2] * a[11]);
    }
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}

/* Symmetric solve A x = b via LDL^T with diagonal pivoting; zero pivots give a
 * zero solution component (the behaviour of Eigen's ldlt().solve(), which the
 * reference relies on when a model has no inliers: RGBDOdometry.cpp:435). */
#define ORC_LDLT(T, NAME, TINY)                                                              \
    static inline void NAME(int n, const T *Ain, const T *b, T *x)                           \
    {                                                                                        \
        T A[36], y[6], d[6];                                                                 \
        int perm[6];                                                                         \
        for (int i = 0; i < n * n; i++) A[i] = Ain[i];                                       \
        for (int i = 0; i < n; i++) perm[i] = i;                                             \
        for (int k = 0; k < n; k++) {                                                        \
            int p = k;                                                                       \
            T best = A[k * n + k] < 0 ? -A[k * n + k] : A[k * n + k];                        \
            for (int i = k + 1; i < n; i++) {                                                \
                T v = A[i * n + i] < 0 ? -A[i * n + i] : A[i * n + i];                       \
                if (v > best) { best = v; p = i; }                                           \
            }                                                                                \
            if (p != k) { /* symmetric swap of rows/cols k and p (full storage) */          \
                for (int j = 0; j < n; j++) { T t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; } \
                for (int i = 0; i < n; i++) { T t = A[i * n + k]; A[i * n + k] = A[i * n + p]; A[i * n + p] = t; } \
                int t = perm[k]; perm[k] = perm[p]; perm[p] = t;                             \
            }                                                                                \
            T akk = A[k * n + k];                                                            \
            d[k] = akk;                                                                      \
            T aabs = akk < 0 ? -akk : akk;                                                   \
            if (aabs > (T)TINY) {                                                            \
                for (int i = k + 1; i < n; i++) A[i * n + k] = A[i * n + k] / akk;           \
                for (int i = k + 1; i < n; i++)                                              \
                    for (int j = k + 1; j <= i; j++) {                                       \
                        A[i * n + j] = A[i * n + j] - A[i * n + k] * akk * A[j * n + k];     \
                        A[j * n + i] = A[i * n + j];                                         \
                    }                                                                        \
            } else {                                                                         \
                for (int i = k + 1; i < n; i++) A[i * n + k] = 0;                            \
            }                                                                                \
        }                                                                                    \
        for (int i = 0; i < n; i++) {                                                        \
            T s = b[perm[i]];                                                                \
            for (int j = 0; j < i; j++) s = s - A[i * n + j] * y[j];                         \
            y[i] = s;                                                                        \
        }                                                                                    \
        for (int i = 0; i < n; i++) {                                                        \
            T aabs = d[i] < 0 ? -d[i] : d[i];                                                \
            y[i] = (aabs > (T)TINY) ? y[i] / d[i] : (T)0;                                    \
        }                                                                                    \
        for (int i = n - 1; i >= 0; i--) {                                                   \
            T s = y[i];                                                                      \
            for (int j = i + 1; j < n; j++) s = s - A[j * n + i] * y[j];                     \
            y[i] = s;                                                                        \
        }                                                                                    \
        for (int i = 0; i < n; i++) x[perm[i]] = y[i];                                       \
    }
ORC_LDLT(double, orc_ldlt_d, 2.2250738585072014e-308)
ORC_LDLT(float, orc_ldlt_f, 1.17549435e-38f)

#endif /* ORC_MATH_H_ */
