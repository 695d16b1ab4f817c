// gn_ref_host.h -- host algebra of the REFERENCE-ORDER tracker (cf_set_icp_arith 2, track_ref.hip).
//
// RGBDOdometry::getIncrementalTransformation (Core/Utils/RGBDOdometry.cpp:217-477) solves on the host with Eigen's fixed-size
// matrices.  In this mode the library does the same -- on the host, in f64 / f32 as the reference's text says -- and in the operation
// ORDER of the classes that text instantiates: products accumulated left to right in the inner index, 3x3 inverse by cofactors with
// the determinant along the first column, 4x4 inverse by cofactors of 3x3 minors with the determinant along the first row, the
// unblocked left-looking LDL^T with diagonal pivoting (first maximum wins) and its solve (P, L^-1, D^-1 with |d| <= 1 / max -> 0, L^-T,
// P^T), the isometry composition of Isometry3f.  Eigen itself is not in this image: the order is the one the repository's pins are
// generated with (the reference's RGBDOdometry class compiled from /root/reference for tests/golden/ref_odo_v1.npz and
// ref_traj_v1.npz), and tests/test_refpin_gpu.py holds this mode against those fixtures bit for bit.  cos / sin are the C library's,
// as in OdometryProvider::rodrigues (OdometryProvider.h:48-49).  Compiled with -ffp-contract=off like everything else.
#pragma once
#include <float.h>
#include <math.h>
#include <string.h>

namespace cf {
namespace refhost {

template <class T> inline void mul33(const T a[9], const T b[9], T o[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            T s = a[i * 3 + 0] * b[0 * 3 + j];
            s = s + a[i * 3 + 1] * b[1 * 3 + j];
            s = s + a[i * 3 + 2] * b[2 * 3 + j];
            o[i * 3 + j] = s;
        }
}
template <class T> inline void mul33v(const T a[9], const T v[3], T o[3])
{
    for (int i = 0; i < 3; i++) { T s = a[i * 3 + 0] * v[0]; s = s + a[i * 3 + 1] * v[1]; s = s + a[i * 3 + 2] * v[2]; o[i] = s; }
}
inline void mul44(const double a[16], const double b[16], double o[16])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = a[i * 4 + 0] * b[0 * 4 + j];
            for (int k = 1; k < 4; k++) s = s + a[i * 4 + k] * b[k * 4 + j];
            o[i * 4 + j] = s;
        }
}
template <class T> inline void inv33(const T m[9], T o[9])
{
    T cof[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            cof[i][j] = m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
        }
    const T det = (cof[0][0] * m[0] + cof[1][0] * m[3]) + cof[2][0] * m[6];
    const T invdet = T(1) / det;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) o[r * 3 + c] = cof[c][r] * invdet;
}
inline double minor3(const double m[16], int r, int c)
{
    int ri[3], ci[3];
    for (int k = 0, t = 0; k < 4; k++) if (k != r) ri[t++] = k;
    for (int k = 0, t = 0; k < 4; k++) if (k != c) ci[t++] = k;
    auto M = [&](int a, int b) { return m[a * 4 + b]; };
    return M(ri[0], ci[0]) * (M(ri[1], ci[1]) * M(ri[2], ci[2]) - M(ri[1], ci[2]) * M(ri[2], ci[1])) -
           M(ri[0], ci[1]) * (M(ri[1], ci[0]) * M(ri[2], ci[2]) - M(ri[1], ci[2]) * M(ri[2], ci[0])) +
           M(ri[0], ci[2]) * (M(ri[1], ci[0]) * M(ri[2], ci[1]) - M(ri[1], ci[1]) * M(ri[2], ci[0]));
}
inline void inv44(const double m[16], double o[16])
{
    double cofm[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) cofm[r][c] = ((r + c) & 1) ? -minor3(m, r, c) : minor3(m, r, c);
    const double det = ((m[0] * cofm[0][0] + m[1] * cofm[0][1]) + m[2] * cofm[0][2]) + m[3] * cofm[0][3];
    const double invdet = 1.0 / det;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) o[r * 4 + c] = cofm[c][r] * invdet;
}
template <class T> inline T tabs(T v) { return v < 0 ? -v : v; }
template <class T, int N> inline void ldlt_solve(const T* Ain, const T* b, T* x, T tmax)
{
    T a[N * N], y[N], temp[N];
    int tr[N];
    for (int i = 0; i < N * N; i++) a[i] = Ain[i];
    for (int k = 0; k < N; k++) {
        int big = k;
        T best = tabs(a[k * N + k]);
        for (int i = k + 1; i < N; i++) { const T v = tabs(a[i * N + i]); if (v > best) { best = v; big = i; } }
        tr[k] = big;
        if (big != k) {
            for (int j = 0; j < k; j++) { const T t = a[k * N + j]; a[k * N + j] = a[big * N + j]; a[big * N + j] = t; }
            for (int i = big + 1; i < N; i++) { const T t = a[i * N + k]; a[i * N + k] = a[i * N + big]; a[i * N + big] = t; }
            { const T t = a[k * N + k]; a[k * N + k] = a[big * N + big]; a[big * N + big] = t; }
            for (int i = k + 1; i < big; i++) { const T t = a[i * N + k]; a[i * N + k] = a[big * N + i]; a[big * N + i] = t; }
        }
        if (k > 0) {
            for (int j = 0; j < k; j++) temp[j] = a[j * N + j] * a[k * N + j];
            { T s = a[k * N + 0] * temp[0]; for (int j = 1; j < k; j++) s = s + a[k * N + j] * temp[j]; a[k * N + k] = a[k * N + k] - s; }
            for (int i = k + 1; i < N; i++) { T s = a[i * N + 0] * temp[0]; for (int j = 1; j < k; j++) s = s + a[i * N + j] * temp[j]; a[i * N + k] = a[i * N + k] - s; }
        }
        const T akk = a[k * N + k];
        if (tabs(akk) > T(0)) for (int i = k + 1; i < N; i++) a[i * N + k] = a[i * N + k] / akk;
    }
    for (int i = 0; i < N; i++) y[i] = b[i];
    for (int k = 0; k < N; k++) { const T t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 1; i < N; i++) { T s = a[i * N + 0] * y[0]; for (int j = 1; j < i; j++) s = s + a[i * N + j] * y[j]; y[i] = y[i] - s; }
    const T tol = T(1) / tmax;
    for (int i = 0; i < N; i++) { const T d = a[i * N + i]; y[i] = (tabs(d) > tol) ? y[i] / d : T(0); }
    for (int i = N - 2; i >= 0; i--) { T s = a[(i + 1) * N + i] * y[i + 1]; for (int j = i + 2; j < N; j++) s = s + a[j * N + i] * y[j]; y[i] = y[i] - s; }
    for (int k = N - 1; k >= 0; k--) { const T t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < N; i++) x[i] = y[i];
}
// OdometryProvider::rodrigues (OdometryProvider.h:32-67) as written
inline void rodrigues(const double src[3], double R[9])
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
inline void k_matrix(float fx, float fy, float cx, float cy, double K[9])
{
    for (int i = 0; i < 9; i++) K[i] = 0;
    K[0] = fx; K[4] = fy; K[2] = cx; K[5] = cy; K[8] = 1;
}
// reduce.cu:481-498: the 29 f32 totals -> A (row-major, symmetric-filled), b, residual
inline void unpack29(const float h[29], float A[36], float b[6], float* residual)
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

}  // namespace refhost
}  // namespace cf
