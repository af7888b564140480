// ORACLE — TEST INFRASTRUCTURE ONLY. Not linked into, imported by, or called from the product path.
// Parity status: "parity unpinned" against real Ceres/Eigen/PCL (the reference ships no golden vectors and
// cannot be built here); pinned by analytic-vs-numeric Jacobian checks, marginalization
// identities and independent numpy/scipy cross-checks in tests/.
//
// omath.hpp — dependency-free fp64 restatement of the Eigen 3.3.7 operations the reference hot path uses
// (Quaterniond product / inverse / toRotationMatrix / Matrix3d->Quaterniond, LLT, PartialPivLU inverse,
// SelfAdjointEigenSolver, colPivHouseholderQr) and of vins_estimator/utility/utility.h:16-143.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <limits>

namespace ora {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    explicit V3(const double *p) : x(p[0]), y(p[1]), z(p[2]) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

// 3x3 row-major
struct M3 {
    double m[9];
    M3() { std::memset(m, 0, sizeof(m)); }
    double operator()(int r, int c) const { return m[3 * r + c]; }
    double &operator()(int r, int c) { return m[3 * r + c]; }
    static M3 Identity() { M3 a; a.m[0] = a.m[4] = a.m[8] = 1; return a; }
    static M3 from(const double *p) { M3 a; std::memcpy(a.m, p, sizeof(a.m)); return a; }
};
inline M3 operator*(const M3 &a, const M3 &b) {
    M3 c;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += a(i, k) * b(k, j); c(i, j) = s; }
    return c;
}
inline V3 operator*(const M3 &a, V3 v) { return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z, a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z}; }
inline M3 operator*(double s, const M3 &a) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = s * a.m[i]; return c; }
inline M3 operator*(const M3 &a, double s) { return s * a; }
inline M3 operator+(const M3 &a, const M3 &b) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] + b.m[i]; return c; }
inline M3 operator-(const M3 &a, const M3 &b) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] - b.m[i]; return c; }
inline M3 operator-(const M3 &a) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = -a.m[i]; return c; }
inline M3 transpose(const M3 &a) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c(i, j) = a(j, i); return c; }

// Utility::skewSymmetric (utility.h:31-39)
inline M3 skew(V3 q) {
    M3 a;
    a(0, 1) = -q.z; a(0, 2) = q.y;
    a(1, 0) = q.z; a(1, 2) = -q.x;
    a(2, 0) = -q.y; a(2, 1) = q.x;
    return a;
}

// Eigen::Quaterniond (w,x,y,z members; coeffs() order is x,y,z,w)
struct Q4 {
    double w = 1, x = 0, y = 0, z = 0;
    Q4() {}
    Q4(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
    V3 vec() const { return {x, y, z}; }
    static Q4 from_xyzw(const double *p) { return Q4(p[3], p[0], p[1], p[2]); }
    void to_xyzw(double *p) const { p[0] = x; p[1] = y; p[2] = z; p[3] = w; }
};
inline Q4 operator*(const Q4 &a, const Q4 &b) {
    return Q4(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
              a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
inline double sqnorm(const Q4 &q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
inline Q4 conj(const Q4 &q) { return Q4(q.w, -q.x, -q.y, -q.z); }
// Eigen QuaternionBase::inverse(): conjugate / squaredNorm (true inverse, also for non-unit q)
inline Q4 inverse(const Q4 &q) {
    double n2 = sqnorm(q);
    if (n2 > 0) return Q4(q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2);
    return Q4(0, 0, 0, 0);
}
inline Q4 normalized(const Q4 &q) { double n = std::sqrt(sqnorm(q)); return Q4(q.w / n, q.x / n, q.y / n, q.z / n); }
// Eigen QuaternionBase::_transformVector
inline V3 operator*(const Q4 &q, V3 v) {
    V3 uv = cross(q.vec(), v);
    uv = uv + uv;
    return v + q.w * uv + cross(q.vec(), uv);
}
// Eigen QuaternionBase::toRotationMatrix
inline M3 toR(const Q4 &q) {
    M3 r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1 - (txx + tyy);
    return r;
}
// Eigen Quaterniond(Matrix3d) (quaternionbase_assign_impl<Other,3,3>)
inline Q4 fromR(const M3 &m) {
    Q4 q;
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m(2, 1) - m(1, 2)) * t;
        q.y = (m(0, 2) - m(2, 0)) * t;
        q.z = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double v[3];
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m(k, j) - m(j, k)) * t;
        v[j] = (m(j, i) + m(i, j)) * t;
        v[k] = (m(k, i) + m(i, k)) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

// Utility::deltaQ (utility.h:16-29): UNNORMALISED first-order quaternion (1, theta/2)
inline Q4 deltaQ(V3 theta) { return Q4(1.0, theta.x / 2.0, theta.y / 2.0, theta.z / 2.0); }

// 4x4 row-major, quaternion order (w, x, y, z). Utility::Qleft / Qright (utility.h:51-69); positify = identity.
struct M4 { double m[16]; double operator()(int r, int c) const { return m[4 * r + c]; } double &operator()(int r, int c) { return m[4 * r + c]; } };
inline M4 Qleft(const Q4 &q) {
    M4 a;
    M3 s = skew(q.vec());
    a(0, 0) = q.w; a(0, 1) = -q.x; a(0, 2) = -q.y; a(0, 3) = -q.z;
    a(1, 0) = q.x; a(2, 0) = q.y; a(3, 0) = q.z;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a(1 + i, 1 + j) = (i == j ? q.w : 0.0) + s(i, j);
    return a;
}
inline M4 Qright(const Q4 &p) {
    M4 a;
    M3 s = skew(p.vec());
    a(0, 0) = p.w; a(0, 1) = -p.x; a(0, 2) = -p.y; a(0, 3) = -p.z;
    a(1, 0) = p.x; a(2, 0) = p.y; a(3, 0) = p.z;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a(1 + i, 1 + j) = (i == j ? p.w : 0.0) - s(i, j);
    return a;
}
inline M4 operator*(const M4 &a, const M4 &b) {
    M4 c;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += a(i, k) * b(k, j); c(i, j) = s; }
    return c;
}
inline M3 bottomRight3(const M4 &a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = a(1 + i, 1 + j); return r; }

// Utility::R2ypr / ypr2R (utility.h:70-114), degrees
inline V3 R2ypr(const M3 &R) {
    V3 n(R(0, 0), R(1, 0), R(2, 0)), o(R(0, 1), R(1, 1), R(2, 1)), a(R(0, 2), R(1, 2), R(2, 2));
    double y = std::atan2(n.y, n.x);
    double p = std::atan2(-n.z, n.x * std::cos(y) + n.y * std::sin(y));
    double r = std::atan2(a.x * std::sin(y) - a.y * std::cos(y), -o.x * std::sin(y) + o.y * std::cos(y));
    return V3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
inline M3 ypr2R(V3 ypr) {
    double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
    M3 Rz, Ry, Rx;
    Rz(0, 0) = std::cos(y); Rz(0, 1) = -std::sin(y); Rz(1, 0) = std::sin(y); Rz(1, 1) = std::cos(y); Rz(2, 2) = 1;
    Ry(0, 0) = std::cos(p); Ry(0, 2) = std::sin(p); Ry(1, 1) = 1; Ry(2, 0) = -std::sin(p); Ry(2, 2) = std::cos(p);
    Rx(0, 0) = 1; Rx(1, 1) = std::cos(r); Rx(1, 2) = -std::sin(r); Rx(2, 1) = std::sin(r); Rx(2, 2) = std::cos(r);
    return Rz * Ry * Rx;
}
// Utility::normalizeAngle (utility.h:135-143), degrees
inline double normalizeAngle(double a) {
    const double two_pi = 360.0;
    if (a > 0) return a - two_pi * std::floor((a + 180.0) / two_pi);
    return a + two_pi * std::floor((-a + 180.0) / two_pi);
}

// ---------------------------------------------------------------------------------------------------
// Dynamic dense matrix, row-major.
struct Mat {
    int r = 0, c = 0;
    std::vector<double> d;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
    double &operator()(int i, int j) { return d[(size_t)i * c + j]; }
    void resize(int r_, int c_) { r = r_; c = c_; d.assign((size_t)r_ * c_, 0.0); }
    double *row(int i) { return d.data() + (size_t)i * c; }
    const double *row(int i) const { return d.data() + (size_t)i * c; }
};
inline Mat matmul(const Mat &a, const Mat &b) {
    Mat o(a.r, b.c);
    for (int i = 0; i < a.r; i++)
        for (int k = 0; k < a.c; k++) {
            double aik = a(i, k);
            if (aik == 0.0) continue;
            const double *bk = b.row(k);
            double *oi = o.row(i);
            for (int j = 0; j < b.c; j++) oi[j] += aik * bk[j];
        }
    return o;
}
inline Mat transpose(const Mat &a) { Mat o(a.c, a.r); for (int i = 0; i < a.r; i++) for (int j = 0; j < a.c; j++) o(j, i) = a(i, j); return o; }

// In-place lower Cholesky A = L L^T (Eigen LLT semantics: fails on non-positive pivot). Returns false on failure.
inline bool cholesky_lower(Mat &a) {
    int n = a.r;
    for (int j = 0; j < n; j++) {
        double s = a(j, j);
        for (int k = 0; k < j; k++) s -= a(j, k) * a(j, k);
        if (!(s > 0.0)) return false;
        double l = std::sqrt(s);
        a(j, j) = l;
        for (int i = j + 1; i < n; i++) {
            double t = a(i, j);
            const double *ai = a.row(i), *aj = a.row(j);
            for (int k = 0; k < j; k++) t -= ai[k] * aj[k];
            a(i, j) = t / l;
        }
    }
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) a(i, j) = 0.0;
    return true;
}
inline void chol_solve(const Mat &L, double *b) {  // solves L L^T x = b in place
    int n = L.r;
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L(i, k) * b[k]; b[i] = s / L(i, i); }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L(k, i) * b[k]; b[i] = s / L(i, i); }
}
// General inverse by LU with partial pivoting (Eigen MatrixBase::inverse() for dynamic/large fixed = PartialPivLU)
inline bool inverse_pplu(const Mat &a_in, Mat &inv) {
    int n = a_in.r;
    Mat a = a_in;
    inv.resize(n, n);
    for (int i = 0; i < n; i++) inv(i, i) = 1.0;
    for (int k = 0; k < n; k++) {
        int p = k; double best = std::fabs(a(k, k));
        for (int i = k + 1; i < n; i++) if (std::fabs(a(i, k)) > best) { best = std::fabs(a(i, k)); p = i; }
        if (best == 0.0) return false;
        if (p != k) for (int j = 0; j < n; j++) { std::swap(a(k, j), a(p, j)); std::swap(inv(k, j), inv(p, j)); }
        double piv = a(k, k);
        for (int i = k + 1; i < n; i++) {
            double f = a(i, k) / piv;
            if (f == 0.0) continue;
            for (int j = k; j < n; j++) a(i, j) -= f * a(k, j);
            for (int j = 0; j < n; j++) inv(i, j) -= f * inv(k, j);
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double piv = a(k, k);
        for (int j = 0; j < n; j++) inv(k, j) /= piv;
        for (int i = 0; i < k; i++) {
            double f = a(i, k);
            if (f == 0.0) continue;
            for (int j = 0; j < n; j++) inv(i, j) -= f * inv(k, j);
        }
    }
    return true;
}

// Symmetric eigen-decomposition A = V diag(w) V^T, eigenvalues ascending (Eigen SelfAdjointEigenSolver order).
// Householder tridiagonalisation + implicit QL (own restatement of the classic tred2/tql2 scheme, which is the
// same algorithm family Eigen uses: tridiagonalisation + implicit symmetric QR).
void sym_eigen(const Mat &A, std::vector<double> &w, Mat &V);
// 3x3 specialisation through the same routine
inline void sym_eigen3(const M3 &A, double w[3], M3 &V) {
    Mat a(3, 3), v; std::vector<double> ww;
    for (int i = 0; i < 9; i++) a.d[i] = A.m[i];
    sym_eigen(a, ww, v);
    for (int i = 0; i < 3; i++) w[i] = ww[i];
    for (int i = 0; i < 9; i++) V.m[i] = v.d[i];
}
// Least-squares solve of the 5x3 system by column-pivoted Householder QR (Eigen colPivHouseholderQr().solve()).
V3 colpiv_qr_solve_5x3(const double A[15], const double b[5]);

}  // namespace ora
