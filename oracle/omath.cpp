// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.hpp header). "parity unpinned" vs Eigen.
#include "omath.hpp"

namespace ora {

// Householder tridiagonalisation followed by implicit-shift QL with accumulated eigenvectors.
// Stands in for Eigen::SelfAdjointEigenSolver (marginalization_factor.cpp:268,283; EstimationMapping.hpp:150).
void sym_eigen(const Mat &A, std::vector<double> &w, Mat &V) {
    const int n = A.r;
    V = A;
    std::vector<double> d(n), e(n);
    if (n == 0) { w.clear(); return; }
    for (int j = 0; j < n; j++) d[j] = V(n - 1, j);
    for (int i = n - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; k++) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; j++) { d[j] = V(i - 1, j); V(i, j) = 0.0; V(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; k++) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h = h - f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; j++) e[j] = 0.0;
            for (int j = 0; j < i; j++) {
                f = d[j];
                V(j, i) = f;
                g = e[j] + V(j, j) * f;
                for (int k = j + 1; k <= i - 1; k++) { g += V(k, j) * d[k]; e[k] += V(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; j++) { e[j] /= h; f += e[j] * d[j]; }
            double hh = f / (h + h);
            for (int j = 0; j < i; j++) e[j] -= hh * d[j];
            for (int j = 0; j < i; j++) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; k++) V(k, j) -= (f * e[k] + g * d[k]);
                d[j] = V(i - 1, j);
                V(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; i++) {
        V(n - 1, i) = V(i, i);
        V(i, i) = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; k++) d[k] = V(k, i + 1) / h;
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
                for (int k = 0; k <= i; k++) g += V(k, i + 1) * V(k, j);
                for (int k = 0; k <= i; k++) V(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; k++) V(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; j++) { d[j] = V(n - 1, j); V(n - 1, j) = 0.0; }
    V(n - 1, n - 1) = 1.0;
    e[0] = 0.0;

    for (int i = 1; i < n; i++) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::numeric_limits<double>::epsilon();
    for (int l = 0; l < n; l++) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; m++; }
        if (m > l) {
            int iter = 0;
            do {
                iter++;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; i++) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; i--) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; k++) {
                        h = V(k, i + 1);
                        V(k, i + 1) = s * V(k, i) + c * h;
                        V(k, i) = c * V(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
        }
        d[l] = d[l] + f;
        e[l] = 0.0;
    }
    // ascending sort (selection), permuting eigenvector columns
    for (int i = 0; i < n - 1; i++) {
        int k = i; double p = d[i];
        for (int j = i + 1; j < n; j++) if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i]; d[i] = p;
            for (int j = 0; j < n; j++) std::swap(V(j, i), V(j, k));
        }
    }
    w = d;
}

// Column-pivoted Householder QR least squares for the 5x3 plane fit (EstimationMapping.hpp:198).
V3 colpiv_qr_solve_5x3(const double A_in[15], const double b_in[5]) {
    const int R = 5, C = 3;
    double a[5][3], b[5];
    for (int i = 0; i < R; i++) { for (int j = 0; j < C; j++) a[i][j] = A_in[3 * i + j]; b[i] = b_in[i]; }
    int perm[3] = {0, 1, 2};
    double colnorm2[3];
    for (int j = 0; j < C; j++) { colnorm2[j] = 0; for (int i = 0; i < R; i++) colnorm2[j] += a[i][j] * a[i][j]; }
    double maxpivot = 0.0;
    double rdiag[3] = {0, 0, 0};
    int rank = 0;
    for (int k = 0; k < C; k++) {
        // pick the remaining column of largest (recomputed) squared norm
        int best = k; double bn = -1;
        for (int j = k; j < C; j++) { double s = 0; for (int i = k; i < R; i++) s += a[i][j] * a[i][j]; colnorm2[j] = s; if (s > bn) { bn = s; best = j; } }
        if (best != k) { for (int i = 0; i < R; i++) std::swap(a[i][k], a[i][best]); std::swap(perm[k], perm[best]); std::swap(colnorm2[k], colnorm2[best]); }
        // Householder on column k, rows k..R-1
        double nrm = std::sqrt(colnorm2[k]);
        if (nrm == 0.0) { rdiag[k] = 0; continue; }
        double alpha = a[k][k] > 0 ? -nrm : nrm;
        double v[5];
        for (int i = 0; i < R; i++) v[i] = 0;
        v[k] = a[k][k] - alpha;
        for (int i = k + 1; i < R; i++) v[i] = a[i][k];
        double vtv = 0; for (int i = k; i < R; i++) vtv += v[i] * v[i];
        if (vtv > 0) {
            for (int j = k; j < C; j++) {
                double s = 0; for (int i = k; i < R; i++) s += v[i] * a[i][j];
                s = 2 * s / vtv;
                for (int i = k; i < R; i++) a[i][j] -= s * v[i];
            }
            double s = 0; for (int i = k; i < R; i++) s += v[i] * b[i];
            s = 2 * s / vtv;
            for (int i = k; i < R; i++) b[i] -= s * v[i];
        }
        rdiag[k] = a[k][k];
        if (std::fabs(rdiag[k]) > maxpivot) maxpivot = std::fabs(rdiag[k]);
    }
    const double thresh = std::numeric_limits<double>::epsilon() * 3.0 * maxpivot;
    for (int k = 0; k < C; k++) if (std::fabs(rdiag[k]) > thresh) rank++;
    double z[3] = {0, 0, 0};
    for (int k = rank - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < rank; j++) s -= a[k][j] * z[j];
        z[k] = s / a[k][k];
    }
    double x[3] = {0, 0, 0};
    for (int k = 0; k < C; k++) x[perm[k]] = z[k];
    return V3(x[0], x[1], x[2]);
}

}  // namespace ora
