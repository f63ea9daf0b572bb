// oracle_geometry.cpp — TEST INFRASTRUCTURE ONLY (see vo_oracle.h; parity unpinned).
// CPU restatement of the OpenCV 3.4.x calib3d/core code behind the reference's geometry calls:
//   cv::findEssentialMat (calib3d/src/five-point.cpp + ptsetreg.cpp)   SURVEY.md A.4-A.6
//   cv::recoverPose / decomposeEssentialMat / triangulatePoints        SURVEY.md A.7
//   cv::SVD (core/src/lapack.cpp JacobiSVDImpl_), cv::solvePoly (core/src/mathfuncs.cpp)
// Reference call sites: src/matcher/matcher.cpp:42, src/visual_odometry/vo_pose.h:14,19,22,
// src/sfm/triangulation.cpp:13-22,34-44,46-66.
#include "oracle_common.h"

#include <cfloat>
#include <complex>

namespace vo_oracle {

// ---------------------------------------------------------------------------------------
// cv::SVD::compute — one-sided Jacobi (Hestenes) on the rows of At, as JacobiSVDImpl_<double>.
// At: n rows of length m (row stride astep); Vt: n x n. n1 = number of At rows to finish as
// orthonormal vectors (rows >= n are completed from a fixed pseudo-random sequence).
// ---------------------------------------------------------------------------------------
static void jacobi_svd_impl(double* At, int astep, double* Wout, double* Vt, int vstep, int m, int n,
                            int n1) {
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    std::vector<double> W(n);
    int i, j, k, iter, max_iter = std::max(m, 30);
    double c, s, sd;

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }
    for (iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = hypot(p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = std::sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            if (Vt) {
                for (k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
                for (k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
            }
        }
    }
    for (i = 0; i < n; i++) Wout[i] = W[i];
    if (!Vt) return;

    RNG rng(0x12345678);
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            // zero singular value: take a pseudo-random +-1/m vector, remove its projections on the
            // vectors already computed, normalise (two passes)
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) {
                double val = (rng.next() & 256) != 0 ? val0 : -val0;
                At[i * astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += std::abs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
                sd = 0;
                for (k = 0; k < m; k++) {
                    double t = At[i * astep + k];
                    sd += t * t;
                }
                sd = std::sqrt(sd);
            }
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

// A: rows x cols row-major.  w: min(rows,cols).  U: rows x (full ? rows : min), Vt: (full ? cols :
// min) x cols.  Either of U / Vt may be null.
static void svd_compute(const double* A, int rows, int cols, double* w, double* U, double* Vt,
                        bool full_uv) {
    int m = rows, n = cols;
    bool at = false;
    if (m < n) {
        std::swap(m, n);
        at = true;
    }
    int urows = full_uv ? m : n;
    std::vector<double> tu((size_t)urows * m, 0.0);  // first n rows double as temp_a
    std::vector<double> tv((size_t)n * n, 0.0);
    if (!at) {
        for (int i = 0; i < n; i++)
            for (int k = 0; k < m; k++) tu[(size_t)i * m + k] = A[(size_t)k * cols + i];
    } else {
        for (int i = 0; i < n; i++)
            for (int k = 0; k < m; k++) tu[(size_t)i * m + k] = A[(size_t)i * cols + k];
    }
    std::vector<double> wv(n);
    jacobi_svd_impl(tu.data(), m, wv.data(), tv.data(), n, m, n, urows);
    if (w)
        for (int i = 0; i < n; i++) w[i] = wv[i];
    if (!at) {
        if (U)  // U = temp_u^T : m x urows
            for (int i = 0; i < m; i++)
                for (int j = 0; j < urows; j++) U[(size_t)i * urows + j] = tu[(size_t)j * m + i];
        if (Vt) memcpy(Vt, tv.data(), sizeof(double) * n * n);
    } else {
        if (U)  // U = temp_v^T : n x n
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) U[(size_t)i * n + j] = tv[(size_t)j * n + i];
        if (Vt) memcpy(Vt, tu.data(), sizeof(double) * urows * m);
    }
}

// ---------------------------------------------------------------------------------------
// cv::solvePoly — Durand-Kerner, fixed start (1,0)*(1,1)^i.  coeffs[k] multiplies z^k.
// ---------------------------------------------------------------------------------------
typedef std::complex<double> C;
static int solve_poly(const double* coeffs0, int n0, C* roots, int maxIters) {
    int n = n0;
    std::vector<C> coeffs(n + 1);
    for (int i = 0; i <= n; i++) coeffs[i] = C(coeffs0[i], 0);
    for (; n > 1; n--)
        if (std::abs(coeffs[n].real()) + std::abs(coeffs[n].imag()) > DBL_EPSILON) break;
    C p(1, 0), r(1, 1);
    for (int i = 0; i < n; i++) {
        roots[i] = p;
        p = p * r;
    }
    maxIters = maxIters <= 0 ? 1000 : maxIters;
    for (int iter = 0; iter < maxIters; iter++) {
        double maxDiff = 0;
        for (int i = 0; i < n; i++) {
            p = roots[i];
            C num = coeffs[n], denom = coeffs[n];
            int num_same_root = 1;
            for (int j = 0; j < n; j++) {
                num = num * p + coeffs[n - j - 1];
                if (j != i) {
                    C d = p - roots[j];
                    if (d.real() != 0 || d.imag() != 0) denom = denom * d;
                    else num_same_root++;
                }
            }
            num /= denom;
            // cv additionally takes a k-th root of the step when k iterates coincide exactly
            // (num_same_root > 1); generic data never produces bit-identical iterates, so the
            // plain step is kept here.
            (void)num_same_root;
            roots[i] = p - num;
            maxDiff = std::max(maxDiff, std::abs(num));
        }
        if (maxDiff <= 0) break;
    }
    for (int i = 0; i < n; i++)
        if (std::fabs(roots[i].imag()) < 1e-100) roots[i] = C(roots[i].real(), 0);
    return n;
}

// ---------------------------------------------------------------------------------------
// Nister five-point solver (EMEstimatorCallback::runKernel)
// ---------------------------------------------------------------------------------------
struct Poly3 {  // polynomial in (x,y,z), total degree <= 3, dense by exponent
    double c[4][4][4];
    Poly3() { memset(c, 0, sizeof(c)); }
};
static Poly3 pmul(const Poly3& a, const Poly3& b) {
    Poly3 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j + i < 4; j++)
            for (int k = 0; k + j + i < 4; k++) {
                double av = a.c[i][j][k];
                if (av == 0) continue;
                for (int u = 0; u + i < 4; u++)
                    for (int v = 0; v + j < 4; v++)
                        for (int q = 0; q + k < 4; q++) {
                            if (i + j + k + u + v + q > 3) continue;
                            r.c[i + u][j + v][k + q] += av * b.c[u][v][q];
                        }
            }
    return r;
}
static Poly3 padd(const Poly3& a, const Poly3& b, double sb = 1.0) {
    Poly3 r;
    for (int i = 0; i < 64; i++) (&r.c[0][0][0])[i] = (&a.c[0][0][0])[i] + sb * (&b.c[0][0][0])[i];
    return r;
}
// column order of the 10 x 20 constraint matrix (Nister): the first ten are eliminated
static const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0},
                                 {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1},
                                 {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2},
                                 {0, 0, 1}, {0, 0, 0}};

static void poly_mul1d(const double* a, int da, const double* b, int db, double* r) {  // lowest first
    for (int i = 0; i <= da + db; i++) r[i] = 0;
    for (int i = 0; i <= da; i++)
        for (int j = 0; j <= db; j++) r[i + j] += a[i] * b[j];
}

// x1, x2: 5 x 2 normalised points; returns number of models, E_out row-major 3x3 each.
static int five_point(const double* x1, const double* x2, double* E_out) {
    // Q rows: [x2x1, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1]  (x2^T E x1 = 0, E row-major)
    double Q[5 * 9];
    for (int i = 0; i < 5; i++) {
        double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double* q = Q + 9 * i;
        q[0] = u2 * u1; q[1] = u2 * v1; q[2] = u2;
        q[3] = v2 * u1; q[4] = v2 * v1; q[5] = v2;
        q[6] = u1;      q[7] = v1;      q[8] = 1.0;
    }
    double W[5], Vt[81];
    svd_compute(Q, 5, 9, W, nullptr, Vt, true);
    const double* EE[4] = {Vt + 5 * 9, Vt + 6 * 9, Vt + 7 * 9, Vt + 8 * 9};  // null-space basis

    // E(x,y,z) = x EE0 + y EE1 + z EE2 + EE3
    Poly3 Ep[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Ep[i][j].c[1][0][0] = EE[0][3 * i + j];
            Ep[i][j].c[0][1][0] = EE[1][3 * i + j];
            Ep[i][j].c[0][0][1] = EE[2][3 * i + j];
            Ep[i][j].c[0][0][0] = EE[3][3 * i + j];
        }
    Poly3 cons[10];
    // det(E) = 0
    {
        Poly3 m0 = padd(pmul(Ep[1][1], Ep[2][2]), pmul(Ep[1][2], Ep[2][1]), -1.0);
        Poly3 m1 = padd(pmul(Ep[1][0], Ep[2][2]), pmul(Ep[1][2], Ep[2][0]), -1.0);
        Poly3 m2 = padd(pmul(Ep[1][0], Ep[2][1]), pmul(Ep[1][1], Ep[2][0]), -1.0);
        cons[0] = padd(padd(pmul(Ep[0][0], m0), pmul(Ep[0][1], m1), -1.0), pmul(Ep[0][2], m2));
    }
    // (E E^T - 1/2 trace(E E^T) I) E = 0
    {
        Poly3 EEt[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) EEt[i][j] = padd(EEt[i][j], pmul(Ep[i][k], Ep[j][k]));
        Poly3 tr = padd(padd(EEt[0][0], EEt[1][1]), EEt[2][2]);
        for (int i = 0; i < 3; i++) EEt[i][i] = padd(EEt[i][i], tr, -0.5);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                Poly3 acc;
                for (int k = 0; k < 3; k++) acc = padd(acc, pmul(EEt[i][k], Ep[k][j]));
                cons[1 + 3 * i + j] = acc;
            }
    }
    double A[10][20];
    for (int r = 0; r < 10; r++)
        for (int cidx = 0; cidx < 20; cidx++)
            A[r][cidx] = cons[r].c[kMono[cidx][0]][kMono[cidx][1]][kMono[cidx][2]];

    // A <- A[:, :10]^-1 A[:, 10:]  (LU with partial pivoting, as Mat::inv(DECOMP_LU))
    for (int col = 0; col < 10; col++) {
        int piv = col;
        for (int r = col + 1; r < 10; r++)
            if (std::abs(A[r][col]) > std::abs(A[piv][col])) piv = r;
        if (std::abs(A[piv][col]) < DBL_EPSILON * 1e-3) return 0;  // singular: cv's inv() yields zeros
        if (piv != col)
            for (int k = 0; k < 20; k++) std::swap(A[piv][k], A[col][k]);
        double d = 1.0 / A[col][col];
        for (int k = 0; k < 20; k++) A[col][k] *= d;
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            double f = A[r][col];
            if (f == 0) continue;
            for (int k = 0; k < 20; k++) A[r][k] -= f * A[col][k];
        }
    }
    // B(z): rows k = e - z f, l = g - z h, m = i - z j; entries lowest degree first
    double B[3][3][5];
    for (int i = 0; i < 3; i++) {
        const double* r1 = &A[4 + 2 * i][10];
        const double* r2 = &A[5 + 2 * i][10];
        // x block (degree 3), y block (degree 3): highest-first [-f0, r0-f1, r1-f2, r2]
        for (int blk = 0; blk < 2; blk++) {
            const double* a = r1 + 3 * blk;
            const double* f = r2 + 3 * blk;
            B[i][blk][3] = -f[0];
            B[i][blk][2] = a[0] - f[1];
            B[i][blk][1] = a[1] - f[2];
            B[i][blk][0] = a[2];
            B[i][blk][4] = 0;
        }
        const double* a = r1 + 6;
        const double* f = r2 + 6;
        B[i][2][4] = -f[0];
        B[i][2][3] = a[0] - f[1];
        B[i][2][2] = a[1] - f[2];
        B[i][2][1] = a[2] - f[3];
        B[i][2][0] = a[3];
    }
    // det B(z): degree 10
    double c[11];
    for (int k = 0; k <= 10; k++) c[k] = 0;
    auto minor2 = [&](int r0, int r1, int c0, int d0, int c1, int d1, double* out) {
        // B[r0][c0]*B[r1][c1] - B[r0][c1]*B[r1][c0]; degree d0+d1
        double t0[9], t1[9];
        poly_mul1d(B[r0][c0], d0, B[r1][c1], d1, t0);
        poly_mul1d(B[r0][c1], d1, B[r1][c0], d0, t1);
        for (int k = 0; k <= d0 + d1; k++) out[k] = t0[k] - t1[k];
    };
    {
        double m[9], t[12];
        minor2(1, 2, 1, 3, 2, 4, m);  // B11 B22 - B12 B21 (deg 7)
        poly_mul1d(B[0][0], 3, m, 7, t);
        for (int k = 0; k <= 10; k++) c[k] += t[k];
        minor2(1, 2, 0, 3, 2, 4, m);  // B10 B22 - B12 B20 (deg 7)
        poly_mul1d(B[0][1], 3, m, 7, t);
        for (int k = 0; k <= 10; k++) c[k] -= t[k];
        minor2(1, 2, 0, 3, 1, 3, m);  // B10 B21 - B11 B20 (deg 6)
        poly_mul1d(B[0][2], 4, m, 6, t);
        for (int k = 0; k <= 10; k++) c[k] += t[k];
    }
    C roots[10];
    int nroots = solve_poly(c, 10, roots, 300);
    int count = 0;
    for (int i = 0; i < nroots; i++) {
        if (std::fabs(roots[i].imag()) > 1e-10) continue;
        double z1 = roots[i].real();
        double z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; j++) {
            bz[3 * j + 0] = B[j][0][3] * z3 + B[j][0][2] * z2 + B[j][0][1] * z1 + B[j][0][0];
            bz[3 * j + 1] = B[j][1][3] * z3 + B[j][1][2] * z2 + B[j][1][1] * z1 + B[j][1][0];
            bz[3 * j + 2] = B[j][2][4] * z4 + B[j][2][3] * z3 + B[j][2][2] * z2 + B[j][2][1] * z1 + B[j][2][0];
        }
        double w3[3], vt3[9];
        svd_compute(bz, 3, 3, w3, nullptr, vt3, false);  // SVD::solveZ: last row of Vt
        const double* xy1 = vt3 + 6;
        if (std::fabs(xy1[2]) < 1e-10) continue;
        double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2], zs = z1;
        double Ev[9], nrm = 0;
        for (int k = 0; k < 9; k++) {
            Ev[k] = EE[0][k] * xs + EE[1][k] * ys + EE[2][k] * zs + EE[3][k];
            nrm += Ev[k] * Ev[k];
        }
        nrm = std::sqrt(nrm);
        for (int k = 0; k < 9; k++) E_out[count * 9 + k] = Ev[k] / nrm;
        count++;
    }
    return count;
}

// EMEstimatorCallback::computeError
static void sampson_err(const double* E, const double* x1, const double* x2, int n, float* err) {
    for (int i = 0; i < n; i++) {
        double a1 = x1[2 * i], b1 = x1[2 * i + 1], a2 = x2[2 * i], b2 = x2[2 * i + 1];
        double Ex1[3], Etx2[3];
        for (int r = 0; r < 3; r++) Ex1[r] = E[3 * r] * a1 + E[3 * r + 1] * b1 + E[3 * r + 2] * 1.0;
        for (int r = 0; r < 3; r++) Etx2[r] = E[r] * a2 + E[3 + r] * b2 + E[6 + r] * 1.0;
        double x2tEx1 = a2 * Ex1[0] + b2 * Ex1[1] + 1.0 * Ex1[2];
        double a = Ex1[0] * Ex1[0];
        double b = Ex1[1] * Ex1[1];
        double c = Etx2[0] * Etx2[0];
        double d = Etx2[1] * Etx2[1];
        err[i] = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
    }
}

static int find_inliers(const double* E, const double* x1, const double* x2, int n, double thresh,
                        std::vector<float>& err, std::vector<uint8_t>& mask) {
    err.resize(n);
    mask.resize(n);
    sampson_err(E, x1, x2, n, err.data());
    float t = (float)(thresh * thresh);
    int good = 0;
    for (int i = 0; i < n; i++) {
        int f = err[i] <= t;
        mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

static int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cvRound(num / denom);
}

// PointSetRegistrator::getSubset for the E callback (checkSubset always true)
static void get_subset(RNG& rng, int count, int idx[5]) {
    for (int i = 0; i < 5; i++) {
        for (;;) {
            int idx_i = idx[i] = rng.uniform(0, count);
            int j;
            for (j = 0; j < i; j++)
                if (idx_i == idx[j]) break;
            if (j == i) break;
        }
    }
}

static int find_essential(const float* p0, const float* p1, int n, const double* K, int method,
                          double prob, double threshold, uint64_t seed, int max_iters, double* E,
                          uint8_t* mask, int* stats) {
    const int modelPoints = 5;
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> x1(2 * (size_t)n), x2(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        x1[2 * i] = ((double)p0[2 * i] - cx) / fx;
        x1[2 * i + 1] = ((double)p0[2 * i + 1] - cy) / fy;
        x2[2 * i] = ((double)p1[2 * i] - cx) / fx;
        x2[2 * i + 1] = ((double)p1[2 * i + 1] - cy) / fy;
    }
    threshold /= (fx + fy) / 2;
    if (mask) memset(mask, 0, n);
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (n < modelPoints) return VO_ERR_FAILED;

    RNG rng(seed);
    std::vector<float> err;
    std::vector<uint8_t> cur_mask, best_mask(n, 0);
    double models[10 * 9], best[9];
    double ms1[10], ms2[10];
    int idx[5];

    if (n == modelPoints) {
        int nm = five_point(x1.data(), x2.data(), models);
        if (nm <= 0) return VO_ERR_FAILED;
        memcpy(E, models, 9 * sizeof(double));  // cv returns all models stacked; the first is kept here
        if (mask) memset(mask, 1, n);
        return VO_OK;
    }

    if (method == VO_RANSAC) {
        int niters = std::max(max_iters, 1), maxGood = 0, iter, nmodels_total = 0;
        for (iter = 0; iter < niters; iter++) {
            get_subset(rng, n, idx);
            for (int i = 0; i < 5; i++) {
                ms1[2 * i] = x1[2 * idx[i]]; ms1[2 * i + 1] = x1[2 * idx[i] + 1];
                ms2[2 * i] = x2[2 * idx[i]]; ms2[2 * i + 1] = x2[2 * idx[i] + 1];
            }
            int nm = five_point(ms1, ms2, models);
            if (nm <= 0) continue;
            nmodels_total += nm;
            for (int i = 0; i < nm; i++) {
                int good = find_inliers(models + 9 * i, x1.data(), x2.data(), n, threshold, err, cur_mask);
                if (good > std::max(maxGood, modelPoints - 1)) {
                    std::swap(cur_mask, best_mask);
                    memcpy(best, models + 9 * i, sizeof(best));
                    maxGood = good;
                    niters = ransac_update_num_iters(prob, (double)(n - good) / n, modelPoints, niters);
                }
            }
        }
        if (stats) {
            stats[0] = iter; stats[1] = nmodels_total; stats[2] = maxGood; stats[3] = niters;
        }
        if (maxGood <= 0) return VO_ERR_FAILED;
        memcpy(E, best, sizeof(best));
        if (mask) memcpy(mask, best_mask.data(), n);
        return VO_OK;
    } else if (method == VO_LMEDS) {
        const double outlierRatio = 0.45;
        int niters = ransac_update_num_iters(prob, outlierRatio, modelPoints, max_iters);
        niters = std::max(niters, 3);
        double minMedian = DBL_MAX;
        int nmodels_total = 0;
        std::vector<float> errs(n);
        for (int iter = 0; iter < niters; iter++) {
            get_subset(rng, n, idx);
            for (int i = 0; i < 5; i++) {
                ms1[2 * i] = x1[2 * idx[i]]; ms1[2 * i + 1] = x1[2 * idx[i] + 1];
                ms2[2 * i] = x2[2 * idx[i]]; ms2[2 * i + 1] = x2[2 * idx[i] + 1];
            }
            int nm = five_point(ms1, ms2, models);
            if (nm <= 0) continue;
            nmodels_total += nm;
            for (int i = 0; i < nm; i++) {
                sampson_err(models + 9 * i, x1.data(), x2.data(), n, errs.data());
                // errors are >= 0 (or NaN): cv sorts the float bit patterns as int32
                int32_t* ip = reinterpret_cast<int32_t*>(errs.data());
                std::sort(ip, ip + n);
                double median = n % 2 != 0 ? (double)errs[n / 2]
                                           : (double)(errs[n / 2 - 1] + errs[n / 2]) * 0.5;
                if (median < minMedian) {
                    minMedian = median;
                    memcpy(best, models + 9 * i, sizeof(best));
                }
            }
        }
        if (stats) {
            stats[0] = niters; stats[1] = nmodels_total; stats[3] = niters;
        }
        if (!(minMedian < DBL_MAX)) return VO_ERR_FAILED;
        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - modelPoints)) * std::sqrt(minMedian);
        sigma = std::max(sigma, 0.001);
        int good = find_inliers(best, x1.data(), x2.data(), n, sigma, err, cur_mask);
        if (stats) stats[2] = good;
        memcpy(E, best, sizeof(best));
        if (mask) memcpy(mask, cur_mask.data(), n);
        return VO_OK;  // cv copies the best model out even when fewer than 5 points are inliers
    }
    return VO_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------
// decomposeEssentialMat / triangulatePoints / recoverPose
// ---------------------------------------------------------------------------------------
static double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}
static void mul3(const double* A, const double* B, double* Cm) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            Cm[3 * i + j] = s;
        }
}

static void decompose_essential(const double* E, double* R1, double* R2, double* t) {
    double D[3], U[9], Vt[9];
    svd_compute(E, 3, 3, D, U, Vt, false);
    if (det3(U) < 0)
        for (int i = 0; i < 9; i++) U[i] *= -1.;
    if (det3(Vt) < 0)
        for (int i = 0; i < 9; i++) Vt[i] *= -1.;
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9];
    mul3(U, Wm, T);
    mul3(T, Vt, R1);
    mul3(U, Wt, T);
    mul3(T, Vt, R2);
    t[0] = U[2];
    t[1] = U[5];
    t[2] = U[8];
}

// x0, x1: N x 2 (interleaved), pts4d: 4 x N row-major
static void triangulate_points(const double* P0, const double* P1, const double* x0, const double* x1,
                               int n, double* pts4d) {
    const double* P[2] = {P0, P1};
    const double* X[2] = {x0, x1};
    for (int i = 0; i < n; i++) {
        double A[16];
        for (int j = 0; j < 2; j++) {
            double x = X[j][2 * i], y = X[j][2 * i + 1];
            for (int k = 0; k < 4; k++) {
                A[(j * 2 + 0) * 4 + k] = x * P[j][8 + k] - P[j][k];
                A[(j * 2 + 1) * 4 + k] = y * P[j][8 + k] - P[j][4 + k];
            }
        }
        double w[4], Vt[16];
        svd_compute(A, 4, 4, w, nullptr, Vt, false);
        for (int k = 0; k < 4; k++) pts4d[(size_t)k * n + i] = Vt[12 + k];
    }
}

static int recover_pose(const double* E, const float* p0, const float* p1, int n, const double* K,
                        double distanceThresh, uint8_t* mask_inout, double* R, double* t,
                        double* pts4d, int* n_good) {
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> x1(2 * (size_t)n), x2(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        x1[2 * i] = ((double)p0[2 * i] - cx) / fx;
        x1[2 * i + 1] = ((double)p0[2 * i + 1] - cy) / fy;
        x2[2 * i] = ((double)p1[2 * i] - cx) / fx;
        x2[2 * i + 1] = ((double)p1[2 * i + 1] - cy) / fy;
    }
    double R1[9], R2[9], tt[3];
    decompose_essential(E, R1, R2, tt);
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double Pc[4][12];
    const double* Rs[4] = {R1, R2, R1, R2};
    const double sg[4] = {1, 1, -1, -1};
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 3; r++) {
            for (int k = 0; k < 3; k++) Pc[c][4 * r + k] = Rs[c][3 * r + k];
            Pc[c][4 * r + 3] = sg[c] * tt[r];
        }
    std::vector<double> Q[4];
    std::vector<uint8_t> masks[4];
    int good[4];
    for (int c = 0; c < 4; c++) {
        Q[c].resize(4 * (size_t)n);
        masks[c].resize(n);
        triangulate_points(P0, Pc[c], x1.data(), x2.data(), n, Q[c].data());
        good[c] = 0;
        for (int i = 0; i < n; i++) {
            double X = Q[c][i], Y = Q[c][(size_t)n + i], Z = Q[c][2 * (size_t)n + i],
                   Wh = Q[c][3 * (size_t)n + i];
            bool m = Z * Wh > 0;
            double xn = X / Wh, yn = Y / Wh, zn = Z / Wh, wn = Wh / Wh;
            m = (zn < distanceThresh) && m;
            double z2 = 0;
            {
                const double q[4] = {xn, yn, zn, wn};
                for (int k = 0; k < 4; k++) z2 += Pc[c][8 + k] * q[k];
            }
            m = (z2 > 0) && m;
            m = (z2 < distanceThresh) && m;
            uint8_t mv = m ? 255 : 0;
            if (mask_inout) mv = mv & mask_inout[i];
            masks[c][i] = mv;
            good[c] += mv != 0;
        }
    }
    int win;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) win = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) win = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) win = 2;
    else win = 3;
    memcpy(R, Rs[win], 9 * sizeof(double));
    for (int r = 0; r < 3; r++) t[r] = sg[win] * tt[r];
    if (mask_inout) memcpy(mask_inout, masks[win].data(), n);
    if (pts4d) memcpy(pts4d, Q[win].data(), sizeof(double) * 4 * (size_t)n);
    if (n_good) *n_good = good[win];
    return VO_OK;
}

}  // namespace vo_oracle

using namespace vo_oracle;

extern "C" {

void vo_oracle_rng_next(uint64_t seed, int n, uint32_t* out) {
    RNG rng(seed);
    for (int i = 0; i < n; i++) out[i] = rng.next();
}

int vo_oracle_sample_subsets(uint64_t seed, int npoints, int niters, int32_t* idx) {
    if (npoints < 5 || niters < 0 || !idx) return VO_ERR_INVALID;
    RNG rng(seed);
    for (int it = 0; it < niters; it++) {
        int s[5];
        get_subset(rng, npoints, s);
        for (int k = 0; k < 5; k++) idx[it * 5 + k] = s[k];
    }
    return VO_OK;
}

int vo_oracle_five_point(const double* x1, const double* x2, double* E_out) {
    return five_point(x1, x2, E_out);
}

void vo_oracle_sampson_err(const double* E, const double* x1, const double* x2, int n, float* err) {
    sampson_err(E, x1, x2, n, err);
}

int vo_oracle_find_essential(const float* p0, const float* p1, int n, const double* K, int method,
                             double prob, double threshold, uint64_t seed, int max_iters,
                             double* E, uint8_t* mask, int* stats) {
    if (n < 0 || !K || !E || (n > 0 && (!p0 || !p1))) return VO_ERR_INVALID;
    if (!(prob > 0 && prob < 1)) return VO_ERR_INVALID;
    return find_essential(p0, p1, n, K, method, prob, threshold, seed, max_iters, E, mask, stats);
}

void vo_oracle_decompose_essential(const double* E, double* R1, double* R2, double* t) {
    decompose_essential(E, R1, R2, t);
}

int vo_oracle_triangulate_points(const double* P0, const double* P1, const double* x0,
                                 const double* x1, int n, double* pts4d) {
    if (n < 0 || !P0 || !P1) return VO_ERR_INVALID;
    triangulate_points(P0, P1, x0, x1, n, pts4d);
    return VO_OK;
}

int vo_oracle_recover_pose(const double* E, const float* p0, const float* p1, int n,
                           const double* K, double dist_thresh, uint8_t* mask_inout, double* R,
                           double* t, double* pts4d, int* n_good) {
    if (n < 0 || !E || !K || !R || !t) return VO_ERR_INVALID;
    return recover_pose(E, p0, p1, n, K, dist_thresh, mask_inout, R, t, pts4d, n_good);
}

int vo_oracle_triangulate(const float* p0, const float* p1, int n, const double* P0,
                          const double* P1, double* pts3d) {
    if (n < 0 || !P0 || !P1) return VO_ERR_INVALID;
    if (n == 0) return VO_OK;
    std::vector<double> x0(2 * (size_t)n), x1(2 * (size_t)n), q(4 * (size_t)n);
    for (int i = 0; i < 2 * n; i++) {
        x0[i] = p0[i];
        x1[i] = p1[i];
    }
    triangulate_points(P0, P1, x0.data(), x1.data(), n, q.data());
    for (int i = 0; i < n; i++) {
        double w = q[3 * (size_t)n + i];
        pts3d[3 * i + 0] = q[i] / w;
        pts3d[3 * i + 1] = q[(size_t)n + i] / w;
        pts3d[3 * i + 2] = q[2 * (size_t)n + i] / w;
    }
    return VO_OK;
}

void vo_oracle_projection_matrix(const double* K, const double* pose, double* P) {
    // P = K [R^T | -R^T t], pose = [R | t] (3x4, camera-to-world)
    double M[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) M[4 * i + j] = pose[4 * j + i];
        double s = 0;
        for (int k = 0; k < 3; k++) s += pose[4 * k + i] * pose[4 * k + 3];
        M[4 * i + 3] = -s;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[3 * i + k] * M[4 * k + j];
            P[4 * i + j] = s;
        }
}

}  // extern "C"
