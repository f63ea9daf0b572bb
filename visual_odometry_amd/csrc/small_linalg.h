// small_linalg.h — register-resident small dense linear algebra for the geometry kernels.
// jacobi_svd<N>: one-sided (Hestenes) Jacobi SVD of a square N x N matrix, the algorithm behind
// cv::SVD::compute for doubles (rotation order, convergence test and singular-value sort are the
// same, so results track the CPU path to rounding).  Every loop has a compile-time trip count so
// all arrays stay in VGPRs.
#ifndef VO_SMALL_LINALG_H
#define VO_SMALL_LINALG_H

#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define VO_HD __host__ __device__ __forceinline__
#else
#define VO_HD inline
#endif

// In:  At[i][k] = A[k][i] (rows of At are the columns of A).
// Out: W descending; At rows = columns of U (normalised); Vt rows = right singular vectors.
//
// mirror (optional, in / out): the decomposition of A with some columns NEGATED, for free.  Bit r of *mirror
// says that column r (row r of At) is negated in the mirrored problem.  Every step of the algorithm is odd or even
// in those signs — IEEE rounding is symmetric, so p = <a_i, a_j> of the mirrored problem is exactly +-p, beta and
// gamma are the same, and of (c, s) exactly one changes sign when p does: s on the branch beta >= 0 (the rotated
// rows keep their signs), c on the branch beta < 0 (the two rows EXCHANGE their signs) — hence the mirrored
// run performs the same rotations on the same magnitudes and differs only in a sign per row, which is all that
// needs tracking.  On return bit r says that row r of the mirrored Vt is the negative of D * Vt[r], D the diagonal
// sign matrix of the negated columns: Vt'[r][k] = (bit r ? -1 : 1) * (column k negated ? -1 : 1) * Vt[r][k].
// k_pose_points uses it for recoverPose's candidates [R | t] and [R | -t], whose DLT systems differ in the sign
// of the fourth column: two decompositions per point instead of four, the same bits.
template <int N>
VO_HD void jacobi_svd(double (&At)[N][N], double (&W)[N], double (&Vt)[N][N], unsigned* mirror = nullptr) {
    unsigned sg = mirror ? *mirror : 0u;
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) sd += At[i][k] * At[i][k];
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < N; k++) Vt[i][k] = (i == k) ? 1.0 : 0.0;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++)
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < N; k++) p += At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot(p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                    if (mirror && (((sg >> i) ^ (sg >> j)) & 1u)) sg ^= (1u << i) | (1u << j);  // the rows exchange signs
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double t0 = c * At[i][k] + s * At[j][k];
                    const double t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0;
                    At[j][k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double t0 = c * Vt[i][k] + s * Vt[j][k];
                    const double t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0;
                    Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) sd += At[i][k] * At[i][k];
        W[i] = sqrt(sd);
    }
    // selection sort, descending, first maximum wins (same permutation as cv's)
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (wj < W[k]) {
                j = k;
                wj = W[k];
            }
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (j == k) {
                const double tw = W[i];
                W[i] = W[k];
                W[k] = tw;
                if (mirror && (((sg >> i) ^ (sg >> k)) & 1u)) sg ^= (1u << i) | (1u << k);
#pragma unroll
                for (int q = 0; q < N; q++) {
                    double t = At[i][q];
                    At[i][q] = At[k][q];
                    At[k][q] = t;
                    t = Vt[i][q];
                    Vt[i][q] = Vt[k][q];
                    Vt[k][q] = t;
                }
            }
    }
    // left singular vectors: normalise the rows.  A row whose singular value is exactly zero (an
    // exactly rank-deficient E) is replaced the way cv does it: a +-1/N vector from cv::RNG(0x12345678)
    // (bit 8 of successive outputs), orthogonalised against the earlier rows in two passes.
    unsigned long long rng = 0x12345678ull;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / N;
#pragma unroll
            for (int k = 0; k < N; k++) {
                rng = (unsigned long long)(unsigned)rng * 4164903690ull + (rng >> 32);
                At[i][k] = ((unsigned)rng & 256u) != 0 ? val0 : -val0;
            }
#pragma unroll
            for (int iter = 0; iter < 2; iter++) {
#pragma unroll
                for (int j = 0; j < N; j++) {
                    if (j < i) {
                        sd = 0;
#pragma unroll
                        for (int k = 0; k < N; k++) sd += At[i][k] * At[j][k];
                        double asum = 0;
#pragma unroll
                        for (int k = 0; k < N; k++) {
                            const double t = At[i][k] - sd * At[j][k];
                            At[i][k] = t;
                            asum += fabs(t);
                        }
                        asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                        for (int k = 0; k < N; k++) At[i][k] *= asum;
                    }
                }
                sd = 0;
#pragma unroll
                for (int k = 0; k < N; k++) sd += At[i][k] * At[i][k];
                sd = sqrt(sd);
            }
        }
        const double s = sd > minval ? 1 / sd : 0.;
#pragma unroll
        for (int k = 0; k < N; k++) At[i][k] *= s;
    }
    if (mirror) *mirror = sg;
}

VO_HD double det3x3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

VO_HD void mul3x3(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}

// cv::decomposeEssentialMat
VO_HD void decompose_essential(const double* E, double* R1, double* R2, double* t) {
    double At[3][3], W[3], Vt[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) At[i][k] = E[3 * k + i];
    jacobi_svd<3>(At, W, Vt);
    double U[9], V[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            U[3 * i + j] = At[j][i];
            V[3 * i + j] = Vt[i][j];
        }
    if (det3x3(U) < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) U[i] *= -1.;
    }
    if (det3x3(V) < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) V[i] *= -1.;
    }
    const double Wm[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9];
    mul3x3(U, Wm, T);
    mul3x3(T, V, R1);
    mul3x3(U, Wt, T);
    mul3x3(T, V, R2);
    t[0] = U[2];
    t[1] = U[5];
    t[2] = U[8];
}

// One point of cv::triangulatePoints: null vector of the 4x4 DLT system, as row 3 of Vt.
// Xm (optional): the null vector of the system with the fourth column of P0 and P1 negated, i.e. of the cameras
// [R0 | -t0], [R1 | -t1], from the same decomposition (jacobi_svd's mirror).
VO_HD void triangulate_point(const double* P0, const double* P1, double x0, double y0, double x1,
                             double y1, double* X, double* Xm = nullptr) {
    double At[4][4], W[4], Vt[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        At[k][0] = x0 * P0[8 + k] - P0[k];
        At[k][1] = y0 * P0[8 + k] - P0[4 + k];
        At[k][2] = x1 * P1[8 + k] - P1[k];
        At[k][3] = y1 * P1[8 + k] - P1[4 + k];
    }
    unsigned mirror = 1u << 3;
    jacobi_svd<4>(At, W, Vt, Xm ? &mirror : nullptr);
#pragma unroll
    for (int k = 0; k < 4; k++) X[k] = Vt[3][k];
    if (Xm) {
        const bool neg = (mirror >> 3) & 1u;
#pragma unroll
        for (int k = 0; k < 4; k++) Xm[k] = (neg != (k == 3)) ? -Vt[3][k] : Vt[3][k];
    }
}

#endif
