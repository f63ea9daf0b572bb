// five_point.h — Nister's five-point relative-pose solver, one hypothesis per GPU lane (fp64).
//
// Same mathematics as cv's EMEstimatorCallback::runKernel (calib3d/src/five-point.cpp; SURVEY.md
// A.5): null space of the 5x9 epipolar system, the ten cubic constraints det(E) = 0 and
// 2 E E^T E - tr(E E^T) E = 0 in Nister's monomial order, elimination of the ten leading
// monomials, the 3x3 polynomial matrix B(z), det B(z) = 0 (degree 10), back-substitution.
// The numerical methods are chosen for a lane, not copied from the CPU path:
//   * null space by Householder QR of the 9x5 transpose (static loops, registers),
//   * constraint expansion with compile-time monomial tables (registers),
//   * Gauss-Jordan with partial pivoting on the 10x20 matrix held in a per-lane LDS column
//     (dynamic row indices are free in LDS, impossible in registers),
//   * real roots only, by bracketing between the roots of successive derivatives and
//     safeguarded Newton (no complex arithmetic, no iteration over the complex roots cv's
//     Durand-Kerner also finds and then discards).
// `S` is the per-lane scratch accessor: S(i) is element i of FP_SCRATCH doubles.
// The header compiles for the host too (tests/host_five_point.cpp checks it against the oracle
// without a GPU).
#ifndef VO_FIVE_POINT_H
#define VO_FIVE_POINT_H

#include "small_linalg.h"

#define FP_SCRATCH 236  // 10x20 constraint matrix + 4x9 null-space basis

// reciprocal for Newton steps (the step only has to be roughly right; the root is fixed by f)
#if defined(__HIP_DEVICE_COMPILE__)
#define VO_RCP(x) __builtin_amdgcn_rcp(x)
#else
#define VO_RCP(x) (1.0 / (x))
#endif

// phase timing hook for tools/micro/solve_time.hip; compiles to nothing in the library
#ifndef FP_TICK
#define FP_TICK(i)
#define FP_COUNT(i)
#endif
// early-exit hook of the same tool (cumulative phase timing without code motion across the marks)
#ifndef FP_STOP
#define FP_STOP(i, v)
#endif

namespace fp_detail {

VO_HD void mul_ll_acc(const double* a, const double* b, double* q, double sgn) {
    constexpr int LL[4][4] = {{0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) q[LL[i][j]] += sgn * (a[i] * b[j]);
}

VO_HD void mul_ql_acc(const double* q, const double* l, double* c, double sgn) {
    constexpr int QL[10][4] = {{0, 2, 4, 5},     {3, 1, 6, 7},    {10, 13, 16, 17}, {2, 3, 8, 9},
                               {4, 8, 10, 11},   {8, 6, 13, 14},  {5, 9, 11, 12},   {9, 7, 14, 15},
                               {11, 14, 17, 18}, {12, 15, 18, 19}};
#pragma unroll
    for (int i = 0; i < 10; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) c[QL[i][j]] += sgn * (q[i] * l[j]);
}

constexpr double falling(int n, int k) {  // n! / (n-k)!
    double r = 1;
    for (int i = 0; i < k; i++) r *= (double)(n - i);
    return r;
}

template <int D>
VO_HD void horner(const double (&a)[D + 1], double x, double& f, double& df) {
    f = a[D];
    df = 0;
#pragma unroll
    for (int i = D - 1; i >= 0; i--) {
        df = fma(df, x, f);
        f = fma(f, x, a[i]);
    }
}

// root of a (degree D) inside [lo, hi] given a sign change; safeguarded Newton / bisection
// `tol`: relative step size at which the iteration stops.  Only the roots of c itself need full
// precision; the roots of its derivatives merely bracket them.
template <int D>
VO_HD double bracket_root(const double (&a)[D + 1], double lo, double hi, double flo, double tol) {
    double xl, xh;
    if (flo < 0) {
        xl = lo;
        xh = hi;
    } else {
        xl = hi;
        xh = lo;
    }
    FP_COUNT(2);
    double rts = 0.5 * (lo + hi);
    double dxold = fabs(hi - lo), dx = dxold, f, df;
    horner<D>(a, rts, f, df);
    for (int it = 0; it < 200; it++) {
        if ((((rts - xh) * df - f) * ((rts - xl) * df - f) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df))) {
            dxold = dx;
            dx = 0.5 * (xh - xl);
            rts = xl + dx;
            if (xl == rts) break;
        } else {
            dxold = dx;
            dx = f * VO_RCP(df);
            const double temp = rts;
            rts -= dx;
            if (temp == rts) break;
        }
        if (fabs(dx) <= tol * fabs(rts)) break;
        FP_COUNT(0);
        horner<D>(a, rts, f, df);
        if (f < 0) xl = rts;
        else xh = rts;
    }
    return rts;
}

// Outward search from x0 (value f0) in direction dir (-1 / +1) for a point where the sign of the
// degree-D polynomial differs from that of f0: the step doubles, so the bracket it returns is about
// as wide as its distance from x0 and Newton converges in a few steps.  (Starting Newton at the
// far end of [-bound, bound] instead costs ~D ln(bound) steps per outer root: the iterate only
// shrinks by (1 - 1/D) per step while it is far from all roots.)  Returns false if |x| passes
// `bound` (a root bound of c and hence of all its derivatives) without a sign change.
template <int D>
VO_HD bool outward_bracket(const double (&a)[D + 1], double x0, double f0, double dir, double bound,
                           double& xin, double& xout, double& fout) {
    double step = 0.25 * fmax(fabs(x0), 0.01);
    xin = x0;
    for (;;) {
        xout = xin + dir * step;
        double d;
        FP_COUNT(1);
        horner<D>(a, xout, fout, d);
        if ((fout < 0) != (f0 < 0)) return true;
        if (!(fabs(xout) <= bound)) return false;
        xin = xout;
        step *= 2;
    }
}

// Real roots of the (10-D)-th derivative of c (degree D), bracketed by the roots of the next
// derivative, which are S(prev..prev+nprev) in ascending order (with none, 0 splits the line).
// The outer intervals have no far end: the sign at infinity is that of the leading coefficient.
// Writes S(cur..), returns count.
template <int D, class S>
VO_HD int level_roots(const double (&c)[11], double bound, int nprev, const S& s, int prev, int cur) {
    double a[D + 1];
#pragma unroll
    for (int i = 0; i <= D; i++) a[i] = c[i + (10 - D)] * falling(i + (10 - D), 10 - D);
    const double tol = D == 10 ? 4e-16 : 1e-9;
    const bool neg_at_minf = (D & 1) ? (a[D] > 0) : (a[D] < 0), neg_at_pinf = a[D] < 0;
    int ncur = 0;
    double lo = nprev > 0 ? s(prev) : 0.0, flo, d;
    horner<D>(a, lo, flo, d);
    // interval 0: (-inf, p0]; k = 1 .. npts-1: [p(k-1), p(k)]; npts: [p(npts-1), +inf).  One loop with
    // one call site per helper: the kernel's code is larger than the instruction cache and mostly
    // straight-line, so every inlined copy is paid for in instruction fetch misses.
    const int npts = nprev > 0 ? nprev : 1;
    for (int k = 0; k <= npts; k++) {
        double blo = 0, bhi = 0, bf = 0;
        bool have;
        if (k == 0 || k == npts) {
            const bool left = k == 0;
            have = (flo < 0) != (left ? neg_at_minf : neg_at_pinf);
            if (have) {
                double xin, xout, fout;
                have = outward_bracket<D>(a, lo, flo, left ? -1.0 : 1.0, bound, xin, xout, fout);
                blo = left ? xout : xin;
                bhi = left ? xin : xout;
                bf = left ? fout : flo;
            }
        } else {
            const double hi = s(prev + k);
            double fhi;
            horner<D>(a, hi, fhi, d);
            have = (flo < 0) != (fhi < 0);
            blo = lo;
            bhi = hi;
            bf = flo;
            lo = hi;
            flo = fhi;
        }
        if (have) s(cur + ncur++) = bracket_root<D>(a, blo, bhi, bf, tol);
    }
    return ncur;
}

// ---- real roots by a Sturm chain ------------------------------------------------------------
// P0 = c, P1 = c', P(k+1) = -rem(P(k-1), P(k)), every member scaled to max |coefficient| = 1 (positive
// factors do not change sign counts).  With quotients q_k(x) = a_k x + b_k and g_k = the scale of the
// remainder, P(k-1)(x) = q_k(x) P(k)(x) - g_k P(k+1)(x): the whole chain is evaluated at a point from
// its two last members upwards in 3 operations per member, and V(x) = number of sign changes along
// it.  The number of real roots in (a, b] is V(a) - V(b).  Roots are isolated in ascending order by
// bisection on t in (-1, 1), x = t / (1 - |t|) (the whole real line, resolution where the roots
// are), until an interval holds exactly one; that root is then polished on c itself by the
// safeguarded Newton iteration above.  Every lane does the same few steps per root (little
// divergence) and a step is ~45 instructions, against ~270 Horner evaluations with data-dependent
// branching for the derivative chain (145 of the solver's 240 us on MI355X), which stays as the
// fallback for chains whose degrees do not drop one by one (a vanishing leading coefficient).
struct SturmChain {
    double a[9], b[9], g[9];  // step k = index k - 1: quotient and remainder scale of P(k-1) / P(k)
    double p9[2], p10;        // the last two members, highest coefficient first
    unsigned neg_pinf, neg_minf;  // bit k: P(k) is negative at +inf / -inf
};

VO_HD bool sturm_build(const double (&c)[11], SturmChain& ch) {
    double u[11], v[11];  // P(k-1), P(k), highest coefficient first
#pragma unroll
    for (int i = 0; i <= 10; i++) u[i] = c[10 - i];
    {
        double m = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            v[i] = u[i] * (double)(10 - i);
            m = fmax(m, fabs(v[i]));
        }
        if (!(m > 0)) return false;
        const double sc = 1.0 / m;
#pragma unroll
        for (int i = 0; i < 10; i++) v[i] *= sc;
        v[10] = 0;
    }
    ch.neg_pinf = (u[0] < 0 ? 1u : 0u) | (v[0] < 0 ? 2u : 0u);
    ch.neg_minf = (u[0] < 0 ? 1u : 0u) | (v[0] < 0 ? 0u : 2u);  // degree 10 even, degree 9 odd
#pragma unroll
    for (int k = 1; k <= 9; k++) {
        const int d = 10 - k;  // degree of v = P(k); u = P(k-1) has degree d + 1
        if (!(fabs(v[0]) > 1e-11)) return false;
        const double inv = 1.0 / v[0];
        const double qa = u[0] * inv;
        const double qb = (u[1] - qa * v[1]) * inv;
        double r[11], m = 0;
#pragma unroll
        for (int j = 0; j < d; j++) {  // remainder, degree d - 1 - j
            const double vj2 = j + 2 <= d ? v[j + 2] : 0.0;
            r[j] = u[j + 2] - qa * vj2 - qb * v[j + 1];
            m = fmax(m, fabs(r[j]));
        }
        if (!(m > 1e-13)) return false;  // c and c' share a factor: a multiple root
        const double sc = 1.0 / m;
        ch.a[k - 1] = qa;
        ch.b[k - 1] = qb;
        ch.g[k - 1] = m;
#pragma unroll
        for (int j = 0; j <= d; j++) u[j] = v[j];
#pragma unroll
        for (int j = 0; j < d; j++) v[j] = -r[j] * sc;
        v[d] = 0;
        const bool neg = v[0] < 0;
        ch.neg_pinf |= neg ? (1u << (k + 1)) : 0u;
        ch.neg_minf |= (neg != (((d - 1) & 1) != 0)) ? (1u << (k + 1)) : 0u;
    }
    ch.p9[0] = u[0];
    ch.p9[1] = u[1];
    ch.p10 = v[0];
    if (!(fabs(ch.p10) > 0)) return false;
    return true;
}

VO_HD int sign_changes(unsigned bits) {  // among the 11 chain members
    return __builtin_popcount((bits ^ (bits >> 1)) & 0x3ffu);
}

VO_HD int sturm_count(const SturmChain& ch, double x) {
    double e1 = ch.p10, e0 = fma(ch.p9[0], x, ch.p9[1]);  // P10(x), P9(x)
    unsigned bits = (e1 < 0 ? (1u << 10) : 0u) | (e0 < 0 ? (1u << 9) : 0u);
#pragma unroll
    for (int k = 9; k >= 1; k--) {
        const double e = fma(fma(ch.a[k - 1], x, ch.b[k - 1]), e0, -(ch.g[k - 1] * e1));  // P(k-1)(x)
        e1 = e0;
        e0 = e;
        bits |= e < 0 ? (1u << (k - 1)) : 0u;
    }
    return sign_changes(bits);
}

// Real roots of c in ascending order into S(out..); -1 if the chain is degenerate.
template <class S>
VO_HD int sturm_roots(const double (&c)[11], const S& s, int out) {
    SturmChain ch;
    if (!sturm_build(c, ch)) return -1;
    const int v_minf = sign_changes(ch.neg_minf), v_pinf = sign_changes(ch.neg_pinf);
    const int total = v_minf - v_pinf;
    if (total < 0 || total > 10) return -1;
    int found = 0, n = 0;
    double ta = -1.0;  // roots up to here are done; V(ta) = va
    int va = v_minf;
    while (found < total) {
        double a = ta, b = 1.0;
        int ca = va, cb = v_pinf;
        int step = 0;
        // the leftmost root of (a, b]: halve towards it until it is alone and the ends are finite
        while ((ca - cb != 1 || a <= -1.0 || b >= 1.0) && step < 80) {
            const double m = 0.5 * (a + b);
            FP_COUNT(1);
            const int cm = sturm_count(ch, m / (1.0 - fabs(m)));
            if (ca - cm >= 1) {
                b = m;
                cb = cm;
            } else {
                a = m;
                ca = cm;
            }
            step++;
        }
        const int here = ca - cb;
        if (here < 1) return -1;  // the counts are not monotone: numerical trouble, use the other finder
        if (here == 1 && a > -1.0 && b < 1.0) {
            const double xa = a / (1.0 - fabs(a)), xb = b / (1.0 - fabs(b));
            double fa, fb, d;
            horner<10>(c, xa, fa, d);
            horner<10>(c, xb, fb, d);
            if ((fa < 0) != (fb < 0)) s(out + n++) = bracket_root<10>(c, xa, xb, fa, 4e-16);
        }
        // (a cluster that 80 halvings cannot split is a multiple root without a sign change: skipped,
        // as the derivative chain skips it)
        found += here;
        ta = b;
        va = cb;
    }
    return n;
}

}  // namespace fp_detail

// x1, x2: five K-normalised correspondences (x, y interleaved), x2^T E x1 = 0.
// E_out: up to 10 models, 9 doubles each (row-major, unit Frobenius norm), ordered by the root z.
template <class S>
VO_HD int five_point_solve(const double* x1, const double* x2, const S& s, double* E_out) {
    using namespace fp_detail;
    // ---- null space of Q (5x9) by Householder QR of A = Q^T (9x5) ------------------------------
    {
        double A[9][5], beta[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double u1 = x1[2 * j], v1 = x1[2 * j + 1], u2 = x2[2 * j], v2 = x2[2 * j + 1];
            A[0][j] = u2 * u1; A[1][j] = u2 * v1; A[2][j] = u2;
            A[3][j] = v2 * u1; A[4][j] = v2 * v1; A[5][j] = v2;
            A[6][j] = u1;      A[7][j] = v1;      A[8][j] = 1.0;
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            double nrm2 = 0;
#pragma unroll
            for (int i = k; i < 9; i++) nrm2 += A[i][k] * A[i][k];
            const double alpha = A[k][k] >= 0 ? -sqrt(nrm2) : sqrt(nrm2);
            A[k][k] -= alpha;
            double vn2 = 0;
#pragma unroll
            for (int i = k; i < 9; i++) vn2 += A[i][k] * A[i][k];
            beta[k] = vn2 > 0 ? 2.0 / vn2 : 0.0;
#pragma unroll
            for (int j = k + 1; j < 5; j++) {
                double d = 0;
#pragma unroll
                for (int i = k; i < 9; i++) d += A[i][k] * A[i][j];
                d *= beta[k];
#pragma unroll
                for (int i = k; i < 9; i++) A[i][j] -= d * A[i][k];
            }
        }
        // orthonormal basis of the row space of Q: the first five columns of the Householder Q
        double basis[9][9];
#pragma unroll
        for (int j = 0; j < 5; j++) {
#pragma unroll
            for (int i = 0; i < 9; i++) basis[j][i] = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 4; k >= 0; k--) {
                double d = 0;
#pragma unroll
                for (int i = k; i < 9; i++) d += A[i][k] * basis[j][i];
                d *= beta[k];
#pragma unroll
                for (int i = k; i < 9; i++) basis[j][i] -= d * A[i][k];
            }
        }
        // The null-space basis fixes the (x, y, z) parametrisation and with it the conditioning of
        // everything downstream, so it is completed the way cv::SVD(FULL_UV) completes it: fixed
        // pseudo-random +-1/9 vectors (cv::RNG(0x12345678), bit 8 of successive outputs),
        // orthogonalised against all earlier vectors in two passes.  The result depends on the row
        // space only, not on which orthonormal basis of it is used.
        constexpr int SGN[36] = {0, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1,
                                 1, 0, 0, 0, 1, 1, 0, 1, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0};
#pragma unroll
        for (int i = 5; i < 9; i++) {
            const double val0 = 1. / 9;
#pragma unroll
            for (int k = 0; k < 9; k++) basis[i][k] = SGN[9 * (i - 5) + k] ? val0 : -val0;
            double sd = 0;
#pragma unroll
            for (int iter = 0; iter < 2; iter++) {
#pragma unroll
                for (int j = 0; j < i; j++) {
                    sd = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) sd += basis[i][k] * basis[j][k];
                    double asum = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        const double t = basis[i][k] - sd * basis[j][k];
                        basis[i][k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > DBL_EPSILON * 10 * 100 ? 1 / asum : 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) basis[i][k] *= asum;
                }
                sd = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) sd += basis[i][k] * basis[i][k];
                sd = sqrt(sd);
            }
            const double sc = sd > DBL_MIN ? 1 / sd : 0.;
#pragma unroll
            for (int k = 0; k < 9; k++) basis[i][k] *= sc;
#pragma unroll
            for (int k = 0; k < 9; k++) s(200 + 9 * (i - 5) + k) = basis[i][k];
        }
    }
    FP_TICK(1);
    FP_STOP(1, s(200) + s(235));
    // ---- ten cubic constraints -> 10x20 matrix in S (row r at 20 r) -----------------------------
    {
        double e[9][4];  // entry (r,c) of E(x,y,z) = x N0 + y N1 + z N2 + N3
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) e[i][k] = s(200 + 9 * k + i);
        {  // det(E)
            double cub[20];
#pragma unroll
            for (int i = 0; i < 20; i++) cub[i] = 0;
            double m[10];
#pragma unroll
            for (int i = 0; i < 10; i++) m[i] = 0;
            mul_ll_acc(e[4], e[8], m, 1.0);
            mul_ll_acc(e[5], e[7], m, -1.0);
            mul_ql_acc(m, e[0], cub, 1.0);
#pragma unroll
            for (int i = 0; i < 10; i++) m[i] = 0;
            mul_ll_acc(e[3], e[8], m, 1.0);
            mul_ll_acc(e[5], e[6], m, -1.0);
            mul_ql_acc(m, e[1], cub, -1.0);
#pragma unroll
            for (int i = 0; i < 10; i++) m[i] = 0;
            mul_ll_acc(e[3], e[7], m, 1.0);
            mul_ll_acc(e[4], e[6], m, -1.0);
            mul_ql_acc(m, e[2], cub, 1.0);
#pragma unroll
            for (int i = 0; i < 20; i++) s(i) = cub[i];
        }
        // L = E E^T - 1/2 tr(E E^T) I (symmetric: 00 01 02 11 12 22)
        double L[6][10];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int i = 0; i < 10; i++) L[a][i] = 0;
        constexpr int SI[6] = {0, 0, 0, 1, 1, 2}, SJ[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) mul_ll_acc(e[3 * SI[a] + k], e[3 * SJ[a] + k], L[a], 1.0);
        {
            double tr[10];
#pragma unroll
            for (int i = 0; i < 10; i++) tr[i] = 0.5 * (L[0][i] + L[3][i] + L[5][i]);
#pragma unroll
            for (int i = 0; i < 10; i++) {
                L[0][i] -= tr[i];
                L[3][i] -= tr[i];
                L[5][i] -= tr[i];
            }
        }
        constexpr int SYM[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double cub[20];
#pragma unroll
                for (int q = 0; q < 20; q++) cub[q] = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) mul_ql_acc(L[SYM[i][k]], e[3 * k + j], cub, 1.0);
#pragma unroll
                for (int q = 0; q < 20; q++) s(20 * (1 + 3 * i + j) + q) = cub[q];
            }
    }
    FP_TICK(2);
    FP_STOP(2, s(0) + s(199));
    // ---- Gauss-Jordan on the first ten columns, partial pivoting ---------------------------------
    // Only rows 4..9 of the reduced right-hand block are used afterwards, so finished rows 0..3
    // are not updated any further.
    // Row loops have compile-time trip counts (rows that take no part are skipped by a uniform
    // branch) so that the LDS reads of a row are issued back to back instead of one per FMA.
    for (int col = 0; col < 10; col++) {
        double cv[10];
#pragma unroll
        for (int r = 0; r < 10; r++) cv[r] = s(20 * r + col);
        int piv = col;
        double best = -1.0;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const double v = fabs(cv[r]);
            if (r >= col && v > best) {
                best = v;
                piv = r;
            }
        }
        if (!(best > 1e-300)) return 0;
        double prow[20];
#pragma unroll
        for (int k = 0; k < 20; k++) prow[k] = s(20 * piv + k);
        if (piv != col) {
            double crow[20];
#pragma unroll
            for (int k = 0; k < 20; k++) crow[k] = s(20 * col + k);
#pragma unroll
            for (int k = 0; k < 20; k++) s(20 * piv + k) = crow[k];
            // the old row `col` now sits at `piv`: its entry in this column moves with it
#pragma unroll
            for (int r = 0; r < 10; r++)
                if (r == piv) cv[r] = crow[col];
        }
        const double inv = 1.0 / prow[col];
#pragma unroll
        for (int k = 0; k < 20; k++) {
            prow[k] *= inv;
            s(20 * col + k) = prow[k];
        }
#pragma unroll
        for (int r = 1; r < 10; r++) {
            if (r > col || (r >= 4 && r < col)) {
                const double f = cv[r];
                double row[20];
#pragma unroll
                for (int k = 1; k < 20; k++)
                    if (k >= 10 || k > col) row[k] = s(20 * r + k);
#pragma unroll
                for (int k = 1; k < 20; k++)
                    if (k >= 10 || k > col) s(20 * r + k) = row[k] - f * prow[k];
            }
        }
    }
    FP_TICK(3);
    FP_STOP(3, s(90) + s(199));
    // ---- B(z): rows e - z f, g - z h, i - z j; coefficients lowest degree first ------------------
    double c[11];
    {
        double B[3][3][5];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double r1[10], r2[10];
#pragma unroll
            for (int k = 0; k < 10; k++) {
                r1[k] = s(20 * (4 + 2 * i) + 10 + k);
                r2[k] = s(20 * (5 + 2 * i) + 10 + k);
            }
#pragma unroll
            for (int blk = 0; blk < 2; blk++) {
                B[i][blk][3] = -r2[3 * blk + 0];
                B[i][blk][2] = r1[3 * blk + 0] - r2[3 * blk + 1];
                B[i][blk][1] = r1[3 * blk + 1] - r2[3 * blk + 2];
                B[i][blk][0] = r1[3 * blk + 2];
                B[i][blk][4] = 0;
            }
            B[i][2][4] = -r2[6];
            B[i][2][3] = r1[6] - r2[7];
            B[i][2][2] = r1[7] - r2[8];
            B[i][2][1] = r1[8] - r2[9];
            B[i][2][0] = r1[9];
        }
        // park B for the back-substitution: S(20 + 15 i + 5 j + k)
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 5; k++) s(20 + 15 * i + 5 * j + k) = B[i][j][k];
        // det B(z), degree 10
#pragma unroll
        for (int k = 0; k <= 10; k++) c[k] = 0;
        // cofactors along row 0: columns (1,2), (0,2), (0,1) of rows 1,2
        constexpr int CA[3] = {1, 0, 0}, CB[3] = {2, 2, 1};
        constexpr int DEG[3] = {3, 3, 4};
#pragma unroll
        for (int t = 0; t < 3; t++) {
            double m[9];
#pragma unroll
            for (int k = 0; k < 9; k++) m[k] = 0;
#pragma unroll
            for (int p = 0; p <= DEG[CA[t]]; p++)
#pragma unroll
                for (int q = 0; q <= DEG[CB[t]]; q++)
                    m[p + q] += B[1][CA[t]][p] * B[2][CB[t]][q] - B[1][CB[t]][q] * B[2][CA[t]][p];
            const double sg = (t == 1) ? -1.0 : 1.0;
#pragma unroll
            for (int p = 0; p <= DEG[t]; p++)
#pragma unroll
                for (int q = 0; q <= DEG[CA[t]] + DEG[CB[t]]; q++) c[p + q] += sg * (B[0][t][p] * m[q]);
        }
    }
    FP_TICK(4);
    FP_STOP(4, c[0] + c[10] + c[5]);
    // ---- real roots of c ---------------------------------------------------------------------
    int nroots;
    {
        double cmax = 0;
#pragma unroll
        for (int k = 0; k <= 10; k++) cmax = fmax(cmax, fabs(c[k]));
        if (!(cmax > 0) || !(fabs(c[10]) > 0)) return 0;
        const double sc = 1.0 / cmax;
        double lead = fabs(c[10] * sc), bmax = 0;
#pragma unroll
        for (int k = 0; k <= 10; k++) c[k] *= sc;
#pragma unroll
        for (int k = 0; k < 10; k++) bmax = fmax(bmax, fabs(c[k]));
        const double bound = 1.0 + bmax / lead;  // Cauchy bound
        if (!(bound < 1e300)) return 0;
#ifdef FP_NO_STURM  // test builds: the derivative chain alone, to compare the two finders
        nroots = -1;
#else
        nroots = sturm_roots(c, s, 10);  // roots at S(10..)
#endif
        if (nroots < 0) {
            // roots of successive derivatives; two ping-pong arrays S(0..9) / S(10..19)
            int n = level_roots<1>(c, bound, 0, s, 10, 0);
            n = level_roots<2>(c, bound, n, s, 0, 10);
            n = level_roots<3>(c, bound, n, s, 10, 0);
            n = level_roots<4>(c, bound, n, s, 0, 10);
            n = level_roots<5>(c, bound, n, s, 10, 0);
            n = level_roots<6>(c, bound, n, s, 0, 10);
            n = level_roots<7>(c, bound, n, s, 10, 0);
            n = level_roots<8>(c, bound, n, s, 0, 10);
            n = level_roots<9>(c, bound, n, s, 10, 0);
            nroots = level_roots<10>(c, bound, n, s, 0, 10);  // roots at S(10..)
        }
    }
    FP_TICK(5);
    FP_STOP(5, (double)nroots + s(10));
    // ---- back-substitution ---------------------------------------------------------------------
    int count = 0;
    for (int i = 0; i < nroots; i++) {
        const double z1 = s(10 + i);
        const double z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int o = 20 + 15 * j;
            bz[j][0] = s(o + 3) * z3 + s(o + 2) * z2 + s(o + 1) * z1 + s(o + 0);
            bz[j][1] = s(o + 8) * z3 + s(o + 7) * z2 + s(o + 6) * z1 + s(o + 5);
            bz[j][2] = s(o + 14) * z4 + s(o + 13) * z3 + s(o + 12) * z2 + s(o + 11) * z1 + s(o + 10);
        }
        // null vector of Bz: the largest of the three row cross products
        double best[3] = {0, 0, 0}, bn = -1;
        constexpr int PA[3] = {0, 0, 1}, PB[3] = {1, 2, 2};
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const double* a = bz[PA[t]];
            const double* b = bz[PB[t]];
            const double cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2],
                         cz = a[0] * b[1] - a[1] * b[0];
            const double n2 = cx * cx + cy * cy + cz * cz;
            if (n2 > bn) {
                bn = n2;
                best[0] = cx;
                best[1] = cy;
                best[2] = cz;
            }
        }
        if (!(bn > 0)) continue;
        const double invn = 1.0 / sqrt(bn);
        if (fabs(best[2] * invn) < 1e-10) continue;
        const double xs = best[0] / best[2], ys = best[1] / best[2];
        double Ev[9], nrm = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            Ev[k] = s(200 + k) * xs + s(209 + k) * ys + s(218 + k) * z1 + s(227 + k);
            nrm += Ev[k] * Ev[k];
        }
        nrm = sqrt(nrm);
        if (!(nrm > 0)) continue;
#pragma unroll
        for (int k = 0; k < 9; k++) E_out[9 * count + k] = Ev[k] / nrm;
        count++;
    }
    return count;
}

// cv's EMEstimatorCallback::computeError for one point (Sampson distance, rounded to float)
VO_HD float sampson_error(const double* E, double a1, double b1, double a2, double b2) {
    const double Ex0 = E[0] * a1 + E[1] * b1 + E[2] * 1.0;
    const double Ex1 = E[3] * a1 + E[4] * b1 + E[5] * 1.0;
    const double Ex2 = E[6] * a1 + E[7] * b1 + E[8] * 1.0;
    const double Et0 = E[0] * a2 + E[3] * b2 + E[6] * 1.0;
    const double Et1 = E[1] * a2 + E[4] * b2 + E[7] * 1.0;
    const double x2tEx1 = a2 * Ex0 + b2 * Ex1 + 1.0 * Ex2;
    const double a = Ex0 * Ex0;
    const double b = Ex1 * Ex1;
    const double c = Et0 * Et0;
    const double d = Et1 * Et1;
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

// sampson_error(...) <= thr without the division, for the RANSAC counts (the same decision, bit for bit).
// (float)q <= thr for a double q holds iff q lies below M, the midpoint of thr and the float after it (a tie at M
// goes to the even neighbour and is left to the exact expression).  With q = fl(num / den): num < M den (1 - 2^-48)
// puts the real quotient, and with it its rounding, below M; num > M den (1 + 2^-48) puts them above; the sliver in
// between, a product that underflows, a zero denominator and NaNs take the division.  sampson_bound(thr) is M.
VO_HD double sampson_bound(float thr) {
    if (!(thr < 3.0e38f)) return (double)thr;  // inf (LMedS with five points: everything is an inlier), NaN
    union {
        float f;
        unsigned u;
    } v;
    v.f = thr;  // thr >= 0: a squared distance
    v.u += 1;
    return ((double)thr + (double)v.f) * 0.5;
}
VO_HD bool sampson_inlier(const double* E, double a1, double b1, double a2, double b2, float thr, double M) {
    const double Ex0 = E[0] * a1 + E[1] * b1 + E[2] * 1.0;
    const double Ex1 = E[3] * a1 + E[4] * b1 + E[5] * 1.0;
    const double Ex2 = E[6] * a1 + E[7] * b1 + E[8] * 1.0;
    const double Et0 = E[0] * a2 + E[3] * b2 + E[6] * 1.0;
    const double Et1 = E[1] * a2 + E[4] * b2 + E[7] * 1.0;
    const double x2tEx1 = a2 * Ex0 + b2 * Ex1 + 1.0 * Ex2;
    const double a = Ex0 * Ex0;
    const double b = Ex1 * Ex1;
    const double c = Et0 * Et0;
    const double d = Et1 * Et1;
    const double num = x2tEx1 * x2tEx1, den = a + b + c + d;
    const double p = M * den;
    const double eps = 3.5527136788005009e-15;  // 2^-48
    if (p > 1e-290 && num < p * (1.0 - eps)) return true;
    if (p > 1e-290 && num > p * (1.0 + eps)) return false;
    return (float)(num / den) <= thr;
}

#endif
