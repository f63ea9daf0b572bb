// five_point_coop.h — the five-point solver of five_point.h with ONE HYPOTHESIS PER 16-LANE GROUP.
//
// One hypothesis per lane (five_point.h) is a ~25 k-operation dependent chain: 180 us for the first RANSAC
// round of a batch, whatever the batch size, and the one stage nothing hides when frames arrive one at a
// time.  Here the same mathematics is spread over a group of 16 lanes (four groups per wavefront), so the
// chain is a few thousand operations long:
//   null space      lane = component (9 of 16): Householder QR with group reductions for the dot products;
//                   the basis is completed by two passes of classical Gram-Schmidt (all projections of a
//                   pass from one batched reduction) instead of cv's one-by-one sweep: same subspace, and
//                   the same vectors to rounding, because both start from cv's fixed +-1/9 vectors
//   constraints     lane = one of the 6 entries of E E^T - tr/2 or one of the 3 minors, then lane = row of the
//                   10 x 20 matrix; operands through the group's LDS slab, monomial tables compile-time
//   elimination     lane = row, rows stay in registers and are never swapped: the pivot lane of a column
//                   (arg-max over the lanes not used yet) publishes its scaled row through LDS, every other
//                   lane eliminates; ten rounds of ~20 FMAs instead of ~2000 sequential ones
//   determinant     the three cofactor terms on three lanes, summed by a group reduction
//   roots           every lane builds the Sturm chain (identical, redundant: cheaper than passing it round),
//                   lane k isolates the k-th real root by bisection on the compactified line and polishes
//                   it on the polynomial; the per-lane finder's rule for inseparable clusters is kept
//   models          lane k back-substitutes its root; models leave in ascending root order
// Results agree with five_point.h to rounding (tests/test_gpu_geometry.py compares the two kernels on
// thousands of samples); the RANSAC / LMedS selection downstream is unchanged.
// Device only.  `slab`: this group's 288 doubles of LDS.
#ifndef VO_FIVE_POINT_COOP_H
#define VO_FIVE_POINT_COOP_H

#include "five_point.h"

#define FPC_G 16         // lanes per hypothesis
#define FPC_SLAB 288     // doubles of LDS per hypothesis
// slab layout (doubles)
#define FPC_E 0          // [9][4]  null-space basis: entry i of N_k at 4 i + k
#define FPC_L 36         // [9][10] L entries 0..5 and the three minors 6..8
#define FPC_PROW 126     // [20]    pivot row of the current elimination round
#define FPC_RHS 146      // [6][10] right-hand block of the reduced rows 4..9
#define FPC_B 206        // [3][3][5] B(z)
#define FPC_ROOTS 252    // [20]    scratch of the fallback root finder / [10] roots
#define FPC_C 272        // [11]    det B(z)

namespace fpc {

// Exchange inside a 16-lane DPP row.  STEP 0, 1: xor 1, xor 2 (quad permutes); STEP 2: mirror inside each half
// (lane i <-> 7 - i), STEP 3: mirror of the row (i <-> 15 - i).  Used in this order the four steps are a
// butterfly: after step s all lanes of a block of 2^(s+1) hold the same value, so the mirrors pair blocks just
// as xor 4 and xor 8 would.  VALU data movement only: no trip through the LDS crossbar.
template <int STEP>
__device__ __forceinline__ int xchg(int v) {
    constexpr int ctrl = STEP == 0 ? 0xB1 : STEP == 1 ? 0x4E : STEP == 2 ? 0x141 : 0x140;
    return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true);
}
template <int STEP>
__device__ __forceinline__ double xchg(double v) {
    return __hiloint2double(xchg<STEP>(__double2hiint(v)), xchg<STEP>(__double2loint(v)));
}
__device__ __forceinline__ double gsum(double v) {  // every lane of the group gets the same bits (a + b == b + a)
    v += xchg<0>(v);
    v += xchg<1>(v);
    v += xchg<2>(v);
    v += xchg<3>(v);
    return v;
}
__device__ __forceinline__ int gor(int v) {
    v |= xchg<0>(v);
    v |= xchg<1>(v);
    v |= xchg<2>(v);
    v |= xchg<3>(v);
    return v;
}
template <int STEP>
__device__ __forceinline__ void argmax_step(double& v, int& who) {
    const double ov = xchg<STEP>(v);
    const int ow = xchg<STEP>(who);
    if (ov > v || (ov == v && ow < who)) {
        v = ov;
        who = ow;
    }
}
__device__ __forceinline__ void lds_sync() {
    // the group is inside one wavefront and a wavefront's LDS operations complete in order: ordering the
    // compiler is all that is needed between a lane's store and another lane's load
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct LdsScratch {  // S(i) accessor over a piece of the slab, for fp_detail's sequential root finders
    double* base;
    __device__ __forceinline__ double& operator()(int i) const { return base[i]; }
};

}  // namespace fpc

// x1, x2: the five K-normalised correspondences (x, y interleaved), the same in every lane of the group.
// gl: lane index inside the group (0..15).  Writes up to 10 models (9 doubles each, ascending root) to E_out
// and returns their number (the same value in every lane of the group).
__device__ __forceinline__ int five_point_solve_coop(const double* x1, const double* x2, double* slab, int gl,
                                                     double* E_out) {
    using namespace fp_detail;
    using namespace fpc;
    const bool comp = gl < 9;  // lanes that hold a component of the 9-vectors

    // ---- null space: Householder QR of A = Q^T (9 x 5), lane = row -------------------------------
    double basis[9];  // component gl of basis vector 0..8 (0..4 row space, 5..8 null space)
    {
        double A[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double u1 = x1[2 * j], v1 = x1[2 * j + 1], u2 = x2[2 * j], v2 = x2[2 * j + 1];
            const double a = gl < 3 ? u2 : (gl < 6 ? v2 : 1.0);
            const int c3 = gl % 3;
            const double b = c3 == 0 ? u1 : (c3 == 1 ? v1 : 1.0);
            A[j] = comp ? a * b : 0.0;
        }
        double hv[5], beta[5];  // Householder vectors (component gl) and their 2 / |v|^2
#pragma unroll
        for (int k = 0; k < 5; k++) {
            double x = gl >= k ? A[k] : 0.0;
            const double nrm2 = gsum(x * x);
            const double akk = gsum(gl == k ? A[k] : 0.0);
            const double alpha = akk >= 0 ? -sqrt(nrm2) : sqrt(nrm2);
            if (gl == k) x -= alpha;
            const double vn2 = gsum(x * x);
            beta[k] = vn2 > 0 ? 2.0 / vn2 : 0.0;
            hv[k] = x;
#pragma unroll
            for (int j = k + 1; j < 5; j++) {
                const double d = gsum(x * A[j]) * beta[k];
                A[j] -= d * x;
            }
        }
#pragma unroll
        for (int j = 0; j < 5; j++) basis[j] = gl == j ? 1.0 : 0.0;
#pragma unroll
        for (int k = 4; k >= 0; k--) {
            double d[5];
#pragma unroll
            for (int j = 0; j < 5; j++) d[j] = hv[k] * basis[j];
#pragma unroll
            for (int j = 0; j < 5; j++) d[j] = gsum(d[j]) * beta[k];
#pragma unroll
            for (int j = 0; j < 5; j++) basis[j] -= d[j] * hv[k];
        }
        // completion from cv's fixed +-1/9 vectors (five_point.h): bit 9 (n - 5) + i of SGN for vector n, component i
        constexpr unsigned long long SGN =
            (1ull << 2) | (1ull << 7) | (1ull << 8) | (1ull << 9) | (1ull << 11) | (1ull << 12) | (1ull << 13) |
            (1ull << 14) | (1ull << 15) | (1ull << 17) | (1ull << 18) | (1ull << 22) | (1ull << 23) | (1ull << 25) |
            (1ull << 26) | (1ull << 27) | (1ull << 30) | (1ull << 32) | (1ull << 33) | (1ull << 34);
#pragma unroll
        for (int n = 5; n < 9; n++) {
            double w = comp ? (((SGN >> (9 * (n - 5) + gl)) & 1ull) ? 1.0 / 9 : -1.0 / 9) : 0.0;
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                double d[8];
#pragma unroll
                for (int j = 0; j < n; j++) d[j] = w * basis[j];
#pragma unroll
                for (int j = 0; j < n; j++) d[j] = gsum(d[j]);
#pragma unroll
                for (int j = 0; j < n; j++) w -= d[j] * basis[j];
                // keep the scale near one between the passes, as cv's sweep does
                const double nn = gsum(w * w);
                w *= nn > DBL_MIN ? 1.0 / sqrt(nn) : 0.0;
            }
            basis[n] = w;
        }
        if (comp) {
#pragma unroll
            for (int k = 0; k < 4; k++) slab[FPC_E + 4 * gl + k] = basis[5 + k];
        }
    }
    lds_sync();

    // ---- constraints ------------------------------------------------------------------------------
    // lanes 0..5: L[a] = sum_k N(3 SI[a] + k) x N(3 SJ[a] + k); lanes 6..8: the minors of det(E)
    {
        constexpr int SI[6] = {0, 0, 0, 1, 1, 2}, SJ[6] = {0, 1, 2, 1, 2, 2};
        int ia[3], ib[3];
        double sg[3];
        if (gl < 6) {
            int si = 0, sj = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
                if (gl == a) {
                    si = SI[a];
                    sj = SJ[a];
                }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                ia[k] = 3 * si + k;
                ib[k] = 3 * sj + k;
                sg[k] = 1.0;
            }
        } else {
            // minor 0: e4 e8 - e5 e7; 1: e3 e8 - e5 e6; 2: e3 e7 - e4 e6
            const int t = gl - 6;
            ia[0] = t == 0 ? 4 : 3;
            ib[0] = t == 2 ? 7 : 8;
            ia[1] = t == 2 ? 4 : 5;
            ib[1] = t == 0 ? 7 : 6;
            ia[2] = 0;
            ib[2] = 0;
            sg[0] = 1.0;
            sg[1] = -1.0;
            sg[2] = 0.0;
        }
        double q[10];
#pragma unroll
        for (int i = 0; i < 10; i++) q[i] = 0;
        if (comp) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double a[4], b[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    a[m] = slab[FPC_E + 4 * ia[k] + m];
                    b[m] = slab[FPC_E + 4 * ib[k] + m];
                }
                mul_ll_acc(a, b, q, sg[k]);
            }
#pragma unroll
            for (int i = 0; i < 10; i++) slab[FPC_L + 10 * gl + i] = q[i];
        }
        lds_sync();
        // L = E E^T - 1/2 tr(E E^T) I: the diagonal entries 0, 3, 5 lose half the trace
        if (gl == 0 || gl == 3 || gl == 5) {
            double tr[10];
#pragma unroll
            for (int i = 0; i < 10; i++)
                tr[i] = 0.5 * (slab[FPC_L + i] + slab[FPC_L + 30 + i] + slab[FPC_L + 50 + i]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 10; i++) slab[FPC_L + 10 * gl + i] = q[i] - tr[i];
        }
        lds_sync();
    }
    // lane r < 10 builds row r of the 10 x 20 matrix: row 0 = det(E), row 1 + 3 i + j = (L E)(i, j)
    double row[20];
    {
        constexpr int SYM[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
        int src[3], ie[3];
        double sg[3];
        if (gl == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                src[k] = 6 + k;
                ie[k] = k;
                sg[k] = k == 1 ? -1.0 : 1.0;
            }
        } else {
            const int r = min(gl, 9) - 1, i = r / 3, j = r - 3 * i;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                int s = 0;
#pragma unroll
                for (int a = 0; a < 3; a++)
                    if (i == a) s = SYM[a][k];
                src[k] = s;
                ie[k] = 3 * k + j;
                sg[k] = 1.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 20; i++) row[i] = 0;
        if (gl < 10) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double q[10], l[4];
#pragma unroll
                for (int m = 0; m < 10; m++) q[m] = slab[FPC_L + 10 * src[k] + m];
#pragma unroll
                for (int m = 0; m < 4; m++) l[m] = slab[FPC_E + 4 * ie[k] + m];
                mul_ql_acc(q, l, row, sg[k]);
            }
        }
    }

    // ---- Gauss-Jordan on the first ten columns, lane = row, partial pivoting -------------------------
    int my_col = -1;  // the column this lane's row was the pivot of
    bool ok = true;
#pragma unroll
    for (int col = 0; col < 10; col++) {
        double v = (gl < 10 && my_col < 0) ? fabs(row[col]) : -1.0;
        int who = gl;
        argmax_step<0>(v, who);
        argmax_step<1>(v, who);
        argmax_step<2>(v, who);
        argmax_step<3>(v, who);
        if (!(v > 1e-300)) ok = false;
        if (gl == who) {
            const double inv = 1.0 / row[col];
#pragma unroll
            for (int k = col + 1; k < 20; k++) {
                row[k] *= inv;
                slab[FPC_PROW + k] = row[k];
            }
            my_col = col;
        }
        lds_sync();
        if (gl != who) {
            const double f = row[col];
#pragma unroll
            for (int k = col + 1; k < 20; k++) row[k] -= f * slab[FPC_PROW + k];
        }
        lds_sync();
    }
    if (!ok) return 0;
    if (my_col >= 4) {
#pragma unroll
        for (int k = 0; k < 10; k++) slab[FPC_RHS + 10 * (my_col - 4) + k] = row[10 + k];
    }
    lds_sync();

    // ---- B(z): row i on lane i; det B(z): cofactor term t on lane t, summed over the group ----------------
    // (the same instructions on every lane, the operands chosen by address: three different compile-time
    // index sets would run one after the other)
    double c[11];
    {
        const int i = gl < 3 ? gl : 0;
        double r1[10], r2[10];
#pragma unroll
        for (int k = 0; k < 10; k++) {
            r1[k] = slab[FPC_RHS + 10 * (2 * i) + k];
            r2[k] = slab[FPC_RHS + 10 * (2 * i + 1) + k];
        }
        if (gl < 3) {
            double* o = slab + FPC_B + 15 * i;
#pragma unroll
            for (int blk = 0; blk < 2; blk++) {
                o[5 * blk + 3] = -r2[3 * blk + 0];
                o[5 * blk + 2] = r1[3 * blk + 0] - r2[3 * blk + 1];
                o[5 * blk + 1] = r1[3 * blk + 1] - r2[3 * blk + 2];
                o[5 * blk + 0] = r1[3 * blk + 2];
                o[5 * blk + 4] = 0;
            }
            o[14] = -r2[6];
            o[13] = r1[6] - r2[7];
            o[12] = r1[7] - r2[8];
            o[11] = r1[8] - r2[9];
            o[10] = r1[9];
        }
        lds_sync();
        // cofactors along row 0: columns (1,2), (0,2), (0,1) of rows 1, 2; columns 0, 1 carry a zero z^4 term
        const int t = i, ca = t == 0 ? 1 : 0, cb = t == 2 ? 1 : 2;
        double b0[5], b1a[5], b1b[5], b2a[5], b2b[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            b0[k] = slab[FPC_B + 5 * t + k];
            b1a[k] = slab[FPC_B + 15 + 5 * ca + k];
            b1b[k] = slab[FPC_B + 15 + 5 * cb + k];
            b2a[k] = slab[FPC_B + 30 + 5 * ca + k];
            b2b[k] = slab[FPC_B + 30 + 5 * cb + k];
        }
        double m[9], cc[13];
#pragma unroll
        for (int k = 0; k < 9; k++) m[k] = 0;
#pragma unroll
        for (int k = 0; k < 13; k++) cc[k] = 0;
#pragma unroll
        for (int p = 0; p < 5; p++)
#pragma unroll
            for (int q = 0; q < 5; q++) m[p + q] += b1a[p] * b2b[q] - b1b[q] * b2a[p];
        const double sgn = gl >= 3 ? 0.0 : (t == 1 ? -1.0 : 1.0);
#pragma unroll
        for (int p = 0; p < 5; p++)
#pragma unroll
            for (int q = 0; q < 9; q++) cc[p + q] += sgn * (b0[p] * m[q]);
#pragma unroll
        for (int k = 0; k <= 10; k++) c[k] = gsum(cc[k]);
    }

    // ---- real roots: lane k finds the k-th ---------------------------------------------------------
    double cmax = 0;
#pragma unroll
    for (int k = 0; k <= 10; k++) cmax = fmax(cmax, fabs(c[k]));
    if (!(cmax > 0) || !(fabs(c[10]) > 0)) return 0;
    {
        const double sc = 1.0 / cmax;
#pragma unroll
        for (int k = 0; k <= 10; k++) c[k] *= sc;
    }
    double bmax = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) bmax = fmax(bmax, fabs(c[k]));
    const double bound = 1.0 + bmax / fabs(c[10]);  // Cauchy bound
    if (!(bound < 1e300)) return 0;

    double root = 0;
    bool have_root = false;
    SturmChain ch;
    int total = -1;
    if (sturm_build(c, ch)) {
        const int v_minf = sign_changes(ch.neg_minf), v_pinf = sign_changes(ch.neg_pinf);
        total = v_minf - v_pinf;
        if (total < 0 || total > 10) total = -1;
        if (total > gl) {
            // roots <= x number v_minf - V(x); root k is alone in (a, b] when the counts at the ends are k and k + 1
            double a = -1.0, b = 1.0;
            int ca = 0, cb = total, step = 0;
            while ((ca != gl || cb != gl + 1 || a <= -1.0 || b >= 1.0) && step < 80) {
                const double m = 0.5 * (a + b);
                const int cm = v_minf - sturm_count(ch, m / (1.0 - fabs(m)));
                if (cm >= gl + 1) {
                    b = m;
                    cb = cm;
                } else {
                    a = m;
                    ca = cm;
                }
                step++;
            }
            if (ca == gl && cb == gl + 1 && a > -1.0 && b < 1.0) {
                const double xa = a / (1.0 - fabs(a)), xb = b / (1.0 - fabs(b));
                double fa, fb, d;
                horner<10>(c, xa, fa, d);
                horner<10>(c, xb, fb, d);
                if ((fa < 0) != (fb < 0)) {
                    root = bracket_root<10>(c, xa, xb, fa, 4e-16);
                    have_root = true;
                }
            }
        }
    }
    // any lane of the group in trouble (or a degenerate chain) sends the group to the derivative-chain finder
    {
        const int bad = gor(total < 0 ? 1 : 0);
        if (bad) {
            lds_sync();
            if (gl == 0) {
                LdsScratch s{slab + FPC_ROOTS};
                int n = level_roots<1>(c, bound, 0, s, 10, 0);
                n = level_roots<2>(c, bound, n, s, 0, 10);
                n = level_roots<3>(c, bound, n, s, 10, 0);
                n = level_roots<4>(c, bound, n, s, 0, 10);
                n = level_roots<5>(c, bound, n, s, 10, 0);
                n = level_roots<6>(c, bound, n, s, 0, 10);
                n = level_roots<7>(c, bound, n, s, 10, 0);
                n = level_roots<8>(c, bound, n, s, 0, 10);
                n = level_roots<9>(c, bound, n, s, 10, 0);
                n = level_roots<10>(c, bound, n, s, 0, 10);  // roots at slab[FPC_ROOTS + 10 ..]
                slab[FPC_C] = (double)n;
            }
            lds_sync();
            const int n = (int)slab[FPC_C];
            have_root = gl < n;
            root = have_root ? slab[FPC_ROOTS + 10 + gl] : 0.0;
        }
    }

    // ---- back-substitution: lane k -> model of root k -------------------------------------------------
    double Ev[9];
    bool valid = false;
    if (have_root) {
        const double z1 = root, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double* o = slab + FPC_B + 15 * j;
            bz[j][0] = o[3] * z3 + o[2] * z2 + o[1] * z1 + o[0];
            bz[j][1] = o[8] * z3 + o[7] * z2 + o[6] * z1 + o[5];
            bz[j][2] = o[14] * z4 + o[13] * z3 + o[12] * z2 + o[11] * z1 + o[10];
        }
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
        if (bn > 0 && !(fabs(best[2] * (1.0 / sqrt(bn))) < 1e-10)) {
            const double xs = best[0] / best[2], ys = best[1] / best[2];
            double nrm = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const double* e = slab + FPC_E + 4 * k;
                Ev[k] = e[0] * xs + e[1] * ys + e[2] * z1 + e[3];
                nrm += Ev[k] * Ev[k];
            }
            nrm = sqrt(nrm);
            if (nrm > 0) {
                valid = true;
#pragma unroll
                for (int k = 0; k < 9; k++) Ev[k] /= nrm;
            }
        }
    }
    const unsigned long long bal = __ballot(valid);
    const unsigned group_bits = (unsigned)(bal >> (((threadIdx.x & 63) / FPC_G) * FPC_G)) & 0xffffu;
    if (valid) {
        const int at = __builtin_popcount(group_bits & ((1u << gl) - 1u));
#pragma unroll
        for (int k = 0; k < 9; k++) E_out[9 * at + k] = Ev[k];
    }
    return __builtin_popcount(group_bits);
}

#endif
