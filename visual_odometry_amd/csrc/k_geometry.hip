// k_geometry.hip — gfx950 kernels for cv::findEssentialMat (RANSAC / LMedS), cv::recoverPose and
// cv::triangulatePoints, all in fp64.
//
//   k_solve5        one 5-point hypothesis per lane; the 10x20 elimination matrix lives in a
//                   per-lane LDS column (120 KB per 64-lane block, one block per CU)
//   k_score         one wave per hypothesis: Sampson error of every point against its <= 10 models,
//                   inlier counts by wave ballot + popcount
//   k_replay        one wave per pair: replays cv's sequential acceptance rule (strictly more
//                   inliers wins, the iteration cap shrinks) over the counts, so the result is the
//                   one the sequential CPU loop reaches
//   k_lmeds_*       per-model median of the float errors by bitwise selection in LDS; arg-min
//   k_est_finish    inlier mask of the winning model
//   k_pose_*        decomposeEssentialMat, 4 x N DLT triangulations (4x4 Jacobi SVD per point),
//                   cheirality masks, winner selection
// Reference call sites: src/matcher/matcher.cpp:42, src/visual_odometry/vo_pose.h:14,19,
// src/sfm/triangulation.cpp:64.
#include <cstdlib>

#include "kernels.h"
#include "geom_types.h"
#include "five_point.h"
#include "five_point_coop.h"


namespace {

template <int LANES>
struct LaneScratch {
    double* base;  // lds + lane; element i at base[LANES i]
    __device__ __forceinline__ double& operator()(int i) const { return base[i * LANES]; }
};

__device__ __forceinline__ int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = fmax(p, 0.);
    p = fmin(p, 1.);
    ep = fmax(ep, 0.);
    ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : __double2int_rn(num / denom);
}

}  // namespace

__global__ void k_est_init(EstState* __restrict__ st, const int* __restrict__ d_n, int n_host, int n_cap,
                           int niters, float thr_f, int npairs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    EstState s;
    int n = d_n ? d_n[p] : n_host;
    n = min(n, n_cap);
    s.n = n;
    s.niters = n < 5 ? 0 : (n == 5 ? 1 : niters);
    s.best_count = 0;
    s.best_h = -1;
    s.best_m = 0;
    s.next_h = 0;
    s.status = VO_ERR_FAILED;
    s.n_inliers = 0;
    s.thr_f = thr_f;
    s.best_median = DBL_MAX;
    for (int i = 0; i < 9; i++) s.E[i] = 0;
    st[p] = s;
}

// K-normalised double coordinates: x = (u - cx) / fx, y = (v - cy) / fy
__global__ __launch_bounds__(256) void k_normalize(const float2* __restrict__ p0, const float2* __restrict__ p1,
                                                   const CamK* __restrict__ cam, const EstState* __restrict__ st,
                                                   const int* __restrict__ d_n, int n_host, int n_cap,
                                                   double2* __restrict__ x1, double2* __restrict__ x2) {
    const int pair = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int n = st ? st[pair].n : (d_n ? d_n[pair] : n_host);
    n = min(n, n_cap);
    if (i >= n) return;
    const CamK k = cam[pair];
    const size_t o = (size_t)pair * n_cap + i;
    const float2 a = p0[o], b = p1[o];
    x1[o] = make_double2(((double)a.x - k.cx) / k.fx, ((double)a.y - k.cy) / k.fy);
    x2[o] = make_double2(((double)b.x - k.cx) / k.fx, ((double)b.y - k.cy) / k.fy);
}

// LANES hypotheses per block (one partially filled wave when LANES < 64): small launches use few
// lanes per wave so that the hypotheses spread over many CUs and diverge less.
template <int LANES>
__global__ __launch_bounds__(LANES) void k_solve5(int h0, int h1, const EstState* __restrict__ st,
                                               const uint16_t* __restrict__ samples, int samples_iters,
                                               const double2* __restrict__ x1, const double2* __restrict__ x2,
                                               int n_cap, double* __restrict__ models, int* __restrict__ nmodels,
                                               int H) {
    __shared__ double lds[FP_SCRATCH * LANES];
    const int pair = blockIdx.y, lane = threadIdx.x;
    const int hb = h0 + blockIdx.x * LANES;
    const int cap = min(h1, st[pair].niters);
    if (hb >= cap) return;
    const int h = hb + lane;
    if (h >= cap) return;
    const int n = st[pair].n;
    const uint16_t* sidx = samples + ((size_t)n * samples_iters + h) * 5;
    double a[10], b[10];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int id = sidx[k];
        const double2 u = x1[(size_t)pair * n_cap + id], v = x2[(size_t)pair * n_cap + id];
        a[2 * k] = u.x;
        a[2 * k + 1] = u.y;
        b[2 * k] = v.x;
        b[2 * k + 1] = v.y;
    }
    LaneScratch<LANES> s{lds + lane};
    const size_t slot = (size_t)pair * H + h;
    nmodels[slot] = five_point_solve(a, b, s, models + slot * 90);
}

// One hypothesis per 16-lane group, four per wave (five_point_coop.h): the dependent chain of a hypothesis is
// a few thousand operations instead of ~25 k, which is what a launch of a few thousand hypotheses (the first
// RANSAC rounds, LMedS on one pair) waits for.
__global__ __launch_bounds__(64) void k_solve5_coop(int h0, int h1, const EstState* __restrict__ st,
                                                    const uint16_t* __restrict__ samples, int samples_iters,
                                                    const double2* __restrict__ x1, const double2* __restrict__ x2,
                                                    int n_cap, double* __restrict__ models,
                                                    int* __restrict__ nmodels, int H) {
    __shared__ double lds[FPC_SLAB * 4];
    const int pair = blockIdx.y, grp = threadIdx.x / FPC_G, gl = threadIdx.x % FPC_G;
    const int cap = min(h1, st[pair].niters);
    const int h = h0 + blockIdx.x * 4 + grp;
    if (h >= cap) return;
    const int n = st[pair].n;
    const uint16_t* sidx = samples + ((size_t)n * samples_iters + h) * 5;
    double a[10], b[10];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int id = sidx[k];
        const double2 u = x1[(size_t)pair * n_cap + id], v = x2[(size_t)pair * n_cap + id];
        a[2 * k] = u.x;
        a[2 * k + 1] = u.y;
        b[2 * k] = v.x;
        b[2 * k + 1] = v.y;
    }
    const size_t slot = (size_t)pair * H + h;
    const int nm = five_point_solve_coop(a, b, lds + FPC_SLAB * grp, gl, models + slot * 90);
    if (gl == 0) nmodels[slot] = nm;
}

// vo_five_point: `count` independent minimal problems (10 doubles per view and problem), either solver.
template <bool COOP>
__global__ __launch_bounds__(64) void k_five_point_batch(int count, const double* __restrict__ x1,
                                                         const double* __restrict__ x2, double* __restrict__ models,
                                                         int* __restrict__ nmodels) {
    constexpr int PER_BLOCK = COOP ? 4 : 16;
    __shared__ double lds[COOP ? FPC_SLAB * 4 : FP_SCRATCH * 16];
    const int sub = COOP ? threadIdx.x / FPC_G : threadIdx.x, gl = threadIdx.x % FPC_G;
    const int h = blockIdx.x * PER_BLOCK + sub;
    if (h >= count || (!COOP && threadIdx.x >= 16)) return;
    double a[10], b[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        a[k] = x1[(size_t)h * 10 + k];
        b[k] = x2[(size_t)h * 10 + k];
    }
    if (COOP) {
        const int nm = five_point_solve_coop(a, b, lds + FPC_SLAB * sub, gl, models + (size_t)h * 90);
        if (gl == 0) nmodels[h] = nm;
    } else {
        LaneScratch<16> s{lds + sub};
        nmodels[h] = five_point_solve(a, b, s, models + (size_t)h * 90);
    }
}

// One wave per hypothesis; its models are staged in LDS, every point is loaded once.
__global__ __launch_bounds__(256) void k_score(int h0, int h1, const EstState* __restrict__ st,
                                               const double2* __restrict__ x1, const double2* __restrict__ x2,
                                               int n_cap, const double* __restrict__ models,
                                               const int* __restrict__ nmodels, int* __restrict__ counts, int H) {
    __shared__ double Es[4][90];
    const int pair = blockIdx.y, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = h0 + blockIdx.x * 4 + wv;
    const EstState& s = st[pair];
    if (h >= min(h1, s.niters)) return;
    const size_t slot = (size_t)pair * H + h;
    const int nm = nmodels[slot];
    for (int i = lane; i < nm * 9; i += 64) Es[wv][i] = models[slot * 90 + i];
    __builtin_amdgcn_wave_barrier();
    const int n = s.n;
    const float thr = s.thr_f;
    const double thr_m = sampson_bound(thr);
    int cnt[10];
#pragma unroll
    for (int m = 0; m < 10; m++) cnt[m] = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = i < n;
        const double2 u = ok ? x1[(size_t)pair * n_cap + i] : make_double2(0, 0);
        const double2 v = ok ? x2[(size_t)pair * n_cap + i] : make_double2(0, 0);
#pragma unroll
        for (int m = 0; m < 10; m++) {
            if (m < nm) {
                const bool inl = sampson_inlier(&Es[wv][9 * m], u.x, u.y, v.x, v.y, thr, thr_m);
                cnt[m] += __popcll(__ballot(ok && inl));
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < 10; m++)
            if (m < nm) counts[slot * 10 + m] = cnt[m];
    }
}

// cv's RANSAC acceptance loop over hypotheses [h0, h1): a model replaces the best one iff it has
// strictly more inliers (and more than 4); every acceptance shrinks the iteration cap.
__global__ __launch_bounds__(64) void k_replay(int h0, int h1, EstState* __restrict__ st,
                                               const int* __restrict__ nmodels, const int* __restrict__ counts,
                                               int H, double prob) {
    const int pair = blockIdx.x, lane = threadIdx.x;
    EstState& s = st[pair];
    int niters = s.niters, best = s.best_count, best_h = s.best_h, best_m = s.best_m;
    const int n = s.n;
    for (int base = max(h0, s.next_h); base < min(h1, niters); base += 64) {
        const int h = base + lane;
        int v = 0, vm = 0;
        if (h < min(h1, niters)) {
            const size_t slot = (size_t)pair * H + h;
            const int nm = nmodels[slot];
            for (int m = 0; m < nm; m++) {
                const int c = counts[slot * 10 + m];
                if (c > v) {
                    v = c;
                    vm = m;
                }
            }
        }
        // exclusive prefix maximum over lower lanes, seeded with the running best and cv's floor of 4
        int pm = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(pm, o);
            if (lane >= o) pm = max(pm, t);
        }
        int ex = __shfl_up(pm, 1);
        if (lane == 0) ex = 0;
        ex = max(ex, max(best, 4));
        unsigned long long rec = __ballot(v > ex);
        while (rec) {
            const int bpos = __ffsll((long long)rec) - 1;
            rec &= rec - 1;
            if (base + bpos >= niters) {
                rec = 0;
                break;
            }
            best = __shfl(v, bpos);
            best_m = __shfl(vm, bpos);
            best_h = base + bpos;
            niters = ransac_update_num_iters(prob, (double)(n - best) / n, 5, niters);
        }
    }
    if (lane == 0) {
        s.niters = niters;
        s.best_count = best;
        s.best_h = best_h;
        s.best_m = best_m;
        s.next_h = h1;
    }
}

// LMedS: median of the float Sampson errors of one model.  Non-negative floats order like unsigned
// integers, so the median is a radix selection on the error bit patterns held in LDS: four passes of
// eight bits, each a 256-bin histogram (LDS atomics) of the elements that still match the prefix, a wave
// scan of the bins and a step into the bin that holds the wanted rank.  One wave per model.
namespace {

// k-th smallest (0-based) of e[0..n): every lane returns it
__device__ uint32_t wave_radix_select(const uint32_t* e, int n, int k, uint32_t* hist, int lane) {
    uint32_t prefix = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        reinterpret_cast<uint4*>(hist)[lane] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n; i += 64) {
            const uint32_t v = e[i];
            if (shift == 24 || (v >> (shift + 8)) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint4 h = reinterpret_cast<const uint4*>(hist)[lane];  // bins 4 lane .. 4 lane + 3
        const int t = (int)(h.x + h.y + h.z + h.w);
        int inc = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        const int owner = __ffsll((long long)__ballot(inc > k)) - 1;  // first lane whose bins reach rank k
        const int ex = __shfl(inc - t, owner);
        const uint32_t hx = __shfl(h.x, owner), hy = __shfl(h.y, owner), hz = __shfl(h.z, owner);
        int r = k - ex, b = 4 * owner;
        if (r >= (int)hx) {
            r -= hx, b++;
            if (r >= (int)hy) {
                r -= hy, b++;
                if (r >= (int)hz) r -= hz, b++;
            }
        }
        k = r;
        prefix = (prefix << 8) | (uint32_t)b;
        __builtin_amdgcn_wave_barrier();
    }
    return prefix;
}

}  // namespace

__global__ __launch_bounds__(256) void k_lmeds_medians(const EstState* __restrict__ st,
                                                       const double2* __restrict__ x1,
                                                       const double2* __restrict__ x2, int n_cap,
                                                       const double* __restrict__ models,
                                                       const int* __restrict__ nmodels,
                                                       double* __restrict__ medians, int H) {
    extern __shared__ uint32_t errs[];  // 4 waves x (n_cap errors + 256 bins)
    const int pair = blockIdx.y, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int idx = blockIdx.x * 4 + wv;
    const int h = idx / 10, m = idx % 10;
    const EstState& s = st[pair];
    if (h >= s.niters) return;
    const size_t slot = (size_t)pair * H + h;
    if (m >= nmodels[slot]) return;
    const int n = s.n;
    const int ecap = (n_cap + 3) & ~3;
    uint32_t* e = errs + (size_t)wv * (ecap + 256);
    uint32_t* hist = e + ecap;  // 16-byte aligned
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; k++) E[k] = models[slot * 90 + 9 * m + k];
    for (int i = lane; i < n; i += 64) {
        const double2 u = x1[(size_t)pair * n_cap + i], v = x2[(size_t)pair * n_cap + i];
        e[i] = __float_as_uint(sampson_error(E, u.x, u.y, v.x, v.y));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // upper median = the (n/2)-th smallest; for even n the lower one is the largest element below it
    // unless the upper median's value also occupies rank n/2 - 1
    const int kb = n / 2;
    const uint32_t rb = wave_radix_select(e, n, kb, hist, lane);
    uint32_t ra = rb;
    if (!(n & 1)) {
        int below = 0;
        uint32_t mx = 0;
        for (int i = lane; i < n; i += 64) {
            const uint32_t v = e[i];
            if (v < rb) {
                below++;
                mx = max(mx, v);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            below += __shfl_xor(below, o);
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        }
        if (below == kb) ra = mx;  // rank n/2 - 1 is the largest of the elements below
    }
    if (lane == 0) {
        const float fa = __uint_as_float(ra), fb = __uint_as_float(rb);
        medians[slot * 10 + m] = (n & 1) ? (double)fb : (double)(fa + fb) * 0.5;
    }
}

// The same median by a whole block (four waves) per model, for calls with few pairs: there the launch is a
// few hundred waves on a thousand SIMDs and lasts as long as ONE wave needs for its model (n fp64 Sampson errors,
// four histogram passes over them); four waves share both.  Same selection, same value.
__global__ __launch_bounds__(256) void k_lmeds_medians_blk(const EstState* __restrict__ st,
                                                           const double2* __restrict__ x1,
                                                           const double2* __restrict__ x2, int n_cap,
                                                           const double* __restrict__ models,
                                                           const int* __restrict__ nmodels,
                                                           double* __restrict__ medians, int H) {
    extern __shared__ uint32_t errs[];  // n_cap errors + 256 bins
    __shared__ int s_below[4];
    __shared__ uint32_t s_mx[4];
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = blockIdx.x / 10, m = blockIdx.x % 10;
    const EstState& s = st[pair];
    if (h >= s.niters) return;
    const size_t slot = (size_t)pair * H + h;
    if (m >= nmodels[slot]) return;
    const int n = s.n;
    const int ecap = (n_cap + 3) & ~3;
    uint32_t* e = errs;
    uint32_t* hist = e + ecap;  // 16-byte aligned
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; k++) E[k] = models[slot * 90 + 9 * m + k];
    for (int i = tid; i < n; i += 256) {
        const double2 u = x1[(size_t)pair * n_cap + i], v = x2[(size_t)pair * n_cap + i];
        e[i] = __float_as_uint(sampson_error(E, u.x, u.y, v.x, v.y));
    }
    const int kb = n / 2;
    // radix selection of rank kb: every thread ends with the value
    uint32_t prefix = 0;
    int k = kb;
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();  // (also: the errors are written)
        for (int i = tid; i < n; i += 256) {
            const uint32_t v = e[i];
            if (shift == 24 || (v >> (shift + 8)) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        const uint4 hq = reinterpret_cast<const uint4*>(hist)[lane];  // bins 4 lane .. 4 lane + 3, every wave
        const int t = (int)(hq.x + hq.y + hq.z + hq.w);
        int inc = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        const int owner = __ffsll((long long)__ballot(inc > k)) - 1;  // first lane whose bins reach rank k
        const int ex = __shfl(inc - t, owner);
        const uint32_t hx = __shfl(hq.x, owner), hy = __shfl(hq.y, owner), hz = __shfl(hq.z, owner);
        int r = k - ex, b = 4 * owner;
        if (r >= (int)hx) {
            r -= hx, b++;
            if (r >= (int)hy) {
                r -= hy, b++;
                if (r >= (int)hz) r -= hz, b++;
            }
        }
        k = r;
        prefix = (prefix << 8) | (uint32_t)b;
        __syncthreads();  // the bins are read before they are cleared again
    }
    const uint32_t rb = prefix;
    uint32_t ra = rb;
    if (!(n & 1)) {
        int below = 0;
        uint32_t mx = 0;
        for (int i = tid; i < n; i += 256) {
            const uint32_t v = e[i];
            if (v < rb) {
                below++;
                mx = max(mx, v);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            below += __shfl_xor(below, o);
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        }
        if (lane == 0) {
            s_below[wv] = below;
            s_mx[wv] = mx;
        }
        __syncthreads();
        below = s_below[0] + s_below[1] + s_below[2] + s_below[3];
        mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
        if (below == kb) ra = mx;  // rank n/2 - 1 is the largest of the elements below
    }
    if (tid == 0) {
        const float fa = __uint_as_float(ra), fb = __uint_as_float(rb);
        medians[slot * 10 + m] = (n & 1) ? (double)fb : (double)(fa + fb) * 0.5;
    }
}

// first strictly smaller median wins (hypotheses, then models, in order); sets the inlier threshold
__global__ __launch_bounds__(256) void k_lmeds_select(EstState* __restrict__ st, const int* __restrict__ nmodels,
                                                      const double* __restrict__ medians, int H) {
    __shared__ double s_best[4];
    __shared__ int s_bi[4];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    EstState& s = st[pair];
    if (s.n == 5) {
        // cv solves once on all five points and marks them all inliers; the first model is kept
        if (tid == 0) {
            s.best_h = nmodels[(size_t)pair * H] > 0 ? 0 : -1;
            s.best_m = 0;
            s.thr_f = __builtin_inff();
        }
        return;
    }
    const int total = s.niters * 10;
    double best = DBL_MAX;
    int bi = 0x7fffffff;
    // a slot's model count and its median are requested together (the median of a slot without a model is whatever
    // the table holds and is not looked at): one round trip for the thread's slots instead of two per slot
    for (int i0 = 0; i0 < total; i0 += 1024) {
        int nm[4];
        double mv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = min(i0 + 256 * u + tid, total - 1);
            const size_t slot = (size_t)pair * H + idx / 10;
            nm[u] = nmodels[slot];
            mv[u] = medians[slot * 10 + idx % 10];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = i0 + 256 * u + tid;
            if (idx < total && idx % 10 < nm[u] && mv[u] < best) {
                best = mv[u];
                bi = idx;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob < best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (lane == 0) {
        s_best[tid >> 6] = best;
        s_bi[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; w++)
            if (s_best[w] < best || (s_best[w] == best && s_bi[w] < bi)) {
                best = s_best[w];
                bi = s_bi[w];
            }
        if (best < DBL_MAX) {
            s.best_h = bi / 10;
            s.best_m = bi % 10;
            s.best_median = best;
            double sigma = 2.5 * 1.4826 * (1 + 5. / (s.n - 5)) * sqrt(best);
            sigma = fmax(sigma, 0.001);
            s.thr_f = (float)(sigma * sigma);
        } else {
            s.best_h = -1;
        }
    }
}

// inlier mask + count of the winning model, E into the state
__global__ __launch_bounds__(256) void k_est_finish(EstState* __restrict__ st, const double2* __restrict__ x1,
                                                    const double2* __restrict__ x2, int n_cap,
                                                    const double* __restrict__ models, uint8_t* __restrict__ mask,
                                                    int H) {
    __shared__ int total;
    const int pair = blockIdx.x;
    EstState& s = st[pair];
    const int n = s.n;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    uint8_t* mk = mask + (size_t)pair * n_cap;
    if (s.best_h < 0) {
        for (int i = threadIdx.x; i < n; i += 256) mk[i] = 0;
        return;
    }
    double E[9];
    const size_t slot = (size_t)pair * H + s.best_h;
#pragma unroll
    for (int k = 0; k < 9; k++) E[k] = models[slot * 90 + 9 * s.best_m + k];
    const float thr = s.thr_f;
    const double thr_m = sampson_bound(thr);
    int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + threadIdx.x;
        bool inl = false;
        if (i < n) {
            const double2 u = x1[(size_t)pair * n_cap + i], v = x2[(size_t)pair * n_cap + i];
            inl = sampson_inlier(E, u.x, u.y, v.x, v.y, thr, thr_m);
            mk[i] = inl ? 1 : 0;
        }
        cnt += __popcll(__ballot(inl));
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) s.E[k] = E[k];
        s.n_inliers = total;
        s.status = VO_OK;
    }
}

// ---------------------------------------------------------------------------------------------
// recoverPose
// ---------------------------------------------------------------------------------------------
__global__ void k_pose_setup(const double* __restrict__ E, size_t e_stride, PoseState* __restrict__ ps,
                             int npairs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const double* e = E + p * e_stride;
    PoseState& s = ps[p];
    double Ev[9], R1[9], R2[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) Ev[k] = e[k];
    decompose_essential(Ev, R1, R2, t);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        s.R1[k] = R1[k];
        s.R2[k] = R2[k];
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const double* R = (c & 1) ? R2 : R1;
        const double sg = c < 2 ? 1.0 : -1.0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int k = 0; k < 3; k++) s.P[c][4 * r + k] = R[3 * r + k];
            s.P[c][4 * r + 3] = sg * t[r];
        }
        s.good[c] = 0;
    }
#pragma unroll
    for (int r = 0; r < 3; r++) s.t[r] = t[r];
}

__global__ __launch_bounds__(256) void k_pose_points(PoseState* __restrict__ ps, const double2* __restrict__ x1,
                                                     const double2* __restrict__ x2, const int* __restrict__ d_n,
                                                     int n_host, int n_cap, const uint8_t* __restrict__ mask_in,
                                                     double dist, double* __restrict__ q4,
                                                     uint8_t* __restrict__ cmask) {
    const int pair = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = min(d_n ? d_n[pair] : n_host, n_cap);
    const bool ok = i < n;
    PoseState& s = ps[pair];
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double2 u = make_double2(0, 0), v = u;
    if (ok) {
        u = x1[(size_t)pair * n_cap + i];
        v = x2[(size_t)pair * n_cap + i];
    }
    const uint8_t min_ = (ok && mask_in) ? mask_in[(size_t)pair * n_cap + i] : 0xff;
    {
        // blockIdx.z = 0, 1: the rotation R1 / R2.  Candidates c = z ([R | t]) and c = z + 2 ([R | -t]) share one
        // decomposition: their DLT systems differ in the sign of the fourth column, and jacobi_svd's mirror gives
        // the second null vector from the first with the bits four separate calls would produce (the oracle makes
        // the four calls; tests/test_gpu_geometry.py holds the masks and points to it).
        const int c = blockIdx.z;
        bool m = false, mm = false;
        if (ok) {
            double P1[12], X[4], Xm[4];
#pragma unroll
            for (int k = 0; k < 12; k++) P1[k] = s.P[c][k];
            triangulate_point(P0, P1, u.x, u.y, v.x, v.y, X, Xm);
            // candidate c, then candidate c + 2 with the translation column negated
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const double* Xh = h ? Xm : X;
                const int ch = c + 2 * h;
                double* q = q4 + (((size_t)pair * 4 + ch) * 4) * n_cap + i;
#pragma unroll
                for (int k = 0; k < 4; k++) q[(size_t)k * n_cap] = Xh[k];
                bool mh = Xh[2] * Xh[3] > 0;
                const double xn = Xh[0] / Xh[3], yn = Xh[1] / Xh[3], zn = Xh[2] / Xh[3], wn = Xh[3] / Xh[3];
                mh = (zn < dist) && mh;
                double z2 = 0;
                z2 += P1[8] * xn;
                z2 += P1[9] * yn;
                z2 += P1[10] * zn;
                z2 += (h ? -P1[11] : P1[11]) * wn;
                mh = (z2 > 0) && mh;
                mh = (z2 < dist) && mh;
                const uint8_t mv = (uint8_t)((mh ? 255 : 0) & min_);
                cmask[((size_t)pair * 4 + ch) * n_cap + i] = mv;
                if (h) mm = mv != 0;
                else m = mv != 0;
            }
        }
        const int cnt = __popcll(__ballot(m)), cntm = __popcll(__ballot(mm));
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s.good[c], cnt);
        if ((threadIdx.x & 63) == 0 && cntm) atomicAdd(&s.good[c + 2], cntm);
    }
}

__global__ __launch_bounds__(256) void k_pose_select(PoseState* __restrict__ ps, const int* __restrict__ d_n,
                                                     int n_host, int n_cap, const double* __restrict__ q4,
                                                     const uint8_t* __restrict__ cmask, uint8_t* __restrict__ mask_out,
                                                     double* __restrict__ pts4) {
    const int pair = blockIdx.y;
    PoseState& s = ps[pair];
    const int n = min(d_n ? d_n[pair] : n_host, n_cap);
    const int g0 = s.good[0], g1 = s.good[1], g2 = s.good[2], g3 = s.good[3];
    int win;
    if (g0 >= g1 && g0 >= g2 && g0 >= g3) win = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3) win = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3) win = 2;
    else win = 3;
    {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < n) {
            mask_out[(size_t)pair * n_cap + i] = cmask[((size_t)pair * 4 + win) * n_cap + i];
#pragma unroll
            for (int k = 0; k < 4; k++)
                pts4[((size_t)pair * 4 + k) * n_cap + i] = q4[(((size_t)pair * 4 + win) * 4 + k) * n_cap + i];
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double* R = (win & 1) ? s.R2 : s.R1;
        const double sg = win < 2 ? 1.0 : -1.0;
#pragma unroll
        for (int k = 0; k < 9; k++) s.R[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; k++) s.tt[k] = sg * s.t[k];
        s.win = win;
        s.n_good = win == 0 ? g0 : win == 1 ? g1 : win == 2 ? g2 : g3;
    }
}

// triangulate(): cv::triangulatePoints on pixel coordinates + points4DToVec
__global__ __launch_bounds__(256) void k_triangulate(const double* __restrict__ P, const float2* __restrict__ p0,
                                                     const float2* __restrict__ p1, int n,
                                                     double* __restrict__ pts3) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double P0[12], P1[12], X[4];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        P0[k] = P[k];
        P1[k] = P[12 + k];
    }
    const float2 a = p0[i], b = p1[i];
    triangulate_point(P0, P1, (double)a.x, (double)a.y, (double)b.x, (double)b.y, X);
    pts3[3 * (size_t)i + 0] = X[0] / X[3];
    pts3[3 * (size_t)i + 1] = X[1] / X[3];
    pts3[3 * (size_t)i + 2] = X[2] / X[3];
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
void launch_est_init(hipStream_t st, EstState* s, const int* d_n, int n_host, int n_cap, int niters, float thr_f,
                     int npairs) {
    hipLaunchKernelGGL(k_est_init, dim3(cdiv(npairs, 64)), dim3(64), 0, st, s, d_n, n_host, n_cap, niters, thr_f,
                       npairs);
}

void launch_normalize(hipStream_t st, const float2* p0, const float2* p1, const CamK* cam, const EstState* s,
                      const int* d_n, int n_host, int n_cap, int n_grid, double2* x1, double2* x2, int npairs) {
    if (n_grid <= 0) return;
    hipLaunchKernelGGL(k_normalize, dim3(cdiv(n_grid, 256), npairs), dim3(256), 0, st, p0, p1, cam, s, d_n, n_host,
                       n_cap, x1, x2);
}

void launch_solve5(hipStream_t st, int h0, int h1, const EstState* s, const uint16_t* samples, int samples_iters,
                   const double2* x1, const double2* x2, int n_cap, double* models, int* nmodels, int H,
                   int npairs, int per_lane) {
    // Which solver runs is a property of the context (VO_OPT_SOLVER), never of the launch size: the two are
    // not bit-identical (tests/test_gpu_geometry.py::test_five_point_lane_and_group_solvers_agree), so a rule
    // that looked at (h1 - h0) * npairs would make a pair's E depend on how many pairs share its chunk.  The
    // group solver is the default at every size: 48 pairs x 680 hypotheses take 52 us with it against 193 us
    // one per lane (profiles/r3a_solver.txt), because a lane's 10 x 20 system needs 1.6 KB of LDS and one
    // wave fills a CU.
    if (!per_lane) {
        hipLaunchKernelGGL(k_solve5_coop, dim3(cdiv(h1 - h0, 4), npairs), dim3(64), 0, st, h0, h1, s, samples,
                           samples_iters, x1, x2, n_cap, models, nmodels, H);
        return;
    }
    int lanes = 64;
    while (lanes > 4 && (long)(h1 - h0) * npairs / lanes < 1024) lanes /= 2;
    static const int env_lanes = getenv("VO_SOLVE_LANES") ? atoi(getenv("VO_SOLVE_LANES")) : 0;  // tools/prof_solve.py
    if (env_lanes) lanes = env_lanes;
#define VO_SOLVE5(L)                                                                                           \
    hipLaunchKernelGGL(k_solve5<L>, dim3(cdiv(h1 - h0, L), npairs), dim3(L), 0, st, h0, h1, s, samples, samples_iters, \
                       x1, x2, n_cap, models, nmodels, H)
    switch (lanes) {
        case 1: VO_SOLVE5(1); break;
        case 2: VO_SOLVE5(2); break;
        case 4: VO_SOLVE5(4); break;
        case 8: VO_SOLVE5(8); break;
        case 16: VO_SOLVE5(16); break;
        case 32: VO_SOLVE5(32); break;
        default: VO_SOLVE5(64); break;
    }
#undef VO_SOLVE5
}

void launch_five_point_batch(hipStream_t st, int count, bool coop, const double* x1, const double* x2, double* models,
                             int* nmodels) {
    if (count <= 0) return;
    if (coop)
        hipLaunchKernelGGL(k_five_point_batch<true>, dim3(cdiv(count, 4)), dim3(64), 0, st, count, x1, x2, models,
                           nmodels);
    else
        hipLaunchKernelGGL(k_five_point_batch<false>, dim3(cdiv(count, 16)), dim3(64), 0, st, count, x1, x2, models,
                           nmodels);
}

void launch_score(hipStream_t st, int h0, int h1, const EstState* s, const double2* x1, const double2* x2, int n_cap,
                  const double* models, const int* nmodels, int* counts, int H, int npairs) {
    hipLaunchKernelGGL(k_score, dim3(cdiv(h1 - h0, 4), npairs), dim3(256), 0, st, h0, h1, s, x1, x2, n_cap, models,
                       nmodels, counts, H);
}

void launch_replay(hipStream_t st, int h0, int h1, EstState* s, const int* nmodels, const int* counts, int H,
                   double prob, int npairs) {
    hipLaunchKernelGGL(k_replay, dim3(npairs), dim3(64), 0, st, h0, h1, s, nmodels, counts, H, prob);
}

void launch_lmeds(hipStream_t st, EstState* s, const double2* x1, const double2* x2, int n_cap, const double* models,
                  const int* nmodels, double* medians, int H, int niters, int npairs) {
    if (npairs <= 4) {
        // a block per model: few pairs leave most of the chip idle, and the launch lasts one model's time
        const size_t lds = (size_t)(((n_cap + 3) & ~3) + 256) * sizeof(uint32_t);
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lmeds_medians_blk),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_lmeds_medians_blk, dim3(niters * 10, npairs), dim3(256), lds, st, s, x1, x2, n_cap,
                           models, nmodels, medians, H);
    } else {
        const size_t lds = (size_t)4 * (((n_cap + 3) & ~3) + 256) * sizeof(uint32_t);
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lmeds_medians),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_lmeds_medians, dim3(cdiv(niters * 10, 4), npairs), dim3(256), lds, st, s, x1, x2, n_cap,
                           models, nmodels, medians, H);
    }
    hipLaunchKernelGGL(k_lmeds_select, dim3(npairs), dim3(256), 0, st, s, nmodels, medians, H);
}

void launch_est_finish(hipStream_t st, EstState* s, const double2* x1, const double2* x2, int n_cap,
                       const double* models, uint8_t* mask, int H, int npairs) {
    hipLaunchKernelGGL(k_est_finish, dim3(npairs), dim3(256), 0, st, s, x1, x2, n_cap, models, mask, H);
}

void launch_pose(hipStream_t st, const double* E, size_t e_stride, PoseState* ps, const double2* x1,
                 const double2* x2, const int* d_n, int n_host, int n_cap, int n_grid, const uint8_t* mask_in,
                 double dist, double* q4, uint8_t* cmask, uint8_t* mask_out, double* pts4, int npairs) {
    hipLaunchKernelGGL(k_pose_setup, dim3(cdiv(npairs, 64)), dim3(64), 0, st, E, e_stride, ps, npairs);
    if (n_grid > 0)
        hipLaunchKernelGGL(k_pose_points, dim3(cdiv(n_grid, 256), npairs, 2), dim3(256), 0, st, ps, x1, x2, d_n, n_host,
                           n_cap, mask_in, dist, q4, cmask);
    hipLaunchKernelGGL(k_pose_select, dim3(std::max(cdiv(n_grid, 256), 1), npairs), dim3(256), 0, st, ps, d_n, n_host,
                       n_cap, q4, cmask, mask_out, pts4);
}

__global__ __launch_bounds__(256) void k_pack_results(PackArgs a, uint32_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int t = 0; t < VO_PACK_TABLES; t++) {
        const int j = i - a.at[t];
        if (j >= 0 && j < a.ndw[t]) out[i] = a.src[t][j];
    }
}

void launch_pack_results(hipStream_t st, const PackArgs& a, int total_dw, uint32_t* out) {
    if (total_dw <= 0) return;
    hipLaunchKernelGGL(k_pack_results, dim3(cdiv(total_dw, 256)), dim3(256), 0, st, a, out);
}

// cv::RNG (multiply-with-carry) + PointSetRegistrator::getSubset: the 5-subsets of iteration 0 .. iters-1 for
// every point count n.  The generator state is carried from one iteration to the next and a
// draw is repeated while it collides with an earlier pick, so the sequence of one n is sequential: one
// thread per n (8 449 threads of ~50 k integer operations at the defaults, a fraction of a millisecond).
__global__ __launch_bounds__(64) void k_fill_subsets(unsigned long long seed, int iters, int maxn,
                                                     uint16_t* __restrict__ out) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n > maxn) return;
    uint16_t* o = out + (size_t)n * iters * 5;
    if (n < 5) {
        for (int i = 0; i < iters * 5; i++) o[i] = 0;
        return;
    }
    unsigned long long state = seed ? seed : 0xffffffffull;
    for (int it = 0; it < iters; it++) {
        int idx[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            for (;;) {
                state = (unsigned long long)(uint32_t)state * 4164903690ull + (uint32_t)(state >> 32);
                const int v = (int)((uint32_t)state % (uint32_t)n);
                bool dup = false;
#pragma unroll
                for (int j = 0; j < 5; j++) dup |= j < i && idx[j] == v;
                idx[i] = v;
                if (!dup) break;
            }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) o[(size_t)it * 5 + i] = (uint16_t)idx[i];
    }
}

void launch_fill_subsets(hipStream_t st, uint64_t seed, int iters, int maxn, uint16_t* out) {
    hipLaunchKernelGGL(k_fill_subsets, dim3(cdiv(maxn + 1, 64)), dim3(64), 0, st, (unsigned long long)seed, iters, maxn,
                       out);
}

void launch_triangulate(hipStream_t st, const double* P, const float2* p0, const float2* p1, int n, double* pts3) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_triangulate, dim3(cdiv(n, 256)), dim3(256), 0, st, P, p0, p1, n, pts3);
}
