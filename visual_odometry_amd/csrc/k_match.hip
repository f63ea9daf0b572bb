// k_match.hip — brute-force Hamming 2-NN (BFMatcher(NORM_HAMMING).knnMatch(k=2)) and the
// reference's ratio / displacement filter, for gfx950.
//
// k_knn2_partial: one query per lane held in registers (4 or 8 x u64), the train range of the
//   block tiled through LDS (every lane reads the same train row -> LDS broadcast reads),
//   64-bit xor + popcount, running (best, second) kept as packed keys dist << 23 | index so that
//   min / max on the key is the (distance, lower-index-first) order cv's insertion rule gives.
// k_knn2_merge: lexicographic merge of the per-split partial top-2 lists.
// k_match_filter: ratio test + displacement gate + order-preserving compaction (block scan).
// Reference: src/matcher/matcher.cpp:17-38; SURVEY.md A.3.
#include "kernels.h"

#define KNN_TILE 256  // train descriptors per LDS tile

// W32: uint32 words per descriptor (8 or 16); Q: queries per lane.  A block of 256 threads owns
// 256 * Q queries (query i of a lane is q0 + i * 256 + thread) and scans rows [tbeg, tend) of the
// train set.  The tile for the next step is fetched into registers while the current one is
// scanned (two LDS buffers, one barrier per tile).  Per (query, row): W32 xor + W32 bcnt (the
// popcount instruction accumulates) + key pack + min / max / min.
template <int W32, int Q>
__global__ __launch_bounds__(256) void k_knn2_partial(KnnArgs a) {
    constexpr int V = W32 / 4;                   // uint4 per descriptor
    constexpr int LD = KNN_TILE * V / 256;       // uint4 per thread per tile
    __shared__ uint4 tile[2][KNN_TILE * V];
    const int pair = blockIdx.z;
    const int nq = a.d_n ? a.d_n[pair] : a.nq;
    const int nt = a.d_n ? a.d_n[pair + 1] : a.nt;
    const int q0 = blockIdx.x * (256 * Q);
    if (q0 >= nq) return;
    const uint4* Qp = reinterpret_cast<const uint4*>(a.q + pair * a.q_stride);
    const uint4* T = reinterpret_cast<const uint4*>(a.t + pair * a.t_stride);

    uint32_t qv[Q][W32];
#pragma unroll
    for (int i = 0; i < Q; i++) {
        const int qr = min(q0 + i * 256 + (int)threadIdx.x, nq - 1);
#pragma unroll
        for (int k = 0; k < V; k++) {
            const uint4 v = Qp[(size_t)qr * V + k];
            qv[i][4 * k] = v.x;
            qv[i][4 * k + 1] = v.y;
            qv[i][4 * k + 2] = v.z;
            qv[i][4 * k + 3] = v.w;
        }
    }
    const int chunk = cdiv(nt, a.splits);
    const int tbeg = blockIdx.y * chunk;
    const int tend = min(nt, tbeg + chunk);
    uint32_t k1[Q], k2[Q];
#pragma unroll
    for (int i = 0; i < Q; i++) k1[i] = k2[i] = VO_KEY_INVALID;

    uint4 nxt[LD];
    auto fetch = [&](int base) {
        const int cnt = min(KNN_TILE, tend - base) * V;
#pragma unroll
        for (int r = 0; r < LD; r++) {
            const int i = r * 256 + threadIdx.x;
            nxt[r] = i < cnt ? T[(size_t)base * V + i] : make_uint4(0, 0, 0, 0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int r = 0; r < LD; r++) tile[buf][r * 256 + threadIdx.x] = nxt[r];
    };
    int buf = 0;
    if (tbeg < tend) {
        fetch(tbeg);
        stash(0);
    }
    __syncthreads();
    for (int base = tbeg; base < tend; base += KNN_TILE) {
        const int n = min(KNN_TILE, tend - base);
        const bool more = base + KNN_TILE < tend;
        if (more) fetch(base + KNN_TILE);
        const uint4* tl = tile[buf];
#pragma unroll 4
        for (int j = 0; j < n; j++) {
            uint32_t tr[W32];
#pragma unroll
            for (int k = 0; k < V; k++) {
                const uint4 v = tl[j * V + k];
                tr[4 * k] = v.x;
                tr[4 * k + 1] = v.y;
                tr[4 * k + 2] = v.z;
                tr[4 * k + 3] = v.w;
            }
            const uint32_t idx = (uint32_t)(base + j);
#pragma unroll
            for (int i = 0; i < Q; i++) {
                uint32_t d = 0;
#pragma unroll
                for (int k = 0; k < W32; k++) d += __builtin_popcount(qv[i][k] ^ tr[k]);
                const uint32_t key = (d << VO_KEY_SHIFT) | idx;
                const uint32_t hi = max(key, k1[i]);
                k1[i] = min(key, k1[i]);
                k2[i] = min(k2[i], hi);
            }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < Q; i++) {
        const int qi = q0 + i * 256 + threadIdx.x;
        if (qi < nq) a.part[((size_t)pair * a.splits + blockIdx.y) * a.nq_cap + qi] = make_uint2(k1[i], k2[i]);
    }
}

__global__ __launch_bounds__(256) void k_knn2_merge(KnnArgs a) {
    const int pair = blockIdx.y;
    const int nq = a.d_n ? a.d_n[pair] : a.nq;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t k1 = VO_KEY_INVALID, k2 = VO_KEY_INVALID;
    for (int s = 0; s < a.splits; s++) {
        const uint2 p = a.part[((size_t)pair * a.splits + s) * a.nq_cap + qi];
        uint32_t lo = min(p.x, k1), hi = max(p.x, k1);
        k2 = min(k2, hi);
        k1 = lo;
        lo = min(p.y, k1);
        hi = max(p.y, k1);
        k2 = min(k2, hi);
        k1 = lo;
    }
    a.knn[(size_t)pair * a.nq_cap + qi] = make_uint2(k1, k2);
}

void launch_knn2_partial(hipStream_t st, const KnnArgs& a) {
    if (a.nq <= 0 || a.npairs <= 0) return;
    dim3 grid(cdiv(a.nq, 256 * a.qpl), a.splits, a.npairs);
#define VO_KNN(W, QQ) hipLaunchKernelGGL((k_knn2_partial<W, QQ>), grid, dim3(256), 0, st, a)
    if (a.words == 8) {
        if (a.qpl == 4) VO_KNN(8, 4);
        else if (a.qpl == 2) VO_KNN(8, 2);
        else VO_KNN(8, 1);
    } else {
        if (a.qpl >= 2) VO_KNN(16, 2);
        else VO_KNN(16, 1);
    }
#undef VO_KNN
}

void launch_knn2_merge(hipStream_t st, const KnnArgs& a) {
    if (a.nq <= 0 || a.npairs <= 0) return;
    hipLaunchKernelGGL(k_knn2_merge, dim3(cdiv(a.nq, 256), a.npairs), dim3(256), 0, st, a);
}

// One block per pair; queries processed in chunks of 1024 in index order.
__global__ __launch_bounds__(1024) void k_match_filter(FilterArgs a) {
    __shared__ int wave_cnt[16];
    __shared__ int running;
    const int pair = blockIdx.x;
    const int nq = a.d_n ? a.d_n[pair] : a.nq;
    const uint2* knn = a.knn + (size_t)pair * a.nq_cap;
    const vo_keypoint* kq = a.kq + pair * a.kp_stride;
    const vo_keypoint* kt = a.kt + pair * a.kp_stride;
    GoodMatch* good = a.good + (size_t)pair * a.nq_cap;
    float2* p0 = a.p0 + (size_t)pair * a.nq_cap;
    float2* p1 = a.p1 + (size_t)pair * a.nq_cap;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024) {
        const int qi = q0 + threadIdx.x;
        bool keep = false;
        GoodMatch gm = {0, 0, 0, 0};
        float2 a0 = make_float2(0.f, 0.f), a1 = a0;
        if (qi < nq) {
            const uint2 k = knn[qi];
            // rows with fewer than two neighbours: undefined behaviour in the reference, skipped
            if (k.x != VO_KEY_INVALID && k.y != VO_KEY_INVALID) {
                const float d0 = (float)(k.x >> VO_KEY_SHIFT), d1 = (float)(k.y >> VO_KEY_SHIFT);
                if (d0 < a.ratio * d1) {
                    const int ti = (int)(k.x & ((1u << VO_KEY_SHIFT) - 1u));
                    a0 = make_float2(kq[qi].x, kq[qi].y);
                    a1 = make_float2(kt[ti].x, kt[ti].y);
                    const float dx = a0.x - a1.x, dy = a0.y - a1.y;
                    const double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
                    if (nrm < (double)a.max_disp) {
                        keep = true;
                        gm.q = qi;
                        gm.t = ti;
                        gm.dist = (int)(k.x >> VO_KEY_SHIFT);
                    }
                }
            }
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int before = running;
        for (int i = 0; i < wv; i++) before += wave_cnt[i];
        if (keep) {
            const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
            good[pos] = gm;
            p0[pos] = a0;
            p1[pos] = a1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = running;
            for (int i = 0; i < 16; i++) s += wave_cnt[i];
            running = s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.ngood[pair] = running;
}

void launch_match_filter(hipStream_t st, const FilterArgs& a) {
    if (a.npairs <= 0) return;
    hipLaunchKernelGGL(k_match_filter, dim3(a.npairs), dim3(1024), 0, st, a);
}
