// vo_match.cpp — host side of the matcher entry points (knnMatch(k=2) + ratio/displacement
// filter).  Replaces the cv calls in matcher() (reference: src/matcher/matcher.cpp:10,17,23-38).
#include <algorithm>
#include <cstdlib>

#include "ctx.h"

// Blocks of 64 queries x 4 waves; 1536 blocks put six waves on every SIMD of the chip, which is the
// batched path's shape (48 pairs x 2000 queries) with no split at all.  A single small problem is cut
// along the train range until the chip is about that full (every wave keeps >= 64 rows) and merged by
// k_knn2_merge.  The matrix-core kernel (two waves per SIMD, a head of ~64 operand dwords per block) wants
// fewer, longer blocks: about 512 and at most 8 splits (tools/knn_sweep.py: 8192 x 8192 26.7 -> 21.0 us,
// 4096 13.9 -> 11.6, 2000 9.3 -> 8.8 against the popcount kernel's rule).
int knn_splits(int nq, int nt, int npairs, int dirs, bool mfma) {
    const int blocks = cdiv(std::max(nq, 1), 64) * std::max(npairs, 1) * dirs;
    int s = ((mfma ? 512 : 1536) + blocks / 2) / blocks;
    s = std::min(s, std::max(nt, 1) / 256);
    return std::max(1, std::min(s, mfma ? 8 : 32));
}

int match_knn2_dev(vo_ctx* c, const uint32_t* d_q, const uint32_t* d_t, size_t q_stride, size_t t_stride,
                   const int* d_n, int nq, int nt, int words, int npairs, int nq_cap, bool cross) {
    KnnArgs a;
    a.q = d_q;
    a.t = d_t;
    a.q_stride = q_stride;
    a.t_stride = t_stride;
    a.d_n = d_n;
    a.nq = nq;
    a.nt = nt;
    a.words = words;
    a.cross = cross ? 1 : 0;
    const int dirs = cross ? 2 : 1;
    const int env_s = c->opt_knn_splits;  // tuning hook for tools/bench_knn.py
    a.splits = knn_splits(cross ? std::max(nq, nt) : nq, cross ? std::min(nq, nt) : nt, npairs, dirs,
                          words == 8 && c->opt_knn_mfma);
    if (env_s) a.splits = std::max(1, std::min(env_s, std::max(nt, 1) / 64));
    a.nq_cap = nq_cap;
    a.npairs = npairs;
    a.mfma = c->opt_knn_mfma;
    if (a.splits > 1) VO_TRY(vo_ensure(c, c->d_part, (size_t)npairs * dirs * a.splits * nq_cap * sizeof(uint2)));
    VO_TRY(vo_ensure(c, c->d_knn, (size_t)npairs * nq_cap * sizeof(uint2)));
    if (cross) VO_TRY(vo_ensure(c, c->d_rev, (size_t)npairs * nq_cap * sizeof(uint2)));
    a.part = c->d_part.as<uint2>();
    a.knn = c->d_knn.as<uint2>();
    a.rev = cross ? c->d_rev.as<uint2>() : nullptr;
    {
        ProfScope ps(c, VO_PROF_KNN);
        launch_knn2(c->stream, a);
    }
    launch_knn2_merge(c->stream, a);
    return vo_check_launch(c, "knn2");
}

int match_filter_dev(vo_ctx* c, const vo_keypoint* d_kq, const vo_keypoint* d_kt, size_t kp_stride,
                     const int* d_n, int nq, int npairs, int nq_cap, float ratio, float max_disp, bool cross) {
    VO_TRY(vo_ensure(c, c->d_good, (size_t)npairs * nq_cap * sizeof(GoodMatch)));
    VO_TRY(vo_ensure(c, c->d_p0, (size_t)npairs * nq_cap * sizeof(float2)));
    VO_TRY(vo_ensure(c, c->d_p1, (size_t)npairs * nq_cap * sizeof(float2)));
    VO_TRY(vo_ensure(c, c->d_ngood, (size_t)(npairs + 1) * sizeof(int)));
    FilterArgs f;
    f.knn = c->d_knn.as<uint2>();
    f.rev = cross ? c->d_rev.as<uint2>() : nullptr;
    f.kq = d_kq;
    f.kt = d_kt;
    f.kp_stride = kp_stride;
    f.d_n = d_n;
    f.nq = nq;
    f.nq_cap = nq_cap;
    f.ratio = ratio;
    f.max_disp = max_disp;
    f.good = c->d_good.as<GoodMatch>();
    f.p0 = c->d_p0.as<float2>();
    f.p1 = c->d_p1.as<float2>();
    f.ngood = c->d_ngood.as<int>();
    f.npairs = npairs;
    launch_match_filter(c->stream, f);
    return vo_check_launch(c, "match filter");
}

// knnMatch(k = 2) of one pair from host memory; rev_nn != nullptr also returns, for every train row,
// the index of its nearest query (same tie rule; -1 if there is no query).
static int knn2_host(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes, vo_dmatch* out,
                     int* n_valid, int* rev_nn, const char* who) {
    if (!c) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || desc_bytes <= 0 || desc_bytes > 64 || (nq > 0 && (!q || !out)) || (nt > 0 && !t))
        return c->fail(VO_ERR_INVALID, "%s: bad argument (descriptor size must be 1..64 bytes)", who);
    if (nq > 2 * c->lim.max_features || nt > 2 * c->lim.max_features)
        return c->fail(VO_ERR_CAPACITY, "%s: %d x %d exceeds 2*max_features = %d", who, nq, nt,
                       2 * c->lim.max_features);
    VO_HIP(c, hipSetDevice(c->device));
    const bool cross = rev_nn != nullptr;
    if (cross)
        for (int i = 0; i < nt; i++) rev_nn[i] = -1;
    if (nq == 0) return VO_OK;
    const int row = desc_bytes <= 32 ? 32 : 64;  // rows zero-padded to 32 or 64 bytes
    const int words = row / 4;
    const int cap = cross ? std::max(nq, nt) : nq;
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(vo_ensure(c, c->d_q, (size_t)nq * row));
    VO_TRY(vo_ensure(c, c->d_t, (size_t)std::max(nt, 1) * row));
    if (desc_bytes == row) {
        VO_TRY(vo_xfer_in(c, c->d_q.p, q, (size_t)nq * row));
        if (nt) VO_TRY(vo_xfer_in(c, c->d_t.p, t, (size_t)nt * row));
    } else {
        VO_HIP(c, hipMemsetAsync(c->d_q.p, 0, (size_t)nq * row, c->stream));
        VO_HIP(c, hipMemcpy2DAsync(c->d_q.p, row, q, desc_bytes, desc_bytes, nq, hipMemcpyHostToDevice, c->stream));
        if (nt) {
            VO_HIP(c, hipMemsetAsync(c->d_t.p, 0, (size_t)nt * row, c->stream));
            VO_HIP(c, hipMemcpy2DAsync(c->d_t.p, row, t, desc_bytes, desc_bytes, nt, hipMemcpyHostToDevice, c->stream));
        }
    }
    VO_TRY(match_knn2_dev(c, c->d_q.as<uint32_t>(), c->d_t.as<uint32_t>(), 0, 0, nullptr, nq, nt, words, 1, cap,
                          cross));
    VO_TRY(vo_ensure_pinned(c, c->h_knn, (size_t)(nq + (cross ? nt : 0)) * sizeof(uint2)));
    VO_HIP(c, hipMemcpyAsync(c->h_knn.p, c->d_knn.p, (size_t)nq * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
    if (cross && nt)
        VO_HIP(c, hipMemcpyAsync(c->h_knn.as<uint2>() + nq, c->d_rev.p, (size_t)nt * sizeof(uint2),
                                 hipMemcpyDeviceToHost, c->stream));
    VO_TRY(vo_xfer_finish(c));
    const uint2* k = c->h_knn.as<uint2>();
    for (int i = 0; i < nq; i++) {
        const uint32_t key[2] = {k[i].x, k[i].y};
        int nv = 0;
        for (int j = 0; j < 2; j++) {
            vo_dmatch m;
            m.queryIdx = i;
            m.imgIdx = 0;
            if (key[j] != VO_KEY_INVALID) {
                m.trainIdx = (int)(key[j] & ((1u << VO_KEY_SHIFT) - 1u));
                m.distance = (float)(key[j] >> VO_KEY_SHIFT);
                nv++;
            } else {
                m.trainIdx = -1;
                m.distance = 3.402823466e+38F;
            }
            out[(size_t)i * 2 + j] = m;
        }
        if (n_valid) n_valid[i] = nv;
    }
    if (cross)
        for (int i = 0; i < nt; i++)
            rev_nn[i] = k[nq + i].x != VO_KEY_INVALID ? (int)(k[nq + i].x & ((1u << VO_KEY_SHIFT) - 1u)) : -1;
    return VO_OK;
}

static int filter_host(vo_ctx* c, const vo_dmatch* knn, const int* n_valid, int nq, const int* rev_nn, int nt,
                       const vo_keypoint* kq, const vo_keypoint* kt, float ratio, float max_disp, vo_dmatch* good,
                       int* n_good, const char* who) {
    if (!c) return VO_ERR_INVALID;
    if (nq < 0 || !n_good || (nq > 0 && (!knn || !kq || !kt || !good)) || (rev_nn && nt < 0))
        return c->fail(VO_ERR_INVALID, "%s: bad argument", who);
    if (nq > 2 * c->lim.max_features || (rev_nn && nt > 2 * c->lim.max_features))
        return c->fail(VO_ERR_CAPACITY, "%s: %d rows exceed 2*max_features", who, std::max(nq, nt));
    VO_HIP(c, hipSetDevice(c->device));
    *n_good = 0;
    if (nq == 0) return VO_OK;
    const bool cross = rev_nn != nullptr;
    // pack the neighbours into keys; the train keypoint array length is not part of the reference
    // interface, so it is taken as max train index + 1 (nt with cross-check)
    VO_TRY(vo_ensure_pinned(c, c->h_knn, (size_t)(nq + (cross ? nt : 0)) * sizeof(uint2)));
    uint2* k = c->h_knn.as<uint2>();
    int max_t = -1;
    for (int i = 0; i < nq; i++) {
        const int nv = n_valid ? n_valid[i] : 2;
        uint32_t key[2] = {VO_KEY_INVALID, VO_KEY_INVALID};
        for (int j = 0; j < 2 && j < nv; j++) {
            const vo_dmatch& m = knn[(size_t)i * 2 + j];
            if (m.trainIdx < 0) continue;
            if (m.trainIdx >= (1 << VO_KEY_SHIFT) || m.distance < 0 || m.distance > 512 ||
                (cross && m.trainIdx >= nt))
                return c->fail(VO_ERR_INVALID, "%s: row %d is not a Hamming match", who, i);
            key[j] = ((uint32_t)m.distance << VO_KEY_SHIFT) | (uint32_t)m.trainIdx;
            if (j == 0) {
                max_t = std::max(max_t, m.trainIdx);
                if (m.queryIdx != i) return c->fail(VO_ERR_INVALID, "%s: queryIdx != row index", who);
            }
        }
        k[i] = make_uint2(key[0], key[1]);
    }
    for (int i = 0; cross && i < nt; i++)
        k[nq + i] = make_uint2(rev_nn[i] >= 0 ? (uint32_t)rev_nn[i] : VO_KEY_INVALID, VO_KEY_INVALID);
    const int nkq = nq, nkt = cross ? nt : max_t + 1;
    const int cap = cross ? std::max(nq, nt) : nq;
    VO_TRY(vo_ensure(c, c->d_knn, (size_t)cap * sizeof(uint2)));
    if (cross) VO_TRY(vo_ensure(c, c->d_rev, (size_t)cap * sizeof(uint2)));
    VO_TRY(vo_ensure(c, c->d_kq, (size_t)nkq * sizeof(vo_keypoint)));
    VO_TRY(vo_ensure(c, c->d_kt, (size_t)std::max(nkt, 1) * sizeof(vo_keypoint)));
    VO_TRY(vo_xfer_begin(c));
    VO_HIP(c, hipMemcpyAsync(c->d_knn.p, k, (size_t)nq * sizeof(uint2), hipMemcpyHostToDevice, c->stream));  // pinned
    if (cross && nt)
        VO_HIP(c, hipMemcpyAsync(c->d_rev.p, k + nq, (size_t)nt * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
    VO_TRY(vo_xfer_in(c, c->d_kq.p, kq, (size_t)nkq * sizeof(vo_keypoint)));
    if (nkt > 0) VO_TRY(vo_xfer_in(c, c->d_kt.p, kt, (size_t)nkt * sizeof(vo_keypoint)));
    VO_TRY(match_filter_dev(c, c->d_kq.as<vo_keypoint>(), c->d_kt.as<vo_keypoint>(), 0, nullptr, nq, 1, cap, ratio,
                            max_disp, cross));
    VO_TRY(vo_ensure_pinned(c, c->h_good, (size_t)nq * sizeof(GoodMatch) + sizeof(int)));
    int* h_n = reinterpret_cast<int*>(c->h_good.as<char>() + (size_t)nq * sizeof(GoodMatch));
    VO_HIP(c, hipMemcpyAsync(h_n, c->d_ngood.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipMemcpyAsync(c->h_good.p, c->d_good.p, (size_t)nq * sizeof(GoodMatch), hipMemcpyDeviceToHost,
                             c->stream));
    VO_TRY(vo_xfer_finish(c));
    memcpy(good, c->h_good.p, (size_t)*h_n * sizeof(GoodMatch));
    *n_good = *h_n;
    return VO_OK;
}

extern "C" {

int vo_knn2_hamming(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes,
                    vo_dmatch* out, int* n_valid) {
    return knn2_host(c, q, nq, t, nt, desc_bytes, out, n_valid, nullptr, "vo_knn2_hamming");
}

int vo_knn2_hamming_cross(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes,
                          vo_dmatch* out, int* n_valid, int* rev_nn) {
    if (c && !rev_nn && nt > 0) return c->fail(VO_ERR_INVALID, "vo_knn2_hamming_cross: rev_nn is NULL");
    int none = -1;
    return knn2_host(c, q, nq, t, nt, desc_bytes, out, n_valid, rev_nn ? rev_nn : &none, "vo_knn2_hamming_cross");
}

int vo_match_filter(vo_ctx* c, const vo_dmatch* knn, const int* n_valid, int nq, const vo_keypoint* kq,
                    const vo_keypoint* kt, float ratio, float max_disp, vo_dmatch* good, int* n_good) {
    return filter_host(c, knn, n_valid, nq, nullptr, 0, kq, kt, ratio, max_disp, good, n_good, "vo_match_filter");
}

int vo_match_filter_cross(vo_ctx* c, const vo_dmatch* knn, const int* n_valid, int nq, const int* rev_nn, int nt,
                          const vo_keypoint* kq, const vo_keypoint* kt, float ratio, float max_disp,
                          vo_dmatch* good, int* n_good) {
    if (c && !rev_nn) return c->fail(VO_ERR_INVALID, "vo_match_filter_cross: rev_nn is NULL");
    return filter_host(c, knn, n_valid, nq, rev_nn, nt, kq, kt, ratio, max_disp, good, n_good,
                       "vo_match_filter_cross");
}

}  // extern "C"
