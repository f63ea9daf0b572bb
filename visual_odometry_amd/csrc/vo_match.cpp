// vo_match.cpp — host side of the matcher entry points (knnMatch(k=2) + ratio/displacement
// filter).  Replaces the cv calls in matcher() (reference: src/matcher/matcher.cpp:10,17,23-38).
#include <algorithm>
#include <cstdlib>

#include "ctx.h"

// Split count for the train range: many short blocks balance best on 256 CUs (measured on MI355X,
// gpurun_out/knn_sweep.log: 32 x 2k pairs 82.8 us at 1024 blocks, 75.3 us at 8192), chunks of >= 64 rows.
static int knn_splits(int nq, int nt, int npairs, int qpl) {
    const int qblocks = cdiv(std::max(nq, 1), 256 * qpl) * std::max(npairs, 1);
    int s = cdiv(8192, qblocks);
    s = std::min(s, cdiv(std::max(nt, 1), 64));
    return std::max(1, std::min(s, 32));
}

int match_knn2_dev(vo_ctx* c, const uint32_t* d_q, const uint32_t* d_t, size_t q_stride, size_t t_stride,
                   const int* d_n, int nq, int nt, int words, int npairs, int nq_cap) {
    KnnArgs a;
    a.q = d_q;
    a.t = d_t;
    a.q_stride = q_stride;
    a.t_stride = t_stride;
    a.d_n = d_n;
    a.nq = nq;
    a.nt = nt;
    a.words = words;
    // tuning hooks for tools/bench_knn.py: queries per lane (1 measured best) and split count
    static const int env_q = getenv("VO_KNN_Q") ? atoi(getenv("VO_KNN_Q")) : 0;
    static const int env_s = getenv("VO_KNN_SPLITS") ? atoi(getenv("VO_KNN_SPLITS")) : 0;
    a.qpl = env_q == 2 || (env_q == 4 && words == 8) ? env_q : 1;
    a.splits = knn_splits(nq, nt, npairs, a.qpl);
    if (env_s) a.splits = std::max(1, std::min(env_s, cdiv(std::max(nt, 1), 64)));
    a.nq_cap = nq_cap;
    a.npairs = npairs;
    VO_TRY(vo_ensure(c, c->d_part, (size_t)npairs * a.splits * nq_cap * sizeof(uint2)));
    VO_TRY(vo_ensure(c, c->d_knn, (size_t)npairs * nq_cap * sizeof(uint2)));
    a.part = c->d_part.as<uint2>();
    a.knn = c->d_knn.as<uint2>();
    {
        ProfScope ps(c, VO_PROF_KNN);
        launch_knn2_partial(c->stream, a);
    }
    launch_knn2_merge(c->stream, a);
    return vo_check_launch(c, "knn2");
}

int match_filter_dev(vo_ctx* c, const vo_keypoint* d_kq, const vo_keypoint* d_kt, size_t kp_stride,
                     const int* d_n, int nq, int npairs, int nq_cap, float ratio, float max_disp) {
    VO_TRY(vo_ensure(c, c->d_good, (size_t)npairs * nq_cap * sizeof(GoodMatch)));
    VO_TRY(vo_ensure(c, c->d_p0, (size_t)npairs * nq_cap * sizeof(float2)));
    VO_TRY(vo_ensure(c, c->d_p1, (size_t)npairs * nq_cap * sizeof(float2)));
    VO_TRY(vo_ensure(c, c->d_ngood, (size_t)(npairs + 1) * sizeof(int)));
    FilterArgs f;
    f.knn = c->d_knn.as<uint2>();
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

extern "C" {

int vo_knn2_hamming(vo_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes,
                    vo_dmatch* out, int* n_valid) {
    if (!c) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || desc_bytes <= 0 || desc_bytes > 64 || (nq > 0 && (!q || !out)) || (nt > 0 && !t))
        return c->fail(VO_ERR_INVALID, "vo_knn2_hamming: bad argument (descriptor size must be 1..64 bytes)");
    if (nq > 2 * c->lim.max_features || nt > 2 * c->lim.max_features)
        return c->fail(VO_ERR_CAPACITY, "vo_knn2_hamming: %d x %d exceeds 2*max_features = %d", nq, nt,
                       2 * c->lim.max_features);
    VO_HIP(c, hipSetDevice(c->device));
    if (nq == 0) return VO_OK;
    const int row = desc_bytes <= 32 ? 32 : 64;  // rows zero-padded to 32 or 64 bytes
    const int words = row / 4;
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
    VO_TRY(match_knn2_dev(c, c->d_q.as<uint32_t>(), c->d_t.as<uint32_t>(), 0, 0, nullptr, nq, nt, words, 1, nq));
    VO_TRY(vo_ensure_pinned(c, c->h_knn, (size_t)nq * sizeof(uint2)));
    VO_HIP(c, hipMemcpyAsync(c->h_knn.p, c->d_knn.p, (size_t)nq * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
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
    return VO_OK;
}

int vo_match_filter(vo_ctx* c, const vo_dmatch* knn, const int* n_valid, int nq, const vo_keypoint* kq,
                    const vo_keypoint* kt, float ratio, float max_disp, vo_dmatch* good, int* n_good) {
    if (!c) return VO_ERR_INVALID;
    if (nq < 0 || !n_good || (nq > 0 && (!knn || !kq || !kt || !good)))
        return c->fail(VO_ERR_INVALID, "vo_match_filter: bad argument");
    if (nq > 2 * c->lim.max_features)
        return c->fail(VO_ERR_CAPACITY, "vo_match_filter: %d rows exceed 2*max_features", nq);
    VO_HIP(c, hipSetDevice(c->device));
    *n_good = 0;
    if (nq == 0) return VO_OK;
    // pack the neighbours into keys; the train keypoint array length is not part of the reference
    // interface, so it is taken as max train index + 1
    VO_TRY(vo_ensure_pinned(c, c->h_knn, (size_t)nq * sizeof(uint2)));
    uint2* k = c->h_knn.as<uint2>();
    int max_t = -1, max_q = -1;
    for (int i = 0; i < nq; i++) {
        const int nv = n_valid ? n_valid[i] : 2;
        uint32_t key[2] = {VO_KEY_INVALID, VO_KEY_INVALID};
        for (int j = 0; j < 2 && j < nv; j++) {
            const vo_dmatch& m = knn[(size_t)i * 2 + j];
            if (m.trainIdx < 0) continue;
            if (m.trainIdx >= (1 << VO_KEY_SHIFT) || m.distance < 0 || m.distance > 511)
                return c->fail(VO_ERR_INVALID, "vo_match_filter: row %d is not a Hamming match", i);
            key[j] = ((uint32_t)m.distance << VO_KEY_SHIFT) | (uint32_t)m.trainIdx;
            if (j == 0) {
                max_t = std::max(max_t, m.trainIdx);
                if (m.queryIdx != i) return c->fail(VO_ERR_INVALID, "vo_match_filter: queryIdx != row index");
                max_q = std::max(max_q, m.queryIdx);
            }
        }
        k[i] = make_uint2(key[0], key[1]);
    }
    const int nkq = nq, nkt = max_t + 1;
    VO_TRY(vo_ensure(c, c->d_knn, (size_t)nq * sizeof(uint2)));
    VO_TRY(vo_ensure(c, c->d_kq, (size_t)nkq * sizeof(vo_keypoint)));
    VO_TRY(vo_ensure(c, c->d_kt, (size_t)std::max(nkt, 1) * sizeof(vo_keypoint)));
    VO_TRY(vo_xfer_begin(c));
    VO_HIP(c, hipMemcpyAsync(c->d_knn.p, k, (size_t)nq * sizeof(uint2), hipMemcpyHostToDevice, c->stream));  // pinned
    VO_TRY(vo_xfer_in(c, c->d_kq.p, kq, (size_t)nkq * sizeof(vo_keypoint)));
    if (nkt > 0) VO_TRY(vo_xfer_in(c, c->d_kt.p, kt, (size_t)nkt * sizeof(vo_keypoint)));
    VO_TRY(match_filter_dev(c, c->d_kq.as<vo_keypoint>(), c->d_kt.as<vo_keypoint>(), 0, nullptr, nq, 1, nq, ratio,
                            max_disp));
    VO_TRY(vo_ensure_pinned(c, c->h_good, (size_t)nq * sizeof(GoodMatch) + sizeof(int)));
    int* h_n = reinterpret_cast<int*>(c->h_good.as<char>() + (size_t)nq * sizeof(GoodMatch));
    VO_HIP(c, hipMemcpyAsync(h_n, c->d_ngood.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipMemcpyAsync(c->h_good.p, c->d_good.p, (size_t)nq * sizeof(GoodMatch), hipMemcpyDeviceToHost,
                             c->stream));
    VO_TRY(vo_xfer_finish(c));
    const GoodMatch* g = c->h_good.as<GoodMatch>();
    for (int i = 0; i < *h_n; i++) {
        good[i].queryIdx = g[i].q;
        good[i].trainIdx = g[i].t;
        good[i].imgIdx = 0;
        good[i].distance = (float)g[i].dist;
    }
    *n_good = *h_n;
    return VO_OK;
}

}  // extern "C"
