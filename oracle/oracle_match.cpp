// oracle_match.cpp — TEST INFRASTRUCTURE ONLY (see vo_oracle.h; parity unpinned).
// CPU restatement of BFMatcher(NORM_HAMMING).knnMatch(k=2) (features2d/src/matchers.cpp ->
// core/src/batch_distance.cpp) and of the reference's ratio/displacement
// filter (reference: src/matcher/matcher.cpp:17-38).
#include "oracle_common.h"

#include <climits>

extern "C" {

int vo_oracle_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes,
                           vo_dmatch* out, int* n_valid) {
    if (nq < 0 || nt < 0 || desc_bytes <= 0 || (nq > 0 && (!q || !out))) return VO_ERR_INVALID;
    const int K = 2;
    for (int i = 0; i < nq; i++) {
        int dist[K] = {INT_MAX, INT_MAX};
        int idx[K] = {-1, -1};
        const uint8_t* a = q + (size_t)i * desc_bytes;
        for (int j = 0; j < nt; j++) {
            const uint8_t* b = t + (size_t)j * desc_bytes;
            int d = 0;
            for (int k = 0; k < desc_bytes; k++) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
            // batchDistance K-nearest update: insert keeping earlier (lower-index) entries first
            if (d < dist[K - 1]) {
                int k;
                for (k = K - 2; k >= 0 && dist[k] > d; k--) {
                    dist[k + 1] = dist[k];
                    idx[k + 1] = idx[k];
                }
                dist[k + 1] = d;
                idx[k + 1] = j;
            }
        }
        int nv = 0;
        for (int k = 0; k < K; k++) {
            vo_dmatch m;
            m.queryIdx = i;
            m.trainIdx = idx[k];
            m.imgIdx = 0;
            m.distance = idx[k] >= 0 ? (float)dist[k] : 3.402823466e+38F;
            out[(size_t)i * K + k] = m;
            if (idx[k] >= 0) nv++;
        }
        if (n_valid) n_valid[i] = nv;
    }
    return VO_OK;
}

int vo_oracle_match_filter(const vo_dmatch* knn, const int* n_valid, int nq, const vo_keypoint* kq,
                           const vo_keypoint* kt, float ratio, float max_disp, vo_dmatch* good,
                           int* n_good) {
    if (nq < 0 || !n_good || (nq > 0 && (!knn || !kq || !kt || !good))) return VO_ERR_INVALID;
    int n = 0;
    for (int k = 0; k < nq; k++) {
        // the reference indexes matches[k][1] unconditionally (matcher.cpp:25); rows with fewer
        // than two neighbours are undefined behaviour there and are skipped here.
        if (n_valid && n_valid[k] < 2) continue;
        const vo_dmatch& m0 = knn[(size_t)k * 2];
        const vo_dmatch& m1 = knn[(size_t)k * 2 + 1];
        if (m0.distance < ratio * m1.distance) {
            const vo_keypoint& p0 = kq[m0.queryIdx];
            const vo_keypoint& p1 = kt[m0.trainIdx];
            // cv::norm(Mat(p0.pt) - Mat(p1.pt)): float difference, double L2 accumulation
            float dx = p0.x - p1.x, dy = p0.y - p1.y;
            double nrm = std::sqrt((double)dx * (double)dx + (double)dy * (double)dy);
            if (nrm < (double)max_disp) good[n++] = m0;
        }
    }
    *n_good = n;
    return VO_OK;
}

}  // extern "C"
