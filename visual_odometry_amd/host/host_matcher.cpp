// host_matcher.cpp — matcher() and createMatchMatrix() of the host mirror (reference: src/matcher/matcher.cpp)
// over vo_knn2_hamming / vo_match_filter / vo_find_essential.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>
#include <unordered_map>

#include "host_internal.h"
#include "matcher.h"

// ---- matcher ----------------------------------------------------------------------------------
static const float kMatchRatio = 0.7f;

std::vector<vo::MatchesInfo> matcher(const std::vector<vo::ImageFeatures>& features, const vo::Matd& K) {
    vo_ctx* ctx = vo::default_ctx();
    std::vector<vo::MatchesInfo> pairwise_matches;
    const int nimg = (int)features.size();
    for (int i = 0; i + 1 < nimg; i++) {
        for (int j = i + 1; j < std::min(nimg, i + 3); j++) {
            const auto& fi = features[i];
            const auto& fj = features[j];
            const int nq = fi.descriptors.rows, nt = fj.descriptors.rows;
            if (nq == 0) continue;
            std::vector<vo_dmatch> knn((size_t)nq * 2);
            std::vector<int> nvalid(nq);
            check(ctx, vo_knn2_hamming(ctx, fi.descriptors.d.data(), nq, fj.descriptors.d.data(), nt,
                                       fi.descriptors.cols, knn.data(), nvalid.data()));
            std::vector<vo_dmatch> good(nq);
            int ngood = 0;
            check(ctx, vo_match_filter(ctx, knn.data(), nvalid.data(), nq,
                                       reinterpret_cast<const vo_keypoint*>(fi.keypoints.data()),
                                       reinterpret_cast<const vo_keypoint*>(fj.keypoints.data()), kMatchRatio, 200.f,
                                       good.data(), &ngood));
            if (ngood < 5) continue;
            vo::MatchesInfo info;
            std::vector<float> p0((size_t)ngood * 2), p1((size_t)ngood * 2);
            for (int k = 0; k < ngood; k++) {
                vo::DMatch m;
                m.queryIdx = good[k].queryIdx;
                m.trainIdx = good[k].trainIdx;
                m.imgIdx = good[k].imgIdx;
                m.distance = good[k].distance;
                info.matches.push_back(m);
                p0[2 * k] = fi.keypoints[m.queryIdx].pt.x;
                p0[2 * k + 1] = fi.keypoints[m.queryIdx].pt.y;
                p1[2 * k] = fj.keypoints[m.trainIdx].pt.x;
                p1[2 * k + 1] = fj.keypoints[m.trainIdx].pt.y;
            }
            double E[9];
            info.inliers_mask.assign(ngood, 0);
            const int rc = vo_find_essential(ctx, p0.data(), p1.data(), ngood, K.d.data(), VO_RANSAC, 0.999, 1.0,
                                             ~(uint64_t)0, 1000, E, info.inliers_mask.data());
            if (rc != VO_OK && rc != VO_ERR_FAILED) check(ctx, rc);
            for (uint8_t v : info.inliers_mask) info.num_inliers += v;
            info.src_img_idx = i;
            info.dst_img_idx = j;
            pairwise_matches.push_back(std::move(info));
        }
    }
    return pairwise_matches;
}

// Feature tracks across the window (reference: src/matcher/matcher.cpp:55-86).  The reference's rule, kept
// exactly: an inlier match (q in image a, t in image b) joins the FIRST track, in creation order, whose
// entry for a is q or whose entry for b is t, overwriting both entries; otherwise it starts a track.
// The reference finds that track by scanning all of them (O(matches x tracks)); here every (image, keypoint)
// value keeps the ordered set of tracks that currently hold it, so the first one is a lookup, and an
// overwrite moves the track between sets.  (A plain union-find would not do: overwriting can leave two
// tracks holding the same keypoint, and can drop a keypoint from a track.)
std::vector<std::vector<int>> createMatchMatrix(const std::vector<vo::MatchesInfo> pairwise_matches,
                                                const int pose_count) {
    std::vector<std::vector<int>> tracks;
    std::vector<std::unordered_map<int, std::set<int>>> holders((size_t)std::max(pose_count, 0));
    auto first_holder = [&](int img, int kp) {
        const auto it = holders[img].find(kp);
        return it == holders[img].end() || it->second.empty() ? -1 : *it->second.begin();
    };
    auto set_entry = [&](int row, int img, int kp) {
        int& cell = tracks[row][img];
        if (cell == kp) return;
        if (cell != -1) {
            auto it = holders[img].find(cell);
            it->second.erase(row);
            if (it->second.empty()) holders[img].erase(it);
        }
        cell = kp;
        holders[img][kp].insert(row);
    };
    for (const auto& pw : pairwise_matches) {
        const int a = pw.src_img_idx, b = pw.dst_img_idx;
        if (a < 0 || b < 0 || a >= pose_count || b >= pose_count)
            throw vo::Error(VO_ERR_INVALID, "createMatchMatrix: image index outside pose_count");
        for (size_t i = 0; i < pw.matches.size(); i++) {
            if (pw.inliers_mask[i] != 1) continue;
            const int q = pw.matches[i].queryIdx, t = pw.matches[i].trainIdx;
            const int ra = first_holder(a, q), rb = first_holder(b, t);
            int row = ra < 0 ? rb : (rb < 0 ? ra : std::min(ra, rb));
            if (row < 0) {
                tracks.emplace_back(pose_count, -1);
                row = (int)tracks.size() - 1;
            }
            set_entry(row, a, q);
            set_entry(row, b, t);
        }
    }
    return tracks;
}

