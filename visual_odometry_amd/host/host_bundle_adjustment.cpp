// host_bundle_adjustment.cpp — the front end of BundleAdjustment (reference: src/sfm/bundle_adjustment.cpp:16-130):
// key-frame features, matcher() over the window, the match matrix and the initial 3-D points.  The GTSAM
// optimisation is out of scope.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

#include "bundle_adjustment.h"
#include "host_internal.h"
#include "matcher.h"

// ---- bundle adjustment front end (bundle_adjustment.h) ---------------------------------------------
void BundleAdjustment::init(const vo::Matd& K, size_t max_frames) {
    K_ = K;
    max_frames_ = max_frames;
    count_ = 0;
    features_.clear();
    R_.clear();
    t_.clear();
    pairwise_matches_.clear();
    match_matrix_.clear();
    points_3d_.clear();
    cameras_visible_.clear();
    points_img_.clear();
}

void BundleAdjustment::addKeyFrame(const VOFrame& frame) {
    R_.push_back(frame.pose_R);
    t_.push_back(frame.pose_t);
    vo::ImageFeatures f;
    if (use_akaze_) feature_detector_.detectComputeAKAZE(frame, &f.keypoints, &f.descriptors);  // :41
    else feature_detector_.detectComputeORB(frame, &f.keypoints, &f.descriptors);
    f.img_idx = count_++;
    f.img_size.width = frame.image.width;
    f.img_size.height = frame.image.height;
    features_.push_back(std::move(f));
    if (features_.size() > max_frames_) {
        features_.erase(features_.begin());
        R_.erase(R_.begin());
        t_.erase(t_.begin());
    }
    pairwise_matches_ = matcher(features_, K_);
    match_matrix_ = createMatchMatrix(pairwise_matches_, (int)R_.size());
    setPBAPoints();
}

void BundleAdjustment::setPBAPoints() {
    points_3d_.clear();
    cameras_visible_.clear();
    points_img_.clear();
    struct Track {
        std::vector<int> cams;
        std::vector<vo::Point2f> points;
    };
    // tracks with at least three views, grouped by the two key frames they are triangulated from so that
    // each group is one vo_triangulate call
    std::vector<Track> tracks;
    std::map<std::pair<int, int>, std::vector<int>> groups;
    for (const auto& row : match_matrix_) {
        Track tr;
        for (int cam = 0; cam < (int)row.size(); cam++) {
            if (row[cam] == -1) continue;
            tr.cams.push_back(cam);
            tr.points.push_back(features_[cam].keypoints[row[cam]].pt);
        }
        if (tr.points.size() < 3) continue;
        groups[{tr.cams[0], tr.cams[1]}].push_back((int)tracks.size());
        tracks.push_back(std::move(tr));
    }
    std::vector<vo::Point3d> tri(tracks.size());
    for (const auto& g : groups) {
        const int c0 = g.first.first, c1 = g.first.second;
        std::vector<vo::Point2f> p0, p1;
        for (int i : g.second) {
            p0.push_back(tracks[i].points[0]);
            p1.push_back(tracks[i].points[1]);
        }
        const std::vector<vo::Point3d> x = triangulate(p0, p1, getProjectionMatrix(K_, vo::hconcat(R_[c0], t_[c0])),
                                                       getProjectionMatrix(K_, vo::hconcat(R_[c1], t_[c1])));
        for (size_t k = 0; k < g.second.size(); k++) tri[g.second[k]] = x[k];
    }
    for (size_t i = 0; i < tracks.size(); i++) {
        // the point in the first view's frame: R^T (X - t)
        const vo::Matd& R = R_[tracks[i].cams[0]];
        const vo::Matd& t = t_[tracks[i].cams[0]];
        const double d[3] = {tri[i].x - t(0, 0), tri[i].y - t(1, 0), tri[i].z - t(2, 0)};
        double o[3];
        for (int c = 0; c < 3; c++) o[c] = R(0, c) * d[0] + R(1, c) * d[1] + R(2, c) * d[2];
        const double dist = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
        if (dist < kMax3DDist && o[2] > kMin3DDist && std::fabs(o[0]) < kMax3DWidth) {
            points_3d_.push_back(tri[i]);
            cameras_visible_.push_back(tracks[i].cams);
            points_img_.push_back(tracks[i].points);
        }
    }
}

