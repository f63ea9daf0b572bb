// bundle_adjustment.h — the FRONT END of the reference's sliding-window bundle adjustment
// (src/sfm/bundle_adjustment.h:38-88, bundle_adjustment.cpp:29-130): key frames in, feature tracks and
// triangulated points out, which is everything a GTSAM back end needs as input.  Same class and method
// names; the optimisation itself (slove(): GTSAM, bundle_adjustment.cpp:147-249) and the viz calls are
// out of scope and are not here.
//
//   addKeyFrame   describe the frame, keep a window of max_frames key frames, then
//                 matcher() (GPU) -> createMatchMatrix() (host) -> setPBAPoints() (GPU triangulation)
//   setPBAPoints  a track seen in >= 3 key frames is triangulated from its FIRST TWO views
//                 (the reference stops filling the sfm inputs once it holds two: bundle_adjustment.cpp:89-94)
//                 with getProjectionMatrix(K, [R | t]) of those key frames, and kept when, in the first
//                 view's frame, it is nearer than 200, deeper than 10 and less than 40 off axis (:105-110).
//
// Key frames are described with AKAZE as in the reference (:41, detectComputeAKAZE -> vo_akaze_detect_compute,
// 61-byte MLDB rows into the same matcher()); `akaze = false` describes them with ORB instead (cheaper, and the
// descriptor the rest of the accelerated path uses).  A deviation, stated: the reference indexes its point list by CAMERA index where it means the list
// position (:90: points[cam_idx]) — out of bounds whenever a track skips key frame 0; this mirror uses the
// position, which is what the line computes whenever it is defined.
// A second deviation, stated: the reference triangulates with cv::sfm::triangulatePoints (opencv_contrib's sfm
// module, :100), which is neither vendored in the reference nor installed here; this mirror calls
// triangulate() (triangulation.h -> vo_triangulate, the DLT of cv::triangulatePoints).  For TWO views both are
// the homogeneous DLT of the same 4 x 4 system, so the points agree up to the solver's rounding, but nothing
// here can check that: the BA front end's points are "parity unpinned" (DESIGN.md section 2).
#ifndef VO_HOST_BUNDLE_ADJUSTMENT_H
#define VO_HOST_BUNDLE_ADJUSTMENT_H

#include <vector>

#include "feature_detector.h"
#include "matcher.h"
#include "triangulation.h"

class BundleAdjustment {
public:
    explicit BundleAdjustment(int orb_nfeatures = 2000, bool akaze = true)
        : feature_detector_(orb_nfeatures), use_akaze_(akaze) {}
    int descriptorBytes() const { return use_akaze_ ? 61 : 32; }
    void init(const vo::Matd& K, size_t max_frames);  // bundle_adjustment.cpp:16-27
    void addKeyFrame(const VOFrame& frame);           // :29-61

    // what setPBAPoints() leaves for the optimiser (bundle_adjustment.h:70-74)
    const std::vector<vo::Point3d>& points3D() const { return points_3d_; }
    const std::vector<std::vector<int>>& camerasVisible() const { return cameras_visible_; }
    const std::vector<std::vector<vo::Point2f>>& pointsImg() const { return points_img_; }
    const std::vector<std::vector<int>>& matchMatrix() const { return match_matrix_; }
    const std::vector<vo::MatchesInfo>& pairwiseMatches() const { return pairwise_matches_; }
    const std::vector<vo::ImageFeatures>& features() const { return features_; }

    const float kMin3DDist = 10;  // bundle_adjustment.h:54-56
    const float kMax3DDist = 200;
    const float kMax3DWidth = 40;

private:
    void setPBAPoints();  // :64-130

    FeatureDetector feature_detector_;
    bool use_akaze_ = true;
    vo::Matd K_;
    size_t max_frames_ = 4;
    int count_ = 0;
    std::vector<vo::ImageFeatures> features_;
    std::vector<vo::Matd> R_, t_;
    std::vector<vo::MatchesInfo> pairwise_matches_;
    std::vector<std::vector<int>> match_matrix_;
    std::vector<vo::Point3d> points_3d_;
    std::vector<std::vector<int>> cameras_visible_;
    std::vector<std::vector<vo::Point2f>> points_img_;
};

#endif
