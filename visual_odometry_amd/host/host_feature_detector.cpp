// host_feature_detector.cpp — FeatureDetector of the host mirror (reference: src/features/feature_detector.cpp)
// over vo_fast_detect / vo_orb_detect_compute / vo_orb_compute / vo_akaze_detect_compute.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "feature_detector.h"
#include "host_internal.h"

// ---- FeatureDetector ------------------------------------------------------------------------
FeatureDetector::FeatureDetector(int orb_nfeatures) {
    vo_limits lim;
    vo_limits_default(&lim);
    lim.max_features = std::max(lim.max_features, orb_nfeatures);
    lim.max_points = std::max(lim.max_points, 2 * orb_nfeatures);
    const char* e = std::getenv("VO_HIP_DEVICE");
    const int rc = vo_ctx_create(e ? std::atoi(e) : 0, &lim, &ctx_);
    if (rc != VO_OK || !ctx_)
        throw vo::Error(rc, "vo_ctx_create failed: no usable HIP device (there is no CPU fallback)");
    vo_orb_params_default(&orb_, orb_nfeatures);
}

FeatureDetector::~FeatureDetector() {
    if (ctx_) vo_ctx_destroy(ctx_);
}

void FeatureDetector::detectFAST(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints) {
    const vo::Image& im = frame.image;
    keypoints->clear();
    if (im.empty()) return;
    int cap = std::max(4096, im.width * im.height / 16), n = 0;
    for (;;) {
        keypoints->resize(cap);
        const int rc = vo_fast_detect(ctx_, im.px.data(), im.width, im.height, im.width, fast_threshold_, 1,
                                      reinterpret_cast<vo_keypoint*>(keypoints->data()), cap, &n);
        if (rc == VO_ERR_CAPACITY && n > cap) {  // *n is the true count: retry once with room for it
            cap = n;
            continue;
        }
        check(ctx_, rc);
        break;
    }
    keypoints->resize(n);
}

void FeatureDetector::detectFAST(VOFrame* frame) {
    std::vector<vo::KeyPoint> keypoints;
    detectFAST(*frame, &keypoints);
    for (const auto& kp : keypoints) frame->points.push_back(kp.pt);  // appends, like the reference
}

void FeatureDetector::detectComputeORB(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints,
                                       vo::Descriptors* descriptors) {
    const vo::Image& im = frame.image;
    const int cap = std::max(2 * orb_.nfeatures, 64);
    int n = 0;
    keypoints->resize(cap);
    descriptors->d.resize((size_t)cap * 32);
    check(ctx_, vo_orb_detect_compute(ctx_, im.px.data(), im.width, im.height, im.width, &orb_,
                                      reinterpret_cast<vo_keypoint*>(keypoints->data()), descriptors->d.data(), cap,
                                      &n));
    keypoints->resize(n);
    descriptors->rows = n;
    descriptors->cols = 32;
    descriptors->d.resize((size_t)n * 32);
}

void FeatureDetector::computeORB(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints,
                                 vo::Descriptors* descriptors) {
    keypoints->clear();
    for (const auto& p : frame.points) {
        vo::KeyPoint kp;  // cv::KeyPoint(): size 0, angle -1, response 0, octave 0, class_id -1
        kp.pt = p;
        keypoints->push_back(kp);
    }
    const vo::Image& im = frame.image;
    int n = (int)keypoints->size();
    descriptors->d.resize((size_t)std::max(n, 1) * 32);
    check(ctx_, vo_orb_compute(ctx_, im.px.data(), im.width, im.height, im.width, &orb_,
                               reinterpret_cast<vo_keypoint*>(keypoints->data()), &n, descriptors->d.data()));
    keypoints->resize(n);
    descriptors->rows = n;
    descriptors->cols = 32;
    descriptors->d.resize((size_t)n * 32);
}

// akaze_->detectAndCompute(frame.image, noArray(), *keypoints, *descriptors) with cv::AKAZE::create()'s defaults
// (feature_detector.cpp:7,43-45): MLDB rows of 61 bytes
void FeatureDetector::detectComputeAKAZE(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints,
                                         vo::Descriptors* descriptors) {
    const vo::Image& im = frame.image;
    vo_akaze_params prm;
    vo_akaze_params_default(&prm);
    int cap = 16384, n = 0;
    for (;;) {
        keypoints->resize((size_t)cap);
        descriptors->d.resize((size_t)cap * VO_AKAZE_DESC_BYTES);
        const int rc = vo_akaze_detect_compute(ctx_, im.px.data(), im.width, im.height, im.width, &prm,
                                               reinterpret_cast<vo_keypoint*>(keypoints->data()),
                                               descriptors->d.data(), cap, &n);
        if (rc == VO_ERR_CAPACITY && n > cap) {  // *n is the true count: once more with room for it
            cap = n;
            continue;
        }
        check(ctx_, rc);
        break;
    }
    keypoints->resize((size_t)n);
    descriptors->rows = n;
    descriptors->cols = VO_AKAZE_DESC_BYTES;
    descriptors->d.resize((size_t)n * VO_AKAZE_DESC_BYTES);
}
void FeatureDetector::detectComputeBRISK(const VOFrame&, std::vector<vo::KeyPoint>*, vo::Descriptors*) {
    throw vo::Error(VO_ERR_INVALID, "detectComputeBRISK is not on the accelerated path (use detectComputeORB)");
}

