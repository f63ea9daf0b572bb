// feature_detector.h — the HIP variant of the reference's FeatureDetector: same class and method
// names and argument meaning as src/features/feature_detector.h:12-33 (and its CUDA twin
// src/features/cuda/feature_detector.h), forwarding to the C ABI in include/vo_hip.h.
#ifndef VO_HOST_FEATURE_DETECTOR_H
#define VO_HOST_FEATURE_DETECTOR_H

#include <vector>

#include "vo_frame.h"

class FeatureDetector {
public:
    // reference: feature_detector.cpp:3-10 — FAST(20, nms, TYPE_9_16); ORB nfeatures 5000 (the
    // CUDA variant uses 2000: cuda/feature_detector.cpp:7), every other ORB parameter default.
    explicit FeatureDetector(int orb_nfeatures = 5000);
    ~FeatureDetector();
    FeatureDetector(const FeatureDetector&) = delete;
    FeatureDetector& operator=(const FeatureDetector&) = delete;

    void detectFAST(VOFrame* frame);  // appends to frame->points (feature_detector.cpp:12-20)
    void detectFAST(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints);  // :23-25
    void detectComputeORB(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints,
                          vo::Descriptors* descriptors);  // :27-29
    void computeORB(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints,
                    vo::Descriptors* descriptors);  // :32-42 (keypoints made from frame.points)

    // AKAZE: vo_akaze_detect_compute.  BRISK is never called by the reference: it throws.
    void detectComputeAKAZE(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints, vo::Descriptors* descriptors);
    void detectComputeBRISK(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints, vo::Descriptors* descriptors);

private:
    vo_ctx* ctx_ = nullptr;
    int fast_threshold_ = 20;
    vo_orb_params orb_;
};

#endif
