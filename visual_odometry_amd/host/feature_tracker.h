// feature_tracker.h — the HIP variant of the reference's FeatureTracker: same class and method as
// src/features/feature_tracker.h:13-23 (CUDA twin: src/features/cuda/feature_tracker.h),
// forwarding to vo_lk_track (include/vo_hip.h).
#ifndef VO_HOST_FEATURE_TRACKER_H
#define VO_HOST_FEATURE_TRACKER_H

#include "vo_frame.h"

class FeatureTracker {
public:
    // reference: feature_tracker.cpp:5-7 — cv::SparsePyrLKOpticalFlow::create() with every default.
    // channels: 3 tracks as the reference's CPU build does on the 8UC3 image imread returns for a
    // gray PNG (src/main.cpp:61); 1 as its CUDA build does on the gray gpu_image (vo_frame.h:39-43).
    explicit FeatureTracker(int channels = 3);
    ~FeatureTracker();
    FeatureTracker(const FeatureTracker&) = delete;
    FeatureTracker& operator=(const FeatureTracker&) = delete;

    // calc(vo0->image, vo1->image, vo0->points, vo1->points, status), then removePoints
    // (feature_tracker.cpp:10-17).  When vo0's image is the one the previous call tracked into, its
    // pyramid is still on the device and is not rebuilt.
    float trackPoints(VOFrame* vo0, VOFrame* vo1);

private:
    vo_ctx* ctx_ = nullptr;
    vo_lk_params lk_;
    uint64_t cached_id_ = 0;  // vo::Image::id of the image whose pyramid the context holds
};

#endif
