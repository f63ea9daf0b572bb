// host_visual_odometry.cpp — the VisualOdometry driver of the host mirror (reference:
// src/visual_odometry/visual_odometry.cpp:9-68).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "host_internal.h"
#include "visual_odometry.h"
#include "vo_pose.h"

// ---- VisualOdometry ---------------------------------------------------------------------------
VisualOdometry::VisualOdometry(const vo::Matd& K, size_t min_tracked_points, bool ba, int tracker_channels)
    : feature_tracker_(tracker_channels) {
    if (ba)
        throw vo::Error(VO_ERR_INVALID,
                        "VisualOdometry: bundle adjustment (GTSAM back end) is not part of the accelerated path");
    tracking_ = false;
    min_tracked_points_ = min_tracked_points;
    K_ = K;
}

void VisualOdometry::addImage(const vo::Image& image, vo::Matd* pose, vo::Matd* /*pose_kalman*/) {
    {
        StageScope ts(1);
        VOFrame frame;
        frame.setImage(image);
        if (frame_buffer_.size() == kFrameBufferCapacity) frame_buffer_.pop_front();  // circular_buffer overwrite
        frame_buffer_.push_back(std::move(frame));
    }
    if (frame_buffer_.size() != kFrameBufferCapacity) return;

    VOFrame& vo1 = frame_buffer_[frame_buffer_.size() - 2];
    VOFrame& vo2 = frame_buffer_[frame_buffer_.size() - 1];

    if (!tracking_) {
        StageScope ts(2);
        feature_detector_.detectFAST(&vo1);
        tracking_ = true;
    }

    if (feature_tracker_.trackPoints(&vo1, &vo2) < 0) {  // never true in the reference either
        frame_buffer_.pop_back();
        return;
    }

    if (vo2.points.size() < min_tracked_points_) tracking_ = false;

    updatePose(K_, &vo1, &vo2);

    (*pose) = vo2.pose;
}

