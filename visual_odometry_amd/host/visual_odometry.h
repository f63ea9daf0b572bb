// visual_odometry.h — mirror of the reference's driver class (src/visual_odometry/visual_odometry.h:21-51,
// visual_odometry.cpp:9-68): the per-image state machine that calls the accelerated front end.
// The driver itself is host glue; bundle adjustment (GTSAM) and the Kalman
// filter are out of scope, so `ba` must be false and pose_kalman is left untouched, as in the
// reference where that code is commented out (visual_odometry.cpp:60-65).
#ifndef VO_HOST_VISUAL_ODOMETRY_H
#define VO_HOST_VISUAL_ODOMETRY_H

#include <deque>

#include "feature_detector.h"
#include "feature_tracker.h"
#include "vo_frame.h"

class VisualOdometry {
public:
    VisualOdometry(const vo::Matd& K, size_t min_tracked_points, bool ba, int tracker_channels = 3);

    void addImage(const vo::Image& image, vo::Matd* pose, vo::Matd* pose_kalman);

    // the last two frames of the buffer (valid once three images were added); for tests and drawing
    bool full() const { return frame_buffer_.size() == kFrameBufferCapacity; }
    const VOFrame& previousFrame() const { return frame_buffer_[frame_buffer_.size() - 2]; }
    const VOFrame& currentFrame() const { return frame_buffer_[frame_buffer_.size() - 1]; }
    bool tracking() const { return tracking_; }

private:
    const size_t kFrameBufferCapacity = 3;

    FeatureDetector feature_detector_;
    FeatureTracker feature_tracker_;

    bool tracking_;
    size_t min_tracked_points_;

    std::deque<VOFrame> frame_buffer_;  // boost::circular_buffer<VOFrame>(3) in the reference

    vo::Matd K_;
};

#endif
