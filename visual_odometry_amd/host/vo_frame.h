// vo_frame.h — the per-image record the host driver passes between its stages.
//
// The member NAMES are the ones the reference's callers use (src/visual_odometry/vo_frame.h:14-45:
// visual_odometry.cpp, vo_pose.h, feature_tracker.cpp and triangulation.cpp all reach into the frame by
// name), so code written against the reference's VOFrame compiles against this one.  Everything else is
// this library's: OpenCV-free carrier types (vo_host_types.h), an 8-bit gray image instead of an 8UC3
// Mat plus a GpuMat copy (the HIP library uploads from the host image itself), and no CUDA switch.
#ifndef VO_HOST_VO_FRAME_H
#define VO_HOST_VO_FRAME_H

#include <utility>
#include <vector>

#include "vo_host_types.h"

const bool kHasHIP = true;  // takes the place of the reference's kHasCUDA (always true: there is no CPU path)

class VOFrame {
public:
    // ---- what the frame is built from ----------------------------------------------------------
    vo::Image image;  // gray, row-major; setImage() takes ownership
    void setImage(vo::Image image_in) { image = std::move(image_in); }

    // ---- features carried from frame to frame (FeatureDetector / FeatureTracker) -----------------
    std::vector<vo::Point2f> points;  // pixel positions; trackPoints keeps the two frames' lists aligned
    std::vector<int> tracked_index;   // present for source compatibility; the reference never fills it

    // ---- updatePose() outputs: motion of this frame relative to the one before ---------------------
    vo::Matd E;                          // essential matrix (LMedS), 3 x 3
    vo::Mask mask;                       // inliers of E that also pass recoverPose's cheirality / distance tests
    vo::Matd local_R, local_t;           // relative rotation 3 x 3 and unit translation 3 x 1
    vo::Matd local_pose;                 // [local_R | local_t], 3 x 4
    std::vector<vo::Point3d> points_3d;  // the triangulated inlier points, in this pair's frame
    float scale = 1.0;                   // getScale(): ratio of 3-D distances against the previous pair

    // ---- the chained trajectory: camera-to-world, identity until the third image -----------------
    vo::Matd pose_R = vo::Matd::eye(3, 3);
    vo::Matd pose_t = vo::Matd(3, 1);
    vo::Matd pose = vo::Matd::eye(3, 4);
};

#endif
