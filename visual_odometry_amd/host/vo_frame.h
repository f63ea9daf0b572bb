// vo_frame.h — mirror of the reference's VOFrame carrier
// (reference: src/visual_odometry/vo_frame.h:14-45), same member names; `image` is 8-bit gray.
#ifndef VO_HOST_VO_FRAME_H
#define VO_HOST_VO_FRAME_H

#include "vo_host_types.h"

const bool kHasHIP = true;

class VOFrame {
public:
    // Local transform between frames
    vo::Matd E;           // Essential matrix
    vo::Matd local_R;     // Rotation
    vo::Matd local_t;     // Translation
    vo::Matd local_pose;  // Pose [R|t]
    vo::Mask mask;

    // Global pose
    vo::Matd pose_R = vo::Matd::eye(3, 3);
    vo::Matd pose_t = vo::Matd(3, 1);
    vo::Matd pose = vo::Matd::eye(3, 4);

    float scale = 1.0;

    vo::Image image;
    std::vector<vo::Point2f> points;
    std::vector<int> tracked_index;
    std::vector<vo::Point3d> points_3d;

    void setImage(vo::Image image_in) { image = std::move(image_in); }
};

#endif
