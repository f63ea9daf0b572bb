// vo_pose.h — mirror of src/visual_odometry/vo_pose.h:12-53.
#ifndef VO_HOST_VO_POSE_H
#define VO_HOST_VO_POSE_H

#include "triangulation.h"

const size_t kMinPosePoints = 8;
const float kMax3DDist = 200;

// findEssentialMat(LMEDS, 0.999, 1.0) -> recoverPose(200) -> points4DToVec -> sign/transpose
// convention -> getScale -> pose chaining.  Returns 0, or 1 when too few points survive.
int updatePose(const vo::Matd& K, VOFrame* last_frame, VOFrame* new_frame);

#endif
