// host_vo_pose.cpp — updatePose() of the host mirror (reference: src/visual_odometry/vo_pose.h:12-53) over
// vo_essential_pose.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "host_batch.h"
#include "host_internal.h"
#include "vo_pose.h"

// ---- updatePose -----------------------------------------------------------------------------------
int updatePose(const vo::Matd& K, VOFrame* last_frame, VOFrame* new_frame) {
    vo::BatchGroup* const grp = vo::current_group();  // lock-step mode of vo_host_vo_run_concurrent, else nullptr
    vo_ctx* ctx = grp ? nullptr : vo::default_ctx();
    const int n = (int)new_frame->points.size();
    if ((int)last_frame->points.size() != n) throw vo::Error(VO_ERR_INVALID, "updatePose: point counts differ");
    static_assert(sizeof(vo::Point2f) == 8, "packed points");
    const float* p0 = n ? &last_frame->points[0].x : nullptr;
    const float* p1 = n ? &new_frame->points[0].x : nullptr;

    new_frame->E = vo::Matd(3, 3);
    new_frame->mask.d.assign(n, 0);
    // findEssentialMat(LMEDS) and recoverPose (vo_pose.h:14,19) as one submission: the mask recoverPose
    // receives is the one findEssentialMat left, the mask kept is recoverPose's
    int res = 0;
    vo::Matd points_4d(4, n);
    new_frame->local_R = vo::Matd(3, 3);
    new_frame->local_t = vo::Matd(3, 1);
    int rc;
    {
        StageScope ts(5);
        if (grp) {  // one vo_essential_pose_batch for the whole group, on the group's context
            rc = grp->essential_pose(vo::current_slot(), p0, p1, n, K.d.data(), VO_LMEDS, 0.999, 1.0, ~(uint64_t)0, 1000,
                                     kMax3DDist, new_frame->E.d.data(), new_frame->mask.d.data(),
                                     new_frame->local_R.d.data(), new_frame->local_t.d.data(), points_4d.d.data(), &res);
            if (rc != VO_OK && rc != VO_ERR_FAILED) throw vo::Error(rc, grp->last_error());
        } else {
            rc = vo_essential_pose(ctx, p0, p1, n, K.d.data(), VO_LMEDS, 0.999, 1.0, ~(uint64_t)0, 1000, kMax3DDist,
                                   new_frame->E.d.data(), new_frame->mask.d.data(), new_frame->local_R.d.data(),
                                   new_frame->local_t.d.data(), points_4d.d.data(), &res);
        }
    }
    if (rc == VO_OK) {
        StageScope ts(6);
        new_frame->points_3d = points4DToVec(points_4d);
    } else if (rc != VO_ERR_FAILED) {
        check(ctx, rc);
    }
    if (res > (int)kMinPosePoints) {
        new_frame->local_t = -new_frame->local_t;
        new_frame->local_R = new_frame->local_R.t();
        new_frame->local_pose = vo::hconcat(new_frame->local_R, new_frame->local_t);
        {
            StageScope ts(7);
            new_frame->scale = getScale(*last_frame, *new_frame, kMinPosePoints, 200, kMax3DDist);
        }
        new_frame->pose_t = last_frame->pose_t + (double)new_frame->scale * (last_frame->pose_R * new_frame->local_t);
        new_frame->pose_R = new_frame->local_R * last_frame->pose_R;
        new_frame->pose = vo::hconcat(new_frame->pose_R, new_frame->pose_t);
        return 0;
    }
    new_frame->pose_R = last_frame->pose_R;
    new_frame->pose_t = last_frame->pose_t;
    new_frame->pose = last_frame->pose;
    return 1;
}

