// host_batch.h — lock-step batching of the device calls of several VisualOdometry drivers (internal).
//
// The frames of one sequence are serial (tracked points and the pose chain), sequences are not.  S drivers
// running on S host threads each issue ~30 HIP calls per frame; those calls serialise inside the HIP runtime, and
// the GPU sees S streams of small latency-bound launches: eight threads reach 2.6x one driver, not 8x
// (DESIGN.md 9).  A BatchGroup makes the S threads meet at their two device calls per frame -- the tracker
// (FeatureTracker::trackPoints) and findEssentialMat + recoverPose (updatePose) -- and sends the S requests down as
// ONE vo_lk_track_batch / vo_essential_pose_batch call on the group's context; the host parts of a frame
// (image copy, removePoints, points4DToVec, getScale, pose chaining) stay on the S threads and run in parallel.
// Every request's results are bit-identical to the single calls, so a driver inside a group computes exactly what
// it computes alone.
#ifndef VO_HOST_BATCH_H
#define VO_HOST_BATCH_H

#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "vo_host_types.h"

namespace vo {

class BatchGroup {
public:
    explicit BatchGroup(int members);
    ~BatchGroup();
    BatchGroup(const BatchGroup&) = delete;
    BatchGroup& operator=(const BatchGroup&) = delete;

    // A member with no more frames (or one that failed) stops taking part; the others no longer wait for it.
    void leave(int slot);

    // The member's vo_lk_track call.  prev_cached: the member knows the group's context tracked into `prev` last time.
    int lk_track(int slot, const uint8_t* prev, bool prev_cached, const uint8_t* next, int w, int h, const float* p0,
                 int n, const vo_lk_params* prm, float* p1, uint8_t* status);
    // The member's vo_essential_pose call (LMedS or RANSAC; every member passes the same K and parameters).
    int essential_pose(int slot, const float* p0, const float* p1, int n, const double* K, int method, double prob,
                       double threshold, uint64_t seed, int max_iters, double dist_thresh, double* E, uint8_t* mask,
                       double* R, double* t, double* pts4d, int* n_good);
    const char* last_error() const { return err_.c_str(); }
    // The member's slot of the group's pinned image block (w x h bytes): a member copies its next image there from
    // its own thread and passes that address to lk_track, so the group's upload is one DMA with no gather.
    uint8_t* image_slot(int slot, int w, int h);

private:
    struct Request {
        int kind = 0;  // 0 none, 1 tracker, 2 pose
        const uint8_t *prev = nullptr, *next = nullptr;
        bool prev_cached = false;
        int w = 0, h = 0, n = 0;
        const float *p0 = nullptr, *p1in = nullptr;
        float* p1 = nullptr;
        uint8_t* status = nullptr;
        const vo_lk_params* prm = nullptr;
        const double* K = nullptr;
        int method = 0, max_iters = 0;
        double prob = 0, threshold = 0, dist_thresh = 0;
        uint64_t seed = 0;
        double *E = nullptr, *R = nullptr, *t = nullptr, *pts4d = nullptr;
        uint8_t* mask = nullptr;
        int* n_good = nullptr;
        int rc = 0;
    };
    int rendezvous(int slot, const Request& rq);  // deposit, wait for the group's call, return this member's status
    void run_locked();                             // the batched call for every deposited request (mu_ held)

    vo_ctx* ctx_ = nullptr;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Request> req_;
    std::vector<char> active_;
    int n_active_ = 0, n_waiting_ = 0;
    unsigned long generation_ = 0;
    std::vector<int> last_track_slots_;  // members of the last tracker call, in batch order (cached pyramids)
    std::string err_;
    uint8_t* arena_ = nullptr;  // members x image bytes, pinned
    size_t arena_image_ = 0;
    double t_track_ = 0, t_pose_ = 0;  // seconds inside the two batched calls (VO_HOST_TIMING prints them)
    long n_track_ = 0, n_pose_ = 0;
};

// the group (and the slot in it) of the calling thread; nullptr outside vo_host_vo_run_concurrent's lock-step mode
BatchGroup* current_group();
int current_slot();
void set_current_group(BatchGroup* g, int slot);

}  // namespace vo

#endif
