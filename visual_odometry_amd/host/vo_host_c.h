/* vo_host_c.h — C entry points over the C++ host mirror's VisualOdometry driver, so that a test
 * or a script (ctypes) can push frames through the very class a C++ caller would use
 * (reference: src/main.cpp:53-66 — construct VisualOdometry, addImage per frame, read the pose). */
#ifndef VO_HOST_C_H
#define VO_HOST_C_H

#include <stdint.h>

#include "../../include/vo_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vo_host_vo vo_host_vo;

/* K: 3x3 row-major.  scale_seed seeds getScale's generator (the reference's is unseedable).
 * tracker_channels: 3 = the reference's CPU build, 1 = its CUDA build (FeatureTracker). */
vo_host_vo* vo_host_vo_create(const double* K, int min_tracked_points, uint32_t scale_seed, int tracker_channels,
                              char* err, int err_cap);
void vo_host_vo_destroy(vo_host_vo* v);

/* VisualOdometry::addImage.  pose: 3x4 row-major (identity until three frames were added).
 * Returns 0, or -1 with the message in err. */
int vo_host_vo_add_image(vo_host_vo* v, const uint8_t* gray, int w, int h, double* pose, char* err, int err_cap);

/* S independent VisualOdometry drivers side by side, one host thread (and its own contexts / HIP streams) each:
 * the frames of one sequence are serial through the tracked points and the pose chain, sequences are not
 * ("frame-pair batching").  Driver s is fed frames[s][0 .. n_frames[s]) (w x h gray, tightly
 * packed, HOST memory) through addImage exactly as vo_host_vo_add_image would.  Every driver first runs its
 * first `warmup` frames (context creation, sample table, buffer growth), then all meet at a barrier and the
 * rest is timed: *wall_seconds is the time until the last driver finishes, *timed_frames the frames inside it.
 * poses: NULL or n_seq pointers to n_frames[s] x 12 doubles.  Same results as n_seq separate drivers
 * (getScale's generator is per thread, seeded scale_seed + s).
 * lockstep = G > 0 (G groups, driver s in group s mod G; two let one group's batch run on the GPU while the other
 * group's drivers do their host work): the drivers' two device calls per frame (tracker; findEssentialMat + recoverPose) meet in ONE
 * vo_lk_track_batch / vo_essential_pose_batch call on a context of the group's own instead of S calls on S streams
 * (host_batch.h); per-driver results stay bit-identical, the host parts of a frame still run on the S threads.
 * Returns 0, or -1 with the message in err. */
int vo_host_vo_run_concurrent(const double* K, int min_tracked_points, uint32_t scale_seed, int tracker_channels,
                              const uint8_t* const* frames, const int* n_frames, int n_seq, int w, int h, int warmup,
                              int lockstep, double* const* poses, double* wall_seconds, long* timed_frames, char* err,
                              int err_cap);

/* Wall time per stage of the per-frame driver (what VO_HOST_TIMING=1 prints at exit), readable by a harness.
 * One set of counters per process, not synchronised: meaningful for a single driver thread only.
 * vo_host_timing_get: stage i -> name, calls, microseconds per call; returns the number of stages. */
void vo_host_timing_enable(int on);
void vo_host_timing_reset(void);
int vo_host_timing_get(int i, const char** name, long* calls, double* us_each);

/* State of the last addImage for parity checks: counts first (n_prev == n_cur), then the arrays
 * (any pointer may be NULL; capacities are the counts returned by the first call). */
int vo_host_vo_state(vo_host_vo* v, int* full, int* tracking, int* n_points, int* n_mask, float* scale);
int vo_host_vo_arrays(vo_host_vo* v, float* prev_pts, float* cur_pts, uint8_t* mask, double* E, double* local_R,
                      double* local_t, double* points_3d);

/* loadKittiCalibration / loadPoses through the mirror; K: 9 doubles; poses: cap x 12 doubles. */
int vo_host_load_calibration(const char* calib_file, int line_number, double* K, char* err, int err_cap);
int vo_host_load_poses(const char* file_name, double* poses, int cap, int* n);
int vo_host_save_poses(const char* file_name, const double* poses, int n);

/* removePoints through the mirror (same argument meaning as the oracle's restatement). */
int vo_host_remove_points(float* p0, float* p1, int n, uint8_t* status, double* pts3d, int n3d, uint8_t* mask,
                          int nmask, int* n_out, int* n3d_out, int* nmask_out, int* ret);

/* createMatchMatrix through the mirror (reference: src/matcher/matcher.cpp:55-86).  Pair p joins images
 * src[p] -> dst[p] with the matches off[p] .. off[p+1]-1 of q[] / t[] / inliers[].  out: cap_rows x pose_count,
 * -1 where a track has no keypoint; *n_rows is the true row count. */
int vo_host_create_match_matrix(const int* src, const int* dst, const int* off, int n_pairs, const int* q,
                                const int* t, const uint8_t* inliers, int pose_count, int* out, int cap_rows,
                                int* n_rows);

/* BundleAdjustment front end of the mirror (bundle_adjustment.h): init, addKeyFrame, and what
 * setPBAPoints() leaves for an optimiser.  pose_R 3x3 / pose_t 3: camera-to-world pose of the key frame. */
typedef struct vo_host_ba vo_host_ba;
vo_host_ba* vo_host_ba_create(const double* K, int max_frames, int nfeatures, char* err, int err_cap);
/* descriptor: 0 = AKAZE (what the reference's addKeyFrame uses), 1 = ORB with orb_nfeatures */
vo_host_ba* vo_host_ba_create_ex(const double* K, int max_frames, int descriptor, int orb_nfeatures, char* err,
                                 int err_cap);
int vo_host_ba_desc_bytes(vo_host_ba* b);
void vo_host_ba_destroy(vo_host_ba* b);
int vo_host_ba_add_key_frame(vo_host_ba* b, const uint8_t* gray, int w, int h, const double* pose_R,
                             const double* pose_t, char* err, int err_cap);
int vo_host_ba_counts(vo_host_ba* b, int* n_frames, int* n_pairs, int* n_rows, int* n_points);
int vo_host_ba_features(vo_host_ba* b, int frame, vo_keypoint* kp, uint8_t* desc, int cap, int* n);
int vo_host_ba_pair(vo_host_ba* b, int pair, int* src, int* dst, vo_dmatch* matches, uint8_t* inliers, int cap,
                    int* n);
int vo_host_ba_match_matrix(vo_host_ba* b, int* out /* n_rows x n_frames */);
int vo_host_ba_points(vo_host_ba* b, double* points /* n x 3 */, uint8_t* visible /* n x n_frames */,
                      float* img_pts /* n x n_frames x 2 */);

#ifdef __cplusplus
}
#endif
#endif
