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
