/* vo_host_c.h — C entry points over the C++ host mirror's VisualOdometry driver, so that a test
 * or a script (ctypes) can push frames through the very class a C++ caller would use
 * (reference: src/main.cpp:53-66 — construct VisualOdometry, addImage per frame, read the pose). */
#ifndef VO_HOST_C_H
#define VO_HOST_C_H

#include <stdint.h>

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

#ifdef __cplusplus
}
#endif
#endif
