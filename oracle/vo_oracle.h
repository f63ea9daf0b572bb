/* vo_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C++, no OpenCV) of the OpenCV 3.4.x CPU algorithms that the
 * reference's hot path calls.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (visual_odometry_amd/) never does.
 *
 * PARITY UNPINNED: the arithmetic lives in OpenCV 3.4.2 (+contrib), a third-party dependency
 * that is absent from /root/reference and from this image (README.md:21-35 pins it), the
 * reference cannot be compiled here (needs OpenCV/GTSAM/Boost/gflags/glog/GTest), and the
 * reference's own tests (the two files under src/sfm/test) hold no assertions, golden vectors or fixtures
 * beyond six KITTI PNGs.  The restatement follows SURVEY.md Appendix A (published algorithm,
 * restated from the OpenCV sources from memory) and is anchored on the reference's call
 * sites, cited per function below, and on hand-derived known answers (tests/test_oracle_*.py).
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H

#include "../include/vo_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- features ------------------------------------------------------------------------- */

/* cv::FastFeatureDetector::create(thr, nms, TYPE_9_16)->detect()
 * reference: src/features/feature_detector.cpp:4,15,24. Returns all keypoints (no cap);
 * *n is the true count even when cap is smaller (VO_ERR_CAPACITY then). */
int vo_oracle_fast_detect(const uint8_t* gray, int w, int h, int stride, int threshold, int nms,
                          vo_keypoint* out, int cap, int* n);

/* cv::resize(INTER_LINEAR_EXACT) for 8-bit single channel (ORB pyramid step). */
int vo_oracle_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride,
                                  uint8_t* dst, int dw, int dh, int dstride);

/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) 8-bit fixed-point separable path. */
int vo_oracle_gauss7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);

void vo_oracle_orb_params_default(vo_orb_params* p, int nfeatures);
/* Where KeyPointsFilter::retainBest's nth_element splits: OpenCV 3.4.2 (the reference's pin) uses begin + n,
 * later releases begin + n - 1; process-wide switch of the ORACLE only (the product implements the 3.4.2 form). */
#define VO_ORACLE_RETAIN_CV342 0
#define VO_ORACLE_RETAIN_LATER 1
void vo_oracle_set_retain_best_split(int mode);
int vo_oracle_get_retain_best_split(void);

/* ORB pyramid geometry: per level width/height/scale/feature quota. Returns nlevels. */
int vo_oracle_orb_layout(int w, int h, const vo_orb_params* p, int* lw, int* lh, float* lscale,
                         int* quota);

/* Pyramid level pixels (blurred=0: as used by FAST/Harris/IC-angle; 1: as used by rBRIEF). */
int vo_oracle_orb_level(const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                        int level, int blurred, uint8_t* out /* lw*lh, tight */);

/* Per-level candidates after FAST+NMS+border filter, in FAST (row-major) order, with FAST
 * score and Harris response; used to check the GPU's stage-A output. */
int vo_oracle_orb_candidates(const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                             int level, int* xs, int* ys, float* fast_score, float* harris, int cap,
                             int* n);

/* cv::ORB::create(nfeatures)->detectAndCompute(image, noArray(), kps, desc)
 * reference: src/features/feature_detector.cpp:5,27-29. */
int vo_oracle_orb_detect_compute(const uint8_t* gray, int w, int h, int stride,
                                 const vo_orb_params* p, vo_keypoint* kp, uint8_t* desc, int cap,
                                 int* n);

/* cv::ORB::compute(image, kps, desc) with caller keypoints (angle/octave taken as given)
 * reference: src/features/feature_detector.cpp:32-42. kp is filtered in place. */
int vo_oracle_orb_compute(const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                          vo_keypoint* kp_inout, int* n_inout, uint8_t* desc);

/* ---- AKAZE ---------------------------------------------------------------- */

void vo_oracle_akaze_params_default(vo_akaze_params* p);

/* Evolution levels: sizes, sigma_size, border, esigma and FED step counts. Returns the number of levels. */
int vo_oracle_akaze_layout(int w, int h, const vo_akaze_params* p, int* lw, int* lh, int* sigma_size, int* border,
                           float* esigma, int* nsteps);
/* fed_tau_by_process_time(T, 1, 0.25, reordering): the step sizes of one FED cycle. Returns their number. */
int vo_oracle_akaze_tau(float T, float* tau, int cap);
/* One plane of one evolution level, tight floats. which: 0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet, 5 keypoint mask. */
int vo_oracle_akaze_level(const uint8_t* gray, int w, int h, int stride, const vo_akaze_params* p, int level,
                          int which, float* out);
/* cv::AKAZE::create()->detectAndCompute(image, noArray(), kps, desc)
 * reference: src/features/feature_detector.cpp:7,43-45 (from src/sfm/bundle_adjustment.cpp:41).
 * desc: rows of VO_AKAZE_DESC_BYTES (61). kp.class_id = evolution level, kp.octave = its octave. */
int vo_oracle_akaze_detect_compute(const uint8_t* gray, int w, int h, int stride, const vo_akaze_params* p,
                                   vo_keypoint* kp, uint8_t* desc, int cap, int* n);

/* ---- tracker -------------------------------------------------------------------------- */

void vo_oracle_lk_params_default(vo_lk_params* p);

/* Level sizes of buildOpticalFlowPyramid; returns the number of levels actually used. */
int vo_oracle_lk_levels(int w, int h, const vo_lk_params* prm, int* lw, int* lh);

/* One level of the tracker's pyramid (pyrDown chain) and its Scharr derivative image
 * (dx, dy interleaved), tight; either output may be NULL. */
int vo_oracle_lk_level(const uint8_t* img, int w, int h, int stride, const vo_lk_params* prm, int level,
                       uint8_t* out_img, int16_t* out_deriv);

/* optical_flow_->calc(prev, next, prev_pts, next_pts, status)
 * reference: src/features/feature_tracker.cpp:13. Images are gray planes (see vo_lk_params.channels). */
int vo_oracle_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h, int stride, const float* prev_pts,
                       int n, const vo_lk_params* prm, float* next_pts, uint8_t* status);

/* removePoints(vo0, vo1, &status); reference: src/features/utils.h:9-58. Compacts p0, p1 (n x 2),
 * pts3d (n3d x 3) and mask (nmask) in place, clears status of points with negative coordinates,
 * *ret = the function's return value. */
int vo_oracle_remove_points(float* p0, float* p1, int n, uint8_t* status, double* pts3d, int n3d, uint8_t* mask,
                            int nmask, int* n_out, int* n3d_out, int* nmask_out, int* ret);

/* ---- matcher -------------------------------------------------------------------------- */

/* BFMatcher(NORM_HAMMING).knnMatch(q, t, matches, 2); reference: src/matcher/matcher.cpp:10,17.
 * out is nq x 2; n_valid[i] = min(2, nt). */
int vo_oracle_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes,
                           vo_dmatch* out, int* n_valid);

/* ratio + displacement gate; reference: src/matcher/matcher.cpp:23-38. */
int vo_oracle_match_filter(const vo_dmatch* knn, const int* n_valid, int nq, const vo_keypoint* kq,
                           const vo_keypoint* kt, float ratio, float max_disp, vo_dmatch* good,
                           int* n_good);

/* ---- geometry ------------------------------------------------------------------------- */

/* cv::RNG(seed): first n outputs of next(). */
void vo_oracle_rng_next(uint64_t seed, int n, uint32_t* out);

/* RANSAC/LMedS 5-subsets for iterations 0..niters-1 (depends on seed and npoints only). */
int vo_oracle_sample_subsets(uint64_t seed, int npoints, int niters, int32_t* idx /* niters x 5 */);

/* Nister five-point solver on 5 normalised correspondences (x2^T E x1 = 0).
 * E_out: up to 10 row-major 3x3 matrices, unit Frobenius norm. Returns the count. */
int vo_oracle_five_point(const double* x1 /*5x2*/, const double* x2 /*5x2*/, double* E_out);

/* Sampson error of cv's EMEstimatorCallback::computeError, as float. */
void vo_oracle_sampson_err(const double* E, const double* x1, const double* x2, int n, float* err);

/* cv::findEssentialMat(p0, p1, K, method, prob, threshold, mask)
 * reference: src/matcher/matcher.cpp:42 (RANSAC), src/visual_odometry/vo_pose.h:14 (LMEDS).
 * stats (optional, 4 ints): iterations run, models evaluated, best inlier count, final niters. */
int vo_oracle_find_essential(const float* p0, const float* p1, int n, const double* K, int method,
                             double prob, double threshold, uint64_t seed, int max_iters,
                             double* E, uint8_t* mask, int* stats);

/* cv::decomposeEssentialMat */
void vo_oracle_decompose_essential(const double* E, double* R1, double* R2, double* t);

/* cv::triangulatePoints(P0, P1, x0, x1) -> 4xN row-major homogeneous points. */
int vo_oracle_triangulate_points(const double* P0, const double* P1, const double* x0 /*N x 2*/,
                                 const double* x1, int n, double* pts4d);

/* cv::recoverPose(E, p0, p1, K, R, t, distanceThresh, mask, triangulatedPoints)
 * reference: src/visual_odometry/vo_pose.h:19. mask_inout may be NULL. */
int vo_oracle_recover_pose(const double* E, const float* p0, const float* p1, int n,
                           const double* K, double dist_thresh, uint8_t* mask_inout, double* R,
                           double* t, double* pts4d, int* n_good);

/* triangulate(points0, points1, P0, P1); reference: src/sfm/triangulation.cpp:46-66.
 * out: N x 3 dehomogenised points. */
int vo_oracle_triangulate(const float* p0, const float* p1, int n, const double* P0,
                          const double* P1, double* pts3d);

/* getProjectionMatrix(K, pose); reference: src/sfm/triangulation.cpp:34-44. */
void vo_oracle_projection_matrix(const double* K, const double* pose /*3x4*/, double* P /*3x4*/);

#ifdef __cplusplus
}
#endif
#endif
