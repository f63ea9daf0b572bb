/* vo_hip.h — C ABI of libvo_hip.so, the MI355X (gfx950) implementation of the reference's
 * per-frame-pair hot path.
 *
 * The reference (maym86/visual_odometry) has no FFI: its seam is a compile-time source-variant
 * switch between src/features/ and src/features/cuda/ with identical C++ signatures
 * (reference: CMakeLists.txt:33-37, src/visual_odometry/visual_odometry.h:8-14).  A third
 * variant forwards those same methods to the entry points below (INTEGRATION.md shows the
 * adapter); each entry point cites the reference call it replaces.
 *
 * Conventions: plain pointers and sizes, caller-allocated outputs with capacity + count-out,
 * int status (VO_OK == 0), vo_last_error(ctx) for the message.  Unless the name ends in _dev,
 * pointers are HOST pointers and the call is synchronous (upload, kernels, download).  One
 * vo_ctx = one HIP stream + workspace; a ctx must not be used from two threads at once.
 * The library never falls back to the CPU: without a usable HIP device vo_ctx_create fails.
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include "vo_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vo_ctx vo_ctx;

typedef struct vo_limits {
    int32_t max_width;    /* largest image width  (default 1241) */
    int32_t max_height;   /* largest image height (default 376)  */
    int32_t max_features; /* largest ORB nfeatures / descriptor count per set (default 8192) */
    int32_t max_points;   /* largest point count for E estimation / recoverPose (default 8192) */
    int32_t max_batch;    /* frames per vo_pairs_run_dev call (default 64) */
} vo_limits;

void vo_limits_default(vo_limits* l);

/* No reference counterpart (the reference keeps cv::Ptr detectors as members:
 * src/features/feature_detector.cpp:3-10). limits may be NULL. */
int vo_ctx_create(int device, const vo_limits* limits, vo_ctx** out);
void vo_ctx_destroy(vo_ctx* ctx);
const char* vo_last_error(const vo_ctx* ctx);
int vo_ctx_synchronize(vo_ctx* ctx);

void vo_orb_params_default(vo_orb_params* p, int nfeatures);

/* ---- features: replaces FeatureDetector (src/features/feature_detector.cpp) ------------- */

/* detector_->detect(image, keypoints) with FastFeatureDetector::create(thr, nms, TYPE_9_16)
 * reference: src/features/feature_detector.cpp:4,15,24 (cuda variant :6,17,26).
 * Keypoints come back in cv's order (rows ascending, x ascending). *n is the true count;
 * VO_ERR_CAPACITY if it exceeds cap (the first cap are written). */
int vo_fast_detect(vo_ctx* ctx, const uint8_t* gray, int w, int h, int stride, int threshold,
                   int nms, vo_keypoint* out, int cap, int* n);

/* descriptor_->detectAndCompute(image, noArray(), keypoints, descriptors)
 * reference: src/features/feature_detector.cpp:5,27-29 (cuda variant :7,31-35).
 * desc is cap x 32 bytes. Up to nfeatures plus response ties are returned (cv semantics). */
int vo_orb_detect_compute(vo_ctx* ctx, const uint8_t* gray, int w, int h, int stride,
                          const vo_orb_params* p, vo_keypoint* kp, uint8_t* desc, int cap, int* n);

/* descriptor_->compute(image, keypoints, descriptors) on caller keypoints
 * reference: src/features/feature_detector.cpp:32-42. Keypoints closer than edge_threshold to
 * the border are dropped (list compacted in place, *n updated), angle/octave are used as given. */
int vo_orb_compute(vo_ctx* ctx, const uint8_t* gray, int w, int h, int stride,
                   const vo_orb_params* p, vo_keypoint* kp_inout, int* n_inout, uint8_t* desc);

/* ---- tracker: replaces FeatureTracker (src/features/feature_tracker.cpp) ------------------ */

void vo_lk_params_default(vo_lk_params* p);

/* optical_flow_->calc(vo0->image, vo1->image, vo0->points, vo1->points, status) with
 * cv::SparsePyrLKOpticalFlow::create() (21x21 window, 4 levels, 30 iterations / eps 0.01,
 * minEigThreshold 1e-4, no initial flow)
 * reference: src/features/feature_tracker.cpp:6,13 (cuda variant src/features/cuda/feature_tracker.cpp:5,18).
 * prev / next: gray planes of w x h; prev_pts / next_pts: n x 2 floats; status: n bytes of 0/1.
 * The pyramid of `next` stays in the context: a following call with prev == NULL and the same
 * geometry tracks from that image (the per-image loop tracks frame k -> k+1, then k+1 -> k+2,
 * src/visual_odometry/visual_odometry.cpp:39).  removePoints() (src/features/utils.h:9-58) is
 * host bookkeeping and lives in the host mirror. */
int vo_lk_track(vo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
                const float* prev_pts, int n, const vo_lk_params* params, float* next_pts, uint8_t* status);

/* vo_lk_track for n_seq independent image pairs of one size in ONE submission (no counterpart in the reference,
 * which tracks one pair per call: src/features/feature_tracker.cpp:13).  The frames of one sequence are serial,
 * sequences are not: a host that drives several sequences in lock step (visual_odometry_amd/host:
 * vo_host_vo_run_concurrent) sends their frames through the tracker together, so that the pyramids, the
 * derivative images and the point sets of all of them cost one upload, one launch each and one download.  Pair s
 * = (prev[s], next[s]) with n[s] points; every result is bit-identical to vo_lk_track on that pair.  prev may be
 * NULL (or hold NULLs for every pair): track from the `next` images of the previous call, which must have had
 * the same geometry and n_seq.  At most 64 pairs a call. */
int vo_lk_track_batch(vo_ctx* ctx, int n_seq, const uint8_t* const* prev, const uint8_t* const* next, int w, int h,
                      int stride, const float* const* prev_pts, const int* n, const vo_lk_params* params,
                      float* const* next_pts, uint8_t* const* status);

/* Page-locked host memory (hipHostMalloc) for callers that assemble a batch themselves: when the `next` (or `prev`)
 * images of vo_lk_track_batch are tightly packed and lie one after the other in such a block, they are sent by DMA
 * from where they are instead of being gathered into the context's staging buffer first.  NULL on failure.
 * A device must be usable; the memory belongs to the process, not to a context. */
void* vo_host_alloc_pinned(size_t bytes);
void vo_host_free_pinned(void* p);

/* ---- matcher: replaces the body of matcher() (src/matcher/matcher.cpp) ------------------- */

/* matcher->knnMatch(desc_q, desc_t, matches, 2) with "BruteForce-Hamming"
 * reference: src/matcher/matcher.cpp:10,17. out is nq x 2 (row-major), n_valid[i] = min(2, nt)
 * (n_valid may be NULL). desc_bytes <= 64. Ties keep the lower train index first. */
int vo_knn2_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes,
                    vo_dmatch* out, int* n_valid);

/* ratio test d0 < ratio*d1 and ||p0-p1|| < max_disp, order preserving
 * reference: src/matcher/matcher.cpp:23-38 (kMatchRatio :6, 200 px :30). Rows with fewer than
 * two neighbours are skipped (undefined behaviour in the reference). */
int vo_match_filter(vo_ctx* ctx, const vo_dmatch* knn, const int* n_valid, int nq,
                    const vo_keypoint* kq, const vo_keypoint* kt, float ratio, float max_disp,
                    vo_dmatch* good, int* n_good);

/* Cross-check option (BASELINE.json north_star: "ratio and cross-check filtering").  The reference's
 * matcher is BFMatcher(NORM_HAMMING, crossCheck=false) (src/matcher/matcher.cpp:10), so this is off by
 * default everywhere; cv's own crossCheck mode only exists for k = 1 and cannot be combined with the
 * k = 2 ratio test the reference uses, hence the rule here is stated, not inherited: a match
 * (q -> t) that passed the ratio and displacement tests is kept only if q is also the nearest query
 * of train row t (Hamming distance, ties to the lower query index: cv's rule in the other direction).
 * vo_knn2_hamming_cross runs both directions in ONE launch and returns rev_nn[t] = nearest query of
 * train row t (-1 if nq == 0); vo_match_filter_cross applies the rule. */
int vo_knn2_hamming_cross(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes,
                          vo_dmatch* out, int* n_valid, int* rev_nn);
int vo_match_filter_cross(vo_ctx* ctx, const vo_dmatch* knn, const int* n_valid, int nq, const int* rev_nn,
                          int nt, const vo_keypoint* kq, const vo_keypoint* kt, float ratio, float max_disp,
                          vo_dmatch* good, int* n_good);

/* knnMatch(k = 2) for callers that cannot allocate.  NO counterpart in the reference: it exists for code
 * under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same results as
 * vo_knn2_hamming, but: d_q / d_t / d_out are DEVICE pointers (16-byte aligned), rows are exactly 32 or
 * 64 bytes, nq and nt are at most 8192 (VO_ERR_CAPACITY beyond), d_out is nq x 2 vo_dmatch (missing
 * neighbours: trainIdx -1, distance FLT_MAX).  The call allocates nothing: its partial and merged keys
 * live in a 0.81 MiB static device array that every GPU the library runs on holds until the process
 * ends.  It only queues work on the context's stream: results are valid after vo_ctx_synchronize(ctx).
 * All calls on one GPU share that array, so calls that overlap on one GPU (two contexts) race. */
int vo_knn2_hamming_static(vo_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int desc_bytes,
                           vo_dmatch* d_out);

/* ---- sfm: replaces the cv calls in updatePose() and matcher() ---------------------------- */

/* cv::findEssentialMat(p0, p1, K, method, prob, threshold, mask)
 * reference: src/matcher/matcher.cpp:42 (VO_RANSAC, 0.999, 1.0),
 *            src/visual_odometry/vo_pose.h:14 (VO_LMEDS, 0.999, 1.0).
 * p0/p1: n x 2 float pixels; K: 3x3 row-major; seed: cv::RNG seed (cv uses (uint64)-1);
 * max_iters: 1000 in cv 3.4.2. E: 3x3 row-major, unit Frobenius norm; mask: n bytes of 0/1.
 * Returns VO_ERR_FAILED (E zeroed, mask zeroed) where cv returns an empty matrix. */
int vo_find_essential(vo_ctx* ctx, const float* p0, const float* p1, int n, const double* K,
                      int method, double prob, double threshold, uint64_t seed, int max_iters,
                      double* E, uint8_t* mask);

/* cv::recoverPose(E, p0, p1, K, R, t, distanceThresh, mask, triangulatedPoints)
 * reference: src/visual_odometry/vo_pose.h:19. mask_inout (n bytes) may be NULL; pts4d (4 x n
 * row-major, un-normalised homogeneous points of the winning candidate) may be NULL. */
int vo_recover_pose(vo_ctx* ctx, const double* E, const float* p0, const float* p1, int n,
                    const double* K, double dist_thresh, uint8_t* mask_inout, double* R, double* t,
                    double* pts4d, int* n_good);

/* updatePose()'s two cv calls as one submission: cv::findEssentialMat(p0, p1, K, method, prob, threshold, mask)
 * followed by cv::recoverPose(E, p0, p1, K, R, t, dist_thresh, mask, pts4d) on the same points with the mask
 * passed through (reference: src/visual_odometry/vo_pose.h:14,19).  Same outputs as vo_find_essential +
 * vo_recover_pose; the points cross the bus once and the host waits once.  mask (n bytes: the mask after
 * recoverPose), pts4d (4 x n) and n_good may be NULL.  VO_ERR_FAILED when no E is found: E and mask are zero, R
 * and t untouched. */
int vo_essential_pose(vo_ctx* ctx, const float* p0, const float* p1, int n, const double* K, int method,
                      double prob, double threshold, uint64_t seed, int max_iters, double dist_thresh, double* E,
                      uint8_t* mask, double* R, double* t, double* pts4d, int* n_good);

/* vo_essential_pose for n_seq independent point sets in ONE submission (no counterpart in the reference; the
 * companion of vo_lk_track_batch for a host that drives several sequences in lock step).  Set s: p0[s], p1[s] with
 * n[s] points.  Outputs per set: E + 9 s, R + 9 s, t + 3 s, status[s] (VO_OK, or VO_ERR_FAILED where cv returns an
 * empty matrix: then E and mask are zero, R and t untouched, n_good 0), mask[s] (n[s] bytes), pts4d[s] (4 x n[s]),
 * n_good[s]; mask, pts4d, n_good and their entries may be NULL.  Every set's results are bit-identical to
 * vo_essential_pose on that set.  At most 64 sets a call. */
int vo_essential_pose_batch(vo_ctx* ctx, int n_seq, const float* const* p0, const float* const* p1, const int* n,
                            const double* K, int method, double prob, double threshold, uint64_t seed, int max_iters,
                            double dist_thresh, double* E, uint8_t* const* mask, double* R, double* t,
                            double* const* pts4d, int* n_good, int* status);

/* cv::AKAZE::create()->detectAndCompute(image, noArray(), keypoints, descriptors)
 * reference: FeatureDetector::detectComputeAKAZE, src/features/feature_detector.cpp:7,43-45, called by
 * BundleAdjustment::addKeyFrame (src/sfm/bundle_adjustment.cpp:41).  MLDB, full size: desc rows of
 * VO_AKAZE_DESC_BYTES (61); keypoints in cv's order (level after level, row-major), kp.class_id = evolution
 * level, kp.octave = its octave, angle in degrees, size = 2 esigma derivative_factor.  *n is the true count even
 * when cap is smaller (VO_ERR_CAPACITY then).  vo_knn2_hamming takes the rows with desc_bytes = 61. */
void vo_akaze_params_default(vo_akaze_params* p);
int vo_akaze_detect_compute(vo_ctx* ctx, const uint8_t* gray, int w, int h, int stride, const vo_akaze_params* p,
                            vo_keypoint* kp, uint8_t* desc, int cap, int* n);
/* test hook: a plane of an evolution level of the last call (0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet, 5 mask) */
int vo_debug_akaze_level(vo_ctx* ctx, int level, int which, float* out, int* lw, int* lh);

/* The minimal solver inside cv::findEssentialMat (EMEstimatorCallback::runKernel, calib3d/src/five-point.cpp;
 * the reference reaches it only through src/matcher/matcher.cpp:42 and src/visual_odometry/vo_pose.h:14), exposed
 * so that a caller - and the parity tests - can run it on chosen samples.  x1, x2: count x 5 K-normalised
 * correspondences (x, y interleaved, 10 doubles a problem).  variant 0: one problem per lane (the solver of the
 * large RANSAC rounds); 1: one problem per 16-lane group (the solver of the small ones).  models: count x 10 x 9
 * (row-major E, unit Frobenius norm, ascending root; unused slots zero), nmodels: count. */
int vo_five_point(vo_ctx* ctx, const double* x1, const double* x2, int count, int variant, double* models,
                  int* nmodels);

/* triangulate(points0, points1, P0, P1): cv::triangulatePoints + points4DToVec
 * reference: src/sfm/triangulation.cpp:46-66,13-22. pts3d: n x 3. */
int vo_triangulate(vo_ctx* ctx, const float* p0, const float* p1, int n, const double* P0,
                   const double* P1, double* pts3d);

/* getProjectionMatrix(K, pose): K [R^T | -R^T t]; reference: src/sfm/triangulation.cpp:34-44.
 * Twelve flops: computed on the host. */
void vo_projection_matrix(const double* K, const double* pose, double* P);

/* ---- device-resident batched frame-pair path --------------------------------------------
 * One call runs, for n_frames consecutive frames already resident in HBM, ORB detectAndCompute
 * on every frame and, for each consecutive pair (i, i+1): knnMatch(k=2) + ratio/displacement
 * filter + findEssentialMat(RANSAC) (the body of matcher(), src/matcher/matcher.cpp:13-50)
 * followed by recoverPose on the filtered matches (src/visual_odometry/vo_pose.h:19).
 * d_frames: DEVICE pointer, n_frames images of h rows with `stride` bytes per row, frame i at
 * d_frames + i*frame_stride. results: HOST array of n_frames-1 entries. */
typedef struct vo_pair_params {
    vo_orb_params orb;
    float ratio;         /* 0.7f */
    float max_disp;      /* 200 */
    double prob;         /* 0.999 */
    double threshold;    /* 1.0 */
    double dist_thresh;  /* 200 */
    uint64_t seed;       /* (uint64)-1 */
    int32_t max_iters;   /* 1000 */
    int32_t cross_check; /* 0 (the reference: BFMatcher crossCheck=false, src/matcher/matcher.cpp:10); 1: see vo_match_filter_cross */
} vo_pair_params;

typedef struct vo_pair_result {
    int32_t n_kp0, n_kp1;     /* ORB keypoints in frame i, i+1 */
    int32_t n_good;           /* matches after ratio + displacement filter */
    int32_t n_inliers;        /* RANSAC inliers of E (0 if n_good < 5) */
    int32_t n_pose;           /* recoverPose survivors */
    int32_t status;           /* VO_OK, or VO_ERR_FAILED when no E was found */
    double E[9], R[9], t[3];
} vo_pair_result;

void vo_pair_params_default(vo_pair_params* p, int nfeatures);

int vo_pairs_run_dev(vo_ctx* ctx, const uint8_t* d_frames, int n_frames, int w, int h, int stride,
                     size_t frame_stride, const double* K, const vo_pair_params* params,
                     vo_pair_result* results);

/* After vo_pairs_run_dev: copy out the details of pair `pair` for parity checks (any pointer may
 * be NULL). kp/desc are those of frame `pair` (0) and `pair+1` (1). */
int vo_pairs_get_frame(vo_ctx* ctx, int frame, vo_keypoint* kp, uint8_t* desc, int cap, int* n);
int vo_pairs_get_matches(vo_ctx* ctx, int pair, vo_dmatch* good, uint8_t* inlier_mask,
                         uint8_t* pose_mask, int cap, int* n);

/* ---- sequence runner (BASELINE.json configs[3]: a whole sequence pipelined on one GPU) ------
 * Replaces the front-end work of the per-image loop (reference: src/main.cpp:59-66 ->
 * VisualOdometry::addImage, src/visual_odometry/visual_odometry.cpp:20-68) for the descriptor
 * path: n_frames consecutive frames in, one vo_pair_result per consecutive pair out, identical to
 * what vo_pairs_run_dev returns for the same frames.  The sequence is cut into chunks of
 * pairs_per_chunk pairs (neighbouring chunks share a frame); the chunk list is split into n_streams runs of
 * consecutive chunks, run r on stream r with its own workspace (the stream describes the shared frame of two of
 * its chunks once), the streams are fed in turn, so uploads and the latency-bound stages of one chunk overlap the
 * streaming stages of the others.  `mem` says where `frames` lives: VO_MEM_DEVICE (HBM),
 * VO_MEM_HOST (pageable; staged through pinned memory) or VO_MEM_HOST_PINNED (hipHostMalloc'ed /
 * registered memory; copied by DMA directly).  Not thread-safe; one runner per host thread. */
#define VO_MEM_HOST 0
#define VO_MEM_DEVICE 1
#define VO_MEM_HOST_PINNED 2
typedef struct vo_seq vo_seq;
int vo_seq_create(int device, const vo_limits* limits, int n_streams, int pairs_per_chunk, vo_seq** out);
void vo_seq_destroy(vo_seq* s);
const char* vo_seq_last_error(const vo_seq* s);
int vo_seq_streams(const vo_seq* s);
/* context i of the runner (owned by the runner): for the measurement hooks and vo_ctx_*_option */
vo_ctx* vo_seq_ctx(vo_seq* s, int i);
/* host time of the last vo_seq_run spent queueing work (submit) and waiting for results (wait) */
int vo_seq_get_stats(const vo_seq* s, int* chunks, double* submit_ms, double* wait_ms);
int vo_seq_run(vo_seq* s, const uint8_t* frames, int mem, int n_frames, int w, int h, int stride,
               size_t frame_stride, const double* K, const vo_pair_params* params, vo_pair_result* results);

/* vo_seq_run plus the per-pair tables the reference's matcher() / updatePose() hand back: the filtered
 * matches (MatchesInfo::matches, src/matcher/matcher.cpp:33), the RANSAC inlier mask
 * (MatchesInfo::inliers_mask, :44-48) and recoverPose's mask (src/visual_odometry/vo_pose.h:19).
 * HOST arrays of n_frames-1 rows of match_cap entries; row p holds results[p].n_good used entries.
 * Any pointer may be NULL.  VO_ERR_CAPACITY if a pair has more than match_cap matches. */
typedef struct vo_seq_outputs {
    vo_dmatch* matches;
    uint8_t* inlier_mask;
    uint8_t* pose_mask;
    int32_t match_cap;
    int32_t reserved;
} vo_seq_outputs;
int vo_seq_run_ex(vo_seq* s, const uint8_t* frames, int mem, int n_frames, int w, int h, int stride,
                  size_t frame_stride, const double* K, const vo_pair_params* params, vo_pair_result* results,
                  const vo_seq_outputs* out);

/* Several sequences through the same pipeline in one call: the chunks of sequence i + 1 follow those of sequence i
 * without draining the streams in between (a rank that owns several KITTI sequences, BASELINE configs[4]; the
 * reference runs one sequence per process, src/main.cpp:59-66, so this has no counterpart there).  Same results
 * as n_seq calls of vo_seq_run_ex.  results[i]: n_frames[i] - 1 entries; outs: NULL or n_seq entries. */
int vo_seq_run_multi(vo_seq* s, const uint8_t* const* frames, const int* n_frames, int n_seq, int mem, int w, int h,
                     int stride, size_t frame_stride, const double* K, const vo_pair_params* params,
                     vo_pair_result* const* results, const vo_seq_outputs* outs);

/* ---- measurement hooks (bench.py) -------------------------------------------------------
 * When enabled, launches of the matcher kernel are bracketed with HIP events on the ctx stream;
 * vo_prof_get returns the number of launches and their summed duration since the last reset. */
#define VO_PROF_KNN 0
#define VO_PROF_FAST 1
#define VO_PROF_SOLVE 2
#define VO_PROF_FEATURES 3  /* vo_pairs_run_dev: everything from the pyramid to the descriptors of a batch */
#define VO_PROF_COUNT 4
int vo_prof_enable(vo_ctx* ctx, int on);

/* Tuning / test knobs of a context (no reference counterpart).
 * VO_OPT_SEL_CAP: candidates per (frame, level) the device-side keypoint selection holds in LDS
 * (default 6144, at most 13000); a level with more candidates is selected on the host instead,
 * with identical results.  Tests set it low to exercise that path.
 * VO_OPT_LAST_FLAGS (read with vo_ctx_get_option): OR of the selection flags of the last batch
 * (0 = every frame was selected on the device). */
#define VO_OPT_SEL_CAP 0
#define VO_OPT_LAST_FLAGS 1
#define VO_OPT_CAND_CAP 2 /* read only: FAST candidates per frame the lists hold now (they grow on demand) */
/* Switches between kernel schedules that give identical results (the parity tests flip them).  Each starts from
 * the environment variable of the same name without the OPT_ (VO_KNN_MFMA, VO_ORB_SPLIT, VO_SEQ_PREFETCH,
 * VO_AKAZE_GRAPH, VO_SOLVER, VO_KNN_SPLITS, VO_LK_GENERIC), read ONCE when the context is created, never per call. */
#define VO_OPT_KNN_MFMA 3     /* 1 (default): 32-byte rows on the matrix-core matcher; 0: popcount kernel */
#define VO_OPT_ORB_SPLIT 4    /* 0 (default): orientation + description fused; 1: two kernels */
#define VO_OPT_SEQ_PREFETCH 5 /* 0 (default): pinned frames go up on the chunk's stream; 1: copy stream per context */
#define VO_OPT_AKAZE_GRAPH 6  /* 1 (default): AKAZE scale space replayed from a hipGraph; 0: plain launches */
#define VO_OPT_SOLVER 7       /* five-point solver inside findEssentialMat: 0 (default) one hypothesis per 16-lane
                               * group, 1 one per lane.  Fixed per context: it never depends on the batch size, so a
                               * pair's E / masks / pose do not depend on how many pairs share its launch. */
#define VO_OPT_KNN_SPLITS 8   /* 0 (default): automatic; > 0: train-range splits of a single 2-NN problem */
#define VO_OPT_LK_GENERIC 9   /* 0 (default): cv's default 21 x 21 LK window on the kernel specialised for it; 1: on the
                               * kernels every other window size uses (VO_LK_GENERIC) */
int vo_ctx_set_option(vo_ctx* ctx, int option, int value);
int vo_ctx_get_option(vo_ctx* ctx, int option, int* value);

/* Test hook (no reference counterpart): copies level `level` of frame `frame` of the last ORB call's
 * pyramid (which = VO_DBG_PYRAMID: cv::resize INTER_LINEAR_EXACT chain) or blurred pyramid
 * (VO_DBG_BLUR: GaussianBlur 7x7 sigma 2) to `out` as w x h tightly packed bytes. */
#define VO_DBG_PYRAMID 0
#define VO_DBG_BLUR 1
int vo_debug_read_level(vo_ctx* ctx, int which, int frame, int level, uint8_t* out, int cap, int* w, int* h);
int vo_prof_get(vo_ctx* ctx, int which, int reset, int* launches, double* total_ms);

/* Test hook (no reference counterpart): level `level` of the last vo_lk_track call's pyramids
 * (VO_DBG_LK_PREV / VO_DBG_LK_NEXT: cv::pyrDown chain, w x h bytes) or of the Scharr derivative
 * image of prev (VO_DBG_LK_DERIV: w x h x 2 int16, dx dy interleaved), tightly packed; cap in bytes. */
#define VO_DBG_LK_PREV 0
#define VO_DBG_LK_NEXT 1
#define VO_DBG_LK_DERIV 2
int vo_debug_read_lk_level(vo_ctx* ctx, int which, int level, void* out, int cap, int* w, int* h);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
