/* vo_types.h — plain-C data carriers that cross the drop-in boundary.
 *
 * Layouts mirror the OpenCV 3.4 types the reference's interfaces move around
 * (reference: src/features/feature_detector.h:17-25, src/matcher/matcher.h:11,
 * src/visual_odometry/vo_frame.h:14-45), so a cv-typed adapter can memcpy them.
 */
#ifndef VO_TYPES_H
#define VO_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint: {Point2f pt; float size, angle, response; int octave, class_id;} — 28 bytes. */
typedef struct vo_keypoint {
    float x, y;
    float size;
    float angle;     /* degrees in [0,360), -1 when not computed */
    float response;
    int32_t octave;
    int32_t class_id;
} vo_keypoint;

/* cv::DMatch: {int queryIdx, trainIdx, imgIdx; float distance;} — 16 bytes. */
typedef struct vo_dmatch {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} vo_dmatch;

/* cv::ORB::create(...) arguments, in cv order. The reference fixes nfeatures only
 * (src/features/feature_detector.cpp:5 -> 5000; src/features/cuda/feature_detector.cpp:7 -> 2000);
 * every other field keeps the cv default (vo_orb_params_default). */
typedef struct vo_orb_params {
    int32_t nfeatures;
    float scale_factor;    /* 1.2f */
    int32_t nlevels;       /* 8 */
    int32_t edge_threshold;/* 31 */
    int32_t first_level;   /* 0 (only 0 supported) */
    int32_t wta_k;         /* 2 (only 2 supported) */
    int32_t score_type;    /* 0 = HARRIS_SCORE (only 0 supported) */
    int32_t patch_size;    /* 31 (only 31 supported) */
    int32_t fast_threshold;/* 20 */
} vo_orb_params;

/* cv::AKAZE::create(...) arguments, in cv order.  The reference takes every default
 * (src/features/feature_detector.cpp:7, cuda variant src/features/cuda/feature_detector.cpp:9):
 * DESCRIPTOR_MLDB, full size (486 bits = 61 bytes), 3 channels, threshold 0.001, 4 octaves x 4 sublevels,
 * DIFF_PM_G2.  Only MLDB / full size / PM_G2 are supported; the fields below are the ones that may vary. */
typedef struct vo_akaze_params {
    float threshold;          /* 0.001f: detector response threshold */
    int32_t n_octaves;        /* 4 */
    int32_t n_sublevels;      /* 4 (cv: nOctaveLayers) */
    int32_t channels;         /* 3 (1..3) */
    int32_t pattern_size;     /* 10 (AKAZEOptions::descriptor_pattern_size; even) */
    float soffset;            /* 1.6f  base scale offset */
    float derivative_factor;  /* 1.5f */
    float kcontrast_percentile; /* 0.7f */
    int32_t kcontrast_nbins;  /* 300 */
    int32_t reserved;
} vo_akaze_params;
#define VO_AKAZE_DESC_BYTES 61

/* cv::SparsePyrLKOpticalFlow::create(...) arguments. The reference takes every default
 * (src/features/feature_tracker.cpp:6, cuda variant src/features/cuda/feature_tracker.cpp:5);
 * vo_lk_params_default fills them in. flags are fixed at 0 (no initial flow, no eigenvalue output). */
typedef struct vo_lk_params {
    int32_t win_w, win_h;      /* 21 x 21 (3 .. 32 supported) */
    int32_t max_level;         /* 3 */
    int32_t max_count;         /* 30 (TermCriteria COUNT) */
    double epsilon;            /* 0.01 (TermCriteria EPS) */
    double min_eig_threshold;  /* 1e-4 */
    int32_t channels;          /* 3: the image is the 8UC3 that cv::imread makes of a gray PNG (what the
                                  reference's CPU build tracks on, src/main.cpp:61) given as its single
                                  gray plane; 1: a true single-channel image (the cuda variant's gpu_image) */
    int32_t reserved;
} vo_lk_params;

/* cv method enum values used by findEssentialMat (reference: src/matcher/matcher.cpp:42 RANSAC,
 * src/visual_odometry/vo_pose.h:14 LMEDS). */
#define VO_LMEDS 4
#define VO_RANSAC 8

/* status codes */
#define VO_OK 0
#define VO_ERR_INVALID 1    /* bad argument (null pointer, size out of range, unsupported param) */
#define VO_ERR_CAPACITY 2   /* an output or the context workspace is too small */
#define VO_ERR_DEVICE 3     /* HIP runtime error */
#define VO_ERR_FAILED 4     /* the estimator found no model (cv returns an empty Mat) */

#define VO_ORB_MAX_LEVELS 16
/* Smallest edge_threshold accepted: the descriptor window reaches 18 pixels from a keypoint, so below 19 it
 * would leave the image (cv pads every level by more than that and reads reflected pixels there; this
 * library keeps a 3-pixel frame only and refuses instead). The reference never changes the default, 31. */
#define VO_ORB_MIN_EDGE 19
#define VO_DESC_BYTES 32

#ifdef __cplusplus
}
#endif
#endif /* VO_TYPES_H */
