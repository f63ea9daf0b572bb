// geom_types.h — device-side state of the essential-matrix / pose kernels and their launch
// wrappers (k_geometry.hip).  Internal.
#ifndef VO_GEOM_TYPES_H
#define VO_GEOM_TYPES_H

#include <hip/hip_runtime.h>

#include <cstdint>

struct CamK {
    double fx, fy, cx, cy;
};

// one per frame pair
struct EstState {
    int n;            // points in this pair
    int niters;       // RANSAC: current iteration cap; LMedS: fixed iteration count
    int best_count;   // RANSAC: inliers of the best model so far
    int best_h, best_m;
    int next_h;       // first hypothesis not yet replayed
    int status;       // VO_OK once a model was found
    int n_inliers;
    float thr_f;      // (float)(threshold^2) in normalised units
    int pad;
    double best_median;
    double E[9];
};

struct PoseState {
    double P[4][12];  // [R1|t] [R2|t] [R1|-t] [R2|-t]
    double R1[9], R2[9], t[3];
    double R[9], tt[3];  // winner
    int good[4];
    int win, n_good;
};

void launch_est_init(hipStream_t st, EstState* s, const int* d_n, int n_host, int n_cap, int niters, float thr_f,
                     int npairs);
void launch_normalize(hipStream_t st, const float2* p0, const float2* p1, const CamK* cam, const EstState* s,
                      const int* d_n, int n_host, int n_cap, int n_grid, double2* x1, double2* x2, int npairs);
void launch_five_point_batch(hipStream_t st, int count, bool coop, const double* x1, const double* x2, double* models,
                             int* nmodels);
void launch_solve5(hipStream_t st, int h0, int h1, const EstState* s, const uint16_t* samples, int samples_iters,
                   const double2* x1, const double2* x2, int n_cap, double* models, int* nmodels, int H, int npairs, int per_lane);
void launch_score(hipStream_t st, int h0, int h1, const EstState* s, const double2* x1, const double2* x2, int n_cap,
                  const double* models, const int* nmodels, int* counts, int H, int npairs);
void launch_replay(hipStream_t st, int h0, int h1, EstState* s, const int* nmodels, const int* counts, int H,
                   double prob, int npairs);
void launch_lmeds(hipStream_t st, EstState* s, const double2* x1, const double2* x2, int n_cap, const double* models,
                  const int* nmodels, double* medians, int H, int niters, int npairs);
void launch_est_finish(hipStream_t st, EstState* s, const double2* x1, const double2* x2, int n_cap,
                       const double* models, uint8_t* mask, int H, int npairs);
void launch_pose(hipStream_t st, const double* E, size_t e_stride, PoseState* ps, const double2* x1,
                 const double2* x2, const int* d_n, int n_host, int n_cap, int n_grid, const uint8_t* mask_in,
                 double dist, double* q4, uint8_t* cmask, uint8_t* mask_out, double* pts4, int npairs);
void launch_fill_subsets(hipStream_t st, uint64_t seed, int iters, int maxn, uint16_t* out);
void launch_triangulate(hipStream_t st, const double* P, const float2* p0, const float2* p1, int n, double* pts3);

#endif
