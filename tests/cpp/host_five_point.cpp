// Host build of the device five-point solver and Jacobi SVD (visual_odometry_amd/csrc/five_point.h,
// small_linalg.h) so that their numerics can be checked against the oracle without a GPU.
// Test-only: built by tests/test_host_solver.py into tests/cpp/libhost_solver.so.
#include "../../visual_odometry_amd/csrc/five_point.h"

struct HostScratch {
    double* base;
    double& operator()(int i) const { return base[i]; }
};

extern "C" int host_five_point(const double* x1, const double* x2, double* E_out) {
    double scratch[FP_SCRATCH];
    HostScratch s{scratch};
    return five_point_solve(x1, x2, s, E_out);
}

extern "C" void host_decompose_essential(const double* E, double* R1, double* R2, double* t) {
    decompose_essential(E, R1, R2, t);
}

extern "C" void host_triangulate_point(const double* P0, const double* P1, const double* x0,
                                       const double* x1, double* X) {
    triangulate_point(P0, P1, x0[0], x0[1], x1[0], x1[1], X);
}

// X: null vector for the cameras P0, P1; Xm: the one jacobi_svd's mirror derives for the same cameras with their
// fourth columns negated (what k_pose_points uses for recoverPose's candidates [R | -t]).
extern "C" void host_triangulate_point_mirror(const double* P0, const double* P1, const double* x0,
                                              const double* x1, double* X, double* Xm) {
    triangulate_point(P0, P1, x0[0], x0[1], x1[0], x1[1], X, Xm);
}

// the two forms of the RANSAC inlier test: (float)error <= thr, and the division-free decision used by k_score
extern "C" void host_sampson_both(const double* E, const double* x1, const double* x2, int n, float thr,
                                  unsigned char* by_error, unsigned char* by_bound, float* err) {
    const double M = sampson_bound(thr);
    for (int i = 0; i < n; i++) {
        err[i] = sampson_error(E, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]);
        by_error[i] = err[i] <= thr;
        by_bound[i] = sampson_inlier(E, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1], thr, M);
    }
}
