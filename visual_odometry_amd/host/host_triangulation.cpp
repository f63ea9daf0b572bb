// host_triangulation.cpp — triangulate / getProjectionMatrix / points4DToVec / getScale of the host mirror
// (reference: src/sfm/triangulation.cpp) over vo_triangulate / vo_projection_matrix.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>

#include "host_internal.h"
#include "triangulation.h"

// ---- triangulation ------------------------------------------------------------------------------
std::vector<vo::Point3d> points4DToVec(const vo::Matd& p) {
    std::vector<vo::Point3d> out;
    out.reserve(p.cols);
    for (int i = 0; i < p.cols; i++) {
        vo::Point3d q;
        q.x = p(0, i) / p(3, i);
        q.y = p(1, i) / p(3, i);
        q.z = p(2, i) / p(3, i);
        out.push_back(q);
    }
    return out;
}

std::vector<vo::Point3d> points3DToVec(const vo::Matd& p) {
    std::vector<vo::Point3d> out;
    out.reserve(p.cols);
    for (int i = 0; i < p.cols; i++) {
        vo::Point3d q;
        q.x = p(0, i);
        q.y = p(1, i);
        q.z = p(2, i);
        out.push_back(q);
    }
    return out;
}

std::vector<vo::Point3d> triangulate(const std::vector<vo::Point2f>& points0, const std::vector<vo::Point2f>& points1,
                                     const vo::Matd& P0, const vo::Matd& P1) {
    if (points0.empty() || points1.empty()) return {};
    if (points0.size() != points1.size()) throw vo::Error(VO_ERR_INVALID, "triangulate: point counts differ");
    vo_ctx* ctx = vo::default_ctx();
    const int n = (int)points0.size();
    std::vector<vo::Point3d> out(n);
    static_assert(sizeof(vo::Point3d) == 24 && sizeof(vo::Point2f) == 8, "packed points");
    check(ctx, vo_triangulate(ctx, &points0[0].x, &points1[0].x, n, P0.d.data(), P1.d.data(), &out[0].x));
    return out;
}

vo::Matd getProjectionMatrix(const vo::Matd& K, const vo::Matd& pose) {
    vo::Matd P(3, 4);
    vo_projection_matrix(K.d.data(), pose.d.data(), P.d.data());
    return P;
}

namespace {
// The reference keeps one process-global generator (src/sfm/triangulation.cpp:10-11), which makes getScale
// non-reentrant.  Here every host thread has its own, so several VisualOdometry drivers can run side by side,
// one per thread (bench.py's lk_pipeline); setScaleSeed seeds the calling thread's generator.
std::mt19937& scale_rng() {
    thread_local std::mt19937 rng{std::random_device{}()};
    return rng;
}
double norm3(const vo::Point3d& p) { return std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z); }
double dist3(const vo::Point3d& a, const vo::Point3d& b) {
    return norm3({a.x - b.x, a.y - b.y, a.z - b.z});
}
}  // namespace

void setScaleSeed(uint32_t seed) { scale_rng().seed(seed); }

float getScale(const VOFrame& f0, const VOFrame& f1, size_t min_points, size_t max_points, float max_3d_dist) {
    if (f0.points_3d.empty() || f1.points_3d.empty()) return 1;
    std::uniform_int_distribution<int> uni(0, (int)f0.points_3d.size() - 1);
    auto usable = [&](const VOFrame& f, int i) {
        return f.mask.at(i) && f.points_3d[i].z > 0 && norm3(f.points_3d[i]) < max_3d_dist;
    };
    std::vector<int> picks;
    int last = -1;
    for (size_t i = 0; i < max_points; i++)
        for (int attempt = 0; attempt < 10; attempt++) {
            const int idx = uni(scale_rng());
            if (usable(f1, idx) && usable(f0, idx) && idx != last) {
                last = idx;
                picks.push_back(idx);
                break;
            }
        }
    if (picks.empty()) return 1;
    std::vector<double> ratios;
    for (size_t i = 0; i + 1 < picks.size(); i++)
        ratios.push_back(dist3(f0.points_3d[picks[i]], f0.points_3d[picks[i + 1]]) /
                         dist3(f1.points_3d[picks[i]], f1.points_3d[picks[i + 1]]));
    if (ratios.size() < min_points) return 1;
    std::sort(ratios.begin(), ratios.end());
    const double s = ratios[ratios.size() / 2];
    if (std::isnan(s) || std::isinf(s) || s == 0 || s > 5) return 1;
    return (float)s;
}

