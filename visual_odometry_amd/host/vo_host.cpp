// vo_host.cpp — host-side mirror of the reference's features / matcher / sfm interfaces over the
// C ABI of libvo_hip.so.  Everything that computes runs in the device library; this file only
// converts carriers and chains poses (the same glue the reference keeps on the host).
// Reference files mirrored: src/features/feature_detector.cpp, src/matcher/matcher.cpp,
// src/sfm/triangulation.cpp, src/visual_odometry/vo_pose.h.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <random>

#include "feature_detector.h"
#include "matcher.h"
#include "vo_pose.h"

namespace vo {

Matd operator*(const Matd& a, const Matd& b) {
    if (a.cols != b.rows) throw Error(VO_ERR_INVALID, "Matd product: inner dimensions differ");
    Matd m(a.rows, b.cols);
    for (int r = 0; r < a.rows; r++)
        for (int c = 0; c < b.cols; c++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += a(r, k) * b(k, c);
            m(r, c) = s;
        }
    return m;
}
Matd operator+(const Matd& a, const Matd& b) {
    if (a.rows != b.rows || a.cols != b.cols) throw Error(VO_ERR_INVALID, "Matd sum: shapes differ");
    Matd m = a;
    for (size_t i = 0; i < m.d.size(); i++) m.d[i] += b.d[i];
    return m;
}
Matd operator*(double s, const Matd& a) {
    Matd m = a;
    for (auto& v : m.d) v *= s;
    return m;
}
Matd operator-(const Matd& a) { return -1.0 * a; }
Matd hconcat(const Matd& a, const Matd& b) {
    if (a.rows != b.rows) throw Error(VO_ERR_INVALID, "hconcat: row counts differ");
    Matd m(a.rows, a.cols + b.cols);
    for (int r = 0; r < a.rows; r++) {
        for (int c = 0; c < a.cols; c++) m(r, c) = a(r, c);
        for (int c = 0; c < b.cols; c++) m(r, a.cols + c) = b(r, c);
    }
    return m;
}

namespace {
struct CtxHolder {
    vo_ctx* ctx = nullptr;
    ~CtxHolder() {
        if (ctx) vo_ctx_destroy(ctx);
    }
};
int env_device() {
    const char* e = std::getenv("VO_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}
vo_ctx* make_ctx() {
    vo_ctx* c = nullptr;
    const int rc = vo_ctx_create(env_device(), nullptr, &c);
    if (rc != VO_OK || !c) throw Error(rc, "vo_ctx_create failed: no usable HIP device (there is no CPU fallback)");
    return c;
}
}  // namespace

vo_ctx* default_ctx() {
    thread_local CtxHolder h;
    if (!h.ctx) h.ctx = make_ctx();
    return h.ctx;
}

void check(vo_ctx* ctx, int rc) {
    if (rc != VO_OK) throw Error(rc, vo_last_error(ctx));
}

}  // namespace vo

using vo::check;

// ---- FeatureDetector ------------------------------------------------------------------------
FeatureDetector::FeatureDetector(int orb_nfeatures) {
    vo_limits lim;
    vo_limits_default(&lim);
    lim.max_features = std::max(lim.max_features, orb_nfeatures);
    lim.max_points = std::max(lim.max_points, 2 * orb_nfeatures);
    const char* e = std::getenv("VO_HIP_DEVICE");
    const int rc = vo_ctx_create(e ? std::atoi(e) : 0, &lim, &ctx_);
    if (rc != VO_OK || !ctx_)
        throw vo::Error(rc, "vo_ctx_create failed: no usable HIP device (there is no CPU fallback)");
    vo_orb_params_default(&orb_, orb_nfeatures);
}

FeatureDetector::~FeatureDetector() {
    if (ctx_) vo_ctx_destroy(ctx_);
}

void FeatureDetector::detectFAST(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints) {
    const vo::Image& im = frame.image;
    keypoints->clear();
    if (im.empty()) return;
    int cap = std::max(4096, im.width * im.height / 16), n = 0;
    for (;;) {
        keypoints->resize(cap);
        const int rc = vo_fast_detect(ctx_, im.px.data(), im.width, im.height, im.width, fast_threshold_, 1,
                                      reinterpret_cast<vo_keypoint*>(keypoints->data()), cap, &n);
        if (rc == VO_ERR_CAPACITY && n > cap) {  // *n is the true count: retry once with room for it
            cap = n;
            continue;
        }
        check(ctx_, rc);
        break;
    }
    keypoints->resize(n);
}

void FeatureDetector::detectFAST(VOFrame* frame) {
    std::vector<vo::KeyPoint> keypoints;
    detectFAST(*frame, &keypoints);
    for (const auto& kp : keypoints) frame->points.push_back(kp.pt);  // appends, like the reference
}

void FeatureDetector::detectComputeORB(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints,
                                       vo::Descriptors* descriptors) {
    const vo::Image& im = frame.image;
    const int cap = std::max(2 * orb_.nfeatures, 64);
    int n = 0;
    keypoints->resize(cap);
    descriptors->d.resize((size_t)cap * 32);
    check(ctx_, vo_orb_detect_compute(ctx_, im.px.data(), im.width, im.height, im.width, &orb_,
                                      reinterpret_cast<vo_keypoint*>(keypoints->data()), descriptors->d.data(), cap,
                                      &n));
    keypoints->resize(n);
    descriptors->rows = n;
    descriptors->cols = 32;
    descriptors->d.resize((size_t)n * 32);
}

void FeatureDetector::computeORB(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints,
                                 vo::Descriptors* descriptors) {
    keypoints->clear();
    for (const auto& p : frame.points) {
        vo::KeyPoint kp;  // cv::KeyPoint(): size 0, angle -1, response 0, octave 0, class_id -1
        kp.pt = p;
        keypoints->push_back(kp);
    }
    const vo::Image& im = frame.image;
    int n = (int)keypoints->size();
    descriptors->d.resize((size_t)std::max(n, 1) * 32);
    check(ctx_, vo_orb_compute(ctx_, im.px.data(), im.width, im.height, im.width, &orb_,
                               reinterpret_cast<vo_keypoint*>(keypoints->data()), &n, descriptors->d.data()));
    keypoints->resize(n);
    descriptors->rows = n;
    descriptors->cols = 32;
    descriptors->d.resize((size_t)n * 32);
}

void FeatureDetector::detectComputeAKAZE(const VOFrame&, std::vector<vo::KeyPoint>*, vo::Descriptors*) {
    throw vo::Error(VO_ERR_INVALID, "detectComputeAKAZE is not on the accelerated path (use detectComputeORB)");
}
void FeatureDetector::detectComputeBRISK(const VOFrame&, std::vector<vo::KeyPoint>*, vo::Descriptors*) {
    throw vo::Error(VO_ERR_INVALID, "detectComputeBRISK is not on the accelerated path (use detectComputeORB)");
}

// ---- matcher ----------------------------------------------------------------------------------
static const float kMatchRatio = 0.7f;

std::vector<vo::MatchesInfo> matcher(const std::vector<vo::ImageFeatures>& features, const vo::Matd& K) {
    vo_ctx* ctx = vo::default_ctx();
    std::vector<vo::MatchesInfo> pairwise_matches;
    const int nimg = (int)features.size();
    for (int i = 0; i + 1 < nimg; i++) {
        for (int j = i + 1; j < std::min(nimg, i + 3); j++) {
            const auto& fi = features[i];
            const auto& fj = features[j];
            const int nq = fi.descriptors.rows, nt = fj.descriptors.rows;
            if (nq == 0) continue;
            std::vector<vo_dmatch> knn((size_t)nq * 2);
            std::vector<int> nvalid(nq);
            check(ctx, vo_knn2_hamming(ctx, fi.descriptors.d.data(), nq, fj.descriptors.d.data(), nt,
                                       fi.descriptors.cols, knn.data(), nvalid.data()));
            std::vector<vo_dmatch> good(nq);
            int ngood = 0;
            check(ctx, vo_match_filter(ctx, knn.data(), nvalid.data(), nq,
                                       reinterpret_cast<const vo_keypoint*>(fi.keypoints.data()),
                                       reinterpret_cast<const vo_keypoint*>(fj.keypoints.data()), kMatchRatio, 200.f,
                                       good.data(), &ngood));
            if (ngood < 5) continue;
            vo::MatchesInfo info;
            std::vector<float> p0((size_t)ngood * 2), p1((size_t)ngood * 2);
            for (int k = 0; k < ngood; k++) {
                vo::DMatch m;
                m.queryIdx = good[k].queryIdx;
                m.trainIdx = good[k].trainIdx;
                m.imgIdx = good[k].imgIdx;
                m.distance = good[k].distance;
                info.matches.push_back(m);
                p0[2 * k] = fi.keypoints[m.queryIdx].pt.x;
                p0[2 * k + 1] = fi.keypoints[m.queryIdx].pt.y;
                p1[2 * k] = fj.keypoints[m.trainIdx].pt.x;
                p1[2 * k + 1] = fj.keypoints[m.trainIdx].pt.y;
            }
            double E[9];
            info.inliers_mask.assign(ngood, 0);
            const int rc = vo_find_essential(ctx, p0.data(), p1.data(), ngood, K.d.data(), VO_RANSAC, 0.999, 1.0,
                                             ~(uint64_t)0, 1000, E, info.inliers_mask.data());
            if (rc != VO_OK && rc != VO_ERR_FAILED) check(ctx, rc);
            for (uint8_t v : info.inliers_mask) info.num_inliers += v;
            info.src_img_idx = i;
            info.dst_img_idx = j;
            pairwise_matches.push_back(std::move(info));
        }
    }
    return pairwise_matches;
}

std::vector<std::vector<int>> createMatchMatrix(const std::vector<vo::MatchesInfo> pairwise_matches,
                                                const int pose_count) {
    std::vector<std::vector<int>> tracks;
    for (const auto& pw : pairwise_matches) {
        const int a = pw.src_img_idx, b = pw.dst_img_idx;
        for (size_t i = 0; i < pw.matches.size(); i++) {
            if (pw.inliers_mask[i] != 1) continue;
            const int q = pw.matches[i].queryIdx, t = pw.matches[i].trainIdx;
            // the first track (in creation order) that already holds either end takes both
            auto it = std::find_if(tracks.begin(), tracks.end(),
                                   [&](const std::vector<int>& r) { return r[a] == q || r[b] == t; });
            if (it == tracks.end()) {
                tracks.emplace_back(pose_count, -1);
                it = tracks.end() - 1;
            }
            (*it)[a] = q;
            (*it)[b] = t;
        }
    }
    return tracks;
}

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
std::mt19937& scale_rng() {
    static std::mt19937 rng{std::random_device{}()};
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

// ---- updatePose -----------------------------------------------------------------------------------
int updatePose(const vo::Matd& K, VOFrame* last_frame, VOFrame* new_frame) {
    vo_ctx* ctx = vo::default_ctx();
    const int n = (int)new_frame->points.size();
    if ((int)last_frame->points.size() != n) throw vo::Error(VO_ERR_INVALID, "updatePose: point counts differ");
    static_assert(sizeof(vo::Point2f) == 8, "packed points");
    const float* p0 = n ? &last_frame->points[0].x : nullptr;
    const float* p1 = n ? &new_frame->points[0].x : nullptr;

    new_frame->E = vo::Matd(3, 3);
    new_frame->mask.d.assign(n, 0);
    int rc = vo_find_essential(ctx, p0, p1, n, K.d.data(), VO_LMEDS, 0.999, 1.0, ~(uint64_t)0, 1000,
                               new_frame->E.d.data(), new_frame->mask.d.data());
    int res = 0;
    vo::Matd points_4d(4, n);
    new_frame->local_R = vo::Matd(3, 3);
    new_frame->local_t = vo::Matd(3, 1);
    if (rc == VO_OK) {
        check(ctx, vo_recover_pose(ctx, new_frame->E.d.data(), p0, p1, n, K.d.data(), kMax3DDist,
                                   new_frame->mask.d.data(), new_frame->local_R.d.data(),
                                   new_frame->local_t.d.data(), points_4d.d.data(), &res));
        new_frame->points_3d = points4DToVec(points_4d);
    } else if (rc != VO_ERR_FAILED) {
        check(ctx, rc);
    }
    if (res > (int)kMinPosePoints) {
        new_frame->local_t = -new_frame->local_t;
        new_frame->local_R = new_frame->local_R.t();
        new_frame->local_pose = vo::hconcat(new_frame->local_R, new_frame->local_t);
        new_frame->scale = getScale(*last_frame, *new_frame, kMinPosePoints, 200, kMax3DDist);
        new_frame->pose_t = last_frame->pose_t + (double)new_frame->scale * (last_frame->pose_R * new_frame->local_t);
        new_frame->pose_R = new_frame->local_R * last_frame->pose_R;
        new_frame->pose = vo::hconcat(new_frame->pose_R, new_frame->pose_t);
        return 0;
    }
    new_frame->pose_R = last_frame->pose_R;
    new_frame->pose_t = last_frame->pose_t;
    new_frame->pose = last_frame->pose;
    return 1;
}
