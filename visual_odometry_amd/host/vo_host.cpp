// vo_host.cpp — host-side mirror of the reference's features / matcher / sfm interfaces over the
// C ABI of libvo_hip.so.  Everything that computes runs in the device library; this file only
// converts carriers and chains poses (the same glue the reference keeps on the host).
// Reference files mirrored: src/features/feature_detector.cpp, src/matcher/matcher.cpp,
// src/sfm/triangulation.cpp, src/visual_odometry/vo_pose.h.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <random>

#include <map>
#include <set>
#include <unordered_map>

#include "bundle_adjustment.h"
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

// ---- VO_HOST_TIMING=1: wall time per stage of the per-frame driver, printed when the process ends -------
namespace {
struct StageTimers {
    static constexpr int N = 10;
    const char* names[N] = {"image copy in", "frame buffer", "detectFAST", "vo_lk_track", "removePoints",
                            "vo_essential_pose", "points4DToVec", "getScale", "pose compose", "other"};
    double sec[N] = {0};
    long calls[N] = {0};
    bool on = std::getenv("VO_HOST_TIMING") != nullptr;
    ~StageTimers() {
        if (!on) return;
        for (int i = 0; i < N; i++)
            if (calls[i])
                std::fprintf(stderr, "[vo_host timing] %-18s %6ld calls  %8.1f us each\n", names[i], calls[i],
                             sec[i] / calls[i] * 1e6);
    }
};
StageTimers g_timers;
struct StageScope {
    int i;
    std::chrono::steady_clock::time_point t0;
    explicit StageScope(int idx) : i(idx) {
        if (g_timers.on) t0 = std::chrono::steady_clock::now();
    }
    ~StageScope() {
        if (!g_timers.on) return;
        g_timers.sec[i] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        g_timers.calls[i]++;
    }
};
}  // namespace

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

// akaze_->detectAndCompute(frame.image, noArray(), *keypoints, *descriptors) with cv::AKAZE::create()'s defaults
// (feature_detector.cpp:7,43-45): MLDB rows of 61 bytes
void FeatureDetector::detectComputeAKAZE(const VOFrame& frame, std::vector<vo::KeyPoint>* keypoints,
                                         vo::Descriptors* descriptors) {
    const vo::Image& im = frame.image;
    vo_akaze_params prm;
    vo_akaze_params_default(&prm);
    int cap = 16384, n = 0;
    for (;;) {
        keypoints->resize((size_t)cap);
        descriptors->d.resize((size_t)cap * VO_AKAZE_DESC_BYTES);
        const int rc = vo_akaze_detect_compute(ctx_, im.px.data(), im.width, im.height, im.width, &prm,
                                               reinterpret_cast<vo_keypoint*>(keypoints->data()),
                                               descriptors->d.data(), cap, &n);
        if (rc == VO_ERR_CAPACITY && n > cap) {  // *n is the true count: once more with room for it
            cap = n;
            continue;
        }
        check(ctx_, rc);
        break;
    }
    keypoints->resize((size_t)n);
    descriptors->rows = n;
    descriptors->cols = VO_AKAZE_DESC_BYTES;
    descriptors->d.resize((size_t)n * VO_AKAZE_DESC_BYTES);
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

// Feature tracks across the window (reference: src/matcher/matcher.cpp:55-86).  The reference's rule, kept
// exactly: an inlier match (q in image a, t in image b) joins the FIRST track, in creation order, whose
// entry for a is q or whose entry for b is t, overwriting both entries; otherwise it starts a track.
// The reference finds that track by scanning all of them (O(matches x tracks)); here every (image, keypoint)
// value keeps the ordered set of tracks that currently hold it, so the first one is a lookup, and an
// overwrite moves the track between sets.  (A plain union-find would not do: overwriting can leave two
// tracks holding the same keypoint, and can drop a keypoint from a track.)
std::vector<std::vector<int>> createMatchMatrix(const std::vector<vo::MatchesInfo> pairwise_matches,
                                                const int pose_count) {
    std::vector<std::vector<int>> tracks;
    std::vector<std::unordered_map<int, std::set<int>>> holders((size_t)std::max(pose_count, 0));
    auto first_holder = [&](int img, int kp) {
        const auto it = holders[img].find(kp);
        return it == holders[img].end() || it->second.empty() ? -1 : *it->second.begin();
    };
    auto set_entry = [&](int row, int img, int kp) {
        int& cell = tracks[row][img];
        if (cell == kp) return;
        if (cell != -1) {
            auto it = holders[img].find(cell);
            it->second.erase(row);
            if (it->second.empty()) holders[img].erase(it);
        }
        cell = kp;
        holders[img][kp].insert(row);
    };
    for (const auto& pw : pairwise_matches) {
        const int a = pw.src_img_idx, b = pw.dst_img_idx;
        if (a < 0 || b < 0 || a >= pose_count || b >= pose_count)
            throw vo::Error(VO_ERR_INVALID, "createMatchMatrix: image index outside pose_count");
        for (size_t i = 0; i < pw.matches.size(); i++) {
            if (pw.inliers_mask[i] != 1) continue;
            const int q = pw.matches[i].queryIdx, t = pw.matches[i].trainIdx;
            const int ra = first_holder(a, q), rb = first_holder(b, t);
            int row = ra < 0 ? rb : (rb < 0 ? ra : std::min(ra, rb));
            if (row < 0) {
                tracks.emplace_back(pose_count, -1);
                row = (int)tracks.size() - 1;
            }
            set_entry(row, a, q);
            set_entry(row, b, t);
        }
    }
    return tracks;
}

// ---- bundle adjustment front end (bundle_adjustment.h) ---------------------------------------------
void BundleAdjustment::init(const vo::Matd& K, size_t max_frames) {
    K_ = K;
    max_frames_ = max_frames;
    count_ = 0;
    features_.clear();
    R_.clear();
    t_.clear();
    pairwise_matches_.clear();
    match_matrix_.clear();
    points_3d_.clear();
    cameras_visible_.clear();
    points_img_.clear();
}

void BundleAdjustment::addKeyFrame(const VOFrame& frame) {
    R_.push_back(frame.pose_R);
    t_.push_back(frame.pose_t);
    vo::ImageFeatures f;
    if (use_akaze_) feature_detector_.detectComputeAKAZE(frame, &f.keypoints, &f.descriptors);  // :41
    else feature_detector_.detectComputeORB(frame, &f.keypoints, &f.descriptors);
    f.img_idx = count_++;
    f.img_size.width = frame.image.width;
    f.img_size.height = frame.image.height;
    features_.push_back(std::move(f));
    if (features_.size() > max_frames_) {
        features_.erase(features_.begin());
        R_.erase(R_.begin());
        t_.erase(t_.begin());
    }
    pairwise_matches_ = matcher(features_, K_);
    match_matrix_ = createMatchMatrix(pairwise_matches_, (int)R_.size());
    setPBAPoints();
}

void BundleAdjustment::setPBAPoints() {
    points_3d_.clear();
    cameras_visible_.clear();
    points_img_.clear();
    struct Track {
        std::vector<int> cams;
        std::vector<vo::Point2f> points;
    };
    // tracks with at least three views, grouped by the two key frames they are triangulated from so that
    // each group is one vo_triangulate call
    std::vector<Track> tracks;
    std::map<std::pair<int, int>, std::vector<int>> groups;
    for (const auto& row : match_matrix_) {
        Track tr;
        for (int cam = 0; cam < (int)row.size(); cam++) {
            if (row[cam] == -1) continue;
            tr.cams.push_back(cam);
            tr.points.push_back(features_[cam].keypoints[row[cam]].pt);
        }
        if (tr.points.size() < 3) continue;
        groups[{tr.cams[0], tr.cams[1]}].push_back((int)tracks.size());
        tracks.push_back(std::move(tr));
    }
    std::vector<vo::Point3d> tri(tracks.size());
    for (const auto& g : groups) {
        const int c0 = g.first.first, c1 = g.first.second;
        std::vector<vo::Point2f> p0, p1;
        for (int i : g.second) {
            p0.push_back(tracks[i].points[0]);
            p1.push_back(tracks[i].points[1]);
        }
        const std::vector<vo::Point3d> x = triangulate(p0, p1, getProjectionMatrix(K_, vo::hconcat(R_[c0], t_[c0])),
                                                       getProjectionMatrix(K_, vo::hconcat(R_[c1], t_[c1])));
        for (size_t k = 0; k < g.second.size(); k++) tri[g.second[k]] = x[k];
    }
    for (size_t i = 0; i < tracks.size(); i++) {
        // the point in the first view's frame: R^T (X - t)
        const vo::Matd& R = R_[tracks[i].cams[0]];
        const vo::Matd& t = t_[tracks[i].cams[0]];
        const double d[3] = {tri[i].x - t(0, 0), tri[i].y - t(1, 0), tri[i].z - t(2, 0)};
        double o[3];
        for (int c = 0; c < 3; c++) o[c] = R(0, c) * d[0] + R(1, c) * d[1] + R(2, c) * d[2];
        const double dist = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
        if (dist < kMax3DDist && o[2] > kMin3DDist && std::fabs(o[0]) < kMax3DWidth) {
            points_3d_.push_back(tri[i]);
            cameras_visible_.push_back(tracks[i].cams);
            points_img_.push_back(tracks[i].points);
        }
    }
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
    // findEssentialMat(LMEDS) and recoverPose (vo_pose.h:14,19) as one submission: the mask recoverPose
    // receives is the one findEssentialMat left, the mask kept is recoverPose's
    int res = 0;
    vo::Matd points_4d(4, n);
    new_frame->local_R = vo::Matd(3, 3);
    new_frame->local_t = vo::Matd(3, 1);
    int rc;
    {
        StageScope ts(5);
        rc = vo_essential_pose(ctx, p0, p1, n, K.d.data(), VO_LMEDS, 0.999, 1.0, ~(uint64_t)0, 1000, kMax3DDist,
                               new_frame->E.d.data(), new_frame->mask.d.data(), new_frame->local_R.d.data(),
                               new_frame->local_t.d.data(), points_4d.d.data(), &res);
    }
    if (rc == VO_OK) {
        StageScope ts(6);
        new_frame->points_3d = points4DToVec(points_4d);
    } else if (rc != VO_ERR_FAILED) {
        check(ctx, rc);
    }
    if (res > (int)kMinPosePoints) {
        new_frame->local_t = -new_frame->local_t;
        new_frame->local_R = new_frame->local_R.t();
        new_frame->local_pose = vo::hconcat(new_frame->local_R, new_frame->local_t);
        {
            StageScope ts(7);
            new_frame->scale = getScale(*last_frame, *new_frame, kMinPosePoints, 200, kMax3DDist);
        }
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

// ---- FeatureTracker ---------------------------------------------------------------------------
#include <atomic>
#include <cstdio>
#include <fstream>
#include <numeric>
#include <sstream>

#include "feature_tracker.h"
#include "kitti.h"
#include "utils.h"
#include "visual_odometry.h"
#include "vo_host_c.h"

uint64_t vo::Image::next_id() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1);
}

FeatureTracker::FeatureTracker(int channels) {
    const char* e = std::getenv("VO_HIP_DEVICE");
    const int rc = vo_ctx_create(e ? std::atoi(e) : 0, nullptr, &ctx_);
    if (rc != VO_OK || !ctx_)
        throw vo::Error(rc, "vo_ctx_create failed: no usable HIP device (there is no CPU fallback)");
    vo_lk_params_default(&lk_);
    lk_.channels = channels;
}

FeatureTracker::~FeatureTracker() {
    if (ctx_) vo_ctx_destroy(ctx_);
}

float FeatureTracker::trackPoints(VOFrame* vo0, VOFrame* vo1) {
    const vo::Image& a = vo0->image;
    const vo::Image& b = vo1->image;
    if (a.width != b.width || a.height != b.height || a.empty())
        throw vo::Error(VO_ERR_INVALID, "trackPoints: images differ in size or are empty");
    const int n = (int)vo0->points.size();
    std::vector<unsigned char> status(n);
    vo1->points.resize(n);
    const bool cached = cached_id_ != 0 && cached_id_ == a.id;
    cached_id_ = 0;
    static_assert(sizeof(vo::Point2f) == 8, "packed points");
    {
        StageScope ts(3);
        check(ctx_, vo_lk_track(ctx_, cached ? nullptr : a.px.data(), b.px.data(), a.width, a.height, a.width,
                                n ? &vo0->points[0].x : nullptr, n, &lk_, n ? &vo1->points[0].x : nullptr,
                                n ? status.data() : nullptr));
    }
    cached_id_ = b.id;
    // Remove bad points
    StageScope ts(4);
    return (float)removePoints(vo0, vo1, &status);
}

// The reference erases from the back of every vector inside the loop; an erase at index i only
// moves elements behind i, which the loop has already visited, so every decision sees original
// data and the result is this stable compaction.  `dists` is sorted first and then erased by the
// ORIGINAL point index: kept as is (utils.h:16,42,57).
namespace {
// Ascending sort of floats that are norms (>= 0, or NaN, which sorts last as a bit pattern): their bit
// patterns order like unsigned integers, so three 11-bit counting passes do it in a fraction of
// std::sort's time for the few thousand points of a frame.  Equal values are indistinguishable, so the
// result is the array std::sort produces.
void sort_nonnegative(std::vector<float>* v) {
    const size_t n = v->size();
    bool plain = n >= 256;
    for (size_t i = 0; i < n && plain; i++) plain = !std::signbit((*v)[i]) && !std::isnan((*v)[i]);
    if (!plain) {
        std::sort(v->begin(), v->end());
        return;
    }
    std::vector<uint32_t> a(n), b(n);
    memcpy(a.data(), v->data(), n * sizeof(float));
    for (int pass = 0; pass < 3; pass++) {
        const int shift = 11 * pass;
        uint32_t count[2049] = {0};
        for (size_t i = 0; i < n; i++) count[((a[i] >> shift) & 2047u) + 1]++;
        for (int k = 0; k < 2048; k++) count[k + 1] += count[k];
        for (size_t i = 0; i < n; i++) b[count[(a[i] >> shift) & 2047u]++] = a[i];
        a.swap(b);
    }
    memcpy(v->data(), a.data(), n * sizeof(float));
}
}  // namespace

int removePoints(VOFrame* vo0, VOFrame* vo1, std::vector<unsigned char>* status) {
    const size_t n = status->size();
    std::vector<float> orig(n);
    for (size_t i = 0; i < n; i++) {
        const float dx = vo0->points[i].x - vo1->points[i].x, dy = vo0->points[i].y - vo1->points[i].y;
        orig[i] = (float)std::sqrt((double)dx * dx + (double)dy * dy);  // cv::norm(Point2f) -> double -> float
    }
    std::vector<float> dists = orig;
    sort_nonnegative(&dists);  // std::sort(dists) in the reference: the sorted ORDER is what the sums below see
    const double sum = std::accumulate(dists.begin(), dists.end(), 0.0);
    const double mean = sum / dists.size();
    const double sq_sum = std::inner_product(dists.begin(), dists.end(), dists.begin(), 0.0);
    const double stdev = std::sqrt(sq_sum / dists.size() - mean * mean);

    std::vector<uint8_t>& mask = vo0->mask.d;
    const size_t n3 = vo0->points_3d.size(), nm = mask.size();
    size_t k = 0, k3 = 0, km = 0;
    std::vector<float> kept;
    kept.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const vo::Point2f pt = vo1->points[i];
        const float dist = orig[i];
        if ((*status)[i] == 0 || pt.x < 0 || pt.y < 0 || std::abs(dist - mean) > stdev * 2) {
            if (pt.x < 0 || pt.y < 0) (*status)[i] = 0;
            continue;
        }
        vo0->points[k] = vo0->points[i];
        vo1->points[k] = pt;
        k++;
        kept.push_back(dists[i]);
        if (i < n3) vo0->points_3d[k3++] = vo0->points_3d[i];
        if (i < nm) mask[km++] = mask[i];
    }
    for (size_t i = n; i < n3; i++) vo0->points_3d[k3++] = vo0->points_3d[i];
    for (size_t i = n; i < nm; i++) mask[km++] = mask[i];
    vo0->points.resize(k);
    vo1->points.resize(k);
    vo0->points_3d.resize(k3);
    mask.resize(km);
    return kept.empty() ? 0 : (int)kept[kept.size() / 2];  // empty: undefined behaviour in the reference
}

// ---- VisualOdometry ---------------------------------------------------------------------------
VisualOdometry::VisualOdometry(const vo::Matd& K, size_t min_tracked_points, bool ba, int tracker_channels)
    : feature_tracker_(tracker_channels) {
    if (ba)
        throw vo::Error(VO_ERR_INVALID,
                        "VisualOdometry: bundle adjustment (GTSAM back end) is not part of the accelerated path");
    tracking_ = false;
    min_tracked_points_ = min_tracked_points;
    K_ = K;
}

void VisualOdometry::addImage(const vo::Image& image, vo::Matd* pose, vo::Matd* /*pose_kalman*/) {
    {
        StageScope ts(1);
        VOFrame frame;
        frame.setImage(image);
        if (frame_buffer_.size() == kFrameBufferCapacity) frame_buffer_.pop_front();  // circular_buffer overwrite
        frame_buffer_.push_back(std::move(frame));
    }
    if (frame_buffer_.size() != kFrameBufferCapacity) return;

    VOFrame& vo1 = frame_buffer_[frame_buffer_.size() - 2];
    VOFrame& vo2 = frame_buffer_[frame_buffer_.size() - 1];

    if (!tracking_) {
        StageScope ts(2);
        feature_detector_.detectFAST(&vo1);
        tracking_ = true;
    }

    if (feature_tracker_.trackPoints(&vo1, &vo2) < 0) {  // never true in the reference either
        frame_buffer_.pop_back();
        return;
    }

    if (vo2.points.size() < min_tracked_points_) tracking_ = false;

    updatePose(K_, &vo1, &vo2);

    (*pose) = vo2.pose;
}

// ---- KITTI formats ------------------------------------------------------------------------------
vo::Matd loadKittiCalibration(std::string calib_file, int line_number) {
    std::string line;
    std::ifstream file_stream(calib_file);
    if (!file_stream.is_open()) throw vo::Error(VO_ERR_INVALID, "Unable to open file " + calib_file);
    int count = 0;
    while (std::getline(file_stream, line)) {
        if (count == line_number) break;
        count++;
    }
    line.erase(0, std::min<size_t>(4, line.size()));
    // boost::algorithm::split(elements, line, is_any_of(" ")): empty tokens are kept and std::stod
    // throws on them, as the reference would
    double data[12] = {0};
    size_t pos = 0;
    int i = 0;
    for (;;) {
        const size_t sp = line.find(' ', pos);
        const std::string tok = line.substr(pos, sp == std::string::npos ? std::string::npos : sp - pos);
        if (i >= 12) throw vo::Error(VO_ERR_INVALID, "calibration line holds more than 12 values");
        try {
            data[i++] = std::stod(tok);
        } catch (const std::exception&) {
            throw vo::Error(VO_ERR_INVALID, "calibration line: '" + tok + "' is not a number");
        }
        if (sp == std::string::npos) break;
        pos = sp + 1;
    }
    vo::Matd K(3, 3);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) K(r, c) = data[4 * r + c];
    return K;
}

vo::Matd kittiResultMat(const vo::Matd& pose) {
    vo::Matd m = vo::Matd::eye(4, 4);
    for (int r = 0; r < pose.rows; r++)
        for (int c = 0; c < pose.cols; c++) m(r, c) = pose(r, c);
    return m;
}

// Ground-truth / result trajectories: text, one pose per line, the first three rows of the 4 x 4
// camera-to-world matrix row-major (README.md:54-94 of the reference describes the layout).  Lines that do
// not hold twelve numbers end the file (a trailing blank line is fine).
std::vector<vo::Matd> loadPoses(const std::string& file_name) {
    std::vector<vo::Matd> poses;
    std::ifstream in(file_name);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream fields(line);
        vo::Matd P = vo::Matd::eye(4, 4);
        int got = 0;
        for (double v; got < 12 && (fields >> v); got++) P(got / 4, got % 4) = v;
        if (got == 0 && line.find_first_not_of(" \t\r") == std::string::npos) continue;  // blank line
        if (got != 12) break;
        poses.push_back(P);
    }
    return poses;
}

void savePoses(const std::string& file_name, const std::vector<vo::Matd>& poses) {
    FILE* fp = fopen(file_name.c_str(), "w");
    if (!fp) throw vo::Error(VO_ERR_INVALID, "Unable to open file " + file_name);
    for (const auto& P : poses) {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) fprintf(fp, "%.9e%s", P(r, c), (r == 2 && c == 3) ? "\n" : " ");
    }
    fclose(fp);
}

// ---- C entry points over the driver ---------------------------------------------------------------
struct vo_host_vo {
    VisualOdometry* vo = nullptr;
    vo::Matd pose = vo::Matd::eye(3, 4);
};

namespace {
void set_err(char* err, int cap, const char* msg) {
    if (err && cap > 0) snprintf(err, cap, "%s", msg);
}
}  // namespace

extern "C" {

vo_host_vo* vo_host_vo_create(const double* K, int min_tracked_points, uint32_t scale_seed, int tracker_channels,
                              char* err, int err_cap) {
    try {
        vo::Matd Km(3, 3);
        for (int i = 0; i < 9; i++) Km.d[i] = K[i];
        vo_host_vo* v = new vo_host_vo();
        v->vo = new VisualOdometry(Km, (size_t)min_tracked_points, false, tracker_channels);
        setScaleSeed(scale_seed);
        return v;
    } catch (const std::exception& e) {
        set_err(err, err_cap, e.what());
        return nullptr;
    }
}

void vo_host_vo_destroy(vo_host_vo* v) {
    if (!v) return;
    delete v->vo;
    delete v;
}

int vo_host_vo_add_image(vo_host_vo* v, const uint8_t* gray, int w, int h, double* pose, char* err, int err_cap) {
    try {
        vo::Image im;
        im.width = w;
        im.height = h;
        {
            StageScope ts(0);
            im.px.assign(gray, gray + (size_t)w * h);
        }
        v->vo->addImage(im, &v->pose, nullptr);
        for (int i = 0; i < 12; i++) pose[i] = v->pose.d[i];
        return 0;
    } catch (const std::exception& e) {
        set_err(err, err_cap, e.what());
        return -1;
    }
}

int vo_host_vo_state(vo_host_vo* v, int* full, int* tracking, int* n_points, int* n_mask, float* scale) {
    *full = v->vo->full() ? 1 : 0;
    *tracking = v->vo->tracking() ? 1 : 0;
    *n_points = *n_mask = 0;
    *scale = 1.f;
    if (!v->vo->full()) return 0;
    const VOFrame& cur = v->vo->currentFrame();
    *n_points = (int)cur.points.size();
    *n_mask = cur.mask.rows();
    *scale = cur.scale;
    return 0;
}

int vo_host_vo_arrays(vo_host_vo* v, float* prev_pts, float* cur_pts, uint8_t* mask, double* E, double* local_R,
                      double* local_t, double* points_3d) {
    if (!v->vo->full()) return -1;
    const VOFrame& prev = v->vo->previousFrame();
    const VOFrame& cur = v->vo->currentFrame();
    if (prev_pts && !prev.points.empty()) memcpy(prev_pts, prev.points.data(), prev.points.size() * sizeof(vo::Point2f));
    if (cur_pts && !cur.points.empty()) memcpy(cur_pts, cur.points.data(), cur.points.size() * sizeof(vo::Point2f));
    if (mask && !cur.mask.d.empty()) memcpy(mask, cur.mask.d.data(), cur.mask.d.size());
    if (E && cur.E.d.size() == 9) memcpy(E, cur.E.d.data(), 9 * sizeof(double));
    if (local_R && cur.local_R.d.size() == 9) memcpy(local_R, cur.local_R.d.data(), 9 * sizeof(double));
    if (local_t && cur.local_t.d.size() == 3) memcpy(local_t, cur.local_t.d.data(), 3 * sizeof(double));
    if (points_3d && !cur.points_3d.empty())
        memcpy(points_3d, cur.points_3d.data(), cur.points_3d.size() * sizeof(vo::Point3d));
    return 0;
}

int vo_host_load_calibration(const char* calib_file, int line_number, double* K, char* err, int err_cap) {
    try {
        const vo::Matd Km = loadKittiCalibration(calib_file, line_number);
        for (int i = 0; i < 9; i++) K[i] = Km.d[i];
        return 0;
    } catch (const std::exception& e) {
        set_err(err, err_cap, e.what());
        return -1;
    }
}

int vo_host_load_poses(const char* file_name, double* poses, int cap, int* n) {
    const auto P = loadPoses(file_name);
    *n = (int)P.size();
    for (int i = 0; i < *n && i < cap; i++)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) poses[12 * i + 4 * r + c] = P[i](r, c);
    return 0;
}

int vo_host_save_poses(const char* file_name, const double* poses, int n) {
    try {
        std::vector<vo::Matd> P;
        for (int i = 0; i < n; i++) {
            vo::Matd m(3, 4);
            for (int k = 0; k < 12; k++) m.d[k] = poses[12 * i + k];
            P.push_back(m);
        }
        savePoses(file_name, P);
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

int vo_host_remove_points(float* p0, float* p1, int n, uint8_t* status, double* pts3d, int n3d, uint8_t* mask,
                          int nmask, int* n_out, int* n3d_out, int* nmask_out, int* ret) {
    VOFrame a, b;
    a.points.resize(n);
    b.points.resize(n);
    if (n) {
        memcpy(static_cast<void*>(a.points.data()), p0, (size_t)n * 8);
        memcpy(static_cast<void*>(b.points.data()), p1, (size_t)n * 8);
    }
    a.points_3d.resize(n3d);
    if (n3d) memcpy(static_cast<void*>(a.points_3d.data()), pts3d, (size_t)n3d * 24);
    a.mask.d.assign(mask, mask + nmask);
    std::vector<unsigned char> st(status, status + n);
    *ret = removePoints(&a, &b, &st);
    *n_out = (int)a.points.size();
    if (*n_out) {
        memcpy(p0, a.points.data(), a.points.size() * 8);
        memcpy(p1, b.points.data(), b.points.size() * 8);
    }
    if (n) memcpy(status, st.data(), n);
    *n3d_out = (int)a.points_3d.size();
    if (*n3d_out) memcpy(pts3d, a.points_3d.data(), a.points_3d.size() * 24);
    *nmask_out = (int)a.mask.d.size();
    if (*nmask_out) memcpy(mask, a.mask.d.data(), a.mask.d.size());
    return 0;
}

// ---- feature tracks and the bundle-adjustment front end -------------------------------------------------
int vo_host_create_match_matrix(const int* src, const int* dst, const int* off, int n_pairs, const int* q,
                                const int* t, const uint8_t* inliers, int pose_count, int* out, int cap_rows,
                                int* n_rows) {
    try {
        std::vector<vo::MatchesInfo> pw((size_t)n_pairs);
        for (int p = 0; p < n_pairs; p++) {
            pw[p].src_img_idx = src[p];
            pw[p].dst_img_idx = dst[p];
            for (int i = off[p]; i < off[p + 1]; i++) {
                vo::DMatch m;
                m.queryIdx = q[i];
                m.trainIdx = t[i];
                pw[p].matches.push_back(m);
                pw[p].inliers_mask.push_back(inliers[i]);
            }
        }
        const auto mm = createMatchMatrix(pw, pose_count);
        *n_rows = (int)mm.size();
        for (int r = 0; r < (int)mm.size() && r < cap_rows; r++)
            for (int c = 0; c < pose_count; c++) out[(size_t)r * pose_count + c] = mm[r][c];
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

struct vo_host_ba {
    BundleAdjustment* ba = nullptr;
};

vo_host_ba* vo_host_ba_create(const double* K, int max_frames, int nfeatures, char* err, int err_cap) {
    return vo_host_ba_create_ex(K, max_frames, 1, nfeatures, err, err_cap);
}

vo_host_ba* vo_host_ba_create_ex(const double* K, int max_frames, int descriptor, int orb_nfeatures, char* err,
                                 int err_cap) {
    try {
        vo_host_ba* b = new vo_host_ba();
        b->ba = new BundleAdjustment(orb_nfeatures, descriptor == 0);
        vo::Matd Km(3, 3);
        for (int i = 0; i < 9; i++) Km.d[i] = K[i];
        b->ba->init(Km, (size_t)max_frames);
        return b;
    } catch (const std::exception& e) {
        set_err(err, err_cap, e.what());
        return nullptr;
    }
}

void vo_host_ba_destroy(vo_host_ba* b) {
    if (!b) return;
    delete b->ba;
    delete b;
}

int vo_host_ba_add_key_frame(vo_host_ba* b, const uint8_t* gray, int w, int h, const double* pose_R,
                             const double* pose_t, char* err, int err_cap) {
    try {
        VOFrame f;
        vo::Image im;
        im.width = w;
        im.height = h;
        im.px.assign(gray, gray + (size_t)w * h);
        f.setImage(std::move(im));
        f.pose_R = vo::Matd(3, 3);
        f.pose_t = vo::Matd(3, 1);
        for (int i = 0; i < 9; i++) f.pose_R.d[i] = pose_R[i];
        for (int i = 0; i < 3; i++) f.pose_t.d[i] = pose_t[i];
        b->ba->addKeyFrame(f);
        return 0;
    } catch (const std::exception& e) {
        set_err(err, err_cap, e.what());
        return -1;
    }
}

int vo_host_ba_counts(vo_host_ba* b, int* n_frames, int* n_pairs, int* n_rows, int* n_points) {
    *n_frames = (int)b->ba->features().size();
    *n_pairs = (int)b->ba->pairwiseMatches().size();
    *n_rows = (int)b->ba->matchMatrix().size();
    *n_points = (int)b->ba->points3D().size();
    return 0;
}

int vo_host_ba_features(vo_host_ba* b, int frame, vo_keypoint* kp, uint8_t* desc, int cap, int* n) {
    const auto& f = b->ba->features()[(size_t)frame];
    *n = (int)f.keypoints.size();
    if (*n > cap) return -1;
    if (kp) memcpy(kp, f.keypoints.data(), (size_t)*n * sizeof(vo_keypoint));
    if (desc) memcpy(desc, f.descriptors.d.data(), (size_t)*n * f.descriptors.cols);
    return 0;
}

int vo_host_ba_desc_bytes(vo_host_ba* b) { return b->ba->descriptorBytes(); }

int vo_host_ba_pair(vo_host_ba* b, int pair, int* src, int* dst, vo_dmatch* matches, uint8_t* inliers, int cap,
                    int* n) {
    const auto& pw = b->ba->pairwiseMatches()[(size_t)pair];
    *src = pw.src_img_idx;
    *dst = pw.dst_img_idx;
    *n = (int)pw.matches.size();
    if (*n > cap) return -1;
    if (matches) memcpy(matches, pw.matches.data(), (size_t)*n * sizeof(vo_dmatch));
    if (inliers) memcpy(inliers, pw.inliers_mask.data(), (size_t)*n);
    return 0;
}

int vo_host_ba_match_matrix(vo_host_ba* b, int* out) {
    const auto& mm = b->ba->matchMatrix();
    const size_t nf = b->ba->features().size();
    for (size_t r = 0; r < mm.size(); r++)
        for (size_t c = 0; c < nf; c++) out[r * nf + c] = mm[r][c];
    return 0;
}

// points: n x 3; visible: n x n_frames (1 where the key frame sees the point); img_pts: n x n_frames x 2
int vo_host_ba_points(vo_host_ba* b, double* points, uint8_t* visible, float* img_pts) {
    const size_t nf = b->ba->features().size();
    const auto& P = b->ba->points3D();
    for (size_t i = 0; i < P.size(); i++) {
        points[3 * i] = P[i].x;
        points[3 * i + 1] = P[i].y;
        points[3 * i + 2] = P[i].z;
        const auto& cams = b->ba->camerasVisible()[i];
        const auto& pts = b->ba->pointsImg()[i];
        for (size_t c = 0; c < nf; c++) visible[i * nf + c] = 0;
        for (size_t k = 0; k < cams.size(); k++) {
            visible[i * nf + cams[k]] = 1;
            img_pts[(i * nf + cams[k]) * 2] = pts[k].x;
            img_pts[(i * nf + cams[k]) * 2 + 1] = pts[k].y;
        }
    }
    return 0;
}

}  // extern "C"
