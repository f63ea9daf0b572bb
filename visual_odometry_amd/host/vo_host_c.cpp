// vo_host_c.cpp — C entry points over the host mirror (vo_host_c.h), for ctypes and C callers.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>

#include <memory>

#include "bundle_adjustment.h"
#include "host_batch.h"
#include "host_internal.h"
#include "kitti.h"
#include "matcher.h"
#include "utils.h"
#include "visual_odometry.h"
#include "vo_host_c.h"

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

int vo_host_vo_run_concurrent(const double* K, int min_tracked_points, uint32_t scale_seed, int tracker_channels,
                              const uint8_t* const* frames, const int* n_frames, int n_seq, int w, int h, int warmup,
                              int lockstep, double* const* poses, double* wall_seconds, long* timed_frames, char* err,
                              int err_cap) {
    if (!K || !frames || !n_frames || n_seq < 1 || n_seq > 64 || w <= 0 || h <= 0 || warmup < 0) {
        set_err(err, err_cap, "vo_host_vo_run_concurrent: bad argument");
        return -1;
    }
    vo::Matd Km(3, 3);
    for (int i = 0; i < 9; i++) Km.d[i] = K[i];
    // lockstep = G > 0: G groups, driver s in group s mod G (slot s / G).  One group sends every driver's calls in one
    // batch but leaves the GPU idle while the drivers do their host work; with two groups one's batch runs while the
    // other's drivers are on the host.
    const int G = std::max(0, std::min(lockstep, n_seq));
    std::vector<std::unique_ptr<vo::BatchGroup>> groups;
    try {
        for (int g = 0; g < G; g++) groups.emplace_back(new vo::BatchGroup((n_seq - g + G - 1) / G));
    } catch (const std::exception& e) {
        set_err(err, err_cap, e.what());
        return -1;
    }
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    std::string first_error;
    using clk = std::chrono::steady_clock;
    clk::time_point t_start;
    std::vector<clk::time_point> t_end((size_t)n_seq);
    long timed = 0;
    auto run = [&](int s) {
        vo::BatchGroup* const group = G ? groups[(size_t)(s % G)].get() : nullptr;
        const int slot = G ? s / G : -1;
        vo::set_current_group(group, slot);  // (nullptr: every driver makes its own device calls)
        try {
            setScaleSeed(scale_seed + (uint32_t)s);  // this thread's generator
            VisualOdometry vo(Km, (size_t)min_tracked_points, false, tracker_channels);
            vo::Matd pose = vo::Matd::eye(3, 4);
            const size_t img = (size_t)w * h;
            auto feed = [&](int f) {
                vo::Image im;
                im.width = w;
                im.height = h;
                im.px.assign(frames[s] + (size_t)f * img, frames[s] + (size_t)(f + 1) * img);
                vo.addImage(im, &pose, nullptr);
                if (poses && poses[s])
                    for (int i = 0; i < 12; i++) poses[s][(size_t)f * 12 + i] = pose.d[i];
            };
            const int nw = std::min(warmup, n_frames[s]);
            for (int f = 0; f < nw; f++) feed(f);
            {
                std::unique_lock<std::mutex> lock(mu);
                if (++arrived == n_seq) {
                    t_start = clk::now();
                    cv.notify_all();
                } else {
                    cv.wait(lock, [&] { return arrived >= n_seq; });
                }
                timed += n_frames[s] - nw;
            }
            for (int f = nw; f < n_frames[s]; f++) feed(f);
            if (group) group->leave(slot);
        } catch (const std::exception& e) {
            if (group) group->leave(slot);  // the others must not wait for a driver that has stopped
            std::lock_guard<std::mutex> lock(mu);
            if (first_error.empty()) first_error = e.what();
            if (arrived < n_seq) {  // never leave the others waiting at the barrier
                arrived = n_seq;
                t_start = clk::now();
                cv.notify_all();
            }
        }
        vo::set_current_group(nullptr, -1);
        t_end[(size_t)s] = clk::now();
    };
    std::vector<std::thread> th;
    for (int s = 0; s < n_seq; s++) th.emplace_back(run, s);
    for (auto& t : th) t.join();
    if (!first_error.empty()) {
        set_err(err, err_cap, first_error.c_str());
        return -1;
    }
    clk::time_point last = t_end[0];
    for (const auto& t : t_end) last = std::max(last, t);
    if (wall_seconds) *wall_seconds = std::chrono::duration<double>(last - t_start).count();
    if (timed_frames) *timed_frames = timed;
    return 0;
}

void vo_host_timing_enable(int on) { g_timers.on = on != 0; }

void vo_host_timing_reset(void) {
    for (int i = 0; i < StageTimers::N; i++) {
        g_timers.sec[i] = 0;
        g_timers.calls[i] = 0;
    }
}

int vo_host_timing_get(int i, const char** name, long* calls, double* us_each) {
    if (i >= 0 && i < StageTimers::N) {
        if (name) *name = g_timers.names[i];
        if (calls) *calls = g_timers.calls[i];
        if (us_each) *us_each = g_timers.calls[i] ? g_timers.sec[i] / g_timers.calls[i] * 1e6 : 0.0;
    }
    return StageTimers::N;
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
