// vo_geometry.cpp — host side of findEssentialMat / recoverPose / triangulate.
// Replaces the cv calls of matcher() and updatePose() (reference: src/matcher/matcher.cpp:42,
// src/visual_odometry/vo_pose.h:14,19,22, src/sfm/triangulation.cpp:46-66).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <mutex>
#include <thread>

#include "ctx.h"
#include "geom_types.h"

namespace {

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)std::lrint(num / denom);
}

// cv::RNG 5-subsets depend only on the seed and the point count, and the kernels learn a
// pair's point count on the device, so the subsets of every n are tabulated in HBM: uint16 [maxn + 1][iters][5]
// (85 MB at the defaults, 346 MB at 4 096 iterations).  The table is filled ON the device (k_fill_subsets: a
// fraction of a millisecond), so a first call or a change of seed costs one short launch, not a host build
// and a copy.  One table per (device, seed, iters, maxn), shared by the contexts that ask for it.
#define VO_SAMPLE_TABLE_MAX_BYTES ((size_t)2 << 30)
std::shared_ptr<SampleTable> samples_acquire(int device, hipStream_t st, uint64_t seed, int iters, int maxn) {
    static std::mutex mu;
    static std::vector<std::weak_ptr<SampleTable>> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& w : cache)
        if (auto t = w.lock())
            if (t->device == device && t->seed == seed && t->iters >= iters && t->maxn >= maxn) return t;
    const size_t bytes = (size_t)(maxn + 1) * iters * 5 * sizeof(uint16_t);
    if (bytes > VO_SAMPLE_TABLE_MAX_BYTES) return nullptr;
    auto t = std::make_shared<SampleTable>();
    t->device = device;
    t->seed = seed;
    t->iters = iters;
    t->maxn = maxn;
    if (hipMalloc(&t->p, bytes) != hipSuccess) return nullptr;
    launch_fill_subsets(st, seed, iters, maxn, static_cast<uint16_t*>(t->p));
    // other contexts of this device may pick the table up from the cache: it must be complete first
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return nullptr;
    cache.erase(std::remove_if(cache.begin(), cache.end(), [](const std::weak_ptr<SampleTable>& w) { return w.expired(); }),
                cache.end());
    cache.push_back(t);
    return t;
}

}  // namespace

SampleTable::~SampleTable() {
    if (p) {
        (void)hipSetDevice(device);
        (void)hipFree(p);
    }
}

// Sample table + per-pair workspace for npairs problems of up to n_cap points, H hypotheses.
int geom_prepare(vo_ctx* c, int npairs, int n_cap, uint64_t seed, int H) {
    if (n_cap > c->lim.max_points)
        return c->fail(VO_ERR_CAPACITY, "%d points exceed max_points %d", n_cap, c->lim.max_points);
    if (H < 1 || H > 65536) return c->fail(VO_ERR_INVALID, "max_iters %d out of range", H);
    if (!c->samples || c->samples->seed != seed || c->samples->iters < H || c->samples->maxn < c->lim.max_points) {
        VO_HIP(c, hipStreamSynchronize(c->stream));  // the stream may still read the old table
        c->samples = samples_acquire(c->device, c->stream, seed, std::max(H, 1000), c->lim.max_points);
        if (!c->samples)
            return c->fail(VO_ERR_CAPACITY, "RANSAC sample table: (max_points + 1) x max_iters x 10 bytes = %.1f GB "
                           "(limit 2 GB, or the device is out of memory): lower vo_limits.max_points or max_iters",
                           (double)(c->lim.max_points + 1) * std::max(H, 1000) * 10 / 1e9);
    }
    const size_t np = (size_t)npairs;
    VO_TRY(vo_ensure(c, c->d_x1, np * n_cap * sizeof(double2)));
    VO_TRY(vo_ensure(c, c->d_x2, np * n_cap * sizeof(double2)));
    VO_TRY(vo_ensure(c, c->d_models, np * H * 90 * sizeof(double)));
    VO_TRY(vo_ensure(c, c->d_nmodels, np * H * sizeof(int)));
    VO_TRY(vo_ensure(c, c->d_counts, np * H * 10 * sizeof(double)));
    VO_TRY(vo_ensure(c, c->d_state, np * sizeof(EstState)));
    VO_TRY(vo_ensure(c, c->d_mask, np * n_cap));
    VO_TRY(vo_ensure(c, c->d_cam, np * sizeof(CamK)));
    return VO_OK;
}

// Essential matrix for npairs point sets resident on the device (points of pair p at
// d_p0 + p*n_cap).  Leaves EstState[p] (E, status, n_inliers) and d_mask on the device.
// The camera table (the same K for every pair) stays on the device between calls; it is sent again only
// when K changes or more pairs are needed.  The caller has sized d_cam.
static int geom_upload_cam(vo_ctx* c, int npairs, const double* K) {
    const double k4[4] = {K[0], K[4], K[2], K[5]};
    if (c->d_cam.p == c->cam_ptr && npairs <= c->cam_pairs && memcmp(k4, c->cam_cache, sizeof(k4)) == 0)
        return VO_OK;  // (a regrown d_cam is a new allocation: its pointer differs)
    VO_TRY(vo_ensure_pinned(c, c->h_misc, std::max((size_t)npairs * sizeof(CamK), (size_t)4096)));
    CamK* hk = c->h_misc.as<CamK>();
    for (int p = 0; p < npairs; p++) hk[p] = {K[0], K[4], K[2], K[5]};
    VO_HIP(c, hipMemcpyAsync(c->d_cam.p, hk, (size_t)npairs * sizeof(CamK), hipMemcpyHostToDevice, c->stream));
    memcpy(c->cam_cache, k4, sizeof(k4));
    c->cam_pairs = npairs;
    c->cam_ptr = c->d_cam.p;
    return VO_OK;
}

int geom_find_essential_dev(vo_ctx* c, int npairs, int n_cap, const float2* d_p0, const float2* d_p1,
                            const int* d_n, int n_host, const double* K, int method, double prob,
                            double threshold, int max_iters) {
    hipStream_t st = c->stream;
    const double fx = K[0], fy = K[4];
    const double thr = threshold / ((fx + fy) / 2);
    const float thr_f = (float)(thr * thr);
    const int H = std::max(max_iters, 1);
    int niters = H;
    if (method == VO_LMEDS) niters = std::min(H, std::max(ransac_update_num_iters(prob, 0.45, 5, H), 3));
    // camera (same K for every pair)
    VO_TRY(geom_upload_cam(c, npairs, K));

    EstState* S = c->d_state.as<EstState>();
    double2* x1 = c->d_x1.as<double2>();
    double2* x2 = c->d_x2.as<double2>();
    double* models = c->d_models.as<double>();
    int* nmodels = c->d_nmodels.as<int>();
    const uint16_t* samples = reinterpret_cast<const uint16_t*>(c->samples->p);
    const int n_grid = d_n ? n_cap : n_host;

    launch_est_init(st, S, d_n, n_host, n_cap, niters, thr_f, npairs);
    launch_normalize(st, d_p0, d_p1, c->d_cam.as<CamK>(), S, nullptr, 0, n_cap, n_grid, x1, x2, npairs);
    if (method == VO_RANSAC) {
        // hypotheses are evaluated in growing chunks; a chunk whose first index is already past
        // the (shrinking) iteration cap exits at once, so typical inputs run only the first one
        // or two, with the same result as evaluating all of them.
        // Every round costs three launches even when it exits at once, so the rounds grow by 4x: 32, 128, 512,
        // 328 for cv's 1000.  The first round is what every pair pays in full: with the matcher's ratio test in
        // front, most pairs reach their iteration cap within twenty hypotheses, and the ones solved and scored
        // beyond it are wasted fp64 work beside the other chunks' kernels (first round 64: 146.5 k pairs/s, 32:
        // 149.7 k, 24: the same, 16: 148.0 k on one box; 128 measured 30 % slower in round 2).
        // A call for one or two pairs is a latency measurement, not a throughput one (BASELINE configs[1]): the chip
        // is idle, 1 000 hypotheses solve and score in the time 64 do, and every extra round is three more dependent
        // launches, so they all go out at once.  Which hypothesis wins does not depend on the rounds: k_replay applies
        // cv's sequential rule to the counts in hypothesis order and ignores those past the (shrinking) cap.
        int h0 = 0, size = npairs <= 2 ? std::min(H, 1024) : 32;
        while (h0 < H) {
            const int h1 = std::min(H, h0 + size);
            {
                ProfScope ps(c, VO_PROF_SOLVE);
                launch_solve5(st, h0, h1, S, samples, c->samples->iters, x1, x2, n_cap, models, nmodels, H, npairs, c->opt_solver);
            }
            launch_score(st, h0, h1, S, x1, x2, n_cap, models, nmodels, c->d_counts.as<int>(), H, npairs);
            launch_replay(st, h0, h1, S, nmodels, c->d_counts.as<int>(), H, prob, npairs);
            h0 = h1;
            size *= 4;
        }
    } else {
        {
            ProfScope ps(c, VO_PROF_SOLVE);
            launch_solve5(st, 0, niters, S, samples, c->samples->iters, x1, x2, n_cap, models, nmodels, H, npairs, c->opt_solver);
        }
        launch_lmeds(st, S, x1, x2, n_cap, models, nmodels, c->d_counts.as<double>(), H, niters, npairs);
    }
    launch_est_finish(st, S, x1, x2, n_cap, models, c->d_mask.as<uint8_t>(), H, npairs);
    return vo_check_launch(c, "findEssentialMat");
}

// recoverPose for npairs point sets; E of pair p at d_E + p*e_stride.  Uses d_x1/d_x2 if
// `normalized` (left by geom_find_essential_dev), else normalises d_p0/d_p1 first.
int geom_recover_pose_dev(vo_ctx* c, int npairs, int n_cap, const double* d_E, size_t e_stride,
                          const float2* d_p0, const float2* d_p1, const int* d_n, int n_host, const double* K,
                          double dist_thresh, bool use_mask) {
    hipStream_t st = c->stream;
    const size_t np = (size_t)npairs;
    VO_TRY(vo_ensure(c, c->d_q4, np * 16 * n_cap * sizeof(double)));
    VO_TRY(vo_ensure(c, c->d_cmask, np * 4 * n_cap));
    VO_TRY(vo_ensure(c, c->d_pose, np * sizeof(PoseState)));
    VO_TRY(vo_ensure(c, c->d_pmask, np * n_cap));
    VO_TRY(vo_ensure(c, c->d_pts4, np * 4 * n_cap * sizeof(double)));
    const int n_grid = d_n ? n_cap : n_host;
    if (d_p0) {
        VO_TRY(geom_upload_cam(c, npairs, K));
        launch_normalize(st, d_p0, d_p1, c->d_cam.as<CamK>(), nullptr, d_n, n_host, n_cap, n_grid,
                         c->d_x1.as<double2>(), c->d_x2.as<double2>(), npairs);
    }
    launch_pose(st, d_E, e_stride, c->d_pose.as<PoseState>(), c->d_x1.as<double2>(), c->d_x2.as<double2>(), d_n,
                n_host, n_cap, n_grid, use_mask ? c->d_mask.as<uint8_t>() : nullptr, dist_thresh,
                c->d_q4.as<double>(), c->d_cmask.as<uint8_t>(), c->d_pmask.as<uint8_t>(), c->d_pts4.as<double>(),
                npairs);
    return vo_check_launch(c, "recoverPose");
}

static int upload_points(vo_ctx* c, const float* p0, const float* p1, int n) {
    VO_TRY(vo_ensure(c, c->d_fp0, (size_t)std::max(n, 1) * sizeof(float2)));
    VO_TRY(vo_ensure(c, c->d_fp1, (size_t)std::max(n, 1) * sizeof(float2)));
    if (n > 0) {
        VO_TRY(vo_xfer_in(c, c->d_fp0.p, p0, (size_t)n * sizeof(float2)));
        VO_TRY(vo_xfer_in(c, c->d_fp1.p, p1, (size_t)n * sizeof(float2)));
    }
    return VO_OK;
}

extern "C" {

int vo_find_essential(vo_ctx* c, const float* p0, const float* p1, int n, const double* K, int method,
                      double prob, double threshold, uint64_t seed, int max_iters, double* E, uint8_t* mask) {
    if (!c) return VO_ERR_INVALID;
    if (n < 0 || !K || !E || (n > 0 && (!p0 || !p1)) || !(prob > 0 && prob < 1) ||
        (method != VO_RANSAC && method != VO_LMEDS) || max_iters < 1)
        return c->fail(VO_ERR_INVALID, "vo_find_essential: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    for (int i = 0; i < 9; i++) E[i] = 0;
    if (mask && n > 0) memset(mask, 0, n);
    if (n < 5) return c->fail(VO_ERR_FAILED, "vo_find_essential: %d points, 5 needed", n);
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(geom_prepare(c, 1, n, seed, max_iters));
    VO_TRY(upload_points(c, p0, p1, n));
    VO_TRY(geom_find_essential_dev(c, 1, n, c->d_fp0.as<float2>(), c->d_fp1.as<float2>(), nullptr, n, K, method,
                                   prob, threshold, max_iters));
    VO_TRY(vo_ensure_pinned(c, c->h_state, sizeof(EstState)));
    VO_HIP(c, hipMemcpyAsync(c->h_state.p, c->d_state.p, sizeof(EstState), hipMemcpyDeviceToHost, c->stream));
    if (mask) VO_TRY(vo_xfer_out(c, mask, c->d_mask.p, n));
    VO_TRY(vo_xfer_finish(c));
    const EstState* s = c->h_state.as<EstState>();
    if (s->status != VO_OK) {
        if (mask) memset(mask, 0, n);
        return c->fail(VO_ERR_FAILED, "vo_find_essential: no model found");
    }
    memcpy(E, s->E, 9 * sizeof(double));
    return VO_OK;
}

int vo_recover_pose(vo_ctx* c, const double* E, const float* p0, const float* p1, int n, const double* K,
                    double dist_thresh, uint8_t* mask_inout, double* R, double* t, double* pts4d, int* n_good) {
    if (!c) return VO_ERR_INVALID;
    if (n < 0 || !E || !K || !R || !t || (n > 0 && (!p0 || !p1)))
        return c->fail(VO_ERR_INVALID, "vo_recover_pose: bad argument");
    if (n > c->lim.max_points)
        return c->fail(VO_ERR_CAPACITY, "vo_recover_pose: %d points exceed max_points %d", n, c->lim.max_points);
    VO_HIP(c, hipSetDevice(c->device));
    const int n_cap = std::max(n, 1);
    VO_TRY(vo_ensure(c, c->d_x1, (size_t)n_cap * sizeof(double2)));
    VO_TRY(vo_ensure(c, c->d_x2, (size_t)n_cap * sizeof(double2)));
    VO_TRY(vo_ensure(c, c->d_mask, (size_t)n_cap));
    VO_TRY(vo_ensure(c, c->d_cam, sizeof(CamK)));
    VO_TRY(vo_ensure(c, c->d_proj, 24 * sizeof(double)));
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(upload_points(c, p0, p1, n));
    VO_TRY(vo_xfer_in(c, c->d_proj.p, E, 9 * sizeof(double)));
    if (mask_inout && n > 0) VO_TRY(vo_xfer_in(c, c->d_mask.p, mask_inout, n));
    VO_TRY(geom_recover_pose_dev(c, 1, n_cap, c->d_proj.as<double>(), 0, c->d_fp0.as<float2>(),
                                 c->d_fp1.as<float2>(), nullptr, n, K, dist_thresh, mask_inout != nullptr));
    VO_TRY(vo_ensure_pinned(c, c->h_pose, sizeof(PoseState)));
    VO_HIP(c, hipMemcpyAsync(c->h_pose.p, c->d_pose.p, sizeof(PoseState), hipMemcpyDeviceToHost, c->stream));
    if (mask_inout && n > 0) VO_TRY(vo_xfer_out(c, mask_inout, c->d_pmask.p, n));
    if (pts4d && n > 0)
        for (int k = 0; k < 4; k++)
            VO_TRY(vo_xfer_out(c, pts4d + (size_t)k * n, c->d_pts4.as<double>() + (size_t)k * n_cap,
                               (size_t)n * sizeof(double)));
    VO_TRY(vo_xfer_finish(c));
    const PoseState* s = c->h_pose.as<PoseState>();
    memcpy(R, s->R, 9 * sizeof(double));
    memcpy(t, s->tt, 3 * sizeof(double));
    if (n_good) *n_good = s->n_good;
    return VO_OK;
}

// findEssentialMat followed by recoverPose on the same points, as updatePose() calls them (reference:
// src/visual_odometry/vo_pose.h:14,19), in ONE submission: the points go up once, E and the inlier mask stay on
// the device between the two, and the host waits once.  Same results as vo_find_essential + vo_recover_pose with
// the mask passed through.
int vo_essential_pose(vo_ctx* c, const float* p0, const float* p1, int n, const double* K, int method, double prob,
                      double threshold, uint64_t seed, int max_iters, double dist_thresh, double* E, uint8_t* mask,
                      double* R, double* t, double* pts4d, int* n_good) {
    if (!c) return VO_ERR_INVALID;
    if (n < 0 || !K || !E || !R || !t || (n > 0 && (!p0 || !p1)) || !(prob > 0 && prob < 1) ||
        (method != VO_RANSAC && method != VO_LMEDS) || max_iters < 1)
        return c->fail(VO_ERR_INVALID, "vo_essential_pose: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    for (int i = 0; i < 9; i++) E[i] = 0;
    if (mask && n > 0) memset(mask, 0, n);
    if (n_good) *n_good = 0;
    if (n < 5) return c->fail(VO_ERR_FAILED, "vo_essential_pose: %d points, 5 needed", n);
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(geom_prepare(c, 1, n, seed, max_iters));
    VO_TRY(upload_points(c, p0, p1, n));
    VO_TRY(geom_find_essential_dev(c, 1, n, c->d_fp0.as<float2>(), c->d_fp1.as<float2>(), nullptr, n, K, method,
                                   prob, threshold, max_iters));
    const double* d_E = reinterpret_cast<const double*>(c->d_state.as<char>() + offsetof(EstState, E));
    VO_TRY(geom_recover_pose_dev(c, 1, n, d_E, 0, nullptr, nullptr, nullptr, n, K, dist_thresh, true));
    VO_TRY(vo_ensure_pinned(c, c->h_state, sizeof(EstState)));
    VO_TRY(vo_ensure_pinned(c, c->h_pose, sizeof(PoseState)));
    VO_HIP(c, hipMemcpyAsync(c->h_state.p, c->d_state.p, sizeof(EstState), hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipMemcpyAsync(c->h_pose.p, c->d_pose.p, sizeof(PoseState), hipMemcpyDeviceToHost, c->stream));
    if (mask) VO_TRY(vo_xfer_out(c, mask, c->d_pmask.p, n));
    if (pts4d) VO_TRY(vo_xfer_out(c, pts4d, c->d_pts4.p, (size_t)4 * n * sizeof(double)));  // n_cap == n: rows adjoin
    VO_TRY(vo_xfer_finish(c));
    const EstState* s = c->h_state.as<EstState>();
    if (s->status != VO_OK) {
        if (mask) memset(mask, 0, n);
        if (pts4d) memset(pts4d, 0, (size_t)4 * n * sizeof(double));
        return c->fail(VO_ERR_FAILED, "vo_essential_pose: no model found");
    }
    memcpy(E, s->E, 9 * sizeof(double));
    const PoseState* ps = c->h_pose.as<PoseState>();
    memcpy(R, ps->R, 9 * sizeof(double));
    memcpy(t, ps->tt, 3 * sizeof(double));
    if (n_good) *n_good = ps->n_good;
    return VO_OK;
}

// vo_essential_pose for n_seq independent point sets in one submission (the lock-step host of several sequences,
// vo_host_vo_run_concurrent): the estimator and pose kernels already take a batch of pairs with per-pair counts
// (the descriptor path runs them on 48 pairs at a time), so this is packing, one upload, the same launches with
// npairs = n_seq, and one download per table.  Per-pair results are those of vo_essential_pose: a pair's kernels
// only ever read that pair's rows, and the row stride does not enter any arithmetic.
int vo_essential_pose_batch(vo_ctx* c, int n_seq, const float* const* p0, const float* const* p1, const int* n,
                            const double* K, int method, double prob, double threshold, uint64_t seed, int max_iters,
                            double dist_thresh, double* E, uint8_t* const* mask, double* R, double* t,
                            double* const* pts4d, int* n_good, int* status) {
    if (!c) return VO_ERR_INVALID;
    if (n_seq < 1 || n_seq > 64 || !p0 || !p1 || !n || !K || !E || !R || !t || !status || !(prob > 0 && prob < 1) ||
        (method != VO_RANSAC && method != VO_LMEDS) || max_iters < 1)
        return c->fail(VO_ERR_INVALID, "vo_essential_pose_batch: bad argument (at most 64 point sets a call)");
    int nmax = 0;
    for (int s = 0; s < n_seq; s++) {
        if (n[s] < 0 || (n[s] > 0 && (!p0[s] || !p1[s])))
            return c->fail(VO_ERR_INVALID, "vo_essential_pose_batch: bad argument for point set %d", s);
        nmax = std::max(nmax, n[s]);
    }
    VO_HIP(c, hipSetDevice(c->device));
    for (int s = 0; s < n_seq; s++) {
        for (int i = 0; i < 9; i++) E[9 * s + i] = 0;
        if (mask && mask[s] && n[s] > 0) memset(mask[s], 0, n[s]);
        if (n_good) n_good[s] = 0;
        status[s] = VO_ERR_FAILED;
    }
    if (nmax < 5) return VO_OK;  // every set is cv's empty-matrix case
    const int n_cap = (nmax + 63) & ~63;
    const size_t np = (size_t)n_seq, tot = np * n_cap;
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(geom_prepare(c, n_seq, n_cap, seed, max_iters));
    VO_TRY(vo_ensure(c, c->d_fp0, tot * sizeof(float2)));
    VO_TRY(vo_ensure(c, c->d_fp1, tot * sizeof(float2)));
    VO_TRY(vo_ensure(c, c->d_ngood, (np + 1) * sizeof(int)));
    // pinned staging: [p0 | p1 | counts] up, [EstState | PoseState | mask | pts4] down
    const size_t up_bytes = 2 * tot * sizeof(float2) + np * sizeof(int);
    const size_t dn_at = (up_bytes + 255) & ~(size_t)255;
    const size_t st_bytes = np * sizeof(EstState), po_bytes = np * sizeof(PoseState);
    const size_t mask_at = dn_at + ((st_bytes + po_bytes + 255) & ~(size_t)255);
    const size_t pts_at = mask_at + ((tot + 255) & ~(size_t)255);
    VO_TRY(vo_ensure_pinned(c, c->h_pts, pts_at + 4 * tot * sizeof(double)));
    char* hp = c->h_pts.as<char>();
    float2* h0 = reinterpret_cast<float2*>(hp);
    float2* h1 = h0 + tot;
    int* hn = reinterpret_cast<int*>(h1 + tot);
    for (int s = 0; s < n_seq; s++) {
        hn[s] = n[s];
        if (n[s]) {
            memcpy(h0 + (size_t)s * n_cap, p0[s], (size_t)n[s] * sizeof(float2));
            memcpy(h1 + (size_t)s * n_cap, p1[s], (size_t)n[s] * sizeof(float2));
        }
    }
    VO_HIP(c, hipMemcpyAsync(c->d_fp0.p, h0, tot * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipMemcpyAsync(c->d_fp1.p, h1, tot * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipMemcpyAsync(c->d_ngood.p, hn, np * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const int* d_n = c->d_ngood.as<int>();
    VO_TRY(geom_find_essential_dev(c, n_seq, n_cap, c->d_fp0.as<float2>(), c->d_fp1.as<float2>(), d_n, 0, K, method, prob,
                                   threshold, max_iters));
    const double* d_E = reinterpret_cast<const double*>(c->d_state.as<char>() + offsetof(EstState, E));
    VO_TRY(geom_recover_pose_dev(c, n_seq, n_cap, d_E, sizeof(EstState) / sizeof(double), nullptr, nullptr, d_n, 0, K,
                                 dist_thresh, true));
    EstState* hs = reinterpret_cast<EstState*>(hp + dn_at);
    PoseState* hq = reinterpret_cast<PoseState*>(hp + dn_at + st_bytes);
    VO_HIP(c, hipMemcpyAsync(hs, c->d_state.p, st_bytes, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipMemcpyAsync(hq, c->d_pose.p, po_bytes, hipMemcpyDeviceToHost, c->stream));
    if (mask) VO_HIP(c, hipMemcpyAsync(hp + mask_at, c->d_pmask.p, tot, hipMemcpyDeviceToHost, c->stream));
    if (pts4d)
        VO_HIP(c, hipMemcpyAsync(hp + pts_at, c->d_pts4.p, 4 * tot * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    VO_TRY(vo_xfer_finish(c));
    const double* hx = reinterpret_cast<const double*>(hp + pts_at);
    for (int s = 0; s < n_seq; s++) {
        if (hs[s].status != VO_OK) {  // n < 5, or no model: E and mask stay zero, R and t untouched
            if (pts4d && pts4d[s] && n[s] > 0) memset(pts4d[s], 0, (size_t)4 * n[s] * sizeof(double));
            continue;
        }
        status[s] = VO_OK;
        memcpy(E + 9 * s, hs[s].E, 9 * sizeof(double));
        memcpy(R + 9 * s, hq[s].R, 9 * sizeof(double));
        memcpy(t + 3 * s, hq[s].tt, 3 * sizeof(double));
        if (n_good) n_good[s] = hq[s].n_good;
        if (mask && mask[s]) memcpy(mask[s], hp + mask_at + (size_t)s * n_cap, n[s]);
        if (pts4d && pts4d[s])
            for (int k = 0; k < 4; k++)
                memcpy(pts4d[s] + (size_t)k * n[s], hx + ((size_t)s * 4 + k) * n_cap, (size_t)n[s] * sizeof(double));
    }
    return VO_OK;
}

int vo_five_point(vo_ctx* c, const double* x1, const double* x2, int count, int variant, double* models,
                  int* nmodels) {
    if (!c) return VO_ERR_INVALID;
    if (count < 0 || (variant != 0 && variant != 1) || (count > 0 && (!x1 || !x2 || !models || !nmodels)))
        return c->fail(VO_ERR_INVALID, "vo_five_point: bad argument");
    if (count > (1 << 20)) return c->fail(VO_ERR_CAPACITY, "vo_five_point: %d problems, at most %d a call", count, 1 << 20);
    if (count == 0) return VO_OK;
    VO_HIP(c, hipSetDevice(c->device));
    const size_t in_bytes = (size_t)count * 10 * sizeof(double);
    VO_TRY(vo_ensure(c, c->d_x1, in_bytes));
    VO_TRY(vo_ensure(c, c->d_x2, in_bytes));
    VO_TRY(vo_ensure(c, c->d_models, (size_t)count * 90 * sizeof(double)));
    VO_TRY(vo_ensure(c, c->d_nmodels, (size_t)count * sizeof(int)));
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(vo_xfer_in(c, c->d_x1.p, x1, in_bytes));
    VO_TRY(vo_xfer_in(c, c->d_x2.p, x2, in_bytes));
    VO_HIP(c, hipMemsetAsync(c->d_models.p, 0, (size_t)count * 90 * sizeof(double), c->stream));
    launch_five_point_batch(c->stream, count, variant == 1, c->d_x1.as<double>(), c->d_x2.as<double>(),
                            c->d_models.as<double>(), c->d_nmodels.as<int>());
    VO_TRY(vo_check_launch(c, "five_point"));
    VO_TRY(vo_xfer_out(c, models, c->d_models.p, (size_t)count * 90 * sizeof(double)));
    VO_TRY(vo_xfer_out(c, nmodels, c->d_nmodels.p, (size_t)count * sizeof(int)));
    return vo_xfer_finish(c);
}

int vo_triangulate(vo_ctx* c, const float* p0, const float* p1, int n, const double* P0, const double* P1,
                   double* pts3d) {
    if (!c) return VO_ERR_INVALID;
    if (n < 0 || !P0 || !P1 || (n > 0 && (!p0 || !p1 || !pts3d)))
        return c->fail(VO_ERR_INVALID, "vo_triangulate: bad argument");
    if (n == 0) return VO_OK;  // the reference returns an empty vector (triangulation.cpp:49-51)
    VO_HIP(c, hipSetDevice(c->device));
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(upload_points(c, p0, p1, n));
    VO_TRY(vo_ensure(c, c->d_proj, 24 * sizeof(double)));
    VO_TRY(vo_ensure(c, c->d_pts4, (size_t)n * 3 * sizeof(double)));
    double P[24];
    memcpy(P, P0, 12 * sizeof(double));
    memcpy(P + 12, P1, 12 * sizeof(double));
    VO_TRY(vo_xfer_in(c, c->d_proj.p, P, sizeof(P)));
    launch_triangulate(c->stream, c->d_proj.as<double>(), c->d_fp0.as<float2>(), c->d_fp1.as<float2>(), n,
                       c->d_pts4.as<double>());
    VO_TRY(vo_check_launch(c, "triangulate"));
    VO_TRY(vo_xfer_out(c, pts3d, c->d_pts4.p, (size_t)n * 3 * sizeof(double)));
    return vo_xfer_finish(c);
}

}  // extern "C"
