// host_batch.cpp — BatchGroup: the device calls of several drivers sent down as one batch (host_batch.h).
#include "host_batch.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace vo {

namespace {
thread_local BatchGroup* t_group = nullptr;
thread_local int t_slot = -1;
}  // namespace

BatchGroup* current_group() { return t_group; }
int current_slot() { return t_slot; }
void set_current_group(BatchGroup* g, int slot) {
    t_group = g;
    t_slot = slot;
}

BatchGroup::BatchGroup(int members) : req_((size_t)members), active_((size_t)members, 1), n_active_(members) {
    const char* e = std::getenv("VO_HIP_DEVICE");
    vo_limits lim;
    vo_limits_default(&lim);
    const int rc = vo_ctx_create(e ? std::atoi(e) : 0, &lim, &ctx_);
    if (rc != VO_OK || !ctx_) throw Error(rc, "vo_ctx_create failed: no usable HIP device (there is no CPU fallback)");
}

uint8_t* BatchGroup::image_slot(int slot, int w, int h) {
    std::lock_guard<std::mutex> lock(mu_);
    const size_t bytes = (size_t)w * h;
    if (!arena_) {  // sized by the first request; a group tracks images of one size
        arena_ = static_cast<uint8_t*>(vo_host_alloc_pinned(bytes * req_.size()));
        arena_image_ = arena_ ? bytes : 0;
    }
    return arena_ && bytes == arena_image_ ? arena_ + (size_t)slot * bytes : nullptr;
}

BatchGroup::~BatchGroup() {
    vo_host_free_pinned(arena_);
    if (std::getenv("VO_HOST_TIMING") && n_track_)
        std::fprintf(stderr, "[vo_host batch] %ld steps: vo_lk_track_batch %.1f us, vo_essential_pose_batch %.1f us per step\n",
                     n_track_, t_track_ / n_track_ * 1e6, n_pose_ ? t_pose_ / n_pose_ * 1e6 : 0.0);
    if (ctx_) vo_ctx_destroy(ctx_);
}

void BatchGroup::leave(int slot) {
    std::unique_lock<std::mutex> lock(mu_);
    if (slot < 0 || slot >= (int)active_.size() || !active_[(size_t)slot]) return;
    active_[(size_t)slot] = 0;
    n_active_--;
    // the others may all be waiting for this member
    if (n_active_ > 0 && n_waiting_ == n_active_) {
        run_locked();
        generation_++;
        n_waiting_ = 0;
        cv_.notify_all();
    }
}

int BatchGroup::rendezvous(int slot, const Request& rq) {
    std::unique_lock<std::mutex> lock(mu_);
    if (slot < 0 || slot >= (int)req_.size() || !active_[(size_t)slot]) return VO_ERR_INVALID;
    req_[(size_t)slot] = rq;
    n_waiting_++;
    if (n_waiting_ == n_active_) {  // the last to arrive makes the call for everybody
        run_locked();
        generation_++;
        n_waiting_ = 0;
        cv_.notify_all();
    } else {
        const unsigned long gen = generation_;
        cv_.wait(lock, [&] { return generation_ != gen; });
    }
    const int rc = req_[(size_t)slot].rc;
    req_[(size_t)slot].kind = 0;
    return rc;
}

void BatchGroup::run_locked() {
    std::vector<int> slots;
    int kind = 0;
    bool mixed = false;
    for (int s = 0; s < (int)req_.size(); s++)
        if (active_[(size_t)s] && req_[(size_t)s].kind) {
            if (kind && req_[(size_t)s].kind != kind) mixed = true;
            kind = req_[(size_t)s].kind;
            slots.push_back(s);
        }
    if (slots.empty()) return;
    const int S = (int)slots.size();
    auto fail_all = [&](int rc, const char* msg) {
        err_ = msg;
        for (int s : slots) req_[(size_t)s].rc = rc;
    };
    if (mixed) return fail_all(VO_ERR_INVALID, "BatchGroup: the drivers of a group must stay in lock step");
    const Request& r0 = req_[(size_t)slots[0]];
    if (kind == 1) {
        std::vector<const uint8_t*> prev(S), next(S);
        std::vector<const float*> p0(S);
        std::vector<float*> p1(S);
        std::vector<uint8_t*> st(S);
        std::vector<int> n(S);
        bool cached = slots == last_track_slots_;  // same members in the same batch order as the pyramids on the device
        float dummy_pt[2] = {0, 0};
        uint8_t dummy_st = 0;
        for (int i = 0; i < S; i++) {
            const Request& r = req_[(size_t)slots[i]];
            if (r.w != r0.w || r.h != r0.h || memcmp(r.prm, r0.prm, sizeof(vo_lk_params)) != 0)
                return fail_all(VO_ERR_INVALID, "BatchGroup: the drivers of a group must share image size and tracker parameters");
            cached = cached && r.prev_cached;
            prev[i] = r.prev;
            next[i] = r.next;
            n[i] = r.n;
            p0[i] = r.n ? r.p0 : dummy_pt;
            p1[i] = r.n ? r.p1 : dummy_pt;
            st[i] = r.n ? r.status : &dummy_st;
        }
        last_track_slots_.clear();
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = vo_lk_track_batch(ctx_, S, cached ? nullptr : prev.data(), next.data(), r0.w, r0.h, r0.w, p0.data(),
                                         n.data(), r0.prm, p1.data(), st.data());
        t_track_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        n_track_++;
        if (rc != VO_OK) return fail_all(rc, vo_last_error(ctx_));
        last_track_slots_ = slots;
        for (int s : slots) req_[(size_t)s].rc = VO_OK;
        return;
    }
    std::vector<const float*> p0(S), p1(S);
    std::vector<int> n(S), n_good(S), status(S);
    std::vector<uint8_t*> mask(S);
    std::vector<double*> pts(S);
    std::vector<double> E((size_t)S * 9), R((size_t)S * 9), t((size_t)S * 3);
    for (int i = 0; i < S; i++) {
        const Request& r = req_[(size_t)slots[i]];
        if (memcmp(r.K, r0.K, 9 * sizeof(double)) != 0 || r.method != r0.method || r.prob != r0.prob ||
            r.threshold != r0.threshold || r.seed != r0.seed || r.max_iters != r0.max_iters ||
            r.dist_thresh != r0.dist_thresh)
            return fail_all(VO_ERR_INVALID, "BatchGroup: the drivers of a group must share K and the estimator parameters");
        p0[i] = r.p0;
        p1[i] = r.p1in;
        n[i] = r.n;
        mask[i] = r.mask;
        pts[i] = r.pts4d;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = vo_essential_pose_batch(ctx_, S, p0.data(), p1.data(), n.data(), r0.K, r0.method, r0.prob, r0.threshold,
                                           r0.seed, r0.max_iters, r0.dist_thresh, E.data(), mask.data(), R.data(), t.data(),
                                           pts.data(), n_good.data(), status.data());
    t_pose_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    n_pose_++;
    if (rc != VO_OK) return fail_all(rc, vo_last_error(ctx_));
    for (int i = 0; i < S; i++) {
        Request& r = req_[(size_t)slots[i]];
        // as vo_essential_pose: E always written (zero on failure), R / t / n_good only on success
        memcpy(r.E, &E[(size_t)i * 9], 9 * sizeof(double));
        if (r.n_good) *r.n_good = n_good[i];
        if (status[i] == VO_OK) {
            memcpy(r.R, &R[(size_t)i * 9], 9 * sizeof(double));
            memcpy(r.t, &t[(size_t)i * 3], 3 * sizeof(double));
        }
        r.rc = status[i];
    }
}

int BatchGroup::lk_track(int slot, const uint8_t* prev, bool prev_cached, const uint8_t* next, int w, int h,
                         const float* p0, int n, const vo_lk_params* prm, float* p1, uint8_t* status) {
    Request r;
    r.kind = 1;
    r.prev = prev;
    r.prev_cached = prev_cached;
    r.next = next;
    r.w = w;
    r.h = h;
    r.n = n;
    r.p0 = p0;
    r.p1 = p1;
    r.status = status;
    r.prm = prm;
    return rendezvous(slot, r);
}

int BatchGroup::essential_pose(int slot, const float* p0, const float* p1, int n, const double* K, int method,
                               double prob, double threshold, uint64_t seed, int max_iters, double dist_thresh, double* E,
                               uint8_t* mask, double* R, double* t, double* pts4d, int* n_good) {
    Request r;
    r.kind = 2;
    r.p0 = p0;
    r.p1in = p1;
    r.n = n;
    r.K = K;
    r.method = method;
    r.prob = prob;
    r.threshold = threshold;
    r.seed = seed;
    r.max_iters = max_iters;
    r.dist_thresh = dist_thresh;
    r.E = E;
    r.mask = mask;
    r.R = R;
    r.t = t;
    r.pts4d = pts4d;
    r.n_good = n_good;
    return rendezvous(slot, r);
}

}  // namespace vo
