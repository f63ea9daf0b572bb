// vo_pipeline.cpp — the device-resident batched frame-pair path: ORB on every frame of the batch,
// then per consecutive pair knnMatch(k=2) + ratio/displacement filter + findEssentialMat(RANSAC)
// (the body of matcher(), reference: src/matcher/matcher.cpp:13-50) and recoverPose
// (reference: src/visual_odometry/vo_pose.h:19).  Frames stay in HBM; the host sees one
// synchronisation for the keypoint selection (vo_features.cpp) and one for the results.
#include <algorithm>
#include <cstdarg>
#include <chrono>
#include <cstddef>

#include "ctx.h"
#include "geom_types.h"

int match_knn2_dev(vo_ctx* c, const uint32_t* d_q, const uint32_t* d_t, size_t q_stride, size_t t_stride,
                   const int* d_n, int nq, int nt, int words, int npairs, int nq_cap, bool cross);
int match_filter_dev(vo_ctx* c, const vo_keypoint* d_kq, const vo_keypoint* d_kt, size_t kp_stride,
                     const int* d_n, int nq, int npairs, int nq_cap, float ratio, float max_disp, bool cross);



// Everything after keypoint selection for the batch, asynchronous up to the final copy of the
// results; max_n bounds the per-frame keypoint count (launch sizes, per-pair strides).
static int pairs_after_selection(vo_ctx* c, int n_frames, int max_n, const double* K, const vo_pair_params* prm) {
    const int npairs = n_frames - 1;
    const int n_cap = std::max(max_n, 8);
    if (n_cap > c->lim.max_points)
        return c->fail(VO_ERR_CAPACITY, "%d keypoints per frame exceed max_points %d", n_cap, c->lim.max_points);
    const size_t kp_cap = (size_t)c->kp_cap;
    VO_TRY(match_knn2_dev(c, c->d_desc.as<uint32_t>(), c->d_desc.as<uint32_t>() + kp_cap * 8, kp_cap * 8,
                          kp_cap * 8, c->d_nkp.as<int>(), max_n, max_n, 8, npairs, n_cap, prm->cross_check != 0));
    VO_TRY(match_filter_dev(c, c->d_kp.as<vo_keypoint>(), c->d_kp.as<vo_keypoint>() + kp_cap, kp_cap,
                            c->d_nkp.as<int>(), max_n, npairs, n_cap, prm->ratio, prm->max_disp, prm->cross_check != 0));
    VO_TRY(geom_prepare(c, npairs, n_cap, prm->seed, prm->max_iters));
    VO_TRY(geom_find_essential_dev(c, npairs, n_cap, c->d_p0.as<float2>(), c->d_p1.as<float2>(),
                                   c->d_ngood.as<int>(), 0, K, VO_RANSAC, prm->prob, prm->threshold,
                                   prm->max_iters));
    const double* d_E = reinterpret_cast<const double*>(c->d_state.as<char>() + offsetof(EstState, E));
    VO_TRY(geom_recover_pose_dev(c, npairs, n_cap, d_E, sizeof(EstState) / sizeof(double), nullptr, nullptr,
                                 c->d_ngood.as<int>(), 0, K, prm->dist_thresh, true));
    VO_TRY(vo_ensure_pinned(c, c->h_state, (size_t)npairs * sizeof(EstState)));
    VO_TRY(vo_ensure_pinned(c, c->h_pose, (size_t)npairs * sizeof(PoseState)));
    VO_TRY(vo_ensure_pinned(c, c->h_good, (size_t)npairs * sizeof(int)));
    // One copy instead of five (eight with the detail tables): the result tables (estimator and pose states,
    // match counts, keypoint counts, selection flags) and, for vo_seq_run_ex, the per-pair tables the
    // reference's matcher() / updatePose() hand back (MatchesInfo::matches, inliers_mask, recoverPose's mask;
    // fixed stride n_cap, pairs_collect copies the used rows out) are gathered into one device buffer and
    // cross the bus together: every separate copy is a launch of its own at the end of the chunk's stream.
    PackArgs pa;
    const size_t rows = c->pend_details ? (size_t)npairs * n_cap : 0;
    const void* src[VO_PACK_TABLES] = {c->d_state.p, c->d_pose.p, c->d_ngood.p, c->d_nkp.p,
                                       c->d_flags.p, c->d_good.p, c->d_mask.p, c->d_pmask.p};
    const size_t bytes[VO_PACK_TABLES] = {(size_t)npairs * sizeof(EstState), (size_t)npairs * sizeof(PoseState),
                                          (size_t)npairs * sizeof(int), (size_t)n_frames * sizeof(int),
                                          (size_t)n_frames * sizeof(int), rows * sizeof(GoodMatch), rows, rows};
    int at = 0;
    for (int i = 0; i < VO_PACK_TABLES; i++) {
        pa.src[i] = static_cast<const uint32_t*>(src[i]);
        pa.ndw[i] = (int)((bytes[i] + 3) / 4);  // a last partial dword lies inside the allocation (256-byte units)
        pa.at[i] = at;
        c->pack_at[i] = (size_t)at * 4;
        c->pack_bytes[i] = bytes[i];
        at += (pa.ndw[i] + 3) & ~3;
    }
    VO_TRY(vo_ensure(c, c->d_pack, (size_t)at * 4));
    VO_TRY(vo_ensure_pinned(c, c->h_pack, (size_t)at * 4));
    launch_pack_results(c->stream, pa, at, c->d_pack.as<uint32_t>());
    VO_TRY(vo_check_launch(c, "result gather"));
    VO_HIP(c, hipMemcpyAsync(c->h_pack.p, c->d_pack.p, (size_t)at * 4, hipMemcpyDeviceToHost, c->stream));
    c->pipe_ncap = n_cap;
    return VO_OK;
}

// after the synchronisation: the gathered tables back into the pinned buffers the rest of the code reads
static void pairs_unpack(vo_ctx* c) {
    void* dst[5] = {c->h_state.p, c->h_pose.p, c->h_good.p, c->h_nkp.p, c->h_flags.p};
    for (int i = 0; i < 5; i++) memcpy(dst[i], c->h_pack.as<char>() + c->pack_at[i], c->pack_bytes[i]);
}

// Queue the whole batch on the context's stream without a host round trip: ORB with the
// device-side keypoint selection, matching, essential matrix, pose, result copies.
int pairs_submit(vo_ctx* c, const uint8_t* d_frames, int n_frames, int w, int h, int stride, size_t frame_stride,
                 const double* K, const vo_pair_params* prm, bool details, bool carry) {
    if (!d_frames || n_frames < 2 || w <= 0 || h <= 0 || stride < w || frame_stride < (size_t)stride * h || !K ||
        !prm)
        return c->fail(VO_ERR_INVALID, "vo_pairs_run_dev: bad argument");
    if (prm->orb.nfeatures > c->lim.max_features)
        return c->fail(VO_ERR_CAPACITY, "nfeatures %d exceeds max_features %d", prm->orb.nfeatures,
                       c->lim.max_features);
    VO_HIP(c, hipSetDevice(c->device));
    const int prev_frames = c->pipe_frames;  // frames of the batch this context collected last
    c->pipe_frames = 0;
    c->pend_frames = 0;
    c->pend_details = details;
    if (carry && (prev_frames < 2 || c->geom_w != w || c->geom_h != h))
        return c->fail(VO_ERR_INVALID, "pairs_submit: no frame to carry over");
    const void *kp_before = c->d_kp.p, *desc_before = c->d_desc.p;
    const int cap_before = c->kp_cap;
    VO_TRY(feat_setup(c, w, h, &prm->orb, n_frames));
    if (carry && (c->d_kp.p != kp_before || c->d_desc.p != desc_before || c->kp_cap != cap_before))
        return c->fail(VO_ERR_INVALID, "pairs_submit: the tables of the frame to carry over were reallocated");
    const int n_bound = feat_n_bound(c, prm->orb.nfeatures);
    // Consecutive chunks of a sequence share a frame.  A context that runs them one after the other still holds
    // the keypoints and descriptors of that frame (final ones: the chunk before was collected, host redo
    // included): they move to row 0 of the tables and the detector starts at frame 1.
    const int off = carry ? 1 : 0;
    if (carry) {
        launch_carry_frame(c->stream, c->d_kp.as<vo_keypoint>(), c->d_desc.as<uint8_t>(), c->d_nkp.as<int>(), c->kp_cap,
                           prev_frames - 1);
        c->feat_off = 1;
    }
    {
        ProfScope ps(c, VO_PROF_FEATURES);
        VO_TRY(feat_detect(c, d_frames, frame_stride, stride, n_frames - off, prm->orb.fast_threshold,
                           prm->orb.edge_threshold, 1, false));
        VO_TRY(feat_select_dev(c, n_frames - off, n_bound));
        VO_TRY(feat_describe(c, n_frames - off, n_bound, true, false));
    }
    VO_TRY(pairs_after_selection(c, n_frames, n_bound, K, prm));
    memcpy(c->pend_K, K, sizeof(c->pend_K));
    c->pend_prm = *prm;
    c->pend_frames = n_frames;
    return VO_OK;
}

// Wait for the batch queued by pairs_submit and fill results[n_frames - 1]; with `out`, also the matches and
// masks of pair p of the batch into row pair0 + p of the caller's tables.
int pairs_collect(vo_ctx* c, vo_pair_result* results, const vo_seq_outputs* out, size_t pair0) {
    const int n_frames = c->pend_frames;
    if (n_frames < 2 || !results) return c->fail(VO_ERR_INVALID, "no batch is pending");
    VO_HIP(c, hipSetDevice(c->device));
    c->pend_frames = 0;
    const int npairs = n_frames - 1;
    VO_HIP(c, hipStreamSynchronize(c->stream));
    pairs_unpack(c);
    int flagged = 0;
    const int nf_feat = n_frames - c->feat_off;  // frames the feature stage ran on (flags are per such frame)
    for (int f = 0; f < nf_feat; f++) flagged |= c->h_flags.as<int>()[f];
    c->last_flags = flagged;
    if (flagged) {
        // a frame the device selection could not finish exactly (VO_SEL_*): host selection, redo
        int max_n = 0;
        VO_TRY(feat_select_host(c, nf_feat, &max_n));
        if (c->feat_off) max_n = std::max(max_n, c->h_nkp.as<int>()[0]);  // the carried frame's count
        if (max_n > 0) VO_TRY(feat_describe(c, nf_feat, max_n, true, true));
        VO_TRY(pairs_after_selection(c, n_frames, max_n, c->pend_K, &c->pend_prm));
        VO_HIP(c, hipStreamSynchronize(c->stream));
        pairs_unpack(c);
    }
    c->pipe_nkp.assign(c->h_nkp.as<int>(), c->h_nkp.as<int>() + n_frames);

    const EstState* es = c->h_state.as<EstState>();
    const PoseState* ps = c->h_pose.as<PoseState>();
    const int* ng = c->h_good.as<int>();
    c->pipe_ngood.assign(ng, ng + npairs);
    for (int p = 0; p < npairs; p++) {
        vo_pair_result& r = results[p];
        r.n_kp0 = c->pipe_nkp[p];
        r.n_kp1 = c->pipe_nkp[p + 1];
        r.n_good = ng[p];
        r.status = es[p].status;
        r.n_inliers = es[p].status == VO_OK ? es[p].n_inliers : 0;
        r.n_pose = es[p].status == VO_OK ? ps[p].n_good : 0;
        memcpy(r.E, es[p].E, sizeof(r.E));
        if (es[p].status == VO_OK) {
            memcpy(r.R, ps[p].R, sizeof(r.R));
            memcpy(r.t, ps[p].tt, sizeof(r.t));
        } else {
            memset(r.R, 0, sizeof(r.R));
            memset(r.t, 0, sizeof(r.t));
        }
    }
    c->pipe_frames = n_frames;
    if (out && c->pend_details) {
        const size_t n_cap = (size_t)c->pipe_ncap;
        const size_t cap = (size_t)out->match_cap;
        // the detail tables are read where they arrived, in the gathered buffer
        const char* hp = c->h_pack.as<char>();
        const GoodMatch* h_good = reinterpret_cast<const GoodMatch*>(hp + c->pack_at[5]);
        const uint8_t* h_mask = reinterpret_cast<const uint8_t*>(hp + c->pack_at[6]);
        const uint8_t* h_pmask = reinterpret_cast<const uint8_t*>(hp + c->pack_at[7]);
        for (int p = 0; p < npairs; p++) {
            const size_t cnt = (size_t)ng[p];
            if (cnt > cap)
                return c->fail(VO_ERR_CAPACITY, "pair %zu has %zu matches, match_cap is %zu", pair0 + p, cnt, cap);
            const size_t at = (pair0 + p) * cap;
            if (out->matches) memcpy(out->matches + at, h_good + p * n_cap, cnt * sizeof(GoodMatch));
            if (out->inlier_mask) memcpy(out->inlier_mask + at, h_mask + p * n_cap, cnt);
            if (out->pose_mask) memcpy(out->pose_mask + at, h_pmask + p * n_cap, cnt);
        }
    }
    return VO_OK;
}

// ---------------------------------------------------------------------------------------------
// Sequence runner: a long sequence cut into chunks of `chunk_pairs` frame pairs (consecutive chunks
// share one frame), a run of consecutive chunks per context (seq_run_chunks).  Every context has its own HIP stream and
// workspace, so the latency-bound stages of one chunk (keypoint selection, five-point solves,
// RANSAC replay) run while the other chunks' streaming stages use the rest of the chip, and the
// upload of a chunk overlaps the kernels of the chunks before it.
// ---------------------------------------------------------------------------------------------
struct vo_seq {
    int device = 0;
    int chunk_pairs = 32;
    std::vector<vo_ctx*> ctx;
    std::vector<PinnedBuf> stage;  // per context: pinned staging of one chunk (pageable input)
    // Pinned, tightly packed host frames: a context owns two device image buffers and a copy stream, so
    // the upload of its next chunk runs while the current chunk computes instead of ahead of it on the
    // same stream.
    struct Prefetch {
        void* img[2] = {nullptr, nullptr};
        size_t cap = 0;
        hipStream_t copy = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        int chunk[2] = {-1, -1};  // chunk whose frames buffer b holds (or is receiving)
    };
    std::vector<Prefetch> pre;
    double submit_ms = 0, wait_ms = 0;  // host time of the last vo_seq_run spent queueing / waiting
    int chunks = 0;
    char err[512];
    int fail(int code, const char* fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, sizeof(err), fmt, ap);
        va_end(ap);
        return code;
    }
    int from(vo_ctx* c, int rc) {
        if (rc != VO_OK) snprintf(err, sizeof(err), "%s", c->err);
        return rc;
    }
};

extern "C" {

int vo_pairs_run_dev(vo_ctx* c, const uint8_t* d_frames, int n_frames, int w, int h, int stride,
                     size_t frame_stride, const double* K, const vo_pair_params* prm, vo_pair_result* results) {
    if (!c) return VO_ERR_INVALID;
    if (!results) return c->fail(VO_ERR_INVALID, "vo_pairs_run_dev: bad argument");
    VO_TRY(pairs_submit(c, d_frames, n_frames, w, h, stride, frame_stride, K, prm, false));
    return pairs_collect(c, results, nullptr, 0);
}

int vo_seq_create(int device, const vo_limits* limits, int n_streams, int pairs_per_chunk, vo_seq** out) {
    if (!out) return VO_ERR_INVALID;
    *out = nullptr;
    if (n_streams < 1 || n_streams > 16 || pairs_per_chunk < 1) return VO_ERR_INVALID;
    vo_limits lim;
    if (limits) lim = *limits;
    else vo_limits_default(&lim);
    lim.max_batch = std::max(lim.max_batch, pairs_per_chunk + 1);
    vo_seq* s = new vo_seq();
    s->device = device;
    s->chunk_pairs = pairs_per_chunk;
    s->err[0] = 0;
    s->stage.resize(n_streams);
    s->pre.resize(n_streams);
    for (int i = 0; i < n_streams; i++) {
        vo_ctx* c = nullptr;
        const int rc = vo_ctx_create(device, &lim, &c);
        if (rc != VO_OK) {
            vo_seq_destroy(s);
            return rc;
        }
        s->ctx.push_back(c);
    }
    *out = s;
    return VO_OK;
}

void vo_seq_destroy(vo_seq* s) {
    if (!s) return;
    for (size_t i = 0; i < s->ctx.size(); i++) {
        (void)hipSetDevice(s->device);
        (void)hipStreamSynchronize(s->ctx[i]->stream);
        if (s->stage[i].p) (void)hipHostFree(s->stage[i].p);
        vo_seq::Prefetch& P = s->pre[i];
        if (P.copy) {
            (void)hipStreamSynchronize(P.copy);
            (void)hipStreamDestroy(P.copy);
        }
        for (int b = 0; b < 2; b++) {
            if (P.ev[b]) (void)hipEventDestroy(P.ev[b]);
            if (P.img[b]) (void)hipFree(P.img[b]);
        }
        vo_ctx_destroy(s->ctx[i]);
    }
    delete s;
}

const char* vo_seq_last_error(const vo_seq* s) { return s ? s->err : "null sequence runner"; }

int vo_seq_streams(const vo_seq* s) { return s ? (int)s->ctx.size() : 0; }

int vo_seq_get_stats(const vo_seq* s, int* chunks, double* submit_ms, double* wait_ms) {
    if (!s) return VO_ERR_INVALID;
    if (chunks) *chunks = s->chunks;
    if (submit_ms) *submit_ms = s->submit_ms;
    if (wait_ms) *wait_ms = s->wait_ms;
    return VO_OK;
}

vo_ctx* vo_seq_ctx(vo_seq* s, int i) { return (s && i >= 0 && i < (int)s->ctx.size()) ? s->ctx[i] : nullptr; }

int vo_seq_run(vo_seq* s, const uint8_t* frames, int mem, int n_frames, int w, int h, int stride,
               size_t frame_stride, const double* K, const vo_pair_params* prm, vo_pair_result* results) {
    return vo_seq_run_ex(s, frames, mem, n_frames, w, h, stride, frame_stride, K, prm, results, nullptr);
}

}  // extern "C"

namespace {
// One chunk of frame pairs of some sequence: what the runner queues, uploads and collects.
struct ChunkDesc {
    const uint8_t* src;         // first frame of the chunk
    int nf;                     // frames (pairs + 1)
    vo_pair_result* res;        // results of its first pair
    const vo_seq_outputs* out;  // per-pair tables of its sequence (NULL: none) ...
    size_t pair0;               // ... and the row of its first pair in them
};

// The pipeline over a list of chunks, at most S in flight, one per context.  The list is cut into S runs of
// consecutive chunks, run r on context r: a context then meets the chunks of a sequence one after the other, and
// the frame that two of them share is described once -- its keypoints are carried over on the device
// (pairs_submit) instead of being computed again by the next chunk (1 / 49 of the detector's work at 48 pairs a
// chunk, and of the upload).  Step i of the loop submits the (i / S)-th chunk of run i mod S, so the S contexts
// are fed in turn as before; chunks of different sequences follow each other without draining the pipeline.
int seq_run_chunks(vo_seq* s, const std::vector<ChunkDesc>& ch, int mem, int w, int h, int stride, size_t frame_stride,
                   const double* K, const vo_pair_params* prm) {
    const int S = (int)s->ctx.size(), B = s->chunk_pairs;
    const int nchunks = (int)ch.size();
    const size_t img = (size_t)w * h;
    int rc = VO_OK;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    s->submit_ms = s->wait_ms = 0;
    s->chunks = nchunks;
    for (auto& P : s->pre) P.chunk[0] = P.chunk[1] = -1;  // nothing of this run is on the device yet
    const bool prefetch = s->ctx[0]->opt_seq_prefetch != 0;
    std::vector<int> rb((size_t)S + 1, 0);  // run r = chunks [rb[r], rb[r + 1])
    for (int r = 0; r < S; r++) rb[(size_t)r + 1] = rb[(size_t)r] + nchunks / S + (r < nchunks % S ? 1 : 0);
    const int steps = S * (nchunks / S + (nchunks % S ? 1 : 0));
    auto chunk_of = [&](int step) {
        if (step < 0 || step >= steps) return -1;
        const int k = rb[(size_t)(step % S)] + step / S;
        return k < rb[(size_t)(step % S) + 1] ? k : -1;
    };
    // chunk k starts with the last frame of chunk k - 1 of its run
    auto carries = [&](int step) {
        const int k = chunk_of(step);
        if (k < 0 || step < S) return false;
        const ChunkDesc &a = ch[(size_t)k - 1], &b = ch[(size_t)k];
        return b.src == a.src + (size_t)(a.nf - 1) * frame_stride;
    };
    int collected = 0;  // steps [0, collected) have their results
    auto collect = [&](int step) {
        const int k = chunk_of(step);
        if (k < 0) return (int)VO_OK;
        vo_ctx* c = s->ctx[(size_t)(step % S)];
        if (c->pend_frames < 2) return (int)VO_OK;
        const auto t0 = clk::now();
        const int r = s->from(c, pairs_collect(c, ch[(size_t)k].res, ch[(size_t)k].out, ch[(size_t)k].pair0));
        s->wait_ms += ms_since(t0);
        return r;
    };
    for (int step = 0; step < steps && rc == VO_OK; step++) {
        const int slot = step % S;
        vo_ctx* c = s->ctx[(size_t)slot];
        if (step - S >= collected) {  // the context still holds the chunk of step - S
            rc = collect(step - S);
            collected = step - S + 1;
            if (rc != VO_OK) break;
        }
        const int k = chunk_of(step);
        if (k < 0) continue;
        const bool details = ch[(size_t)k].out != nullptr;
        const int nf = ch[(size_t)k].nf;
        const bool carry = carries(step) && c->pipe_frames == ch[(size_t)k - 1].nf;
        if (carries(step) && !carry) {
            rc = s->fail(VO_ERR_INVALID, "chunk %d: the context does not hold the chunk before it", k);
            break;
        }
        const int off = carry ? 1 : 0;  // frames of the chunk that are not sent / described again
        const uint8_t* src = ch[(size_t)k].src + (size_t)off * frame_stride;
        const auto t_submit = clk::now();
        struct Acc {
            vo_seq* s;
            clk::time_point t0;
            ~Acc() { s->submit_ms += std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
        } acc{s, t_submit};
        if (mem == VO_MEM_DEVICE) {
            rc = s->from(c, pairs_submit(c, src, nf, w, h, stride, frame_stride, K, prm, details, carry));
            continue;
        }
        // Pinned frames go up on the chunk's own stream in front of its kernels: the copies of the other chunks
        // overlap them, and no further streams compete for the process's hardware queues (108 k pairs/s against
        // 87 k with four chunks in flight).  VO_OPT_SEQ_PREFETCH = 1 (taken from the runner's first context at the
        // start of the run): the earlier scheme, every context sends its next chunk ahead of time on a copy stream
        // of its own into a second image buffer.
        if (mem == VO_MEM_HOST_PINNED && stride == w && frame_stride == img && prefetch) {
            vo_seq::Prefetch& P = s->pre[(size_t)slot];
            const size_t need = (size_t)(B + 1) * img;
            hipError_t pe = hipSuccess;
            if (!P.copy) pe = hipStreamCreateWithFlags(&P.copy, hipStreamNonBlocking);
            for (int b = 0; b < 2 && pe == hipSuccess; b++)
                if (!P.ev[b]) pe = hipEventCreateWithFlags(&P.ev[b], hipEventDisableTiming);
            if (pe == hipSuccess && P.cap < need) {
                (void)hipStreamSynchronize(P.copy);
                (void)hipStreamSynchronize(c->stream);
                for (int b = 0; b < 2 && pe == hipSuccess; b++) {
                    if (P.img[b]) (void)hipFree(P.img[b]);
                    P.img[b] = nullptr;
                    P.chunk[b] = -1;
                    pe = hipMalloc(&P.img[b], need);
                }
                P.cap = pe == hipSuccess ? need : 0;
            }
            auto upload = [&](int st, int b) {  // the frames the chunk of step st will describe
                const int kk = chunk_of(st);
                const int o = carries(st) ? 1 : 0;
                hipError_t e2 = hipMemcpyAsync(P.img[b], ch[(size_t)kk].src + (size_t)o * img,
                                               (size_t)(ch[(size_t)kk].nf - o) * img, hipMemcpyHostToDevice, P.copy);
                if (e2 == hipSuccess) e2 = hipEventRecord(P.ev[b], P.copy);
                P.chunk[b] = kk;
                return e2;
            };
            const int b = (step / S) & 1;
            if (pe == hipSuccess && P.chunk[b] != k) pe = upload(step, b);  // first chunks of a run: nothing prefetched
            if (pe == hipSuccess) pe = hipStreamWaitEvent(c->stream, P.ev[b], 0);
            if (pe != hipSuccess) {
                rc = s->fail(VO_ERR_DEVICE, "frame upload failed: %s", hipGetErrorString(pe));
                break;
            }
            rc = s->from(c, pairs_submit(c, static_cast<const uint8_t*>(P.img[b]), nf, w, h, w, img, K, prm, details,
                                         carry));
            // the other buffer held the chunk of step - S, which has been collected: send the next one into it now
            if (rc == VO_OK && chunk_of(step + S) >= 0 && upload(step + S, b ^ 1) != hipSuccess)
                rc = s->fail(VO_ERR_DEVICE, "frame prefetch failed");
            continue;
        }
        // host frames: tightly packed copy of the chunk into the context's image buffer
        if ((rc = s->from(c, vo_ensure(c, c->d_img, (size_t)(B + 1) * img))) != VO_OK) break;
        hipError_t e = hipSuccess;
        const int nsend = nf - off;
        if (mem == VO_MEM_HOST) {
            // pageable memory: stage through pinned memory so that the copy is one asynchronous DMA
            if ((rc = s->from(c, vo_ensure_pinned(c, s->stage[(size_t)slot], (size_t)(B + 1) * img))) != VO_OK) break;
            uint8_t* st = s->stage[(size_t)slot].as<uint8_t>();
            for (int f = 0; f < nsend; f++)
                for (int y = 0; y < h; y++)
                    memcpy(st + (size_t)f * img + (size_t)y * w, src + (size_t)f * frame_stride + (size_t)y * stride, w);
            e = hipMemcpyAsync(c->d_img.p, st, (size_t)nsend * img, hipMemcpyHostToDevice, c->stream);
        } else if (stride == w && frame_stride == img) {
            e = hipMemcpyAsync(c->d_img.p, src, (size_t)nsend * img, hipMemcpyHostToDevice, c->stream);
        } else {
            for (int f = 0; f < nsend && e == hipSuccess; f++)
                e = hipMemcpy2DAsync(c->d_img.as<uint8_t>() + (size_t)f * img, w, src + (size_t)f * frame_stride, stride,
                                     w, h, hipMemcpyHostToDevice, c->stream);
        }
        if (e != hipSuccess) {
            rc = s->fail(VO_ERR_DEVICE, "frame upload failed: %s", hipGetErrorString(e));
            break;
        }
        rc = s->from(c, pairs_submit(c, c->d_img.as<uint8_t>(), nf, w, h, w, img, K, prm, details, carry));
    }
    if (rc != VO_OK)
        for (auto& P : s->pre)
            if (P.copy) (void)hipStreamSynchronize(P.copy);  // a prefetch nobody will wait for
    // drain in order (also after an error, so that no context is left with work in flight)
    for (int step = collected; step < steps; step++) {
        vo_ctx* c = s->ctx[(size_t)(step % S)];
        if (chunk_of(step) < 0 || c->pend_frames < 2) continue;
        if (rc == VO_OK) rc = collect(step);
        else {
            (void)hipStreamSynchronize(c->stream);
            c->pend_frames = 0;
        }
    }
    return rc;
}

// the chunks of one sequence appended to `ch`
void add_sequence(std::vector<ChunkDesc>& ch, int B, const uint8_t* frames, int n_frames, size_t frame_stride,
                  vo_pair_result* results, const vo_seq_outputs* out) {
    const int npairs = n_frames - 1;
    for (int p0 = 0; p0 < npairs; p0 += B) {
        ChunkDesc d;
        d.src = frames + (size_t)p0 * frame_stride;
        d.nf = std::min(B, npairs - p0) + 1;
        d.res = results + p0;
        d.out = out;
        d.pair0 = (size_t)p0;
        ch.push_back(d);
    }
}
}  // namespace

extern "C" {

int vo_seq_run_ex(vo_seq* s, const uint8_t* frames, int mem, int n_frames, int w, int h, int stride,
                  size_t frame_stride, const double* K, const vo_pair_params* prm, vo_pair_result* results,
                  const vo_seq_outputs* out) {
    if (!s) return VO_ERR_INVALID;
    if (out && (out->match_cap <= 0 || (!out->matches && !out->inlier_mask && !out->pose_mask))) out = nullptr;
    if (!frames || n_frames < 2 || w <= 0 || h <= 0 || stride < w || frame_stride < (size_t)stride * h || !K ||
        !prm || !results || (mem != VO_MEM_HOST && mem != VO_MEM_DEVICE && mem != VO_MEM_HOST_PINNED))
        return s->fail(VO_ERR_INVALID, "vo_seq_run: bad argument");
    if (hipSetDevice(s->device) != hipSuccess) return s->fail(VO_ERR_DEVICE, "hipSetDevice(%d) failed", s->device);
    std::vector<ChunkDesc> ch;
    add_sequence(ch, s->chunk_pairs, frames, n_frames, frame_stride, results, out);
    return seq_run_chunks(s, ch, mem, w, h, stride, frame_stride, K, prm);
}

int vo_seq_run_multi(vo_seq* s, const uint8_t* const* frames, const int* n_frames, int n_seq, int mem, int w, int h,
                     int stride, size_t frame_stride, const double* K, const vo_pair_params* prm,
                     vo_pair_result* const* results, const vo_seq_outputs* outs) {
    if (!s) return VO_ERR_INVALID;
    if (!frames || !n_frames || n_seq < 1 || w <= 0 || h <= 0 || stride < w || frame_stride < (size_t)stride * h || !K ||
        !prm || !results || (mem != VO_MEM_HOST && mem != VO_MEM_DEVICE && mem != VO_MEM_HOST_PINNED))
        return s->fail(VO_ERR_INVALID, "vo_seq_run_multi: bad argument");
    for (int i = 0; i < n_seq; i++)
        if (!frames[i] || n_frames[i] < 2 || !results[i])
            return s->fail(VO_ERR_INVALID, "vo_seq_run_multi: sequence %d: bad argument", i);
    if (hipSetDevice(s->device) != hipSuccess) return s->fail(VO_ERR_DEVICE, "hipSetDevice(%d) failed", s->device);
    std::vector<ChunkDesc> ch;
    for (int i = 0; i < n_seq; i++) {
        const vo_seq_outputs* out = outs ? &outs[i] : nullptr;
        if (out && (out->match_cap <= 0 || (!out->matches && !out->inlier_mask && !out->pose_mask))) out = nullptr;
        add_sequence(ch, s->chunk_pairs, frames[i], n_frames[i], frame_stride, results[i], out);
    }
    return seq_run_chunks(s, ch, mem, w, h, stride, frame_stride, K, prm);
}

int vo_pairs_get_frame(vo_ctx* c, int frame, vo_keypoint* kp, uint8_t* desc, int cap, int* n) {
    if (!c) return VO_ERR_INVALID;
    if (frame < 0 || frame >= c->pipe_frames || !n) return c->fail(VO_ERR_INVALID, "vo_pairs_get_frame: bad frame");
    VO_HIP(c, hipSetDevice(c->device));
    const int cnt = c->pipe_nkp[frame];
    *n = cnt;
    if (cnt > cap) return c->fail(VO_ERR_CAPACITY, "vo_pairs_get_frame: %d keypoints, capacity %d", cnt, cap);
    if (kp && cnt)
        VO_HIP(c, hipMemcpyAsync(kp, c->d_kp.as<vo_keypoint>() + (size_t)frame * c->kp_cap,
                                 (size_t)cnt * sizeof(vo_keypoint), hipMemcpyDeviceToHost, c->stream));
    if (desc && cnt)
        VO_HIP(c, hipMemcpyAsync(desc, c->d_desc.as<uint8_t>() + (size_t)frame * c->kp_cap * 32, (size_t)cnt * 32,
                                 hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

int vo_pairs_get_matches(vo_ctx* c, int pair, vo_dmatch* good, uint8_t* inlier_mask, uint8_t* pose_mask, int cap,
                         int* n) {
    if (!c) return VO_ERR_INVALID;
    if (pair < 0 || pair >= c->pipe_frames - 1 || !n)
        return c->fail(VO_ERR_INVALID, "vo_pairs_get_matches: bad pair");
    VO_HIP(c, hipSetDevice(c->device));
    const int cnt = c->pipe_ngood[pair];
    const size_t n_cap = (size_t)c->pipe_ncap;
    *n = cnt;
    if (cnt > cap) return c->fail(VO_ERR_CAPACITY, "vo_pairs_get_matches: %d matches, capacity %d", cnt, cap);
    if (cnt == 0) return VO_OK;
    if (good)
        VO_HIP(c, hipMemcpyAsync(good, c->d_good.as<GoodMatch>() + pair * n_cap, (size_t)cnt * sizeof(GoodMatch),
                                 hipMemcpyDeviceToHost, c->stream));
    if (inlier_mask)
        VO_HIP(c, hipMemcpyAsync(inlier_mask, c->d_mask.as<uint8_t>() + pair * n_cap, cnt, hipMemcpyDeviceToHost,
                                 c->stream));
    if (pose_mask)
        VO_HIP(c, hipMemcpyAsync(pose_mask, c->d_pmask.as<uint8_t>() + pair * n_cap, cnt, hipMemcpyDeviceToHost,
                                 c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

}  // extern "C"
