// vo_pipeline.cpp — the device-resident batched frame-pair path: ORB on every frame of the batch,
// then per consecutive pair knnMatch(k=2) + ratio/displacement filter + findEssentialMat(RANSAC)
// (the body of matcher(), reference: src/matcher/matcher.cpp:13-50) and recoverPose
// (reference: src/visual_odometry/vo_pose.h:19).  Frames stay in HBM; the host sees one
// synchronisation for the keypoint selection (vo_features.cpp) and one for the results.
#include <algorithm>
#include <cstddef>

#include "ctx.h"
#include "geom_types.h"

int match_knn2_dev(vo_ctx* c, const uint32_t* d_q, const uint32_t* d_t, size_t q_stride, size_t t_stride,
                   const int* d_n, int nq, int nt, int words, int npairs, int nq_cap);
int match_filter_dev(vo_ctx* c, const vo_keypoint* d_kq, const vo_keypoint* d_kt, size_t kp_stride,
                     const int* d_n, int nq, int npairs, int nq_cap, float ratio, float max_disp);



// Everything after keypoint selection for the batch, asynchronous up to the final copy of the
// results; max_n bounds the per-frame keypoint count (launch sizes, per-pair strides).
static int pairs_after_selection(vo_ctx* c, int n_frames, int max_n, const double* K, const vo_pair_params* prm) {
    const int npairs = n_frames - 1;
    const int n_cap = std::max(max_n, 8);
    if (n_cap > c->lim.max_points)
        return c->fail(VO_ERR_CAPACITY, "%d keypoints per frame exceed max_points %d", n_cap, c->lim.max_points);
    const size_t kp_cap = (size_t)c->kp_cap;
    VO_TRY(match_knn2_dev(c, c->d_desc.as<uint32_t>(), c->d_desc.as<uint32_t>() + kp_cap * 8, kp_cap * 8,
                          kp_cap * 8, c->d_nkp.as<int>(), max_n, max_n, 8, npairs, n_cap));
    VO_TRY(match_filter_dev(c, c->d_kp.as<vo_keypoint>(), c->d_kp.as<vo_keypoint>() + kp_cap, kp_cap,
                            c->d_nkp.as<int>(), max_n, npairs, n_cap, prm->ratio, prm->max_disp));
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
    VO_HIP(c, hipMemcpyAsync(c->h_state.p, c->d_state.p, (size_t)npairs * sizeof(EstState), hipMemcpyDeviceToHost,
                             c->stream));
    VO_HIP(c, hipMemcpyAsync(c->h_pose.p, c->d_pose.p, (size_t)npairs * sizeof(PoseState), hipMemcpyDeviceToHost,
                             c->stream));
    VO_HIP(c, hipMemcpyAsync(c->h_good.p, c->d_ngood.p, (size_t)npairs * sizeof(int), hipMemcpyDeviceToHost,
                             c->stream));
    VO_HIP(c, hipMemcpyAsync(c->h_nkp.p, c->d_nkp.p, (size_t)n_frames * sizeof(int), hipMemcpyDeviceToHost,
                             c->stream));
    c->pipe_ncap = n_cap;
    return VO_OK;
}

extern "C" {

int vo_pairs_run_dev(vo_ctx* c, const uint8_t* d_frames, int n_frames, int w, int h, int stride,
                     size_t frame_stride, const double* K, const vo_pair_params* prm, vo_pair_result* results) {
    if (!c) return VO_ERR_INVALID;
    if (!d_frames || n_frames < 2 || w <= 0 || h <= 0 || stride < w || frame_stride < (size_t)stride * h || !K ||
        !prm || !results)
        return c->fail(VO_ERR_INVALID, "vo_pairs_run_dev: bad argument");
    if (prm->orb.nfeatures > c->lim.max_features)
        return c->fail(VO_ERR_CAPACITY, "nfeatures %d exceeds max_features %d", prm->orb.nfeatures,
                       c->lim.max_features);
    VO_HIP(c, hipSetDevice(c->device));
    const int npairs = n_frames - 1;
    c->pipe_frames = 0;

    // ---- the whole batch is queued without a host round trip: ORB with the device-side keypoint
    // selection, matching, essential matrix, pose; one synchronisation for the results ------------
    VO_TRY(feat_setup(c, w, h, &prm->orb, n_frames));
    VO_TRY(feat_detect(c, d_frames, frame_stride, stride, n_frames, prm->orb.fast_threshold,
                       prm->orb.edge_threshold, 1));
    const int n_bound = feat_n_bound(c, prm->orb.nfeatures);
    VO_TRY(feat_select_dev(c, n_frames, n_bound));
    VO_TRY(feat_describe(c, n_frames, n_bound, true, false));
    VO_TRY(pairs_after_selection(c, n_frames, n_bound, K, prm));
    VO_HIP(c, hipMemcpyAsync(c->h_flags.p, c->d_flags.p, (size_t)n_frames * sizeof(int), hipMemcpyDeviceToHost,
                             c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    int flagged = 0;
    for (int f = 0; f < n_frames; f++) flagged |= c->h_flags.as<int>()[f];
    c->last_flags = flagged;
    if (flagged) {
        // a frame the device selection could not finish exactly (VO_SEL_*): host selection, redo
        int max_n = 0;
        VO_TRY(feat_select_host(c, n_frames, &max_n));
        if (max_n > 0) VO_TRY(feat_describe(c, n_frames, max_n, true, true));
        VO_TRY(pairs_after_selection(c, n_frames, max_n, K, prm));
        VO_HIP(c, hipStreamSynchronize(c->stream));
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
    return VO_OK;
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
    std::vector<GoodMatch> tmp(cnt);
    if (good)
        VO_HIP(c, hipMemcpyAsync(tmp.data(), c->d_good.as<GoodMatch>() + pair * n_cap, (size_t)cnt * sizeof(GoodMatch),
                                 hipMemcpyDeviceToHost, c->stream));
    if (inlier_mask)
        VO_HIP(c, hipMemcpyAsync(inlier_mask, c->d_mask.as<uint8_t>() + pair * n_cap, cnt, hipMemcpyDeviceToHost,
                                 c->stream));
    if (pose_mask)
        VO_HIP(c, hipMemcpyAsync(pose_mask, c->d_pmask.as<uint8_t>() + pair * n_cap, cnt, hipMemcpyDeviceToHost,
                                 c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    if (good)
        for (int i = 0; i < cnt; i++) {
            good[i].queryIdx = tmp[i].q;
            good[i].trainIdx = tmp[i].t;
            good[i].imgIdx = 0;
            good[i].distance = (float)tmp[i].dist;
        }
    return VO_OK;
}

}  // extern "C"
