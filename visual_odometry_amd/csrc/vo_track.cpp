// vo_track.cpp — host side of the tracker entry point: pyramid geometry of
// cv::buildOpticalFlowPyramid, kernel sequencing, uploads / downloads.
// Replaces optical_flow_->calc(...) in FeatureTracker::trackPoints
// (reference: src/features/feature_tracker.cpp:6,13; cuda variant src/features/cuda/feature_tracker.cpp:9-30).
#include <algorithm>

#include "ctx.h"

namespace {

bool lk_params_ok(const vo_lk_params* p) {
    return p && p->win_w > 2 && p->win_h > 2 && p->win_w <= TK_PAD && p->win_h <= TK_PAD && p->max_level >= 0 &&
           p->max_level < TK_MAX_LEVELS && (p->channels == 1 || p->channels == 3);
}

// Level sizes as buildOpticalFlowPyramid makes them: halving (rounding up) until max_level or until
// the next level would not be larger than the window.
int track_setup(vo_ctx* c, int w, int h, const vo_lk_params* p) {
    if (w > c->lim.max_width || h > c->lim.max_height)
        return c->fail(VO_ERR_CAPACITY, "image %dx%d exceeds the context limits %dx%d", w, h, c->lim.max_width,
                       c->lim.max_height);
    const bool same = c->tk_w == w && c->tk_h == h && c->tk_win_w == p->win_w && c->tk_win_h == p->win_h &&
                      c->tk_max_level == p->max_level;
    if (same) return VO_OK;
    TrackGeom g;
    memset(&g, 0, sizeof(g));
    unsigned off = 0;
    int lw = w, lh = h;
    for (int level = 0; level <= p->max_level; level++) {
        TrackLevel& L = g.lv[level];
        L.w = lw;
        L.h = lh;
        L.pitch = (lw + 2 * TK_PAD + 63) & ~63;
        L.offset = off;
        off += (unsigned)L.pitch * (lh + 2 * TK_PAD);
        off = (off + 255u) & ~255u;
        g.nlevels = level + 1;
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
        if (lw <= p->win_w || lh <= p->win_h) break;
    }
    g.bytes = off;
    VO_TRY(vo_ensure(c, c->d_tk_pyr[0], g.bytes));
    VO_TRY(vo_ensure(c, c->d_tk_pyr[1], g.bytes));
    VO_TRY(vo_ensure(c, c->d_tk_deriv, (size_t)g.bytes * sizeof(short2)));
    // the frame of the derivative image is zero and the kernels only ever write its interior
    VO_HIP(c, hipMemsetAsync(c->d_tk_deriv.p, 0, (size_t)g.bytes * sizeof(short2), c->stream));
    c->tk_geom = g;
    c->tk_w = w;
    c->tk_h = h;
    c->tk_win_w = p->win_w;
    c->tk_win_h = p->win_h;
    c->tk_max_level = p->max_level;
    c->tk_last = -1;
    return VO_OK;
}

int track_pyramid(vo_ctx* c, const uint8_t* gray, int w, int h, int stride, int slot) {
    VO_TRY(vo_upload_image(c, c->d_img, gray, w, h, stride));
    launch_lk_pyramid(c->stream, c->d_img.as<uint8_t>(), w, c->tk_geom, c->d_tk_pyr[slot].as<uint8_t>());
    return vo_check_launch(c, "tracker pyramid");
}

}  // namespace

extern "C" {

void vo_lk_params_default(vo_lk_params* p) {
    p->win_w = 21;
    p->win_h = 21;
    p->max_level = 3;
    p->max_count = 30;
    p->epsilon = 0.01;
    p->min_eig_threshold = 1e-4;
    p->channels = 3;
    p->reserved = 0;
}

int vo_lk_track(vo_ctx* c, const uint8_t* prev, const uint8_t* next, int w, int h, int stride, const float* prev_pts,
                int n, const vo_lk_params* prm, float* next_pts, uint8_t* status) {
    if (!c) return VO_ERR_INVALID;
    if (!next || w <= 0 || h <= 0 || stride < w || n < 0 || n > (1 << 24) || !lk_params_ok(prm) ||
        (n > 0 && (!prev_pts || !next_pts || !status)))
        return c->fail(VO_ERR_INVALID, "vo_lk_track: bad argument or unsupported tracker parameter");
    VO_HIP(c, hipSetDevice(c->device));
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(track_setup(c, w, h, prm));
    int si, sj;
    if (prev) {
        si = 0;
        sj = 1;
        VO_TRY(track_pyramid(c, prev, w, h, stride, si));
    } else {
        if (c->tk_last < 0)
            return c->fail(VO_ERR_INVALID, "vo_lk_track: prev == NULL needs a previous call with the same geometry");
        si = c->tk_last;
        sj = si ^ 1;
    }
    c->tk_last = -1;
    VO_TRY(track_pyramid(c, next, w, h, stride, sj));
    launch_lk_scharr(c->stream, c->tk_geom, c->d_tk_pyr[si].as<uint8_t>(), c->d_tk_deriv.as<short2>());
    VO_TRY(vo_check_launch(c, "tracker derivatives"));
    if (n > 0) {
        VO_TRY(vo_ensure(c, c->d_fp0, (size_t)n * sizeof(float2)));
        VO_TRY(vo_ensure(c, c->d_fp1, (size_t)n * sizeof(float2)));
        VO_TRY(vo_ensure(c, c->d_mask, (size_t)n));
        VO_TRY(vo_xfer_in(c, c->d_fp0.p, prev_pts, (size_t)n * sizeof(float2)));
        LkArgs a;
        a.win_w = prm->win_w;
        a.win_h = prm->win_h;
        a.max_count = std::min(std::max(prm->max_count, 0), 100);
        a.channels = prm->channels;
        a.generic = c->opt_lk_generic;
        a.min_eig = (float)prm->min_eig_threshold;
        const double eps = std::min(std::max(prm->epsilon, 0.), 10.);
        a.eps2 = eps * eps;
        launch_lk_track(c->stream, c->tk_geom, c->d_tk_pyr[si].as<uint8_t>(), c->d_tk_deriv.as<short2>(),
                        c->d_tk_pyr[sj].as<uint8_t>(), c->d_fp0.as<float2>(), c->d_fp1.as<float2>(),
                        c->d_mask.as<uint8_t>(), n, a);
        VO_TRY(vo_check_launch(c, "tracker"));
        VO_TRY(vo_xfer_out(c, next_pts, c->d_fp1.p, (size_t)n * sizeof(float2)));
        VO_TRY(vo_xfer_out(c, status, c->d_mask.p, (size_t)n));
    }
    VO_TRY(vo_xfer_finish(c));
    c->tk_last = sj;
    return VO_OK;
}

int vo_debug_read_lk_level(vo_ctx* c, int which, int level, void* out, int cap, int* w, int* h) {
    if (!c) return VO_ERR_INVALID;
    if (!out || !w || !h || level < 0 || level >= c->tk_geom.nlevels || c->tk_last < 0 ||
        (which != VO_DBG_LK_PREV && which != VO_DBG_LK_NEXT && which != VO_DBG_LK_DERIV))
        return c->fail(VO_ERR_INVALID, "vo_debug_read_lk_level: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    const TrackLevel& L = c->tk_geom.lv[level];
    *w = L.w;
    *h = L.h;
    const size_t px = L.offset + (size_t)TK_PAD * L.pitch + TK_PAD;
    if (which == VO_DBG_LK_DERIV) {
        if (cap < L.w * L.h * 4) return c->fail(VO_ERR_CAPACITY, "vo_debug_read_lk_level: need %d bytes", L.w * L.h * 4);
        VO_HIP(c, hipMemcpy2DAsync(out, (size_t)L.w * 4, c->d_tk_deriv.as<short2>() + px, (size_t)L.pitch * 4,
                                   (size_t)L.w * 4, L.h, hipMemcpyDeviceToHost, c->stream));
    } else {
        if (cap < L.w * L.h) return c->fail(VO_ERR_CAPACITY, "vo_debug_read_lk_level: need %d bytes", L.w * L.h);
        const int slot = which == VO_DBG_LK_NEXT ? c->tk_last : c->tk_last ^ 1;
        VO_HIP(c, hipMemcpy2DAsync(out, L.w, c->d_tk_pyr[slot].as<uint8_t>() + px, L.pitch, L.w, L.h,
                                   hipMemcpyDeviceToHost, c->stream));
    }
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

}  // extern "C"
