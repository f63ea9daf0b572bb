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
// the next level would not be larger than the window.  Workspace for `nseq` image pairs side by side.
int track_setup(vo_ctx* c, int w, int h, const vo_lk_params* p, int nseq) {
    if (w > c->lim.max_width || h > c->lim.max_height)
        return c->fail(VO_ERR_CAPACITY, "image %dx%d exceeds the context limits %dx%d", w, h, c->lim.max_width,
                       c->lim.max_height);
    const bool same = c->tk_w == w && c->tk_h == h && c->tk_win_w == p->win_w && c->tk_win_h == p->win_h &&
                      c->tk_max_level == p->max_level && nseq <= c->tk_cap;
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
    const size_t ns = (size_t)nseq;
    VO_TRY(vo_ensure(c, c->d_tk_pyr[0], ns * g.bytes));
    VO_TRY(vo_ensure(c, c->d_tk_pyr[1], ns * g.bytes));
    VO_TRY(vo_ensure(c, c->d_tk_deriv, ns * g.bytes * sizeof(short2)));
    // the frame of the derivative image is zero and the kernels only ever write its interior
    VO_HIP(c, hipMemsetAsync(c->d_tk_deriv.p, 0, ns * g.bytes * sizeof(short2), c->stream));
    c->tk_geom = g;
    c->tk_w = w;
    c->tk_h = h;
    c->tk_win_w = p->win_w;
    c->tk_win_h = p->win_h;
    c->tk_max_level = p->max_level;
    c->tk_cap = nseq;
    c->tk_last = -1;  // (a regrown workspace has lost the cached pyramids)
    c->tk_last_count = 0;
    return VO_OK;
}

// nseq host images -> d_img[at ..] in one DMA -> their pyramids.  Pageable memory (any stride) is gathered into
// pinned staging first; tightly packed images that already lie one after the other in pinned memory
// (vo_host_alloc_pinned: a caller that fills the slots of such a block from its own threads, as the lock-step
// host does) are sent from where they are -- the gather of sixteen 1241 x 376 frames costs 0.5 ms on one thread.
int track_pyramids(vo_ctx* c, const uint8_t* const* gray, int nseq, int w, int h, int stride, int slot, int at) {
    const size_t bytes = (size_t)w * h;
    const uint8_t* stage = c->h_img.as<uint8_t>() + (size_t)at * bytes;
    bool direct = stride == w;
    for (int s = 1; direct && s < nseq; s++) direct = gray[s] == gray[0] + (size_t)s * bytes;
    if (direct) {
        hipPointerAttribute_t attr;
        direct = hipPointerGetAttributes(&attr, gray[0]) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!direct) (void)hipGetLastError();  // an unregistered pointer reports an error: not one of ours
    }
    if (direct) {
        stage = gray[0];
    } else {
        uint8_t* st = c->h_img.as<uint8_t>() + (size_t)at * bytes;
        for (int s = 0; s < nseq; s++) {
            uint8_t* d = st + (size_t)s * bytes;
            if (stride == w) memcpy(d, gray[s], bytes);
            else
                for (int y = 0; y < h; y++) memcpy(d + (size_t)y * w, gray[s] + (size_t)y * stride, w);
        }
    }
    uint8_t* dev = c->d_img.as<uint8_t>() + (size_t)at * bytes;
    VO_HIP(c, hipMemcpyAsync(dev, stage, (size_t)nseq * bytes, hipMemcpyHostToDevice, c->stream));
    launch_lk_pyramid(c->stream, dev, bytes, w, c->tk_geom, c->d_tk_pyr[slot].as<uint8_t>(), nseq);
    return vo_check_launch(c, "tracker pyramid");
}

int lk_track_batch(vo_ctx* c, int nseq, const uint8_t* const* prev, const uint8_t* const* next, int w, int h, int stride,
                   const float* const* prev_pts, const int* n, const vo_lk_params* prm, float* const* next_pts,
                   uint8_t* const* status, const char* who) {
    if (!c) return VO_ERR_INVALID;
    if (nseq < 1 || nseq > VO_LK_MAX_BATCH || !next || w <= 0 || h <= 0 || stride < w || !n || !lk_params_ok(prm) ||
        !prev_pts || !next_pts || !status)
        return c->fail(VO_ERR_INVALID, "%s: bad argument or unsupported tracker parameter (at most %d image pairs a call)",
                       who, VO_LK_MAX_BATCH);
    int nmax = 0, with_prev = 0;
    for (int s = 0; s < nseq; s++) {
        if (!next[s] || n[s] < 0 || n[s] > (1 << 24) || (n[s] > 0 && (!prev_pts[s] || !next_pts[s] || !status[s])))
            return c->fail(VO_ERR_INVALID, "%s: bad argument for image pair %d", who, s);
        nmax = std::max(nmax, n[s]);
        with_prev += prev && prev[s] ? 1 : 0;
    }
    if (with_prev != 0 && with_prev != nseq)
        return c->fail(VO_ERR_INVALID, "%s: either every pair brings its prev image or none does", who);
    VO_HIP(c, hipSetDevice(c->device));
    VO_TRY(vo_xfer_begin(c));
    VO_TRY(track_setup(c, w, h, prm, nseq));
    const size_t bytes = (size_t)w * h;
    VO_TRY(vo_ensure(c, c->d_img, 2 * (size_t)c->tk_cap * bytes));
    VO_TRY(vo_ensure_pinned(c, c->h_img, 2 * (size_t)c->tk_cap * bytes));
    int si, sj;
    if (with_prev) {
        si = 0;
        sj = 1;
        VO_TRY(track_pyramids(c, prev, nseq, w, h, stride, si, 0));
    } else {
        if (c->tk_last < 0 || c->tk_last_count != nseq)
            return c->fail(VO_ERR_INVALID, "%s: prev == NULL needs a previous call with the same geometry and batch size", who);
        si = c->tk_last;
        sj = si ^ 1;
    }
    c->tk_last = -1;
    VO_TRY(track_pyramids(c, next, nseq, w, h, stride, sj, c->tk_cap));
    launch_lk_scharr(c->stream, c->tk_geom, c->d_tk_pyr[si].as<uint8_t>(), c->d_tk_deriv.as<short2>(), nseq);
    VO_TRY(vo_check_launch(c, "tracker derivatives"));
    if (nmax > 0) {
        // point sets side by side, `ps` entries apart: one upload, one launch, one download for the whole batch
        const size_t ps = ((size_t)nmax + 63) & ~(size_t)63, tot = ps * nseq;
        VO_TRY(vo_ensure(c, c->d_fp0, tot * sizeof(float2)));
        VO_TRY(vo_ensure(c, c->d_fp1, tot * sizeof(float2)));
        VO_TRY(vo_ensure(c, c->d_mask, tot));
        VO_TRY(vo_ensure_pinned(c, c->h_pts, tot * (2 * sizeof(float2) + 1)));
        float2* h_in = c->h_pts.as<float2>();
        float2* h_out = h_in + tot;
        uint8_t* h_st = reinterpret_cast<uint8_t*>(h_out + tot);
        LkBatch b;
        memset(&b, 0, sizeof(b));
        b.count = nseq;
        b.pts_stride = (int)ps;
        for (int s = 0; s < nseq; s++) {
            b.n[s] = n[s];
            if (n[s]) memcpy(h_in + ps * s, prev_pts[s], (size_t)n[s] * sizeof(float2));
        }
        VO_HIP(c, hipMemcpyAsync(c->d_fp0.p, h_in, ((size_t)(nseq - 1) * ps + n[nseq - 1]) * sizeof(float2),
                                 hipMemcpyHostToDevice, c->stream));
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
                        c->d_mask.as<uint8_t>(), b, a);
        VO_TRY(vo_check_launch(c, "tracker"));
        VO_HIP(c, hipMemcpyAsync(h_out, c->d_fp1.p, tot * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipMemcpyAsync(h_st, c->d_mask.p, tot, hipMemcpyDeviceToHost, c->stream));
        VO_TRY(vo_xfer_finish(c));
        for (int s = 0; s < nseq; s++) {
            if (!n[s]) continue;
            memcpy(next_pts[s], h_out + ps * s, (size_t)n[s] * sizeof(float2));
            memcpy(status[s], h_st + ps * s, (size_t)n[s]);
        }
    } else {
        VO_TRY(vo_xfer_finish(c));
    }
    c->tk_last = sj;
    c->tk_last_count = nseq;
    return VO_OK;
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
    if (n < 0 || (n > 0 && (!prev_pts || !next_pts || !status)))
        return c->fail(VO_ERR_INVALID, "vo_lk_track: bad argument or unsupported tracker parameter");
    // a batch of one; an empty point set still builds (and caches) the pyramids
    float dummy_pt[2] = {0, 0};
    uint8_t dummy_st = 0;
    const float* pp = n ? prev_pts : dummy_pt;
    float* np = n ? next_pts : dummy_pt;
    uint8_t* sp = n ? status : &dummy_st;
    return lk_track_batch(c, 1, prev ? &prev : nullptr, &next, w, h, stride, &pp, &n, prm, &np, &sp, "vo_lk_track");
}

int vo_lk_track_batch(vo_ctx* c, int n_seq, const uint8_t* const* prev, const uint8_t* const* next, int w, int h,
                      int stride, const float* const* prev_pts, const int* n, const vo_lk_params* prm,
                      float* const* next_pts, uint8_t* const* status) {
    return lk_track_batch(c, n_seq, prev, next, w, h, stride, prev_pts, n, prm, next_pts, status, "vo_lk_track_batch");
}

void* vo_host_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void vo_host_free_pinned(void* p) {
    if (p) (void)hipHostFree(p);
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
