// vo_ctx.cpp — context lifetime, workspace growth, profiling hooks of libvo_hip.so.
#include <algorithm>
#include <cstdlib>

#include "ctx.h"

int vo_ensure(vo_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return VO_OK;
    if (b.p) {
        // the stream may still be using the old allocation
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        VO_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = (bytes + 255) & ~(size_t)255;
    VO_HIP(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return VO_OK;
}

int vo_ensure_pinned(vo_ctx* ctx, PinnedBuf& b, size_t bytes) {
    if (bytes <= b.cap) return VO_OK;
    if (b.p) {
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        VO_HIP(ctx, hipHostFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = (bytes + 4095) & ~(size_t)4095;
    VO_HIP(ctx, hipHostMalloc(&b.p, want, hipHostMallocDefault));
    b.cap = want;
    return VO_OK;
}

// Host image (pageable memory, any stride) -> tightly packed device image.  Rows are gathered into a
// pinned staging slot so that the transfer is one DMA; two slots alternate because an entry point
// uploads at most two images (prev, next) before it synchronises.
int vo_upload_image(vo_ctx* c, DevBuf& dst, const uint8_t* gray, int w, int h, int stride) {
    const size_t bytes = (size_t)w * h;
    VO_TRY(vo_ensure(c, dst, bytes));
    VO_TRY(vo_ensure_pinned(c, c->h_img, 2 * bytes));
    uint8_t* slot = c->h_img.as<uint8_t>() + (size_t)(c->h_img_slot ^= 1) * bytes;
    if (stride == w) memcpy(slot, gray, bytes);
    else
        for (int y = 0; y < h; y++) memcpy(slot + (size_t)y * w, gray + (size_t)y * stride, w);
    VO_HIP(c, hipMemcpyAsync(dst.p, slot, bytes, hipMemcpyHostToDevice, c->stream));
    return VO_OK;
}

// ---- pinned transfer arena (see ctx.h) ----------------------------------------------------------
static const size_t kXferAlign = 64;

static uint8_t* xfer_slot(vo_ctx* c, size_t bytes, size_t* offset) {
    if (!c->h_xfer.p) {
        // sized once (growing would move slots that copies in flight still use): every per-point array
        // of an entry point, twice over
        const size_t want = std::max<size_t>(1 << 20, (size_t)c->lim.max_points * 256);
        if (vo_ensure_pinned(c, c->h_xfer, want) != VO_OK) return nullptr;
        c->xfer_used = 0;
    }
    const size_t at = (c->xfer_used + kXferAlign - 1) & ~(kXferAlign - 1);
    if (at + bytes > c->h_xfer.cap) return nullptr;  // the caller falls back to a direct copy
    c->xfer_used = at + bytes;
    if (offset) *offset = at;
    return c->h_xfer.as<uint8_t>() + at;
}

// Start of an entry point: an earlier call that failed between its transfers and its finish may have
// left copies in flight and outputs registered; let them drain and drop them.
int vo_xfer_begin(vo_ctx* c) {
    if (c->xfer_used != 0 || !c->xfer_out.empty()) {
        (void)hipStreamSynchronize(c->stream);
        c->xfer_out.clear();
        c->xfer_used = 0;
    }
    return VO_OK;
}

int vo_xfer_in(vo_ctx* c, void* dev, const void* host, size_t bytes) {
    if (bytes == 0) return VO_OK;
    uint8_t* slot = xfer_slot(c, bytes, nullptr);
    if (slot) {
        memcpy(slot, host, bytes);
        host = slot;
    }
    VO_HIP(c, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return VO_OK;
}

int vo_xfer_out(vo_ctx* c, void* host, const void* dev, size_t bytes) {
    if (bytes == 0) return VO_OK;
    size_t off = 0;
    uint8_t* slot = xfer_slot(c, bytes, &off);
    if (slot) {
        c->xfer_out.push_back({host, off, bytes});
        host = slot;
    }
    VO_HIP(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    return VO_OK;
}

int vo_xfer_finish(vo_ctx* c) {
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        for (const auto& o : c->xfer_out) memcpy(o.host, c->h_xfer.as<uint8_t>() + o.offset, o.bytes);
    c->xfer_out.clear();
    c->xfer_used = 0;
    if (e != hipSuccess) return c->fail(VO_ERR_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString(e));
    return VO_OK;
}

int vo_check_launch(vo_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ctx->fail(VO_ERR_DEVICE, "%s: launch failed: %s", what, hipGetErrorString(e));
    return VO_OK;
}

extern "C" {

void vo_limits_default(vo_limits* l) {
    l->max_width = 1241;
    l->max_height = 376;
    l->max_features = 8192;
    l->max_points = 8448;
    l->max_batch = 64;
}

void vo_orb_params_default(vo_orb_params* p, int nfeatures) {
    p->nfeatures = nfeatures;
    p->scale_factor = 1.2f;
    p->nlevels = 8;
    p->edge_threshold = 31;
    p->first_level = 0;
    p->wta_k = 2;
    p->score_type = 0;
    p->patch_size = 31;
    p->fast_threshold = 20;
}

void vo_pair_params_default(vo_pair_params* p, int nfeatures) {
    vo_orb_params_default(&p->orb, nfeatures);
    p->ratio = 0.7f;
    p->max_disp = 200.f;
    p->prob = 0.999;
    p->threshold = 1.0;
    p->dist_thresh = 200.0;
    p->seed = ~(uint64_t)0;
    p->max_iters = 1000;
    p->cross_check = 0;
}

int vo_ctx_create(int device, const vo_limits* limits, vo_ctx** out) {
    if (!out) return VO_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VO_ERR_DEVICE;  // no CPU fallback
    if (device < 0 || device >= ndev) return VO_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return VO_ERR_DEVICE;
    if (knn_static_init(device) != VO_OK) return VO_ERR_DEVICE;  // vo_knn2_hamming_static's workspace on this device
    vo_ctx* c = new vo_ctx();
    c->device = device;
    c->err[0] = 0;
    memset(&c->tk_geom, 0, sizeof(c->tk_geom));
    if (limits) c->lim = *limits;
    else vo_limits_default(&c->lim);
    // schedule switches: the environment is read here, once per context (vo_ctx_set_option changes them later)
    auto env_int = [](const char* name, int dflt) {
        const char* e = getenv(name);
        return e && *e ? atoi(e) : dflt;
    };
    c->opt_knn_mfma = env_int("VO_KNN_MFMA", 1) != 0;
    c->opt_orb_split = env_int("VO_ORB_SPLIT", 0) != 0;
    c->opt_seq_prefetch = env_int("VO_SEQ_PREFETCH", 0) != 0;
    c->opt_akaze_graph = env_int("VO_AKAZE_GRAPH", 1) != 0;
    c->opt_solver = env_int("VO_SOLVER", 0) != 0;
    c->opt_lk_generic = env_int("VO_LK_GENERIC", 0) != 0;
    c->opt_knn_splits = std::max(0, env_int("VO_KNN_SPLITS", 0));
    if (c->lim.max_points > 65535 || c->lim.max_features > 262143 || c->lim.max_points < 5 ||
        c->lim.max_batch < 2 || c->lim.max_width < 7 || c->lim.max_height < 7) {
        delete c;
        return VO_ERR_INVALID;
    }
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < VO_PROF_COUNT; i++)
        ok = hipEventCreate(&c->ev0[i]) == hipSuccess && hipEventCreate(&c->ev1[i]) == hipSuccess;
    if (!ok) {
        vo_ctx_destroy(c);
        return VO_ERR_DEVICE;
    }
    *out = c;
    return VO_OK;
}

void vo_ctx_destroy(vo_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* dev[] = {&c->d_img,   &c->d_pyr,    &c->d_blur,  &c->d_score,  &c->d_masks, &c->d_segbase, &c->d_selcand, &c->d_selcnt, &c->d_flags,
                     &c->d_lvlofs, &c->d_cand,  &c->d_sel,   &c->d_coef,   &c->d_geom,   &c->d_kp,
                     &c->d_desc,  &c->d_nkp,    &c->d_q,     &c->d_t,      &c->d_part,   &c->d_knn,   &c->d_rev,
                     &c->d_good,  &c->d_ngood,  &c->d_p0,    &c->d_p1,     &c->d_kq,     &c->d_kt,
                     &c->d_x1,   &c->d_x2,    &c->d_models, &c->d_nmodels, &c->d_counts,
                     &c->d_state, &c->d_mask,   &c->d_q4,    &c->d_cmask,  &c->d_pose,   &c->d_pmask,
                     &c->d_pts4,  &c->d_fp0,    &c->d_fp1,   &c->d_cam,    &c->d_proj,
                     &c->d_tk_pyr[0], &c->d_tk_pyr[1], &c->d_tk_deriv, &c->d_pack,
                     &c->d_ak, &c->d_ak_mask, &c->d_ak_tab, &c->d_ak_kp, &c->d_ak_desc};
    if (c->ak_graph) (void)hipGraphExecDestroy(c->ak_graph);
    for (DevBuf* b : dev)
        if (b->p) (void)hipFree(b->p);
    PinnedBuf* pin[] = {&c->h_lvlofs, &c->h_cand, &c->h_sel,  &c->h_nkp,  &c->h_misc, &c->h_flags, &c->h_knn,
                        &c->h_good,   &c->h_state, &c->h_pose, &c->h_mask, &c->h_pts, &c->h_img,
                        &c->h_xfer, &c->h_pack, &c->h_ak};
    for (PinnedBuf* b : pin)
        if (b->p) (void)hipHostFree(b->p);
    for (int i = 0; i < VO_PROF_COUNT; i++) {
        if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]);
        if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* vo_last_error(const vo_ctx* c) { return c ? c->err : "null context"; }

int vo_ctx_synchronize(vo_ctx* c) {
    if (!c) return VO_ERR_INVALID;
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

int vo_prof_enable(vo_ctx* c, int on) {
    if (!c) return VO_ERR_INVALID;
    c->prof_on = on != 0;
    return VO_OK;
}

int vo_ctx_set_option(vo_ctx* c, int option, int value) {
    if (!c) return VO_ERR_INVALID;
    if (option == VO_OPT_SEL_CAP) {
        if (value < 8 || value > 13000) return c->fail(VO_ERR_INVALID, "VO_OPT_SEL_CAP must be in [8, 13000]");
        c->sel_cap = value;
        return VO_OK;
    }
    int* sw = option == VO_OPT_KNN_MFMA ? &c->opt_knn_mfma : option == VO_OPT_ORB_SPLIT ? &c->opt_orb_split
              : option == VO_OPT_SEQ_PREFETCH ? &c->opt_seq_prefetch : option == VO_OPT_AKAZE_GRAPH ? &c->opt_akaze_graph
              : option == VO_OPT_SOLVER ? &c->opt_solver : option == VO_OPT_LK_GENERIC ? &c->opt_lk_generic : nullptr;
    if (sw) {
        if (value != 0 && value != 1) return c->fail(VO_ERR_INVALID, "option %d takes 0 or 1", option);
        *sw = value;
        return VO_OK;
    }
    if (option == VO_OPT_KNN_SPLITS) {
        if (value < 0 || value > 32) return c->fail(VO_ERR_INVALID, "VO_OPT_KNN_SPLITS must be in [0, 32]");
        c->opt_knn_splits = value;
        return VO_OK;
    }
    return c->fail(VO_ERR_INVALID, "unknown or read-only option %d", option);
}

int vo_ctx_get_option(vo_ctx* c, int option, int* value) {
    if (!c || !value) return VO_ERR_INVALID;
    if (option == VO_OPT_SEL_CAP) *value = c->sel_cap;
    else if (option == VO_OPT_LAST_FLAGS) *value = c->last_flags;
    else if (option == VO_OPT_CAND_CAP) *value = c->max_cand;
    else if (option == VO_OPT_KNN_MFMA) *value = c->opt_knn_mfma;
    else if (option == VO_OPT_ORB_SPLIT) *value = c->opt_orb_split;
    else if (option == VO_OPT_SEQ_PREFETCH) *value = c->opt_seq_prefetch;
    else if (option == VO_OPT_AKAZE_GRAPH) *value = c->opt_akaze_graph;
    else if (option == VO_OPT_SOLVER) *value = c->opt_solver;
    else if (option == VO_OPT_KNN_SPLITS) *value = c->opt_knn_splits;
    else if (option == VO_OPT_LK_GENERIC) *value = c->opt_lk_generic;
    else return c->fail(VO_ERR_INVALID, "unknown option %d", option);
    return VO_OK;
}

int vo_prof_get(vo_ctx* c, int which, int reset, int* launches, double* total_ms) {
    if (!c || which < 0 || which >= VO_PROF_COUNT) return VO_ERR_INVALID;
    if (launches) *launches = c->prof[which].launches;
    if (total_ms) *total_ms = c->prof[which].total_ms;
    if (reset) c->prof[which] = ProfSlot();
    return VO_OK;
}

void vo_projection_matrix(const double* K, const double* pose, double* P) {
    // K [R^T | -R^T t], pose = [R | t] camera-to-world (reference: src/sfm/triangulation.cpp:34-44)
    double M[12];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) M[4 * i + j] = pose[4 * j + i];
        double s = 0;
        for (int k = 0; k < 3; k++) s += pose[4 * k + i] * pose[4 * k + 3];
        M[4 * i + 3] = -s;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[3 * i + k] * M[4 * k + j];
            P[4 * i + j] = s;
        }
}

}  // extern "C"
