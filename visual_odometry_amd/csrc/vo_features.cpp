// vo_features.cpp — host side of the FAST / ORB entry points: pyramid geometry (cv's level
// sizes, quotas and INTER_LINEAR_EXACT coefficient tables), kernel sequencing, and the
// response-ordered keypoint selection (KeyPointsFilter::retainBest), whose output permutation is
// defined by libstdc++'s nth_element/partition: reproduced on the device (k_orb_select), with this
// file's host implementation (libstdc++ itself) as the path for inputs the device kernel flags.
// Replaces FeatureDetector's cv calls (reference: src/features/feature_detector.cpp:3-42).
#include <algorithm>
#include <cmath>
#include <thread>

#include "ctx.h"

namespace {

inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_roundf(float v) { return (int)std::lrintf(v); }
inline int cv_floor(double v) {
    int i = (int)v;
    return i - (i > v);
}

bool orb_params_ok(const vo_orb_params* p) {
    return p && p->nfeatures >= 0 && p->nlevels >= 1 && p->nlevels <= VO_ORB_MAX_LEVELS &&
           p->first_level == 0 && p->wta_k == 2 && p->score_type == 0 && p->patch_size == 31 &&
           p->scale_factor > 1.0f && p->edge_threshold >= VO_ORB_MIN_EDGE && p->fast_threshold >= 0;
}

// interpolationLinear<uchar>::getCoeffs of cv's bit-exact resize: entry = ofs << 16 | c1
void linear_table(int ssize, int dsize, uint32_t* out) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1.0 / inv_scale;
    for (int v = 0; v < dsize; v++) {
        const double fval = scale * ((double)v + 0.5) - 0.5;
        const int ival = cv_floor(fval);
        uint32_t ofs, c1;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                ofs = (uint32_t)ival;
                c1 = (uint32_t)cv_round((fval - (double)ival) * 256.0);
            } else {
                ofs = (uint32_t)(ssize - 1);
                c1 = 0;
            }
        } else {
            ofs = 0;
            c1 = 0;
        }
        out[v] = (ofs << 16) | c1;
    }
}

Taps7 gauss_taps() {
    // cv::getGaussianKernel(7, 2, CV_32F) scaled to 8 fractional bits
    Taps7 t;
    const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < 7; i++) {
        const double x = i - 3.0;
        cf[i] = (float)std::exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        t.t[i] = cv_round((double)cf[i] * 256.0);
    }
    return t;
}

struct RespIdx {
    float response;
    int idx;
};

// KeyPointsFilter::retainBest (cv 3.4.2)
void retain_best(std::vector<RespIdx>& v, int n_points) {
    if (n_points >= 0 && v.size() > (size_t)n_points) {
        if (n_points == 0) {
            v.clear();
            return;
        }
        std::nth_element(v.begin(), v.begin() + n_points, v.end(),
                         [](const RespIdx& a, const RespIdx& b) { return a.response > b.response; });
        const float ambiguous = v[n_points - 1].response;
        auto new_end = std::partition(v.begin() + n_points, v.end(),
                                      [ambiguous](const RespIdx& k) { return k.response >= ambiguous; });
        v.resize(new_end - v.begin());
    }
}

}  // namespace

// Pyramid geometry for (w, h, params); p == nullptr -> single level (plain FAST).
int feat_setup(vo_ctx* c, int w, int h, const vo_orb_params* p, int nframes) {
    c->feat_off = 0;
    const int nlevels = p ? p->nlevels : 1;
    const float sf = p ? p->scale_factor : 1.2f;
    const int nfeatures = p ? p->nfeatures : 0;
    // every path that describes keypoints comes through here: the orientation patch (radius 15) and the
    // descriptor window (radius 18) are read without clamping, which the edge threshold has to allow
    if (p && !orb_params_ok(p)) return c->fail(VO_ERR_INVALID, "unsupported ORB parameters");
    if (w < 7 || h < 7) return c->fail(VO_ERR_INVALID, "image %dx%d is smaller than the FAST ring", w, h);
    if (w > c->lim.max_width || h > c->lim.max_height)
        return c->fail(VO_ERR_CAPACITY, "image %dx%d exceeds the context limits %dx%d", w, h,
                       c->lim.max_width, c->lim.max_height);
    if (nframes > c->lim.max_batch)
        return c->fail(VO_ERR_CAPACITY, "%d frames exceed max_batch %d", nframes, c->lim.max_batch);
    const int border = p ? p->edge_threshold : 0;  // the detector skips what the border filter would drop
    const bool same = c->geom_w == w && c->geom_h == h && c->geom_levels == nlevels && c->geom_scale == sf &&
                      c->geom_border == border;
    if (!same) {
        PyrGeom g;
        memset(&g, 0, sizeof(g));
        g.nlevels = nlevels;
        const double scaleFactor = (double)sf;
        unsigned off = 0;
        int row = 0, tile = 0, seg = 0;
        for (int l = 0; l < nlevels; l++) {
            const float s = (float)std::pow(scaleFactor, (double)l);
            LevelGeom& L = g.lv[l];
            L.scale = s;
            L.w = cv_roundf(w / s);
            L.h = cv_roundf(h / s);
            if (L.w < 7 || L.h < 7) return c->fail(VO_ERR_INVALID, "pyramid level %d is too small", l);
            L.pitch = (L.w + 2 * VO_PADX + 63) & ~63;
            L.offset = off;
            L.row0 = row;
            off += (unsigned)L.pitch * (L.h + 2 * VO_PADY);
            off = (off + 255u) & ~255u;
            row += L.h;
            // keypoints live in [b, w - b) x [b, h - b), b = max(border, 3) (FAST itself needs 3); a level
            // not larger than twice the border keeps none (cv's runByImageBorder)
            const int b = std::max(border, 3);
            const bool level_ok = border == 0 || (L.w > 2 * border && L.h > 2 * border);
            L.rx0 = (b - 1) & ~3;
            L.ry0 = b - 1;
            L.rw = level_ok ? std::max(std::min(L.w, L.w - b + 1) - L.rx0, 0) : 0;
            L.rh = level_ok ? std::max(std::min(L.h, L.h - b + 1) - L.ry0, 0) : 0;
            if (L.rw == 0 || L.rh == 0) L.rw = L.rh = 0;
            L.segs = cdiv(L.rw, 64);
            L.seg0 = seg;
            seg += L.segs * L.rh;
            L.tiles_x = L.segs;
            L.tile0 = tile;
            tile += L.tiles_x * cdiv(L.rh, VO_FT_H);
        }
        g.tiles_total = tile;
        g.segs_total = seg;
        g.rows_total = row;
        g.pyr_bytes = off;
        // coefficient tables for levels >= 1, padded by the reflect-101 frame the resize kernel also
        // writes (x = -4 .. w + 7, y = -3 .. h + 2), 16-byte aligned; and the LDS tile that holds the
        // source rectangle of any 256 x PR_ROWS block of the level
        c->coef_host.clear();
        c->coef_x.assign(nlevels, 0);
        c->coef_y.assign(nlevels, 0);
        c->coef_ldw.assign(nlevels, 0);
        c->coef_rows.assign(nlevels, 0);
        auto reflect = [](int i, int n) {
            while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
            return i;
        };
        auto padded_table = [&](int ssize, int dsize, int before, int after) {
            std::vector<uint32_t> raw(dsize);
            linear_table(ssize, dsize, raw.data());
            const size_t at = c->coef_host.size();
            for (int i = -before; i < dsize + after; i++) c->coef_host.push_back(raw[reflect(i, dsize)]);
            while (c->coef_host.size() % 4) c->coef_host.push_back(0);
            return at;
        };
        for (int l = 1; l < nlevels; l++) {
            const int sw = g.lv[l - 1].w, sh = g.lv[l - 1].h, dw = g.lv[l].w, dh = g.lv[l].h;
            c->coef_x[l] = padded_table(sw, dw, 4, 8);
            c->coef_y[l] = padded_table(sh, dh, 3, 3);
            const uint32_t* xt = &c->coef_host[c->coef_x[l]];
            const uint32_t* yt = &c->coef_host[c->coef_y[l]];
            int ldw = 1, rows = 1, lo, hi;
            for (int xa = -4; xa < dw + 3; xa += 256) {
                vo_reflect_range(xa, std::min(xa + 255, dw + 2), dw, lo, hi);
                const int sx0 = (int)(xt[lo + 4] >> 16) & ~3, sx1 = std::min((int)(xt[hi + 4] >> 16) + 1, sw - 1);
                ldw = std::max(ldw, (sx1 - sx0) / 4 + 1);
            }
            for (int ya = -3; ya < dh + 3; ya += PR_ROWS) {
                vo_reflect_range(ya, std::min(ya + PR_ROWS - 1, dh + 2), dh, lo, hi);
                const int sy0 = (int)(yt[lo + 3] >> 16), sy1 = std::min((int)(yt[hi + 3] >> 16) + 1, sh - 1);
                rows = std::max(rows, sy1 - sy0 + 1);
            }
            if (ldw > 256 || vo_resize_lds_bytes(ldw, rows) > VO_RESIZE_LDS_MAX)
                return c->fail(VO_ERR_INVALID, "scale factor %g is too large for the resize tile", (double)sf);
            c->coef_ldw[l] = ldw;
            c->coef_rows[l] = rows;
        }
        VO_HIP(c, hipStreamSynchronize(c->stream));
        VO_TRY(vo_ensure(c, c->d_geom, sizeof(PyrGeom)));
        VO_HIP(c, hipMemcpy(c->d_geom.p, &g, sizeof(PyrGeom), hipMemcpyHostToDevice));
        if (!c->coef_host.empty()) {
            VO_TRY(vo_ensure(c, c->d_coef, c->coef_host.size() * sizeof(uint32_t)));
            VO_HIP(c, hipMemcpy(c->d_coef.p, c->coef_host.data(), c->coef_host.size() * sizeof(uint32_t),
                                hipMemcpyHostToDevice));
        }
        c->geom = g;
        c->geom_w = w;
        c->geom_h = h;
        c->geom_levels = nlevels;
        c->geom_scale = sf;
        c->geom_border = border;
        c->geom_nfeatures = -1;
    }
    if (c->geom_nfeatures != nfeatures) {
        // per-level feature quotas (cv ORB_Impl::detectAndCompute)
        const float factor = (float)(1.0 / (double)sf);
        float ndesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; l++) {
            c->quota[l] = cv_roundf(ndesired);
            sum += c->quota[l];
            ndesired *= factor;
        }
        c->quota[nlevels - 1] = std::max(nfeatures - sum, 0);
        c->geom_nfeatures = nfeatures;
    }
    // workspace
    const size_t nf = (size_t)nframes;
    const int kp_cap = std::max(c->lim.max_features, 16) * 2;  // nfeatures + response ties
    c->kp_cap = kp_cap;
    VO_TRY(vo_ensure(c, c->d_pyr, nf * c->geom.pyr_bytes));
    VO_TRY(vo_ensure(c, c->d_blur, nf * c->geom.pyr_bytes));
    VO_TRY(vo_ensure(c, c->d_score, nf * c->geom.pyr_bytes));
    VO_TRY(vo_ensure(c, c->d_masks, (nf * c->geom.segs_total + VO_MASK_PAD) * sizeof(unsigned long long)));
    VO_TRY(vo_ensure(c, c->d_segbase, nf * c->geom.segs_total * sizeof(int)));
    VO_TRY(vo_ensure(c, c->d_lvlofs, nf * (VO_ORB_MAX_LEVELS + 1) * sizeof(int)));
    VO_TRY(vo_ensure(c, c->d_cand, nf * c->max_cand * sizeof(Cand)));
    VO_TRY(vo_ensure(c, c->d_selcand, nf * (size_t)std::max(c->geom.nlevels, 1) * c->sel_cap * sizeof(Cand)));
    VO_TRY(vo_ensure(c, c->d_selcnt, nf * VO_ORB_MAX_LEVELS * sizeof(int)));
    VO_TRY(vo_ensure(c, c->d_flags, (nf + 1) * sizeof(int)));
    VO_TRY(vo_ensure(c, c->d_sel, nf * kp_cap * sizeof(Cand)));
    VO_TRY(vo_ensure(c, c->d_kp, nf * kp_cap * sizeof(vo_keypoint)));
    VO_TRY(vo_ensure(c, c->d_desc, nf * kp_cap * 32));
    VO_TRY(vo_ensure(c, c->d_nkp, (nf + 1) * sizeof(int)));
    VO_TRY(vo_ensure_pinned(c, c->h_lvlofs, nf * (VO_ORB_MAX_LEVELS + 1) * sizeof(int)));
    VO_TRY(vo_ensure_pinned(c, c->h_nkp, (nf + 1) * sizeof(int)));
    VO_TRY(vo_ensure_pinned(c, c->h_flags, (nf + 1) * sizeof(int)));
    c->feat_frames = nframes;
    return VO_OK;
}

// The frame-wide prefix scan of the mask table and the row-major candidate lists: what the plain FAST entry
// point returns and what the host selection of flagged frames starts from.  The device selection reads the mask
// table itself.  zero_flags: false for the redo of a batch, whose selection flags must survive.
static int feat_make_lists(vo_ctx* c, int nframes, bool zero_flags) {
    const PyrGeom& g = c->geom;
    const PyrGeom* dg = c->d_geom.as<PyrGeom>();
    launch_row_scan(c->stream, dg, c->d_masks.as<unsigned long long>(), c->d_segbase.as<int>(), c->d_lvlofs.as<int>(),
                    zero_flags ? c->d_flags.as<int>() : nullptr, nframes);
    launch_cand_write(c->stream, dg, g, c->d_masks.as<unsigned long long>(), c->d_segbase.as<int>(),
                      c->d_score.as<uint8_t>(), c->cand_nms, c->d_cand.as<Cand>(), c->max_cand, nframes);
    return vo_check_launch(c, "candidate lists");
}

// Stage A (asynchronous): padded pyramid, fused FAST + NMS + border filter tiles -> mask table and score map;
// lists: also the row-major candidate lists.  d_img: nframes images in device memory.
int feat_detect(vo_ctx* c, const uint8_t* d_img, size_t frame_stride, int stride, int nframes, int fast_threshold,
                int border, int nms, bool lists) {
    const PyrGeom& g = c->geom;
    const PyrGeom* dg = c->d_geom.as<PyrGeom>();
    hipStream_t st = c->stream;
    uint8_t* pyr = c->d_pyr.as<uint8_t>();
    if (border != c->geom_border)
        return c->fail(VO_ERR_INVALID, "feat_detect: border %d, geometry was laid out for %d", border, c->geom_border);
    launch_pyr_level0(st, d_img, frame_stride, stride, dg, g, pyr, nframes);
    for (int l = 1; l < g.nlevels; l++)
        launch_pyr_resize(st, dg, g, pyr, l, c->d_coef.as<uint32_t>() + c->coef_x[l],
                          c->d_coef.as<uint32_t>() + c->coef_y[l], c->coef_ldw[l], c->coef_rows[l], nframes);
    {
        ProfScope ps(c, VO_PROF_FAST);
        launch_fast_nms(st, dg, g, pyr, c->d_score.as<uint8_t>(), c->d_masks.as<unsigned long long>(),
                        std::min(std::max(fast_threshold, 0), 255), border, nms, nframes, c->d_flags.as<int>());
    }
    c->cand_nms = nms;
    if (lists) return feat_make_lists(c, nframes, true);
    return vo_check_launch(c, "feature detection");
}

// Level offsets to the host; when a frame has more FAST corners than the candidate lists hold (cv has no
// such limit) the lists grow and are written again from the mask table, segment bases and score map,
// which are complete.  Synchronises.
static int feat_cand_capacity(vo_ctx* c, int nframes, bool* grew = nullptr) {
    if (grew) *grew = false;
    const PyrGeom& g = c->geom;
    hipStream_t st = c->stream;
    const size_t lv_bytes = (size_t)nframes * (VO_ORB_MAX_LEVELS + 1) * sizeof(int);
    VO_HIP(c, hipMemcpyAsync(c->h_lvlofs.p, c->d_lvlofs.p, lv_bytes, hipMemcpyDeviceToHost, st));
    VO_HIP(c, hipStreamSynchronize(st));
    const int* lv = c->h_lvlofs.as<int>();
    int most = 0;
    for (int f = 0; f < nframes; f++) most = std::max(most, lv[f * (VO_ORB_MAX_LEVELS + 1) + g.nlevels]);
    if (most <= c->max_cand) return VO_OK;
    if (grew) *grew = true;  // the regrown lists start without Harris responses: the caller redoes every frame
    const size_t nf = (size_t)c->feat_frames;
    c->max_cand = (most + 65535) & ~65535;
    VO_TRY(vo_ensure(c, c->d_cand, nf * c->max_cand * sizeof(Cand)));
    launch_cand_write(st, c->d_geom.as<PyrGeom>(), g, c->d_masks.as<unsigned long long>(), c->d_segbase.as<int>(),
                      c->d_score.as<uint8_t>(), c->cand_nms, c->d_cand.as<Cand>(), c->max_cand, nframes);
    return vo_check_launch(c, "candidate lists");
}

// Candidate lists to the host (plain FAST output, host selection path).  Synchronises.
int feat_fetch_cands(vo_ctx* c, int nframes, const char* only) {
    const PyrGeom& g = c->geom;
    hipStream_t st = c->stream;
    VO_TRY(feat_cand_capacity(c, nframes));
    VO_TRY(vo_ensure_pinned(c, c->h_cand, (size_t)nframes * c->max_cand * sizeof(Cand)));
    const int* lv = c->h_lvlofs.as<int>();
    for (int f = 0; f < nframes; f++) {
        const int total = lv[f * (VO_ORB_MAX_LEVELS + 1) + g.nlevels];
        if (total > 0 && (!only || only[f]))
            VO_HIP(c, hipMemcpyAsync(c->h_cand.as<Cand>() + (size_t)f * c->max_cand,
                                     c->d_cand.as<Cand>() + (size_t)f * c->max_cand, (size_t)total * sizeof(Cand),
                                     hipMemcpyDeviceToHost, st));
    }
    VO_HIP(c, hipStreamSynchronize(st));
    return VO_OK;
}

// ORB keypoint selection on the device (asynchronous): per frame and level retainBest(2*quota) on
// the FAST score, Harris, retainBest(quota), in cv's order.  n_bound: keypoints per frame the later
// launches are sized for.  d_flags[f] != 0 afterwards means frame f needs the host path.
int feat_select_dev(vo_ctx* c, int nframes, int n_bound) {
    hipStream_t st = c->stream;
    Quotas q;
    for (int l = 0; l < VO_ORB_MAX_LEVELS; l++) q.q[l] = l < c->geom.nlevels ? c->quota[l] : 0;
    // d_flags was zeroed by the detector of this batch (feat_detect)
    launch_orb_select(st, c->d_pyr.as<uint8_t>(), c->d_geom.as<PyrGeom>(), c->geom,
                      c->d_masks.as<unsigned long long>(), c->d_score.as<uint8_t>(), c->cand_nms, q, c->sel_cap,
                      c->d_selcand.as<Cand>(), c->d_selcnt.as<int>(), c->d_flags.as<int>(), c->d_nkp.as<int>() + c->feat_off, n_bound,
                      nframes, c->d_sel.as<Cand>(), c->kp_cap);
    return vo_check_launch(c, "keypoint selection");
}

// The same selection on the host with libstdc++ itself (exact for every input), for the frames whose
// device selection raised a flag (h_flags, h_nkp hold the device's flags and counts of the batch): Harris
// for every candidate, the flagged frames' candidates to the host, selection on host threads, upload of
// their keypoints.  The other frames keep their device picks; d_flags afterwards marks the frames whose
// keypoints come from d_sel.  Synchronises.
int feat_select_host(vo_ctx* c, int nframes, int* max_n_out) {
    const PyrGeom& g = c->geom;
    bool grew = false;
    VO_TRY(feat_make_lists(c, nframes, false));     // the device selection worked from the mask table alone
    VO_TRY(feat_cand_capacity(c, nframes, &grew));  // before Harris: a regrown list starts without responses
    std::vector<char> redo(nframes);
    int* hflags = c->h_flags.as<int>();
    for (int f = 0; f < nframes; f++) redo[f] = grew || hflags[f] != 0;
    // only for the flagged frames (d_flags still holds the device's flags; after a regrowth every frame)
    launch_harris(c->stream, c->d_pyr.as<uint8_t>(), c->d_geom.as<PyrGeom>(), c->d_lvlofs.as<int>(),
                  c->d_cand.as<Cand>(), c->max_cand, c->max_cand, grew ? nullptr : c->d_flags.as<int>(), nframes);
    VO_TRY(vo_check_launch(c, "harris"));
    VO_TRY(feat_fetch_cands(c, nframes, redo.data()));
    VO_TRY(vo_ensure_pinned(c, c->h_sel, (size_t)nframes * c->kp_cap * sizeof(Cand)));
    const int* lv = c->h_lvlofs.as<int>();
    // (feat_off: the batch's frame f is frame feat_off + f of the caller's chunk, whose tables the counts live in)
    int* nkp = c->h_nkp.as<int>() + c->feat_off;
    // frames are independent: the (sequential, libstdc++-defined) selections run on host threads
    auto select_frame = [&](int f, std::vector<RespIdx>& v) -> int {
        const Cand* cand = c->h_cand.as<Cand>() + (size_t)f * c->max_cand;
        Cand* sel = c->h_sel.as<Cand>() + (size_t)f * c->kp_cap;
        int n = 0;
        for (int l = 0; l < g.nlevels; l++) {
            const int b = lv[f * (VO_ORB_MAX_LEVELS + 1) + l], e = lv[f * (VO_ORB_MAX_LEVELS + 1) + l + 1];
            v.resize(e - b);
            for (int i = b; i < e; i++) v[i - b] = {(float)cand[i].score, i};
            retain_best(v, 2 * c->quota[l]);
            for (auto& r : v) r.response = cand[r.idx].harris;
            retain_best(v, c->quota[l]);
            if (n + (int)v.size() > c->kp_cap) return -1;
            for (auto& r : v) sel[n++] = cand[r.idx];
        }
        return n;
    };
    std::vector<int> todo;
    for (int f = 0; f < nframes; f++)
        if (redo[f]) todo.push_back(f);
    const int nthreads = std::max(1, std::min({(int)todo.size(), 16, (int)std::thread::hardware_concurrency()}));
    if (nthreads == 1) {
        std::vector<RespIdx> v;
        for (int f : todo) nkp[f] = select_frame(f, v);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++)
            pool.emplace_back([&, t]() {
                std::vector<RespIdx> v;
                for (size_t i = t; i < todo.size(); i += nthreads) nkp[todo[i]] = select_frame(todo[i], v);
            });
        for (auto& th : pool) th.join();
    }
    int max_n = 0;
    for (int f = 0; f < nframes; f++) {
        const int n = nkp[f];  // frames not redone: the device's count
        if (n < 0) return c->fail(VO_ERR_CAPACITY, "frame %d selects more than %d keypoints", f, c->kp_cap);
        max_n = std::max(max_n, n);
        if (n > 0 && redo[f])
            VO_HIP(c, hipMemcpyAsync(c->d_sel.as<Cand>() + (size_t)f * c->kp_cap,
                                     c->h_sel.as<Cand>() + (size_t)f * c->kp_cap, (size_t)n * sizeof(Cand),
                                     hipMemcpyHostToDevice, c->stream));
    }
    nkp[nframes] = 0;
    VO_HIP(c, hipMemcpyAsync(c->d_nkp.as<int>() + c->feat_off, nkp, (size_t)(nframes + 1) * sizeof(int), hipMemcpyHostToDevice,
                             c->stream));
    for (int f = 0; f < nframes; f++) hflags[f] = redo[f] ? (hflags[f] ? hflags[f] : VO_SEL_OVERSIZE) : 0;
    VO_HIP(c, hipMemcpyAsync(c->d_flags.p, hflags, (size_t)nframes * sizeof(int), hipMemcpyHostToDevice, c->stream));
    *max_n_out = max_n;
    return VO_OK;
}

// Stage B (asynchronous): IC angles (optional), blur, rBRIEF for d_nkp[f] keypoints per frame;
// max_n bounds the launches.
int feat_describe(vo_ctx* c, int nframes, int max_n, bool angles, bool host_selection) {
    static const Taps7 taps = gauss_taps();
    const PyrGeom& g = c->geom;
    const PyrGeom* dg = c->d_geom.as<PyrGeom>();
    // the batch's outputs start at frame feat_off of the chunk's tables (1: frame 0 was carried over from the
    // chunk before, see pairs_submit)
    int* d_nkp = c->d_nkp.as<int>() + c->feat_off;
    vo_keypoint* d_kp = c->d_kp.as<vo_keypoint>() + (size_t)c->feat_off * c->kp_cap;
    uint8_t* d_desc = c->d_desc.as<uint8_t>() + (size_t)c->feat_off * c->kp_cap * 32;
    // host_selection = the redo of a batch whose first pass already blurred the pyramid
    if (!host_selection)
        launch_gauss7(c->stream, c->d_pyr.as<uint8_t>(), dg, g, c->d_blur.as<uint8_t>(), taps, g.nlevels, nframes);
    const bool split = c->opt_orb_split != 0;  // orientation and description as two kernels (VO_OPT_ORB_SPLIT)
    if (angles && !split) {
        // d_sel: the device selection's lists flattened by k_sel_flatten, with the rows of flagged frames
        // replaced by the host selection in the redo
        launch_orient_rbrief(c->stream, c->d_pyr.as<uint8_t>(), c->d_blur.as<uint8_t>(), dg, c->d_sel.as<Cand>(),
                             d_nkp, c->kp_cap, max_n, d_kp, d_desc, nframes);
        return vo_check_launch(c, "feature description");
    }
    if (angles)
        launch_ic_angle(c->stream, c->d_pyr.as<uint8_t>(), dg, host_selection ? c->d_sel.as<Cand>() : nullptr,
                        c->d_flags.as<int>(), g.nlevels * c->sel_cap, c->d_selcand.as<Cand>(), c->d_selcnt.as<int>(),
                        c->sel_cap, d_nkp, c->kp_cap, max_n, d_kp, nframes);
    launch_rbrief(c->stream, dg, c->d_blur.as<uint8_t>(), d_kp, d_nkp, c->kp_cap, max_n, d_desc, nframes);
    return vo_check_launch(c, "feature description");
}

// Launch bound for the stages after the device selection: nfeatures plus room for response ties.
int feat_n_bound(const vo_ctx* c, int nfeatures) {
    return std::min({c->kp_cap, nfeatures + nfeatures / 8 + 64, c->lim.max_points});
}

// ORB keypoints of nframes device-resident frames into d_kp / d_desc / d_nkp: device selection,
// host selection when a flag was raised.  Synchronises once; h_nkp holds the counts.
int feat_orb_frames(vo_ctx* c, const uint8_t* d_img, size_t frame_stride, int stride, int nframes,
                    const vo_orb_params* p, int* max_n_out) {
    VO_TRY(feat_detect(c, d_img, frame_stride, stride, nframes, p->fast_threshold, p->edge_threshold, 1, false));
    const int n_bound = feat_n_bound(c, p->nfeatures);
    VO_TRY(feat_select_dev(c, nframes, n_bound));
    VO_TRY(feat_describe(c, nframes, n_bound, true, false));
    VO_HIP(c, hipMemcpyAsync(c->h_flags.p, c->d_flags.p, (size_t)nframes * sizeof(int), hipMemcpyDeviceToHost,
                             c->stream));
    VO_HIP(c, hipMemcpyAsync(c->h_nkp.p, c->d_nkp.p, (size_t)(nframes + 1) * sizeof(int), hipMemcpyDeviceToHost,
                             c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    int flagged = 0, max_n = 0;
    for (int f = 0; f < nframes; f++) {
        flagged |= c->h_flags.as<int>()[f];
        max_n = std::max(max_n, c->h_nkp.as<int>()[f]);
    }
    c->last_flags = flagged;
    if (flagged) {
        VO_TRY(feat_select_host(c, nframes, &max_n));
        if (max_n > 0) VO_TRY(feat_describe(c, nframes, max_n, true, true));
        VO_HIP(c, hipStreamSynchronize(c->stream));
    }
    *max_n_out = max_n;
    return VO_OK;
}

static int upload_image(vo_ctx* c, const uint8_t* gray, int w, int h, int stride) {
    return vo_upload_image(c, c->d_img, gray, w, h, stride);
}

extern "C" {

int vo_fast_detect(vo_ctx* c, const uint8_t* gray, int w, int h, int stride, int threshold, int nms,
                   vo_keypoint* out, int cap, int* n) {
    if (!c) return VO_ERR_INVALID;
    if (!gray || !n || w <= 0 || h <= 0 || stride < w || cap < 0 || (cap > 0 && !out))
        return c->fail(VO_ERR_INVALID, "vo_fast_detect: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    *n = 0;
    if (w < 7 || h < 7) return VO_OK;  // cv: no pixel has a full ring
    VO_TRY(feat_setup(c, w, h, nullptr, 1));
    VO_TRY(upload_image(c, gray, w, h, stride));
    VO_TRY(feat_detect(c, c->d_img.as<uint8_t>(), (size_t)w * h, w, 1, threshold, 0, nms != 0, true));
    VO_TRY(feat_fetch_cands(c, 1));
    const int total = c->h_lvlofs.as<int>()[1];
    const Cand* cand = c->h_cand.as<Cand>();
    *n = total;
    for (int i = 0; i < total && i < cap; i++) {
        vo_keypoint k;
        k.x = (float)cand[i].x;
        k.y = (float)cand[i].y;
        k.size = 7.f;
        k.angle = -1.f;
        k.response = (float)cand[i].score;
        k.octave = 0;
        k.class_id = -1;
        out[i] = k;
    }
    if (total > cap) return c->fail(VO_ERR_CAPACITY, "vo_fast_detect: %d keypoints, capacity %d", total, cap);
    return VO_OK;
}

int vo_debug_read_level(vo_ctx* c, int which, int frame, int level, uint8_t* out, int cap, int* w, int* h) {
    if (!c) return VO_ERR_INVALID;
    if (!out || !w || !h || frame < 0 || frame >= c->feat_frames || level < 0 || level >= c->geom.nlevels ||
        (which != VO_DBG_PYRAMID && which != VO_DBG_BLUR))
        return c->fail(VO_ERR_INVALID, "vo_debug_read_level: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    const LevelGeom& L = c->geom.lv[level];
    *w = L.w;
    *h = L.h;
    if (cap < L.w * L.h) return c->fail(VO_ERR_CAPACITY, "vo_debug_read_level: need %d bytes", L.w * L.h);
    const uint8_t* base = (which == VO_DBG_PYRAMID ? c->d_pyr.as<uint8_t>() : c->d_blur.as<uint8_t>()) +
                          (size_t)frame * c->geom.pyr_bytes + L.offset + (size_t)VO_PADY * L.pitch + VO_PADX;
    VO_HIP(c, hipMemcpy2DAsync(out, L.w, base, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

int vo_orb_detect_compute(vo_ctx* c, const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                          vo_keypoint* kp, uint8_t* desc, int cap, int* n) {
    if (!c) return VO_ERR_INVALID;
    if (!gray || !n || w <= 0 || h <= 0 || stride < w || cap < 0 || !orb_params_ok(p))
        return c->fail(VO_ERR_INVALID, "vo_orb_detect_compute: bad argument or unsupported ORB parameter");
    if (p->nfeatures > c->lim.max_features)
        return c->fail(VO_ERR_CAPACITY, "nfeatures %d exceeds max_features %d", p->nfeatures,
                       c->lim.max_features);
    VO_HIP(c, hipSetDevice(c->device));
    *n = 0;
    VO_TRY(feat_setup(c, w, h, p, 1));
    VO_TRY(upload_image(c, gray, w, h, stride));
    int max_n = 0;
    VO_TRY(feat_orb_frames(c, c->d_img.as<uint8_t>(), (size_t)w * h, w, 1, p, &max_n));
    const int cnt = c->h_nkp.as<int>()[0];
    *n = cnt;
    if (cnt > cap) return c->fail(VO_ERR_CAPACITY, "vo_orb_detect_compute: %d keypoints, capacity %d", cnt, cap);
    if (cnt == 0) return VO_OK;
    VO_TRY(vo_xfer_begin(c));
    if (kp) VO_TRY(vo_xfer_out(c, kp, c->d_kp.p, (size_t)cnt * sizeof(vo_keypoint)));
    if (desc) VO_TRY(vo_xfer_out(c, desc, c->d_desc.p, (size_t)cnt * 32));
    VO_TRY(vo_xfer_finish(c));
    return VO_OK;
}

int vo_orb_compute(vo_ctx* c, const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                   vo_keypoint* kp, int* n_inout, uint8_t* desc) {
    if (!c) return VO_ERR_INVALID;
    if (!gray || !n_inout || *n_inout < 0 || w <= 0 || h <= 0 || stride < w || !orb_params_ok(p) ||
        (*n_inout > 0 && (!kp || !desc)))
        return c->fail(VO_ERR_INVALID, "vo_orb_compute: bad argument or unsupported ORB parameter");
    VO_HIP(c, hipSetDevice(c->device));
    const int n_in = *n_inout;
    // KeyPointsFilter::runByImageBorder on level-0 coordinates, then stable grouping by octave
    std::vector<vo_keypoint> kept;
    const int b = p->edge_threshold;
    int nlevels = 0;
    for (int i = 0; i < n_in; i++) {
        const vo_keypoint& k = kp[i];
        if (w > 2 * b && h > 2 * b && k.x >= (float)b && k.x < (float)(w - b) && k.y >= (float)b &&
            k.y < (float)(h - b)) {
            if (k.octave < 0 || k.octave >= VO_ORB_MAX_LEVELS)
                return c->fail(VO_ERR_INVALID, "vo_orb_compute: keypoint octave %d out of range", k.octave);
            nlevels = std::max(nlevels, k.octave);
            kept.push_back(k);
        }
    }
    nlevels++;
    std::vector<vo_keypoint> sorted;
    sorted.reserve(kept.size());
    for (int l = 0; l < nlevels; l++)
        for (const auto& k : kept)
            if (k.octave == l) sorted.push_back(k);
    const int cnt = (int)sorted.size();
    *n_inout = cnt;
    if (cnt == 0) return VO_OK;
    if (cnt > 2 * c->lim.max_features)
        return c->fail(VO_ERR_CAPACITY, "vo_orb_compute: %d keypoints exceed the workspace (%d)", cnt,
                       2 * c->lim.max_features);
    vo_orb_params pp = *p;
    pp.nlevels = nlevels;  // only the levels the keypoints use are built
    VO_TRY(feat_setup(c, w, h, &pp, 1));
    VO_TRY(upload_image(c, gray, w, h, stride));
    launch_pyr_level0(c->stream, c->d_img.as<uint8_t>(), (size_t)w * h, w, c->d_geom.as<PyrGeom>(), c->geom,
                      c->d_pyr.as<uint8_t>(), 1);
    for (int l = 1; l < c->geom.nlevels; l++)
        launch_pyr_resize(c->stream, c->d_geom.as<PyrGeom>(), c->geom, c->d_pyr.as<uint8_t>(), l,
                          c->d_coef.as<uint32_t>() + c->coef_x[l], c->d_coef.as<uint32_t>() + c->coef_y[l],
                          c->coef_ldw[l], c->coef_rows[l], 1);
    int* nkp = c->h_nkp.as<int>();
    nkp[0] = cnt;
    nkp[1] = 0;
    VO_HIP(c, hipMemcpyAsync(c->d_nkp.p, nkp, 2 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipMemcpyAsync(c->d_kp.p, sorted.data(), (size_t)cnt * sizeof(vo_keypoint), hipMemcpyHostToDevice,
                             c->stream));
    VO_TRY(feat_describe(c, 1, cnt, false, false));
    VO_HIP(c, hipMemcpyAsync(desc, c->d_desc.p, (size_t)cnt * 32, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(kp, sorted.data(), (size_t)cnt * sizeof(vo_keypoint));
    return VO_OK;
}

}  // extern "C"
