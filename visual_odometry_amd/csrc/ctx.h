// ctx.h — internal context of libvo_hip.so: one HIP stream, grow-only device workspace,
// pinned staging, error string.  Not part of the ABI (include/vo_hip.h is).
#ifndef VO_CTX_H
#define VO_CTX_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/vo_hip.h"
#include "kernels.h"
#include "akaze_types.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// cv::RNG 5-subsets for every point count (uint16 [maxn + 1][iters][5]); depends only on
// (seed, iters, maxn), so the contexts of one device share one copy.
struct SampleTable {
    int device = 0;
    uint64_t seed = 0;
    int iters = 0, maxn = 0;
    void* p = nullptr;
    ~SampleTable();
};

struct ProfSlot {
    int launches = 0;
    double total_ms = 0;
};

struct vo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    vo_limits lim;
    char err[512];
    bool prof_on = false;
    ProfSlot prof[VO_PROF_COUNT];
    hipEvent_t ev0[VO_PROF_COUNT] = {}, ev1[VO_PROF_COUNT] = {};  // per slot: brackets may nest

    // ---- feature workspace (per frame strides: geom.pyr_bytes, max_cand, kp_cap) ---------
    DevBuf d_img;     // uploaded frames (host-pointer entry points)
    DevBuf d_pyr;     // packed padded pyramid, all levels (level 0 = copy of the caller's frame)
    DevBuf d_blur;    // blurred pyramid, all levels
    DevBuf d_score;   // FAST score + 1 at keypoint positions (sparsely written), pyramid layout
    DevBuf d_masks;   // uint64 [frame][segs_total]: keypoint bit per pixel, one word per 64-pixel row segment
    DevBuf d_segbase; // int [frame][segs_total]: candidate-list position of each segment's first keypoint
    DevBuf d_lvlofs;  // int [frame][VO_ORB_MAX_LEVELS + 1]
    DevBuf d_cand;    // Cand [frame][max_cand]: row-major candidates, level after level
    DevBuf d_selcand; // Cand [frame][level][sel_cap]: device selection, the picks of every level (whole records)
    DevBuf d_selcnt;  // int [frame][VO_ORB_MAX_LEVELS]
    DevBuf d_flags;   // int [frame]: VO_SEL_* flags of the device selection
    DevBuf d_sel;     // Cand [frame][kp_cap]: selected keypoints in keypoint order (k_sel_flatten; rows of flagged frames from the host selection)
    DevBuf d_coef;    // resize coefficient tables
    DevBuf d_geom;    // PyrGeom
    DevBuf d_kp;      // vo_keypoint [frame][kp_cap]
    DevBuf d_desc;    // uint8 [frame][kp_cap][32]
    DevBuf d_nkp;     // int [frame] (+1 spare)
    PinnedBuf h_lvlofs, h_cand, h_sel, h_nkp, h_misc, h_flags;
    PinnedBuf h_img;  // two staging slots for vo_upload_image
    int h_img_slot = 0;
    // Pinned arena for the small transfers of an entry point (points, masks, matrices): a copy from or to
    // the caller's pageable memory blocks the host for ~30 us each, a pinned one is queued in ~5.  Inputs
    // are copied into the arena and sent asynchronously; outputs land in the arena and vo_xfer_finish
    // (synchronise + copy out) hands them to the caller.  The arena is recycled at every finish.
    DevBuf d_pack;       // result tables of a batch gathered for one device -> host copy
    PinnedBuf h_pack;
    size_t pack_at[8] = {}, pack_bytes[8] = {};  // byte offset and size of table i in d_pack / h_pack
    double cam_cache[4] = {0, 0, 0, 0};  // camera table on the device: K it was built from, pairs it covers
    int cam_pairs = 0;
    const void* cam_ptr = nullptr;
    PinnedBuf h_xfer;
    size_t xfer_used = 0;
    struct XferOut {
        void* host;
        size_t offset, bytes;
    };
    std::vector<XferOut> xfer_out;
    int max_cand = 65536;  // candidates per frame the lists hold; grows when a frame has more
    int cand_nms = 1;      // nms flag of the last detection (to rewrite the lists after growing)
    int last_flags = 0;  // OR of the VO_SEL_* flags of the last batch
    int feat_off = 0;    // chunk frame that the feature batch's frame 0 is (1: frame 0 carried over from the chunk before)
    int sel_cap = 6144;  // candidates per (frame, level) the device selection holds in LDS (72 KB)
    int kp_cap = 0;
    int feat_frames = 0;
    // Switches between equivalent kernel schedules (vo_ctx_set_option; the environment variables of the same
    // names, VO_KNN_MFMA ..., are read ONCE, when the context is created)
    int opt_knn_mfma = 1;      // 32-byte rows on the matrix-core matcher (0: popcount kernel)
    int opt_orb_split = 0;     // 1: orientation and description as two kernels
    int opt_seq_prefetch = 0;  // 1: pinned frames prefetched on a copy stream of the context's own
    int opt_akaze_graph = 1;   // AKAZE scale space replayed from a captured hipGraph (0: plain launches)
    int opt_solver = 0;        // five-point solver of findEssentialMat: 0 one hypothesis per 16-lane group, 1 per lane
    int opt_lk_generic = 0;    // 1: the default 21 x 21 LK window through the generic kernels
    int opt_knn_splits = 0;    // > 0: train-range splits of a single 2-NN problem (tuning hook)

    PyrGeom geom;  // host copy of d_geom
    int quota[VO_ORB_MAX_LEVELS];
    int geom_w = 0, geom_h = 0, geom_levels = 0, geom_nfeatures = -1, geom_border = -1;
    float geom_scale = 0;
    std::vector<uint32_t> coef_host;
    std::vector<size_t> coef_x, coef_y;  // per level offsets into d_coef (uint32 units)
    std::vector<int> coef_ldw, coef_rows;  // per level LDS tile of the resize kernel (dwords per row, rows)

    // ---- tracker workspace ---------------------------------------------------------------
    DevBuf d_tk_pyr[2];  // packed tracker pyramids of two images (the last `next` image stays cached)
    DevBuf d_tk_deriv;   // short2 per pixel: Scharr derivatives of the `prev` pyramid, zero frame
    TrackGeom tk_geom;
    int tk_w = 0, tk_h = 0, tk_win_w = 0, tk_win_h = 0, tk_max_level = -1;
    int tk_last = -1;    // slot of d_tk_pyr holding the pyramids of the last call's `next` images
    int tk_cap = 0;      // image pairs the tracker workspace holds side by side (vo_lk_track_batch)
    int tk_last_count = 0;  // image pairs of the last call (a prev == NULL call must bring as many)

    // ---- AKAZE workspace (vo_akaze.cpp) ------------------------------------------------------
    DevBuf d_ak;       // float arena: work planes, then Lt / Lsmooth / Lx / Ly / Ldet of every level, AkContrast
    DevBuf d_ak_mask;  // counter + AkExt list: the 3 x 3 maxima of every level with their neighbourhoods
    DevBuf d_ak_tab;   // area-resize tables, descriptor comparison table
    DevBuf d_ak_kp, d_ak_desc;
    PinnedBuf h_ak;    // host copy of the counter and the list
    std::vector<uint8_t> ak_hmask;   // dense masks of all levels on the host, all zero between calls
    std::vector<AkExt> ak_ext_sorted;
    std::vector<int> ak_survivors;   // positions that survived the cross-level filtering (debug hook)
    struct AkGraphKey {  // what the captured launch sequence depends on
        int w, h;
        vo_akaze_params prm;
        const void *arena, *mask, *tab, *img;
        size_t ext_cap;
    };
    AkGraphKey ak_key;
    hipGraphExec_t ak_graph = nullptr;  // phase 1 of vo_akaze_detect_compute, replayed per image
    int ak_levels = 0;
    size_t ak_total = 0, ak_plane0 = 0;
    std::vector<int> ak_lw, ak_lh;
    std::vector<size_t> ak_at;

    // ---- matcher workspace ---------------------------------------------------------------
    DevBuf d_q, d_t;    // descriptor uploads (host-pointer entry points)
    DevBuf d_part;      // uint2 [pair][split][nq_cap]
    DevBuf d_knn;       // uint2 [pair][nq_cap]
    DevBuf d_rev;       // uint2 [pair][nq_cap]: nearest queries of the train rows (cross-check only)
    DevBuf d_good;      // GoodMatch [pair][nq_cap]
    DevBuf d_ngood;     // int [pair]
    DevBuf d_p0, d_p1;  // float2 [pair][nq_cap]
    DevBuf d_kq, d_kt;  // keypoint uploads (host-pointer match filter)
    PinnedBuf h_knn, h_good;

    // ---- geometry workspace ----------------------------------------------------------------
    std::shared_ptr<SampleTable> samples;  // uint16 [max_points + 1][iters][5]
    DevBuf d_x1, d_x2;   // double2 [pair][n_cap]: K-normalised points
    DevBuf d_models;     // double [pair][H][10][9]
    DevBuf d_nmodels;    // int [pair][H]
    DevBuf d_counts;     // int [pair][H][10] inlier counts (RANSAC) / float medians (LMedS)
    DevBuf d_state;      // EstState [pair]
    DevBuf d_mask;       // uint8 [pair][n_cap]
    DevBuf d_q4;         // double [pair][4][4][n_cap]: triangulations of the 4 candidates
    DevBuf d_cmask;      // uint8 [pair][4][n_cap]
    DevBuf d_pose;       // PoseState [pair]
    DevBuf d_pmask;      // uint8 [pair][n_cap]: recoverPose mask
    DevBuf d_pts4;       // double [pair][4][n_cap]: winner's points
    DevBuf d_fp0, d_fp1; // float2 uploads (host-pointer entry points)
    DevBuf d_cam;        // CamK [pair]
    DevBuf d_proj;       // double[24]: P0, P1 (vo_triangulate)
    PinnedBuf h_state, h_pose, h_mask, h_pts;

    // ---- vo_pairs_run_dev bookkeeping --------------------------------------------------------
    int pipe_frames = 0, pipe_ncap = 0;
    int pend_frames = 0;  // frames of the batch queued by pairs_submit and not yet collected
    bool pend_details = false;  // that batch also brings its match / mask tables to the host
    double pend_K[9];
    vo_pair_params pend_prm;
    std::vector<int> pipe_nkp;
    std::vector<int> pipe_ngood;

    int fail(int code, const char* fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, sizeof(err), fmt, ap);
        va_end(ap);
        return code;
    }
};

#define VO_HIP(ctx, expr)                                                                    \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return (ctx)->fail(VO_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,                \
                               hipGetErrorString(e__), __FILE__, __LINE__);                  \
    } while (0)

#define VO_TRY(expr)                    \
    do {                                \
        int rc__ = (expr);              \
        if (rc__ != VO_OK) return rc__; \
    } while (0)

int vo_ensure(vo_ctx* ctx, DevBuf& b, size_t bytes);
int vo_ensure_pinned(vo_ctx* ctx, PinnedBuf& b, size_t bytes);
int vo_check_launch(vo_ctx* ctx, const char* what);
int vo_upload_image(vo_ctx* c, DevBuf& dst, const uint8_t* gray, int w, int h, int stride);
int vo_xfer_begin(vo_ctx* c);                                            // first thing in an entry point
int vo_xfer_in(vo_ctx* c, void* dev, const void* host, size_t bytes);    // host -> device on c->stream
int vo_xfer_out(vo_ctx* c, void* host, const void* dev, size_t bytes);   // device -> host, valid after finish
int vo_xfer_finish(vo_ctx* c);                                           // synchronise, deliver the outputs

// features (vo_features.cpp)
int feat_setup(vo_ctx* c, int w, int h, const vo_orb_params* p, int nframes);
int feat_detect(vo_ctx* c, const uint8_t* d_img, size_t frame_stride, int stride, int nframes, int fast_threshold,
                int border, int nms, bool lists);  // lists: also the row-major candidate lists (plain FAST output)
int feat_fetch_cands(vo_ctx* c, int nframes, const char* only = nullptr);  // only[f] != 0: fetch frame f
int feat_select_dev(vo_ctx* c, int nframes, int n_bound);
int feat_select_host(vo_ctx* c, int nframes, int* max_n_out);
int feat_describe(vo_ctx* c, int nframes, int max_n, bool angles, bool host_selection);
int feat_n_bound(const vo_ctx* c, int nfeatures);
int feat_orb_frames(vo_ctx* c, const uint8_t* d_img, size_t frame_stride, int stride, int nframes,
                    const vo_orb_params* p, int* max_n_out);

// geometry (vo_geometry.cpp)
struct EstArgs;
int geom_prepare(vo_ctx* c, int npairs, int n_cap, uint64_t seed, int max_iters);
int geom_find_essential_dev(vo_ctx* c, int npairs, int n_cap, const float2* d_p0, const float2* d_p1,
                            const int* d_n, int n_host, const double* K, int method, double prob,
                            double threshold, int max_iters);
int geom_recover_pose_dev(vo_ctx* c, int npairs, int n_cap, const double* d_E, size_t e_stride,
                          const float2* d_p0, const float2* d_p1, const int* d_n, int n_host,
                          const double* K, double dist_thresh, bool use_mask);

// device-resident batch (vo_pipeline.cpp): queue everything / wait and fill the results
// carry: frame 0 of the batch is the last frame of the batch this context ran before (same geometry and
// parameters); its keypoints are taken over, d_frames points at frame 1, and n_frames still counts frame 0
int pairs_submit(vo_ctx* c, const uint8_t* d_frames, int n_frames, int w, int h, int stride, size_t frame_stride,
                 const double* K, const vo_pair_params* prm, bool details, bool carry = false);
int pairs_collect(vo_ctx* c, vo_pair_result* results, const vo_seq_outputs* out, size_t pair0);

// profiling bracket (no-ops unless vo_prof_enable(ctx, 1))
struct ProfScope {
    vo_ctx* c;
    int which;
    bool on;
    ProfScope(vo_ctx* ctx, int w) : c(ctx), which(w), on(ctx->prof_on) {
        if (on) (void)hipEventRecord(c->ev0[which], c->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(c->ev1[which], c->stream);
        (void)hipEventSynchronize(c->ev1[which]);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev0[which], c->ev1[which]);
        c->prof[which].launches++;
        c->prof[which].total_ms += ms;
    }
};

#endif
