// kernels.h — device-side data layouts and the launch wrappers of libvo_hip.so's kernels.
// Internal; the ABI is include/vo_hip.h.
#ifndef VO_KERNELS_H
#define VO_KERNELS_H

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/vo_types.h"

static inline __host__ __device__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- ORB pyramid geometry (one copy on the host, one in device memory) ------------------------
// Every frame owns one packed buffer holding all levels (level 0 is a copy of the caller's image).
// A level is stored with a border: VO_PADX bytes left of x = 0 and VO_PADY rows above y = 0, so
// pixel (0,0) sits on a 16-byte boundary, rows are `pitch` (multiple of 64) bytes apart and the
// 3-pixel reflect-101 frame the 7x7 blur reads is real memory.  The same layout is used for the
// pyramid, the blurred pyramid and the (sparsely written) FAST score map.
#define VO_PADX 16
#define VO_PADY 4
#ifndef VO_FT_H
#define VO_FT_H 64  // rows of a detector tile (k_fast_nms); tiles are 64 pixels wide
#endif
struct LevelGeom {
    int w, h;
    int pitch;
    int row0;         // first row of this level in the frame-wide row table
    unsigned offset;  // byte offset of the padded level inside a frame's packed buffer
    float scale;      // cv layerScale
    // Detector region: the rectangle whose pixels can become keypoints under the border filter, plus the
    // one-pixel ring non-maximum suppression reads (rx0 is a multiple of 4).  Tiles, mask segments and
    // mask rows are counted inside it: segment (y, s) covers x = rx0 + 64 s .. + 63 of row ry0 + y.
    int rx0, ry0, rw, rh;
    int seg0, segs;   // first 64-pixel row segment of this level in the frame-wide mask table, segments per row
    int tile0, tiles_x;  // first 64 x VO_FT_H detector tile of this level, tiles per tile row
};
struct PyrGeom {
    int nlevels;
    int rows_total;      // sum of level heights
    unsigned pyr_bytes;  // packed bytes per frame
    int tiles_total;     // 64 x VO_FT_H tiles over all levels
    int segs_total;      // 64-pixel row segments over all levels
    LevelGeom lv[VO_ORB_MAX_LEVELS];
};

// Range [lo, hi] of reflect-101 coordinates that the destination coordinates a..b (a <= b, a >= -n + 1,
// b <= 2n - 2) of an axis of length n map to.
static inline __host__ __device__ void vo_reflect_range(int a, int b, int n, int& lo, int& hi) {
    lo = a;
    hi = b;
    if (a < 0) {
        lo = b >= 0 ? 0 : -b;
        hi = hi > -a ? hi : -a;
    }
    if (b >= n) {
        const int r = 2 * n - 2 - b;
        lo = lo < r ? lo : r;
        hi = n - 1;
    }
    lo = lo < 0 ? 0 : (lo > n - 1 ? n - 1 : lo);
    hi = hi < 0 ? 0 : (hi > n - 1 ? n - 1 : hi);
}
#define PR_ROWS 16  // destination rows per block of the pyramid resize
// LDS tile of the resize kernel: `rows` pair rows of one dword per source pixel (4 * ldw of them) + 4
static inline size_t vo_resize_lds_bytes(int ldw, int rows) { return (size_t)rows * (4 * (size_t)ldw + 4) * 4; }
#define VO_RESIZE_LDS_MAX (160 * 1024)

struct Taps7 {
    int t[7];
};
// blocks of the blur launch: level l owns blocks [b0[l], b0[l+1]), bx[l] of them per block row
#define VO_PACK_TABLES 8
struct PackArgs {  // up to eight dword tables gathered into one buffer, table i at dword at[i] (unused: ndw 0)
    const uint32_t* src[VO_PACK_TABLES];
    int ndw[VO_PACK_TABLES];
    int at[VO_PACK_TABLES];
};
void launch_pack_results(hipStream_t st, const PackArgs& a, int total_dw, uint32_t* out);

struct GaussGrid {
    int nlevels;
    int b0[VO_ORB_MAX_LEVELS];
    int bx[VO_ORB_MAX_LEVELS];
};

// candidate corner / selected keypoint in level coordinates (16 bytes)
// words of slack after the last frame's mask table: k_row_scan reads whole 512-word steps
#define VO_MASK_PAD (20 * 512)

struct Cand {
    int16_t x, y;
    int16_t level;
    int16_t score;  // FAST score
    float harris;   // Harris response (ORB only)
    uint32_t pad;
};

struct Quotas {
    int q[VO_ORB_MAX_LEVELS];
};

// flags raised by the device-side keypoint selection (per frame); any flag sends the batch through
// the host selection path, which is exact for every input
#define VO_SEL_OVERSIZE 1   // a level has more candidates than the LDS selection buffer holds
#define VO_SEL_DEPTH 2      // introselect ran out of its depth budget (heap-select fallback in libstdc++)
#define VO_SEL_OVERFLOW 4   // more keypoints selected than the launch bound of the later stages

// ---- features -----------------------------------------------------------------------------
void launch_pyr_level0(hipStream_t st, const uint8_t* img, size_t frame_stride, int stride, const PyrGeom* d_geom,
                       const PyrGeom& hg, uint8_t* pyr, int nframes);
void launch_pyr_resize(hipStream_t st, const PyrGeom* d_geom, const PyrGeom& hg, uint8_t* pyr, int level,
                       const uint32_t* xtab, const uint32_t* ytab, int ldw, int lrows, int nframes);
void launch_orient_rbrief(hipStream_t st, const uint8_t* pyr, const uint8_t* blur, const PyrGeom* d_geom,
                          const Cand* flat, const int* nkp, int kp_cap, int max_n, vo_keypoint* out, uint8_t* desc,
                          int nframes);
void launch_fast_nms(hipStream_t st, const PyrGeom* d_geom, const PyrGeom& hg, const uint8_t* pyr, uint8_t* score,
                     unsigned long long* masks, int threshold, int border, int nms, int nframes, int* flags);
void launch_row_scan(hipStream_t st, const PyrGeom* d_geom, const unsigned long long* masks, int* segbase,
                     int* lvlofs, int* flags, int nframes);
void launch_cand_write(hipStream_t st, const PyrGeom* d_geom, const PyrGeom& hg, const unsigned long long* masks,
                       const int* segbase, const uint8_t* score, int nms, Cand* cand, int max_cand, int nframes);
void launch_harris(hipStream_t st, const uint8_t* pyr, const PyrGeom* d_geom, const int* lvlofs, Cand* cand,
                   int max_cand, int grid_cand, const int* only_flagged, int nframes);
// selcand: [frame][level][sel_cap] (frame stride nlevels * sel_cap)
void launch_orb_select(hipStream_t st, const uint8_t* pyr, const PyrGeom* d_geom, const PyrGeom& hg,
                       const unsigned long long* masks, const uint8_t* score, int nms, const Quotas& quota,
                       int sel_cap, Cand* selcand, int* selcnt, int* flags, int* nkp, int n_bound, int nframes,
                       Cand* flat, int kp_cap);
void launch_ic_angle(hipStream_t st, const uint8_t* pyr, const PyrGeom* d_geom, const Cand* sel, const int* flags,
                     int sel_fs, const Cand* selcand, const int* selcnt, int sel_cap, const int* nkp, int kp_cap,
                     int max_n, vo_keypoint* out, int nframes);
void launch_carry_frame(hipStream_t st, vo_keypoint* kp, uint8_t* desc, int* nkp, int kp_cap, int from);
void launch_gauss7(hipStream_t st, const uint8_t* pyr, const PyrGeom* d_geom, const PyrGeom& hg, uint8_t* blur,
                   const Taps7& taps, int nlevels_used, int nframes);
void launch_rbrief(hipStream_t st, const PyrGeom* d_geom, const uint8_t* blur, const vo_keypoint* kps,
                   const int* nkp, int kp_cap, int max_n, uint8_t* desc, int nframes);

// ---- tracker ------------------------------------------------------------------------------
// Tracker pyramid (cv::buildOpticalFlowPyramid): every level with a frame of TK_PAD pixels on all
// sides, rows `pitch` (multiple of 64) bytes apart, all levels of an image in one packed buffer.
// The derivative image (short2 per pixel) uses the same pixel indices.
#define TK_PAD 32
#define TK_MAX_LEVELS 8
struct TrackLevel {
    int w, h;
    int pitch;
    unsigned offset;  // pixel index of the padded level inside the packed buffer
};
struct TrackGeom {
    int nlevels;
    unsigned bytes;  // packed pixels per image
    TrackLevel lv[TK_MAX_LEVELS];
};
struct LkArgs {
    int win_w, win_h;
    int max_count;
    int channels;
    int generic;  // 1: never the kernel specialised for the 21 x 21 window
    float min_eig;
    double eps2;
};
// A batch of tracking problems of one geometry (vo_lk_track_batch; a single call is a batch of one): image
// pair i has n[i] points, stored pts_stride entries after those of pair i - 1.
#define VO_LK_MAX_BATCH 64
struct LkBatch {
    int count;
    int pts_stride;
    int n[VO_LK_MAX_BATCH];
};
// nimg images img_stride bytes apart -> nimg pyramids g.bytes apart
void launch_lk_pyramid(hipStream_t st, const uint8_t* img, size_t img_stride, int stride, const TrackGeom& g,
                       uint8_t* pyr, int nimg);
void launch_lk_scharr(hipStream_t st, const TrackGeom& g, const uint8_t* pyr, short2* deriv, int nimg);
void launch_lk_track(hipStream_t st, const TrackGeom& g, const uint8_t* pyrI, const short2* deriv,
                     const uint8_t* pyrJ, const float2* prev_pts, float2* next_pts, uint8_t* status,
                     const LkBatch& b, const LkArgs& a);

// ---- matcher ------------------------------------------------------------------------------
// Packed neighbour key: distance << 22 | train index (numeric order == (distance, index) order).
// 64-byte rows reach distance 512 = 2^9: 9 + 22 bits fit, and train indices stay below 2^22
// (cv packs the image index above 18 bits, so its own limit is 262 144 rows).
#define VO_KEY_INVALID 0xffffffffu
#define VO_KEY_SHIFT 22
struct KnnArgs {
    const uint32_t* q;    // descriptors, `words` uint32 per row
    const uint32_t* t;
    size_t q_stride;      // uint32 between consecutive pairs' query sets
    size_t t_stride;
    const int* d_n;       // optional device counts: nq = d_n[pair], nt = d_n[pair + 1]
    int nq, nt;           // used when d_n == nullptr; otherwise upper bounds for the grid
    int words;            // 8 (32-byte descriptors) or 16 (64-byte)
    int splits;           // train range split across blockIdx.y (1: the launch writes knn / rev itself)
    int cross;            // also run the swapped direction (train rows as queries) into rev
    uint2* part;          // [pair (+ npairs: reverse)][split][nq_cap] partial top-2 keys, splits > 1 only
    uint2* knn;           // [pair][nq_cap] keys of the two nearest train rows of every query
    uint2* rev;           // [pair][nq_cap] keys of the two nearest queries of every train row (cross only)
    int nq_cap;           // row capacity of part / knn / rev (>= nq, and >= nt with cross)
    int npairs;
    int mfma;             // 32-byte rows: 1 matrix-core kernel, 0 popcount kernel (vo_ctx::opt_knn_mfma)
};
#define VO_KNN_STATIC_MAX 8192  // rows a side of vo_knn2_hamming_static (include/vo_hip.h documents it)
int knn_static_init(int device);  // address of this device's static matcher workspace (k_match.hip)
void launch_knn2(hipStream_t st, const KnnArgs& a);
void launch_knn2_merge(hipStream_t st, const KnnArgs& a);  // no-op unless splits > 1

typedef vo_dmatch GoodMatch;  // filtered matches leave the device in cv::DMatch layout
struct FilterArgs {
    const uint2* knn;          // [pair][nq_cap]
    const uint2* rev;          // [pair][nq_cap] nearest queries of the train rows, or nullptr: no cross-check
    const vo_keypoint* kq;     // query keypoints of pair p at kq + p*kp_stride
    const vo_keypoint* kt;
    size_t kp_stride;          // in keypoints
    const int* d_n;            // optional device counts (nq = d_n[pair])
    int nq;
    int nq_cap;
    float ratio, max_disp;
    GoodMatch* good;           // [pair][nq_cap]
    float2* p0;                // [pair][nq_cap] compacted query points
    float2* p1;
    int* ngood;                // [pair]
    int npairs;
};
void launch_match_filter(hipStream_t st, const FilterArgs& a);

#endif
