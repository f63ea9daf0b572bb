// k_features.hip — gfx950 kernels for FAST-9/16 and ORB: padded pyramid, fused FAST score + NMS
// tiles, ordered candidate lists, device-side keypoint selection (cv's retainBest with libstdc++'s
// nth_element / partition permutation reproduced on a wavefront), IC angle, 7x7 blur, rBRIEF.
// Integer pipeline throughout; the few float expressions (Harris epilogue, fastAtan2, pattern
// rotation) follow cv's operation order and the file is compiled with -ffp-contract=off so nothing
// is fused.  Behaviour follows SURVEY.md A.1/A.2; reference call sites:
// src/features/feature_detector.cpp:4-5,15,24,28,41.
#include "kernels.h"

namespace {

// pointer to pixel (0,0) of level l of frame f in a packed buffer
__device__ __forceinline__ const uint8_t* level_px(const uint8_t* buf, const PyrGeom* g, int f, int l) {
    return buf + (size_t)f * g->pyr_bytes + g->lv[l].offset + (size_t)VO_PADY * g->lv[l].pitch + VO_PADX;
}
__device__ __forceinline__ uint8_t* level_px(uint8_t* buf, const PyrGeom* g, int f, int l) {
    return buf + (size_t)f * g->pyr_bytes + g->lv[l].offset + (size_t)VO_PADY * g->lv[l].pitch + VO_PADX;
}

typedef unsigned short vo_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(vo_u16x2, a), __builtin_bit_cast(vo_u16x2, b), c, false);
}

// number of set bits of a wave mask below the calling lane (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ int mbcnt(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Pins a loaded value at this point of the program: without it the compiler sinks a load whose only
// use is conditional into the branch, which turns a batch of independent loads into a chain of
// load -> wait -> load.
// Three-input forms written out: left to itself the compiler re-associates the arc networks of k_fast_nms into
// more two-input instructions (63 and 44 where 40 do).
__device__ __forceinline__ uint32_t and3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x80" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int min3i(int a, int b, int c) {
    int d;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int max3i(int a, int b, int c) {
    int d;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <class T>
__device__ __forceinline__ void pin_loaded(T& v) {
    asm volatile("" : "+v"(v));
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Level 0: copy of the caller's image into the padded layout, with the 3-pixel reflect-101 frame.
// One thread per aligned 16-byte chunk of the destination (x = -16, 0, 16, ...).  Blocks
// blockIdx.x < nint copy interior chunks: one unaligned 16-byte load (caller rows have any alignment)
// and one aligned store.  The last block column assembles the chunks that touch the left or right edge
// byte by byte through reflect-101, one thread per (row, edge chunk), so the waves of the interior never
// run that path (the padding bytes beyond the 3-pixel frame get reflected values too; nothing reads
// them as pixels).
// ---------------------------------------------------------------------------------------------
#define L0_CHUNKS 4  // chunks per thread of the interior blocks: four independent 16-byte loads in flight
__global__ __launch_bounds__(256) void k_pyr_level0(const uint8_t* __restrict__ img, size_t frame_stride, int stride,
                                                    int w, int h, int pitch, unsigned pyr_bytes, unsigned loff,
                                                    int nedge_blocks, unsigned nch_magic, uint8_t* __restrict__ pyr) {
    const int f = blockIdx.z;
    const int nch = w - 15 > 0 ? (w - 16) / 16 + 1 : 0;  // interior chunks of a row: x = 0, 16, ... < end_int
    const int end_int = nch * 16;
    const uint8_t* fimg = img + (size_t)f * frame_stride;
    uint8_t* dst = pyr + (size_t)f * pyr_bytes + loff;
    if ((int)blockIdx.x >= nedge_blocks) {  // the (slow, few) edge blocks go first and run beside the interior
        const int bi = (int)blockIdx.x - nedge_blocks;
        // interior: the (row, chunk) pairs of the frame in one sequence, chunk fastest, so that every lane has
        // a chunk whatever the width; row = index / nch by a 32-bit reciprocal (exact for index < 2^16 * nch)
        const int total = nch * (h + 6);
        uint4 v[L0_CHUNKS];
        int at[L0_CHUNKS];
#pragma unroll
        for (int i = 0; i < L0_CHUNKS; i++) {
            const int idx = min(bi * (256 * L0_CHUNKS) + 256 * i + (int)threadIdx.x, total - 1);
            const int row = nch == 1 ? idx : (int)__umulhi((unsigned)idx, nch_magic), ch = idx - row * nch;
            const int y = row - 3;
            at[i] = y * pitch + 16 * ch;
            __builtin_memcpy(&v[i], fimg + (size_t)reflect101(y, h) * stride + 16 * ch, 16);
        }
#pragma unroll
        for (int i = 0; i < L0_CHUNKS; i++) {
            pin_loaded(v[i].x);
            pin_loaded(v[i].y);
            pin_loaded(v[i].z);
            pin_loaded(v[i].w);
        }
#pragma unroll
        for (int i = 0; i < L0_CHUNKS; i++)  // an index clamped to the last pair stores that pair again: same bytes
            *reinterpret_cast<uint4*>(dst + (ptrdiff_t)at[i]) = v[i];
    } else {
        // edge chunks: x = -16, then end_int, end_int + 16, ... < w + 3.  One thread per (row, edge chunk), the
        // pairs of the frame in one sequence (a row has two or three such chunks: a thread per row and chunk column
        // left 62 lanes of every wave idle, and these blocks were nine tenths of the kernel's instructions)
        const int nedge = 1 + (w + 3 - end_int + 15) / 16;
        const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
        const int row = idx / nedge, e = idx - row * nedge;
        const int y = row - 3;
        if (y >= h + 3) return;
        const int x16 = e == 0 ? -16 : end_int + 16 * (e - 1);
        const uint8_t* src = fimg + (size_t)reflect101(y, h) * stride;
        uint32_t d[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            d[q] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) d[q] |= (uint32_t)src[reflect101(x16 + 4 * q + j, w)] << (8 * j);
        }
        *reinterpret_cast<uint4*>(dst + (ptrdiff_t)y * pitch + x16) = make_uint4(d[0], d[1], d[2], d[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// Pyramid level l from level l-1: cv::resize(INTER_LINEAR_EXACT), 8.8 fixed-point coefficients
// from host tables (entries: ofs << 16 | c1, c0 = 256 - c1).  The tables are padded and already
// reflected: xtab[i] belongs to x = i - 4, ytab[i] to y = i - 3, so the reflect-101 frame of the
// new level is produced by the same code as its interior.  A block makes 256 x PR_ROWS pixels: the
// source rectangle they read (known from the tables, which are monotone in the reflected
// coordinate) is staged in LDS as one dword per source pixel holding the pixel and the one below it
// in its 16-bit halves -- the operand the packed 16-bit arithmetic of the interpolation takes -- so an
// output pixel costs two LDS dword reads (one ds_read2) and five vector instructions.
// ---------------------------------------------------------------------------------------------
struct ResizeArgs {  // geometry of the two levels, from the host copy: nothing to fetch before the tables
    int dw, dh, dpitch, sw, sh, spitch;
    unsigned soff, doff, pyr_bytes;  // level offsets include the padding rows and columns
    int ldw;                         // source dwords per tile row (the level's maximum)
    unsigned magic;                  // ceil(65536 / ldw): tid / ldw for tid < 256 by one multiplication
    int ngroups, grows;              // tile fill: 256 / ldw row groups of `grows` pair rows each
};

__global__ __launch_bounds__(256) void k_pyr_resize(uint8_t* __restrict__ pyr, ResizeArgs a,
                                                    const uint32_t* __restrict__ xtab,
                                                    const uint32_t* __restrict__ ytab) {
    extern __shared__ __attribute__((aligned(16))) uint32_t rs_tile[];
    const int f = blockIdx.z, tid = threadIdx.x;
    const int dw = a.dw, dh = a.dh, dpitch = a.dpitch, sw = a.sw, sh = a.sh, spitch = a.spitch, ldw = a.ldw;
    const int xa = -4 + 256 * (int)blockIdx.x, ya = -3 + PR_ROWS * (int)blockIdx.y;
    // the thread's own coefficients do not depend on the tile: they travel with the tile's range lookups
    const int x4 = xa + 4 * (tid & 63);
    const uint4 tx4 = *reinterpret_cast<const uint4*>(xtab + (x4 < dw + 3 ? x4 : -4) + 4);
    uint32_t tyk[PR_ROWS / 4];
#pragma unroll
    for (int i = 0; i < PR_ROWS / 4; i++) tyk[i] = ytab[min(ya + (tid >> 6) + 4 * i, dh + 2) + 3];
    int lo, hi;
    vo_reflect_range(xa, min(xa + 255, dw + 2), dw, lo, hi);
    const int sx0 = (int)(xtab[lo + 4] >> 16) & ~3, sx1 = min((int)(xtab[hi + 4] >> 16) + 1, sw - 1);
    vo_reflect_range(ya, min(ya + PR_ROWS - 1, dh + 2), dh, lo, hi);
    const int sy0 = (int)(ytab[lo + 3] >> 16), sy1 = min((int)(ytab[hi + 3] >> 16) + 1, sh - 1);
    const int ndw = (sx1 - sx0) / 4 + 1, nrows = sy1 - sy0 + 1;  // <= ldw, <= the rows the launch sized LDS for
    const int LP = 4 * ldw + 4;  // dwords per pair row: one per source pixel + the right neighbour of the last
    const uint8_t* s = pyr + (size_t)f * a.pyr_bytes + a.soff + (ptrdiff_t)sy0 * spitch + sx0;
    // Tile fill: thread (group, c) takes source dword c of the rows group * grows ... + grows (one more than it
    // pairs up), all loads in flight, 32-bit offsets from a block-uniform base; pair row r of the tile is
    // (row r, row r + 1), the last one (row nrows - 1 twice) only ever read with weight 0 on its lower half.
    {
        const int grp = (int)((tid * a.magic) >> 16), c = tid - grp * ldw;
        const int cc = min(c, ndw - 1);  // idle threads (c >= ndw, or a group beyond the last) load in range
        const int rend = (c < ndw && grp < a.ngroups) ? min((grp + 1) * a.grows, nrows) : 0;  // pair rows < rend are mine
        for (int b0 = 0; b0 < a.grows; b0 += 8) {
            const int r0 = grp * a.grows + b0;
            uint32_t d[9];
#pragma unroll
            for (int u = 0; u < 9; u++)
                d[u] = *reinterpret_cast<const uint32_t*>(s + (uint32_t)(__mul24(min(r0 + u, nrows - 1), spitch) + 4 * cc));
#pragma unroll
            for (int u = 0; u < 9; u++) pin_loaded(d[u]);
            uint32_t* at = rs_tile + __mul24(r0, LP) + 4 * c;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (r0 + u < rend) {
                    uint4 e;
                    e.x = __builtin_amdgcn_perm(d[u + 1], d[u], 0x0c040c00u);
                    e.y = __builtin_amdgcn_perm(d[u + 1], d[u], 0x0c050c01u);
                    e.z = __builtin_amdgcn_perm(d[u + 1], d[u], 0x0c060c02u);
                    e.w = __builtin_amdgcn_perm(d[u + 1], d[u], 0x0c070c03u);
                    *reinterpret_cast<uint4*>(at + u * LP) = e;
                }
            }
        }
    }
    __syncthreads();
    if (x4 >= dw + 3) return;
    // Packed 16-bit arithmetic: the (upper row, lower row) samples of a column share a dword, so one
    // v_pk_mul_lo_u16 + v_pk_mad_u16 interpolates both rows horizontally (h <= 256 * 255 fits 16 bits) and
    // one v_dot2_u32_u16 with (cy0, cy1) is the vertical pass; the same integers as the scalar form.  The
    // right / lower neighbour is read unclamped: where cv clamps it (last column / row) its weight is 0.
    const uint32_t tx[4] = {tx4.x, tx4.y, tx4.z, tx4.w};
    vo_u16x2 CX0[4], CX1[4];
    int ox[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        ox[j] = (int)(tx[j] >> 16) - sx0;
        const unsigned short c1 = (unsigned short)(tx[j] & 0xffff), c0 = (unsigned short)(256 - c1);
        CX0[j] = vo_u16x2{c0, c0};
        CX1[j] = vo_u16x2{c1, c1};
    }
    uint8_t* dst = pyr + (size_t)f * a.pyr_bytes + a.doff;
#pragma unroll
    for (int i = 0; i < PR_ROWS / 4; i++) {
        const int y = ya + (tid >> 6) + 4 * i;
        if (y >= dh + 3) break;
        const uint32_t ty = tyk[i];
        const int oy = ty >> 16, cy1 = ty & 0xffff, cy0 = 256 - cy1;
        const uint32_t CY = (uint32_t)cy0 | ((uint32_t)cy1 << 16);
        const uint32_t* row = rs_tile + __umul24(oy - sy0, LP);
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const vo_u16x2 A = __builtin_bit_cast(vo_u16x2, row[ox[j]]), B = __builtin_bit_cast(vo_u16x2, row[ox[j] + 1]);
            const vo_u16x2 hp = A * CX0[j] + B * CX1[j];
            v[j] = udot2(__builtin_bit_cast(uint32_t, hp), CY, 32768u);
        }
        // the weights sum to 2^16, so v < 2^24 and the pixel is byte 2 of v: no clamp, three byte permutes
        const uint32_t lo2 = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), hi2 = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u);
        *reinterpret_cast<uint32_t*>(dst + (ptrdiff_t)y * dpitch + x4) = __builtin_amdgcn_perm(hi2, lo2, 0x05040100u);
    }
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16 + cornerScore<16> + 3x3 strict-greater NMS + border filter, fused per 64 x VO_FT_H (64) tile.
//   phase 0  tile pixels (+4 halo) -> LDS with aligned dword loads
//   phase A  compass test (ring points 0,4,8,12: a 9-arc holds two adjacent ones) on every pixel
//            of the 66 x 66 score region; survivors (about one pixel in eight) are compacted
//   phase B  full 16-point arc test on the survivors, dense lanes: one multiply-add per ring point carries
//            both polarities, three-input AND / OR networks find the arc; corners are compacted in place
//   phase C  cornerScore of the corners (min3 / max3 network on the arc's polarity) -> LDS score map
//            (score + 1, 0 = not a corner)
//   phase D  NMS + border filter of the corners inside the tile -> per-row 64-bit masks in LDS,
//            score byte of each keypoint -> sparse global score map
//   phase E  one 64-bit mask per tile row -> global mask table
// The tile is 64 rows high so that a wave's lists (a quarter of the rows) fill the lanes of phases B and C:
// with 32 rows a wave had ~73 survivors and ~28 corners, i.e. two steps of B at 57 % and one of C at 44 %.
// Phases A, B and C run per wave on the wave's own rows and lists (a wave's LDS operations complete in
// order, so no block barrier separates them); the block meets after phase 0, before the NMS and before E.
// Row-major keypoint order (what cv's FAST emits) is rebuilt from the masks by k_row_scan /
// k_cand_write.
// ---------------------------------------------------------------------------------------------
#define FT_W 64
#define FT_H VO_FT_H
#define SM_W (FT_W + 2)
#define SM_H (FT_H + 2)
#define PX_W (FT_W + 8)
#define PX_H (FT_H + 8)
// tools/micro/fast_phases.sh builds variants that stop after phase n (-DVO_FAST_STOP=n) to count the
// instructions of each phase; the library itself compiles the hook to nothing
#ifdef VO_FAST_STOP
#define VO_FAST_PHASE_END(n)                                                                      \
    if (VO_FAST_STOP == (n)) {                                                                    \
        unsigned acc_ = cntA[tid & 3] + cntB[tid & 3];  /* keep the phases' results alive */      \
        for (int i_ = tid; i_ < 4 * FT_LCAP; i_ += 256) acc_ += listA[i_];                        \
        acc_ += sm[(tid % SM_H) * PX_W + tid % SM_W];                                             \
        if (acc_ == 0x12345678u) rowmask[0][0] = 1;                                               \
        __syncthreads();                                                                          \
        phase_e();                                                                                \
        return;                                                                                   \
    }
#else
#define VO_FAST_PHASE_END(n)
#endif
// entries of a wave's list: its rows of phase A (every fourth of the score region) x 64 + one more step
#define FT_LCAP (((SM_H + 3) / 4) * 64 + 64)

struct FastTiles {  // first tile of every level, by value: the level of a block is found without a load
    int nlevels;
    int tile0[VO_ORB_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void k_fast_nms(const uint8_t* __restrict__ pyr, const PyrGeom* __restrict__ g,
                                                  uint8_t* __restrict__ score, unsigned long long* __restrict__ masks,
                                                  int threshold, int border, int nms, FastTiles ft,
                                                  int* __restrict__ flags) {
    // the per-frame selection flags of the batch start at zero (nframes + 1 entries; the detector is the first
    // kernel of a batch that the selection follows)
    if (flags && blockIdx.x == 0 && threadIdx.x == 0) {
        flags[blockIdx.y] = 0;
        if (blockIdx.y == 0) flags[gridDim.y] = 0;
    }
    __shared__ uint32_t px4[PX_H * (PX_W / 4)];
    // score map, rows PX_W apart like the pixel tile: pixel (sy, sx) of the score region is element
    // sy * PX_W + sx of the map and has its ring centred on element (sy + 3) * PX_W + sx + 3 of the tile
    __shared__ __attribute__((aligned(4))) uint8_t sm[SM_H * PX_W];
    // survivor / corner lists: one private region per wave (no atomics, no cross-wave ordering
    // needed: the lists only feed the score map); entries are that element index (< 2^13; bit 14 marks a dark
    // arc), so no phase has to rebuild an LDS address from coordinates.  The corner list of phase B
    // is written over the survivor list it is compacted from: a step reads entries i0 .. i0 + 63 before
    // it writes, and writes below i0 + 64.
    __shared__ uint16_t listA[4 * FT_LCAP];
    uint16_t* const listB = listA;
    __shared__ uint32_t rowmask[FT_H][2];
#ifdef VO_FAST_STOP
    __shared__ int cntA[4], cntB[4];  // only the phase-counting variants read the survivor counts across waves
#endif
    const uint8_t* px = reinterpret_cast<const uint8_t*>(px4);
    const int f = blockIdx.y, tid = threadIdx.x;
    int l = 0;
#pragma unroll
    for (int i = 1; i < VO_ORB_MAX_LEVELS; i++)
        if (i < ft.nlevels && (int)blockIdx.x >= ft.tile0[i]) l = i;
    const LevelGeom L = g->lv[l];
    const int local = blockIdx.x - L.tile0;
    const int tix = local % L.tiles_x, tiy = local / L.tiles_x;
    const int tx0 = L.rx0 + tix * FT_W, ty0 = L.ry0 + tiy * FT_H;
    const int w = L.w, h = L.h, pitch = L.pitch;
    const uint8_t* img = level_px(pyr, g, f, l);

    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // wave index in an SGPR

    // phase 0: thread (r0, c) of 14 x 18 takes dword c of the rows r0, r0 + 14, ... of the pixel tile.  A tile
    // whose rectangle lies inside the level's padded storage (all but some tiles of the last tile row /
    // column) needs one 32-bit offset per thread and a wave-uniform base per step; the others clamp.
    {
        constexpr int RW = PX_W / 4, RPS = 256 / RW, STEPS = (PX_H + RPS - 1) / RPS;
        const int r0 = (tid * ((65536 + RW - 1) / RW)) >> 16, c4 = tid - r0 * RW;
        static_assert(RW == 18 && RPS == 14, "r0 = tid / 18 by multiplication is exact for tid < 256");
        const bool act = tid < RW * RPS;
        uint32_t d[STEPS] = {};  // all loads of the thread in flight together
        const bool interior = tx0 - 4 + PX_W <= pitch - VO_PADX && ty0 - 4 + PX_H <= h + VO_PADY &&
                              tx0 - 4 >= -VO_PADX && ty0 - 4 >= -VO_PADY;
        if (interior) {
            const uint8_t* base = img + (ptrdiff_t)(ty0 - 4) * pitch + (tx0 - 4);
            const uint32_t voff = (uint32_t)(r0 * pitch + 4 * c4);
#pragma unroll
            for (int s = 0; s < STEPS; s++)
                if (act && r0 + RPS * s < PX_H)
                    d[s] = *reinterpret_cast<const uint32_t*>(base + (ptrdiff_t)(RPS * s) * pitch + voff);
        } else {
            const int gx = min(max(tx0 - 4 + 4 * c4, -VO_PADX), pitch - VO_PADX - 4);
#pragma unroll
            for (int s = 0; s < STEPS; s++) {
                const int gy = min(max(ty0 - 4 + r0 + RPS * s, -VO_PADY), h + VO_PADY - 1);
                if (act && r0 + RPS * s < PX_H)
                    d[s] = *reinterpret_cast<const uint32_t*>(img + (ptrdiff_t)gy * pitch + gx);
            }
        }
#pragma unroll
        for (int s = 0; s < STEPS; s++) pin_loaded(d[s]);
#pragma unroll
        for (int s = 0; s < STEPS; s++)
            if (act && r0 + RPS * s < PX_H) px4[tid + RW * RPS * s] = d[s];
    }
    static_assert((SM_H * PX_W) % 4 == 0, "score map is cleared in dwords");
    {
        constexpr int SM_DW = SM_H * PX_W / 4;
        uint32_t* sm32 = reinterpret_cast<uint32_t*>(&sm[0]);
#pragma unroll
        for (int s = 0; s < SM_DW / 256; s++) sm32[tid + 256 * s] = 0;
        if (tid < SM_DW % 256) sm32[tid + 256 * (SM_DW / 256)] = 0;
    }
    if (tid < FT_H * 2) (&rowmask[0][0])[tid] = 0;
    __syncthreads();
    // phase E: one 64-bit mask per tile row -> global mask table
    auto phase_e = [&]() {
        if (tid < FT_H) {
            const int yr = tiy * FT_H + tid;  // row inside the detector region
            if (yr < L.rh)
                masks[(size_t)f * g->segs_total + L.seg0 + (size_t)yr * L.segs + tix] =
                    (unsigned long long)rowmask[tid][0] | ((unsigned long long)rowmask[tid][1] << 32);
        }
    };
    VO_FAST_PHASE_END(0);

    // A wave's order of LDS operations is the order of its instructions, so between a phase's writes to the
    // wave's own list and the next phase's reads of it only the compiler has to be held back: no block barrier.
    auto wave_lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // phase A: compass test.  Wave w takes the rows w, w + 4, ... of the score region with lane =
    // column; columns 64 and 65 of all rows are one more step for the last waves.
    // Branch-free per row: the LDS reads are in range for every (sy, sx) of the score region.
    int na = 0;  // survivors of this wave (wave-uniform)
    uint16_t* mineA = listA + wv * FT_LCAP;
    constexpr int RING0 = 3 * PX_W + 3;  // from a pixel's element index to the centre of its ring in the tile
    {
        // As in the arc test below, one multiply-add per ring point tests it against both thresholds:
        // P = (r - v + t) + ((v + t - r) << 16) = C - 65535 r with C = 65535 v + 65537 t has bit 15 set iff r is
        // darker than v - t and bit 31 iff brighter than v + t.  A 9-arc holds two adjacent compass points, so
        // the pixel survives iff S = (P0 | P8) & (P4 | P12) has one of the two bits.  The border conditions
        // go into t (a threshold no difference reaches, 16000, where the pixel cannot be a corner: the column
        // part once per lane, the row part wave-uniform), so no lane mask is rebuilt from scalar conditions.
        auto compass = [&](int idx, uint32_t tt) {
            const uint8_t* c = px + idx + RING0;
            uint32_t v = c[0], r0 = c[3 * PX_W], r4 = c[3], r8 = c[-3 * PX_W], r12 = c[-3];
            // the five reads happen for every (sy, sx), outside any branch on the border conditions
            pin_loaded(v), pin_loaded(r0), pin_loaded(r4), pin_loaded(r8), pin_loaded(r12);
            // (24-bit multiplies by name: behind pin_loaded the compiler no longer knows the values are bytes)
            const uint32_t C = (uint32_t)__mul24((int)v, 65535) + tt;
            auto P = [&](uint32_t r) { return C + (uint32_t)__mul24((int)r, -65535); };
            const uint32_t S = (P(r0) | P(r8)) & (P(r4) | P(r12));
            const bool surv = (S & 0x80008000u) != 0;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(surv);
            if (m) {
                if (surv) mineA[na + mbcnt(m)] = (uint16_t)idx;
                na += __popcll(m);
            }
        };
        // a pixel outside the detector region (keypoint rectangle + the ring the NMS reads) needs no score
        const int rx1 = L.rx0 + L.rw, ry1 = L.ry0 + L.rh;
        auto border_tt = [&](int sy, int sx, bool in) {
            const int x = tx0 - 1 + sx, y = ty0 - 1 + sy;
            uint32_t tt = (in & (x >= 3) & (x < w - 3) & (y >= 3) & (y < h - 3) & (x >= L.rx0) & (x < rx1) &
                           (y >= L.ry0) & (y < ry1))
                              ? (uint32_t)threshold * 65537u
                              : 16000u * 65537u;
            asm volatile("" : "+v"(tt));  // opaque: the compiler would turn it back into compare-and-mask
            return tt;
        };
        {
            // rows outside the image or the region (wave-uniform) are skipped, the column condition is the
            // lane's threshold
            const int y_lo = max(3, L.ry0), y_hi = min(h - 3, ry1);  // y_lo <= y < y_hi
            const uint32_t tt = border_tt(y_lo - (ty0 - 1), lane, true);
            const int sy_lo = y_lo - (ty0 - 1), sy_hi = min(y_hi - (ty0 - 1), SM_H);
            const int sy0 = wv + 4 * (max(sy_lo - wv + 3, 0) >> 2);
            int idx = sy0 * PX_W + lane;
            for (int sy = sy0; sy < sy_hi; sy += 4, idx += 4 * PX_W) compass(idx, tt);
        }
        static_assert(2 * SM_H <= 3 * 64, "columns 64 and 65 fit one step of waves 3, 2, 1");
        if ((3 - wv) * 64 < 2 * SM_H) {
            const int q = (3 - wv) * 64 + lane;  // 0 .. 2 * SM_H - 1
            const int sy = min(q >> 1, SM_H - 1), sx = FT_W + (q & 1);
            compass(sy * PX_W + sx, border_tt(sy, sx, q < 2 * SM_H));
        }
    }
    wave_lds_sync();
#ifdef VO_FAST_STOP
    if (lane == 0) cntA[wv] = na;
    __syncthreads();
#endif
    VO_FAST_PHASE_END(1);

    // phase B: full arc test on the wave's own survivors, dense lanes; corners go to the wave's second list
    int nb_w = 0;  // corners of this wave (wave-uniform)
    uint16_t* mineB = listB + wv * FT_LCAP;
    {
        for (int i0 = 0; i0 < na; i0 += 64) {
            const int i = i0 + lane;
            bool corner = false;
            int p = 0;
            if (i < na) {
                p = mineA[i];
                const uint8_t* c = px + p + RING0;
                const uint32_t tt = (uint32_t)threshold * 65537u;
                // One multiply-add per ring point gives both tests: P = (r - v + t) + ((v + t - r) << 16) has bit 15
                // set iff r < v - t (dark) and bit 31 set iff r > v + t (bright) -- the fields stay within 16 bits
                // and the borrow of a negative low field never flips the high one, which is > 2t then.  A 9-arc
                // of one polarity is a non-zero bit 15 / 31 in the AND of nine consecutive words: two levels of
                // three-input ANDs for the 16 arcs, three-input ORs over them.
                const uint32_t C = (uint32_t)c[0] * 65535u + tt;
                uint32_t P[16];
#define VO_RING(k, dx, dy) P[k] = C - 65535u * (uint32_t)c[(dy) * PX_W + (dx)];
                VO_RING(0, -1, 3) VO_RING(1, -2, 2) VO_RING(2, -3, 1) VO_RING(3, -3, 0)
                VO_RING(4, -3, -1) VO_RING(5, -2, -2) VO_RING(6, -1, -3) VO_RING(7, 0, -3)
                VO_RING(8, 1, -3) VO_RING(9, 2, -2) VO_RING(10, 3, -1) VO_RING(11, 3, 0)
                VO_RING(12, 3, 1) VO_RING(13, 2, 2) VO_RING(14, 1, 3) VO_RING(15, 0, 3)
#undef VO_RING
                uint32_t t3[16], a9[16], any = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) t3[k] = and3(P[k], P[(k + 1) & 15], P[(k + 2) & 15]);
#pragma unroll
                for (int k = 0; k < 16; k++) a9[k] = and3(t3[k], t3[(k + 3) & 15], t3[(k + 6) & 15]);
#pragma unroll
                for (int k = 0; k < 16; k += 2) any = or3(any, a9[k], a9[k + 1]);
                corner = (any & 0x80008000u) != 0;
                // a 9-arc of one polarity excludes one of the other (16 ring points): bit 14 = dark arc
                p |= (any >> 1) & 0x4000;
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(corner);
            if (corner) mineB[nb_w + mbcnt(m)] = (uint16_t)p;
            nb_w += __popcll(m);
        }
    }
    wave_lds_sync();
#ifdef VO_FAST_STOP
    if (lane == 0) cntB[wv] = nb_w;
    __syncthreads();
#endif
    VO_FAST_PHASE_END(2);

    // phase C: cornerScore<16> = max(threshold, max over the 16 arcs of min(d), of min(-d)) - 1 with
    // d = v - ring.  Only the polarity that has the arc can exceed the threshold (the other one would
    // need a second 9-arc), so one min network on e = +-d is enough: the same score.
    // Each wave scores its own corners; the block meets only afterwards, when the NMS needs the whole map.
    for (int i = lane; i < nb_w; i += 64) {
        const int p = mineB[i], idx = p & 0x1fff;
        const uint8_t* c = px + idx + RING0;
        // e = v - ring for a dark arc, ring - v for a bright one: one multiply-add per point
        const int nsgn = (p & 0x4000) ? -1 : 1;
        const int sv = -nsgn * (int)c[0];
        int e[16];
#define VO_E(k, dx, dy) e[k] = nsgn * (int)c[(dy) * PX_W + (dx)] + sv;
        VO_E(0, 0, 3) VO_E(1, 1, 3) VO_E(2, 2, 2) VO_E(3, 3, 1) VO_E(4, 3, 0) VO_E(5, 3, -1) VO_E(6, 2, -2)
        VO_E(7, 1, -3) VO_E(8, 0, -3) VO_E(9, -1, -3) VO_E(10, -2, -2) VO_E(11, -3, -1) VO_E(12, -3, 0)
        VO_E(13, -3, 1) VO_E(14, -2, 2) VO_E(15, -1, 3)
#undef VO_E
        // minimum over each of the 16 arcs of nine points: two levels of three-input minima
        int best = threshold;
        int m3[16];
#pragma unroll
        for (int k = 0; k < 16; k++) m3[k] = min3i(e[k], e[(k + 1) & 15], e[(k + 2) & 15]);
        int m9[16];
#pragma unroll
        for (int k = 0; k < 16; k++) m9[k] = min3i(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k += 2) best = max3i(best, m9[k], m9[k + 1]);
        sm[idx] = (uint8_t)best;  // (best - 1) + 1
    }
    __syncthreads();
    VO_FAST_PHASE_END(3);

    // phase D: NMS + border filter on the corners inside the tile, every wave on its own corners
    const int b = max(border, 3);
    const bool level_ok = border == 0 || (w > 2 * border && h > 2 * border);
    uint8_t* sc = level_px(score, g, f, l);
    for (int i = lane; i < nb_w; i += 64) {
        const int idx = mineB[i] & 0x1fff;
        static_assert(SM_H * PX_W * 3641ull < (1ull << 32) && PX_W == 72, "idx / 72 by multiplication");
        const int sy = (int)(__umul24(idx, 3641) >> 18), sx = idx - sy * PX_W;
        const int x = tx0 - 1 + sx, y = ty0 - 1 + sy;
        const bool in = (sx >= 1) & (sx <= FT_W) & (sy >= 1) & (sy <= FT_H) & level_ok & (x >= b) & (x < w - b) &
                        (y >= b) & (y < h - b);
        if (!in) continue;
        const uint8_t* q = &sm[idx];
        const int s1 = q[0];
        bool kp = true;
        if (nms) {
            // The map holds score + 1 (0 = not a corner); cv keeps a corner whose score exceeds the scores of
            // its eight neighbours, a neighbour that is not a corner counting as 0: s1 - 1 > (q ? q - 1 : 0)
            // for all eight, i.e. s1 > max(1, the eight map values).
            constexpr int SP = PX_W;
            const int a0 = q[-SP - 1], a1 = q[-SP], a2 = q[-SP + 1], a3 = q[-1], a4 = q[1], a5 = q[SP - 1], a6 = q[SP],
                      a7 = q[SP + 1];
            kp = s1 > max(max(max(max(a0, a1), a2), max(max(a3, a4), a5)), max(max(a6, a7), 1));
        }
        if (kp) {
            atomicOr(&rowmask[sy - 1][(sx - 1) >> 5], 1u << ((sx - 1) & 31));
            sc[(size_t)y * pitch + x] = (uint8_t)s1;
        }
    }
    __syncthreads();

    phase_e();
}

// ---------------------------------------------------------------------------------------------
// Per frame: segbase[seg] = position of the first keypoint of that 64-pixel segment in the
// frame's row-major candidate list; lvlofs[l] = first candidate of level l.  The mask table is
// laid out level after level, row after row, segment after segment, i.e. in candidate order, so
// this is one exclusive prefix sum of popcounts over a flat array.  One block of 16 waves per
// frame; a wave owns a contiguous chunk, reads it with consecutive lanes on consecutive words
// (pass 1: chunk total, pass 2: in-wave scans on top of the chunk's base).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_row_scan(const PyrGeom* __restrict__ g,
                                                   const unsigned long long* __restrict__ masks,
                                                   int* __restrict__ segbase, int* __restrict__ lvlofs,
                                                   int* __restrict__ flags) {
    __shared__ int wtot[16];
    // the per-frame selection flags start at zero (this kernel runs once per batch before the selection)
    // (flags == NULL: the lists are made for the host redo of a batch whose flags must survive)
    if (flags && threadIdx.x == 0) flags[blockIdx.x] = 0;
    if (flags && threadIdx.x == 1 && blockIdx.x == 0) flags[gridDim.x] = 0;
    __shared__ int s_total;
    const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int total = g->segs_total, nlevels = g->nlevels;
    const int per = ((total + 15) / 16 + 511) & ~511;  // words per wave, whole steps of 512
    const int beg = wv * per, end = min(total, beg + per);
    const unsigned long long* M = masks + (size_t)f * total;
    int* SB = segbase + (size_t)f * total;
    // a lane owns eight consecutive words (64 bytes) per step of 512: serial prefix inside the lane,
    // one wave scan per step.  Up to RS_STEPS steps per wave (the usual geometry needs three) are read
    // once, every load in flight together, and their popcounts stay in registers for the second pass;
    // the table is padded (VO_MASK_PAD words) so these loads need no bounds.
    constexpr int RS_STEPS = 4;
    static_assert((RS_STEPS + 15) * 512 <= VO_MASK_PAD, "the last wave starts at most 15 * 512 words past the table end");
    if (per <= RS_STEPS * 512) {
        int c[RS_STEPS][8], t[RS_STEPS];
        int sum = 0;
        {
            unsigned long long m[RS_STEPS][8];
#pragma unroll
            for (int s = 0; s < RS_STEPS; s++)
#pragma unroll
                for (int u = 0; u < 8; u++) m[s][u] = M[beg + 512 * s + 8 * lane + u];
#pragma unroll
            for (int s = 0; s < RS_STEPS; s++) {
                const int ib = beg + 512 * s + 8 * lane;
                t[s] = 0;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    pin_loaded(m[s][u]);
                    c[s][u] = ib + u < end ? __popcll(m[s][u]) : 0;
                    t[s] += c[s][u];
                }
                sum += t[s];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) wtot[wv] = sum;
        __syncthreads();
        int run = 0;
        for (int k = 0; k < wv; k++) run += wtot[k];
#pragma unroll
        for (int s = 0; s < RS_STEPS; s++) {
            const int ib = beg + 512 * s + 8 * lane;
            int inc = t[s];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o);
                if (lane >= o) inc += v;
            }
            int base = run + inc - t[s];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = ib + u;
                if (i < end) SB[i] = base;
                base += c[s][u];
            }
            run += __shfl(inc, 63);
        }
        if (wv == 15 && lane == 0) s_total = run;
    } else {
        int sum = 0;
        for (int i0 = beg; i0 < end; i0 += 512) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + 8 * lane + u;
                sum += i < end ? __popcll(M[i]) : 0;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) wtot[wv] = sum;
        __syncthreads();
        int run = 0;
        for (int k = 0; k < wv; k++) run += wtot[k];
        for (int i0 = beg; i0 < end; i0 += 512) {
            const int ib = i0 + 8 * lane;
            int c[8], t = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                c[u] = ib + u < end ? __popcll(M[ib + u]) : 0;
                t += c[u];
            }
            int inc = t;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o);
                if (lane >= o) inc += v;
            }
            int base = run + inc - t;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = ib + u;
                if (i < end) SB[i] = base;
                base += c[u];
            }
            run += __shfl(inc, 63);
        }
        if (wv == 15 && lane == 0) s_total = run;
    }
    // the level starts are entries of the table just written (made visible to the block by the barrier)
    __syncthreads();
    if ((int)threadIdx.x < nlevels)
        lvlofs[(size_t)f * (VO_ORB_MAX_LEVELS + 1) + threadIdx.x] =
            g->lv[threadIdx.x].seg0 < total ? SB[g->lv[threadIdx.x].seg0] : s_total;  // trailing empty levels
    else if ((int)threadIdx.x == nlevels)
        lvlofs[(size_t)f * (VO_ORB_MAX_LEVELS + 1) + nlevels] = s_total;
}

// One thread per 64-pixel segment: its keypoints go to the candidate list in x order.
__global__ __launch_bounds__(256) void k_cand_write(const PyrGeom* __restrict__ g,
                                                    const unsigned long long* __restrict__ masks,
                                                    const int* __restrict__ segbase,
                                                    const uint8_t* __restrict__ score, int nms,
                                                    Cand* __restrict__ cand, int max_cand) {
    const int f = blockIdx.y;
    const int sg = blockIdx.x * 256 + threadIdx.x;
    if (sg >= g->segs_total) return;
    unsigned long long m = masks[(size_t)f * g->segs_total + sg];
    int pos = segbase[(size_t)f * g->segs_total + sg];  // with the mask load, not after it
    if (!m) return;
    int l = 0;
    for (int k = 1; k < g->nlevels; k++)
        if (sg >= g->lv[k].seg0) l = k;
    const int local = sg - g->lv[l].seg0;
    const int yr = local / g->lv[l].segs, sx = local - yr * g->lv[l].segs;
    const int y = g->lv[l].ry0 + yr, xs = g->lv[l].rx0 + sx * 64;
    const uint8_t* sc = level_px(score, g, f, l) + (size_t)y * g->lv[l].pitch;
    while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int x = xs + bit;
        if (pos < max_cand) {
            Cand cd;
            cd.x = (int16_t)x;
            cd.y = (int16_t)y;
            cd.level = (int16_t)l;
            cd.score = (int16_t)(nms ? (int)sc[x] - 1 : 0);
            cd.harris = 0.f;
            cd.pad = 0;
            cand[(size_t)f * max_cand + pos] = cd;
        }
        pos++;
    }
}

// ---------------------------------------------------------------------------------------------
// Harris response (7x7 block, k = 0.04f) in cv's float evaluation order.
// ---------------------------------------------------------------------------------------------
// The 9 x 9 patch is read as nine rows of three aligned dwords (27 loads instead of 392 byte
// loads) and walked with three rows in registers: S = up + 2 mid + down and D = down - up per column,
// Ix = S[j+1] - S[j-1], Iy = D[j-1] + 2 D[j] + D[j+1]: the same integers as cv's per-pixel Sobel sums.
__device__ __forceinline__ float harris_response(const uint8_t* img, int pitch, int x0, int y0) {
    const float harris_k = 0.04f;
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const int xs = x0 - 4, o = xs & 3;
    const uint8_t* base = img + (ptrdiff_t)(y0 - 4) * pitch + (xs - o);
    int R[3][9];
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(base + (ptrdiff_t)r * pitch);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, o), w1 = __builtin_amdgcn_alignbyte(d2, d1, o),
                       w2 = __builtin_amdgcn_alignbyte(0u, d2, o);
        int* row = R[r % 3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            row[k] = (w0 >> (8 * k)) & 0xff;
            row[4 + k] = (w1 >> (8 * k)) & 0xff;
        }
        row[8] = w2 & 0xff;
        if (r >= 2) {
            const int* up = R[(r - 2) % 3];
            const int* mid = R[(r - 1) % 3];
            int S[9], D[9];
#pragma unroll
            for (int k = 0; k < 9; k++) {
                S[k] = up[k] + 2 * mid[k] + row[k];
                D[k] = row[k] - up[k];
            }
#pragma unroll
            for (int j = 1; j <= 7; j++) {
                const int Ix = S[j + 1] - S[j - 1];
                const int Iy = D[j - 1] + 2 * D[j] + D[j + 1];
                a += __mul24(Ix, Ix);  // |Ix|, |Iy| <= 1020: 24-bit multiplies are exact and full rate
                b += __mul24(Iy, Iy);
                c += __mul24(Ix, Iy);
            }
        }
    }
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    return (fa * fb - fc * fc - harris_k * (fa + fb) * (fa + fb)) * scale_sq_sq;
}

// Harris response of every candidate: only the host selection path needs all of them.
__global__ __launch_bounds__(256) void k_harris(const uint8_t* __restrict__ pyr, const PyrGeom* __restrict__ g,
                                                const int* __restrict__ lvlofs, Cand* __restrict__ cand,
                                                int max_cand, const int* __restrict__ only_flagged) {
    const int f = blockIdx.y;
    if (only_flagged && only_flagged[f] == 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = min(lvlofs[(size_t)f * (VO_ORB_MAX_LEVELS + 1) + g->nlevels], max_cand);
    if (i >= total) return;
    Cand* c = cand + (size_t)f * max_cand + i;
    c->harris = harris_response(level_px(pyr, g, f, c->level), g->lv[c->level].pitch, c->x, c->y);
}

// ---------------------------------------------------------------------------------------------
// Device-side keypoint selection: per (frame, level)
//     retainBest(2 * quota) on the FAST score, Harris response, retainBest(quota) on it,
// where retainBest = std::nth_element + std::partition of libstdc++ (cv's KeyPointsFilter), whose
// output PERMUTATION defines the keypoint order and with it every match index downstream.
// The permutation is reproduced exactly: introselect's median-of-three + unguarded Hoare
// partition rounds, the final insertion sort, and the bidirectional std::partition.  One Hoare
// round is data-parallel: the j-th stop of the left pointer is the j-th element that is not
// "before" the pivot, the j-th stop of the right pointer the j-th from the right that is not
// "after" it, and the pointers swap exactly while left stop < right stop — so one wavefront finds
// both stop lists with ballots, counts the swaps and performs them at once.
// One block per (frame, level): its four waves share every partition round of the selection on the
// list held in LDS and compute the Harris responses in between.  The depth-limit (heap-select) branch of introselect and
// lists larger than the LDS buffer are flagged and redone by the host path.
// ---------------------------------------------------------------------------------------------
namespace {

struct RI {
    float r;
    int idx;
};

// MODE 0: std::__unguarded_partition(first, last, pivot) with comp(a, b) = a.r > b.r; returns cut.
// MODE 1: std::__partition(first, last, pred) with pred(a) = a.r >= key; returns the new end.
// All threads of the block call it with the same arguments.  Each wave lists the stops of its quarter of the
// range in its own segment of Lp / Ra; the j-th stop overall is found through the four segment counts
// (sh[0..7]), so the lists are never compacted.
// MODE 0 also carries nth_element's pivot move: the caller has chosen v[piv] as the pivot and wants it
// swapped with v[first - 1]; instead of two more barriers around that swap, pass 1 reads the value that
// position piv will hold (sub_r, the old v[first - 1].r) from a register, and thread 0 performs the swap
// in LDS after the first barrier, when nobody reads the old values any more.
template <int MODE>
__device__ int block_partition(RI* v, int first, int last, float key, uint16_t* Lp, uint16_t* Ra, int* sh, int tid,
                               int piv = -1, float sub_r = 0.f) {
    // the block has SEL_THREADS threads; the first 256 (four waves) share a round, the others keep step at the
    // barriers: a round is a chain of LDS round trips and barriers whose length does not shrink with more waves
    const bool worker = tid < 256;
    const int wv = tid >> 6, lane = tid & 63;
    const int span = ((last - first + 255) >> 8) << 6;
    const int wb = first + wv * span, we = min(last, wb + span);
    int nL = 0, nR = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (worker)
        for (int i0 = wb; i0 < we; i0 += 64) {
            const int i = i0 + lane;
            const bool in = i < we;
            float r = in ? v[i].r : 0.f;
            if (MODE == 0 && i == piv) r = sub_r;
            bool isL, isR;
            if (MODE == 0) {
                isL = in && !(r > key);   // the left pointer stops: !comp(*first, pivot)
                isR = in && !(key > r);   // the right pointer stops: !comp(pivot, *last)
            } else {
                isR = in && (r >= key);   // the right pointer stops on pred
                isL = in && !(r >= key);  // the left pointer stops on !pred
            }
            const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
            if (isL) Lp[wb + nL + __popcll(bL & lt)] = (uint16_t)i;
            if (isR) Ra[wb + nR + __popcll(bR & lt)] = (uint16_t)i;
            nL += __popcll(bL);
            nR += __popcll(bR);
        }
    if (lane == 0 && worker) {
        sh[wv] = nL;
        sh[4 + wv] = nR;
    }
    __syncthreads();
    if (MODE == 0 && piv >= 0 && tid == 0) {
        const RI t = v[first - 1];
        v[first - 1] = v[piv];
        v[piv] = t;
    }
    const int l0 = sh[0], l1 = l0 + sh[1], l2 = l1 + sh[2], nLt = l2 + sh[3];
    const int r0 = sh[4], r1 = r0 + sh[5], r2 = r1 + sh[6], nRt = r2 + sh[7];
    auto leftStop = [&](int k) -> int {  // k-th stop of the left pointer
        const int w = (k >= l0) + (k >= l1) + (k >= l2);
        const int pre = w == 0 ? 0 : w == 1 ? l0 : w == 2 ? l1 : l2;
        return Lp[first + w * span + (k - pre)];
    };
    auto rightStop = [&](int k) -> int {  // k-th stop of the right pointer (counted from the right)
        const int q = nRt - 1 - k;
        const int w = (q >= r0) + (q >= r1) + (q >= r2);
        const int pre = w == 0 ? 0 : w == 1 ? r0 : w == 2 ? r1 : r2;
        return Ra[first + w * span + (q - pre)];
    };
    const int m = min(nLt, nRt);
    int Kw = 0;  // number of swaps: stops cross monotonically
    int a0 = 0, b0 = 0;  // the thread's first pair of stops, kept for the swap below
    if (worker)
        for (int k0 = 0; k0 < m; k0 += 256) {
            const int k = k0 + tid;
            const int a = k < m ? leftStop(k) : 0, b = k < m ? rightStop(k) : 0;
            if (k0 == 0) a0 = a, b0 = b;
            const bool ok = k < m && a < b;
            Kw += __popcll(__ballot(ok));
        }
    if (lane == 0 && worker) sh[8 + wv] = Kw;
    __syncthreads();
    const int K = sh[8] + sh[9] + sh[10] + sh[11];
    int ret;
    if (MODE == 0) {
        const int a = K < nLt ? leftStop(K) : last;
        const int b = K > 0 ? rightStop(K - 1) : last;
        ret = min(a, b);
    } else {
        ret = first + nRt;
    }
    if (worker)
        for (int k = tid; k < K; k += 256) {
            const int a = k == tid ? a0 : leftStop(k), b = k == tid ? b0 : rightStop(k);
            const RI t = v[a];
            v[a] = v[b];
            v[b] = t;
        }
    __syncthreads();
    return ret;
}

// The rounds of nth_element on a range of at most 64 elements, by ONE wave with the elements in registers (lane p
// holds position base + p): no block barrier and no LDS round trip per step.  A round of the block version costs
// ~1 us whatever the range (three barriers, seven dependent LDS reads), and about half of a selection's rounds
// run on ranges this small.  The same unguarded Hoare partition: with bL / bR the ballots of the stops of the
// left / right pointer, the left stop at position p with k stops before it is swapped iff more than k right
// stops lie behind it (the k-th swap exists iff a_k < b_k), partners are found through their common rank, and
// the cut is min(a_K, b_{K-1}) as in block_partition<0>.  first, last, nth are relative to base; returns the
// remaining depth budget, or -1 when libstdc++ would enter its heap-select fallback.
__device__ __forceinline__ float wave_get(float v, int l) {  // v of lane l (wave-uniform l): v_readlane
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ int wave_nth_rounds(float& r, int& idx, int& first, int& last, int nth, int depth, int lane) {
    const unsigned long long lt = (1ull << lane) - 1ull, gt = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    // wave-uniform by construction; said so, for the scalar unit
    first = __builtin_amdgcn_readfirstlane(first);
    last = __builtin_amdgcn_readfirstlane(last);
    nth = __builtin_amdgcn_readfirstlane(nth);
    depth = __builtin_amdgcn_readfirstlane(depth);
    while (last - first > 3) {
        if (depth == 0) return -1;
        depth--;
        const int mid = first + (last - first) / 2;
        const float ra = wave_get(r, first + 1), rb = wave_get(r, mid), rc = wave_get(r, last - 1);
        int sel;
        float pivot;
        if (ra > rb) {
            if (rb > rc) sel = mid, pivot = rb;
            else if (ra > rc) sel = last - 1, pivot = rc;
            else sel = first + 1, pivot = ra;
        } else if (ra > rc) sel = first + 1, pivot = ra;
        else if (rb > rc) sel = last - 1, pivot = rc;
        else sel = mid, pivot = rb;
        {  // std::iter_swap(first, sel)
            const float rf = wave_get(r, first), rs = wave_get(r, sel);
            const int xf = __builtin_amdgcn_readlane(idx, first), xs = __builtin_amdgcn_readlane(idx, sel);
            if (lane == first) r = rs, idx = xs;
            else if (lane == sel) r = rf, idx = xf;
        }
        // __unguarded_partition(first + 1, last, pivot)
        const bool in = lane > first && lane < last;
        const bool isL = in && !(r > pivot);  // the left pointer stops: !comp(*first, pivot)
        const bool isR = in && !(pivot > r);  // the right pointer stops: !comp(pivot, *last)
        const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
        const int kL = __popcll(bL & lt), kR = __popcll(bR & gt);  // rank among the left stops / from the right
        const int nLt = __popcll(bL);
        const int K = __popcll(__ballot(isL && __popcll(bR & gt) > kL));
        const bool swL = isL && kL < K, swR = isR && kR < K;
        // partners meet in slot k (a lane index below 32: a swap takes two positions); the others aim at slot 63
        const int to = swR ? kR : (swL ? kL : 63);
        const float fromR_r = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(4 * (swR ? kR : 63), __builtin_bit_cast(int, r)));
        const int fromR_x = __builtin_amdgcn_ds_permute(4 * (swR ? kR : 63), idx);
        const float fromL_r = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(4 * (swL ? kL : 63), __builtin_bit_cast(int, r)));
        const int fromL_x = __builtin_amdgcn_ds_permute(4 * (swL ? kL : 63), idx);
        // slot k now holds the k-th right stop (fromR) and the k-th left stop (fromL); every swapping lane pulls its partner
        const float getR_r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * to, __builtin_bit_cast(int, fromR_r)));
        const int getR_x = __builtin_amdgcn_ds_bpermute(4 * to, fromR_x);
        const float getL_r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * to, __builtin_bit_cast(int, fromL_r)));
        const int getL_x = __builtin_amdgcn_ds_bpermute(4 * to, fromL_x);
        if (swL) r = getR_r, idx = getR_x;
        else if (swR) r = getL_r, idx = getL_x;
        // cut = min(a_K, b_{K-1}) with `last` where the stop does not exist
        const unsigned long long mA = __ballot(isL && kL == K), mB = __ballot(isR && kR == K - 1);
        const int a = (K < nLt && mA) ? __ffsll((long long)mA) - 1 : last;
        const int b = (K > 0 && mB) ? __ffsll((long long)mB) - 1 : last;
        const int cut = min(a, b);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    return depth;
}

// std::nth_element(v + first, v + nth, v + last, greater-by-response); false when libstdc++ would
// enter its heap-select fallback (depth limit reached).  Called by the whole block.
__device__ bool block_nth_element(RI* v, int first, int nth, int last, uint16_t* Lp, uint16_t* Ra, int* sh, int tid) {
    if (first == last || nth == last) return true;
    int depth = 2 * (31 - __clz(last - first));
    while (last - first > 64) {
        if (depth == 0) return false;
        depth--;
        const int mid = first + (last - first) / 2;
        const float ra = v[first + 1].r, rb = v[mid].r, rc = v[last - 1].r, rf = v[first].r;
        int sel;
        float pivot;
        if (ra > rb) {
            if (rb > rc) sel = mid, pivot = rb;
            else if (ra > rc) sel = last - 1, pivot = rc;
            else sel = first + 1, pivot = ra;
        } else if (ra > rc) sel = first + 1, pivot = ra;
        else if (rb > rc) sel = last - 1, pivot = rc;
        else sel = mid, pivot = rb;
        // std::iter_swap(first, sel) happens inside the partition (see there)
        const int cut = block_partition<0>(v, first + 1, last, pivot, Lp, Ra, sh, tid, sel, rf);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    // the remaining rounds (ranges of at most 64) by the first wave alone, elements in registers
    if (last - first > 3) {
        if (tid < 64) {
            const int base = first, cnt = last - first;
            RI e = v[base + min(tid, cnt - 1)];
            int f = 0, l = cnt;
            const int left = wave_nth_rounds(e.r, e.idx, f, l, nth - base, depth, tid);
            // std::__insertion_sort on the remaining <= 3 elements [f, l), still in registers: a stable sort by
            // descending response (an element moves in front of those it is strictly greater than)
            if (left >= 0 && l - f >= 2) {
                float r0 = wave_get(e.r, f), r1 = wave_get(e.r, f + 1), r2 = l - f == 3 ? wave_get(e.r, f + 2) : 0.f;
                int x0 = __builtin_amdgcn_readlane(e.idx, f), x1 = __builtin_amdgcn_readlane(e.idx, f + 1),
                    x2 = l - f == 3 ? __builtin_amdgcn_readlane(e.idx, f + 2) : 0;
                if (r1 > r0) {
                    const float tr = r0;
                    const int tx = x0;
                    r0 = r1, x0 = x1, r1 = tr, x1 = tx;
                }
                if (l - f == 3) {
                    if (r2 > r0) {
                        const float tr = r2;
                        const int tx = x2;
                        r2 = r1, x2 = x1, r1 = r0, x1 = x0, r0 = tr, x0 = tx;
                    } else if (r2 > r1) {
                        const float tr = r2;
                        const int tx = x2;
                        r2 = r1, x2 = x1, r1 = tr, x1 = tx;
                    }
                }
                if (tid == f) e.r = r0, e.idx = x0;
                if (tid == f + 1) e.r = r1, e.idx = x1;
                if (l - f == 3 && tid == f + 2) e.r = r2, e.idx = x2;
            }
            if (tid < cnt) v[base + tid] = e;
            if (tid == 0) sh[0] = left;
        }
        __syncthreads();
        const int left = sh[0];
        __syncthreads();  // (sh is reused by the next partition)
        return left >= 0;
    }
    // std::__insertion_sort on the remaining <= 3 elements (a block round went from more than 64 straight to here)
    if (tid == 0) {
        for (int i = first + 1; i < last; i++) {
            const RI val = v[i];
            if (val.r > v[first].r) {
                for (int j = i; j > first; j--) v[j] = v[j - 1];
                v[first] = val;
            } else {
                int j = i;
                while (val.r > v[j - 1].r) {
                    v[j] = v[j - 1];
                    j--;
                }
                v[j] = val;
            }
        }
    }
    __syncthreads();
    return true;
}

// KeyPointsFilter::retainBest; returns the new size, -1 on the depth-limit fallback
__device__ int block_retain_best(RI* v, int n, int n_points, uint16_t* Lp, uint16_t* Ra, int* sh, int tid) {
    if (n_points < 0 || n <= n_points) return n;
    if (n_points == 0) return 0;
    if (!block_nth_element(v, 0, n_points, n, Lp, Ra, sh, tid)) return -1;
    const float ambiguous = v[n_points - 1].r;
    return block_partition<1>(v, n_points, n, ambiguous, Lp, Ra, sh, tid);
}

}  // namespace

// The list of a (frame, level) is built by its block straight from the detector's mask table — the words of a
// level are its candidates in cv's order (row-major) —, so the device path needs neither the frame-wide prefix
// scan nor the candidate lists in memory (k_row_scan + k_cand_write: two launches, 31 us of a 49-frame batch, and
// an index -> candidate hop in front of every Harris patch); those two serve the plain FAST entry point and the host
// redo of flagged frames.  A wave takes an eighth of the level's words in steps of 512 (a lane owns eight
// consecutive words), counts, and after the block knows the eight totals writes position x | y << 16 of every set
// bit at its rank; the FAST scores are then fetched by rank, eight loads in flight per thread.
// SEL_THREADS threads: the list, the scores, the Harris responses and the output are work for every wave the CU
// can hold (two blocks of 72 KB LDS and eight waves each); the partition rounds use the first four (see there).
#define SEL_THREADS 512
__global__ __launch_bounds__(SEL_THREADS) void k_orb_select(const uint8_t* __restrict__ pyr, const PyrGeom* __restrict__ g,
                                                    const unsigned long long* __restrict__ masks,
                                                    const uint8_t* __restrict__ score, int nms, Quotas quota,
                                                    int sel_cap, Cand* __restrict__ selcand, int sel_fs,
                                                    int* __restrict__ selcnt, int* __restrict__ flags) {
    extern __shared__ unsigned char sel_smem[];
    __shared__ int sh[12];
    static_assert(SEL_THREADS / 64 <= 12, "sh holds one count per wave while the list is built");
    RI* v = reinterpret_cast<RI*>(sel_smem);
    uint16_t* Lp = reinterpret_cast<uint16_t*>(v + sel_cap);
    uint16_t* Ra = Lp + sel_cap;
    // grid (frames, levels): dispatch order is x fastest and round-robin over the 8 XCDs, so the long level-0
    // lists go out first and are spread over all XCDs (with the level in x every level-0 list sat on XCD 0)
    const int l = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
    const int wv = tid >> 6, lane = tid & 63;
    const int seg0 = g->lv[l].seg0, segs = g->lv[l].segs;
    const int W = (l + 1 < g->nlevels ? g->lv[l + 1].seg0 : g->segs_total) - seg0;
    const int rx0 = g->lv[l].rx0, ry0 = g->lv[l].ry0, pitch = g->lv[l].pitch;
    int* cnt = selcnt + (size_t)f * VO_ORB_MAX_LEVELS + l;
    const unsigned long long* M = masks + (size_t)f * g->segs_total + seg0;
    constexpr int NW = SEL_THREADS / 64;
    const int per = ((W + NW - 1) / NW + 511) & ~511;  // words per wave, whole steps of 512 (the table is padded)
    const int beg = wv * per, end = min(W, beg + per);
    int n;
    // every load of the thread goes out before the first popcount: up to SB_STEPS steps of 512 words per wave stay
    // in registers (a level of up to 8 192 words, 524 288 pixels of detector region); larger levels read twice
    constexpr int SB_STEPS = 2;
    const bool held = per <= SB_STEPS * 512;
    unsigned long long mh[SB_STEPS][8];
    {
        int sum = 0;
        if (held) {
#pragma unroll
            for (int s = 0; s < SB_STEPS; s++)
#pragma unroll
                for (int u = 0; u < 8; u++) mh[s][u] = s * 512 < per ? M[beg + 512 * s + 8 * lane + u] : 0ull;
#pragma unroll
            for (int s = 0; s < SB_STEPS; s++)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (beg + 512 * s + 8 * lane + u >= end) mh[s][u] = 0ull;
                    sum += __popcll(mh[s][u]);
                }
        } else {
            for (int i0 = beg; i0 < end; i0 += 512) {
                unsigned long long m[8];
#pragma unroll
                for (int u = 0; u < 8; u++) m[u] = M[i0 + 8 * lane + u];
#pragma unroll
                for (int u = 0; u < 8; u++) sum += i0 + 8 * lane + u < end ? __popcll(m[u]) : 0;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) sh[wv] = sum;
        __syncthreads();
        int run = 0;
        for (int k = 0; k < wv; k++) run += sh[k];
        n = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) n += sh[k];
        if (n > sel_cap) {
            if (tid == 0) {
                atomicOr(&flags[f], VO_SEL_OVERSIZE);
                *cnt = 0;
            }
            return;
        }
        // one step: the lane's eight words -> ranks by a wave scan -> x | y << 16 of every set bit at its rank
        auto scatter = [&](const unsigned long long (&m)[8], int ib) {
            int t = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) t += __popcll(m[u]);
            int inc = t;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int q = __shfl_up(inc, o);
                if (lane >= o) inc += q;
            }
            int pos = run + inc - t;
            run += __shfl(inc, 63);
            if (t == 0) return;
            int yr = ib / segs, sx = ib - yr * segs;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                unsigned long long mm = m[u];
                const uint32_t yx = (uint32_t)(ry0 + yr) << 16 | (uint32_t)(rx0 + sx * 64);
                while (mm) {
                    const int bit = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    v[pos].idx = (int)(yx + bit);
                    pos++;
                }
                if (++sx == segs) sx = 0, yr++;
            }
        };
        if (held) {
#pragma unroll
            for (int s = 0; s < SB_STEPS; s++)
                if (beg + 512 * s < end) scatter(mh[s], beg + 512 * s + 8 * lane);
        } else {
            for (int i0 = beg; i0 < end; i0 += 512) {
                unsigned long long m[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    m[u] = M[i0 + 8 * lane + u];  // second read: L2
                    if (i0 + 8 * lane + u >= end) m[u] = 0;
                }
                scatter(m, i0 + 8 * lane);
            }
        }
    }
    __syncthreads();
    {
        const uint8_t* S = level_px(score, g, f, l);
        for (int i0 = 0; i0 < n; i0 += 8 * SEL_THREADS) {  // eight loads in flight per thread
            int sc8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t yx = (uint32_t)v[min(i0 + SEL_THREADS * u + tid, n - 1)].idx;
                sc8[u] = S[(size_t)(yx >> 16) * pitch + (yx & 0xffff)];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) pin_loaded(sc8[u]);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + SEL_THREADS * u + tid;
                if (i < n) v[i].r = nms ? (float)(sc8[u] - 1) : 0.f;
            }
        }
    }
    __syncthreads();
    n = block_retain_best(v, n, 2 * quota.q[l], Lp, Ra, sh, tid);
    if (n < 0) {
        if (tid == 0) {
            atomicOr(&flags[f], VO_SEL_DEPTH);
            *cnt = 0;
        }
        return;
    }
    const uint8_t* img = level_px(pyr, g, f, l);
    for (int i = tid; i < n; i += SEL_THREADS) {
        const uint32_t yx = (uint32_t)v[i].idx;
        v[i].r = harris_response(img, pitch, (int)(yx & 0xffff), (int)(yx >> 16));
    }
    __syncthreads();
    n = block_retain_best(v, n, quota.q[l], Lp, Ra, sh, tid);
    if (n < 0) {
        if (tid == 0) {
            atomicOr(&flags[f], VO_SEL_DEPTH);
            *cnt = 0;
        }
        return;
    }
    // the picks of level l go to selcand[f][l * sel_cap ...] as whole records, so that the next kernel reads a
    // keypoint with one load
    for (int i = tid; i < n; i += SEL_THREADS) {
        const uint32_t yx = (uint32_t)v[i].idx;
        Cand cd;
        cd.x = (int16_t)(yx & 0xffff);
        cd.y = (int16_t)(yx >> 16);
        cd.level = (int16_t)l;
        cd.score = 0;  // the FAST score has done its work
        cd.harris = v[i].r;
        cd.pad = 0;
        selcand[(size_t)f * sel_fs + (size_t)l * sel_cap + i] = cd;
    }
    if (tid == 0) *cnt = n;
}

// keypoints per frame = sum of the per-level selections, checked against the launch bound
__global__ void k_sel_count(const PyrGeom* __restrict__ g, const int* __restrict__ selcnt, int* __restrict__ flags,
                            int* __restrict__ nkp, int n_bound, int nframes) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f > nframes) return;
    if (f == nframes) {
        nkp[f] = 0;
        return;
    }
    int n = 0;
    for (int l = 0; l < g->nlevels; l++) n += selcnt[(size_t)f * VO_ORB_MAX_LEVELS + l];
    if (n > n_bound) {
        atomicOr(&flags[f], VO_SEL_OVERFLOW);
        n = n_bound;
    }
    nkp[f] = n;
}

// k_sel_count for the detection path that describes with k_orient_rbrief: besides the count, the frame's
// per-level lists are copied into one list in keypoint order (flat[f][k], the layout of the host selection's
// d_sel), so that the describing kernel finds its candidate by index alone instead of walking the level
// counts first -- one dependent load less in front of its pixel loads.  One block per frame (+ one for the
// sentinel count).
__global__ __launch_bounds__(256) void k_sel_flatten(const PyrGeom* __restrict__ g, const int* __restrict__ selcnt,
                                                     int sel_cap, const Cand* __restrict__ selcand,
                                                     int sel_fs, int* __restrict__ flags, int* __restrict__ nkp,
                                                     int n_bound, int nframes, Cand* __restrict__ flat, int kp_cap) {
    const int f = blockIdx.x;
    if (f == nframes) {
        if (threadIdx.x == 0) nkp[f] = 0;
        return;
    }
    const int nl = g->nlevels;
    int cnt[VO_ORB_MAX_LEVELS], ofs[VO_ORB_MAX_LEVELS], n = 0;
#pragma unroll
    for (int i = 0; i < VO_ORB_MAX_LEVELS; i++) {
        cnt[i] = i < nl ? selcnt[(size_t)f * VO_ORB_MAX_LEVELS + i] : 0;
        ofs[i] = i * sel_cap;
        n += cnt[i];
    }
    if (n > n_bound) {
        if (threadIdx.x == 0) atomicOr(&flags[f], VO_SEL_OVERFLOW);
        n = n_bound;
    }
    if (threadIdx.x == 0) nkp[f] = n;
    const uint4* src = reinterpret_cast<const uint4*>(selcand + (size_t)f * sel_fs);
    uint4* dst = reinterpret_cast<uint4*>(flat + (size_t)f * kp_cap);
    static_assert(sizeof(Cand) == 16, "candidates are copied as 16-byte words");
    for (int k = threadIdx.x; k < n; k += 256) {
        int rem = k, at = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < VO_ORB_MAX_LEVELS; i++) {
            if (!found && rem < cnt[i]) {
                found = true;
                at = ofs[i] + rem;
            }
            if (!found) rem -= cnt[i];
        }
        dst[k] = src[at];
    }
}

// ---------------------------------------------------------------------------------------------
// IC angle: integer moments over the radius-15 disc (umax rows), fastAtan2, then the keypoint
// is written in level-0 coordinates (pt *= scale).  One wave per keypoint.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float atan2_p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float atan2_p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float atan2_p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float atan2_p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Byte masks of the disc: entry [|v|][o][k] keeps the bytes of dword k of the 36-byte row window (which
// starts o = 0..3 pixels left of u = -15) whose u lies within the row's half-width.  Compile-time table.
struct IcMaskTab {
    uint32_t m[16][4][12];  // nine dwords used, padded to 48 bytes for 16-byte loads
};
constexpr IcMaskTab make_ic_masks() {
    IcMaskTab t{};
    const int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int a = 0; a < 16; a++)
        for (int o = 0; o < 4; o++)
            for (int k = 0; k < 9; k++) {
                uint32_t m = 0;
                for (int j = 0; j < 4; j++) {
                    const int u = 4 * k + j - 15 - o;
                    if (u >= -um[a] && u <= um[a]) m |= 0xffu << (8 * j);
                }
                t.m[a][o][k] = m;
            }
    return t;
}
__constant__ __attribute__((aligned(16))) IcMaskTab c_ic_masks = make_ic_masks();

// sum over the 16 lanes of a DPP row, left in every lane of the row
__device__ __forceinline__ int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);  // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);  // row_ror:1
    return v;
}

// Per-keypoint gather kernels, grid (blocks per frame, frames): workgroups are dealt to the 8 XCDs round-robin in
// dispatch order (x fastest), each XCD with its own L2.  Renumbered so that the blocks of one frame run on one
// XCD (frame f of a group of eight on XCD f & 7): a frame's pyramid (1.5 MB) then stays in that XCD's L2 while
// its keypoints' patches are gathered, instead of every XCD fetching every frame (traffic and latency only;
// any bijection is correct).
static __device__ __forceinline__ void xcd_frame_block(int& bx, int& f) {
    const int T = gridDim.x, Z = gridDim.y, full = Z & ~7;
    const int L = blockIdx.x + T * blockIdx.y;
    if (L < full * T) {
        f = (L / (8 * T)) * 8 + (L & 7);
        bx = (L % (8 * T)) >> 3;
    } else {
        const int g = Z - full, l2 = L - full * T;
        f = full + l2 % g;
        bx = l2 / g;
    }
}

// Two keypoints per wave: lane r of a half-wave owns row v = r - 15 of the 31 x 31 patch, reads its
// 36-byte window (nine aligned dwords) and its nine mask dwords, and takes sum(b) and sum((4k + j) b)
// with two accumulating v_dot4_u32_u8 per dword: m10 = sum((4k + j) b) - (15 + o) sum(b), m01 = v sum(b).
// Integer sums: the same m10, m01 as a per-pixel loop.  Few dependent loads: the per-level counts and
// offsets of the frame arrive together (uniform addresses), the level's geometry is fetched while the
// keypoint copy is in flight, the row and mask loads go out together.
__global__ __launch_bounds__(256) void k_ic_angle(const uint8_t* __restrict__ pyr, const PyrGeom* __restrict__ g,
                                                  const Cand* __restrict__ sel, const int* __restrict__ flags,
                                                  int sel_fs, const Cand* __restrict__ selcand,
                                                  const int* __restrict__ selcnt, int sel_cap,
                                                  const int* __restrict__ nkp, int kp_cap,
                                                  vo_keypoint* __restrict__ out) {
    int bx, f;
    xcd_frame_block(bx, f);
    const int n = nkp[f];
    // the per-level counts and offsets travel with the keypoint count (uniform addresses, one batch of
    // scalar loads), before the early exit that only needs the count
    int cnt[VO_ORB_MAX_LEVELS];
    {
        const int* sc = selcnt + (size_t)f * VO_ORB_MAX_LEVELS;
#pragma unroll
        for (int i = 0; i < VO_ORB_MAX_LEVELS; i++) cnt[i] = sc[i];
    }
    if ((int)(bx * 8 + 2 * (threadIdx.x >> 6)) >= n) return;  // the whole wave
    const int k = min((int)(bx * 8 + (threadIdx.x >> 5)), n - 1);
    const bool live = (int)(bx * 8 + (threadIdx.x >> 5)) < n;
    const int r = min((int)threadIdx.x & 31, 30);
    Cand cd;
    int lg;
    // sel != NULL: the frames flagged in `flags` take their keypoints from the host selection
    if (sel && flags[f] != 0) {
        cd = sel[(size_t)f * kp_cap + k];
        lg = cd.level;
    } else {
        const int nl = g->nlevels;
        int l = 0, rem = k, base = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < VO_ORB_MAX_LEVELS; i++)
            if (!found && i < nl) {
                if (rem < cnt[i]) {
                    found = true;
                    l = i;
                    base = i * sel_cap;
                } else {
                    rem -= cnt[i];
                }
            }
        if (!found) rem = 0;  // k < nkp = sum of the counts: not reached
        lg = l;
        cd = selcand[(size_t)f * sel_fs + base + rem];
    }
    const int pitch = g->lv[lg].pitch;
    const unsigned lofs = g->lv[lg].offset;
    float lscale = g->lv[lg].scale;
    pin_loaded(lscale);  // with the other loads, not at the end of the wave's life
    pin_loaded(cd.harris);
    const int xw = (cd.x - 15) & ~3, o = (cd.x - 15) - xw;
    const int v = r - 15;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(pyr + (size_t)f * g->pyr_bytes + lofs +
                                                            (size_t)__mul24(VO_PADY + cd.y + v, pitch) + VO_PADX + xw);
    const uint32_t* msk = c_ic_masks.m[v < 0 ? -v : v][o];
    uint32_t dw[9], mk[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        dw[i] = row[i];
        mk[i] = msk[i];
    }
    uint32_t sb = 0, sw = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t d = dw[i] & mk[i];
        sb = __builtin_amdgcn_udot4(d, 0x01010101u, sb, false);
        sw = __builtin_amdgcn_udot4(d, 0x03020100u + 0x04040404u * i, sw, false);
    }
    int m10 = (int)sw - __mul24(15 + o, (int)sb), m01 = __mul24(v, (int)sb);
    if (!live || (threadIdx.x & 31) == 31) m10 = m01 = 0;
    m10 = row16_sum(m10);
    m01 = row16_sum(m01);
    // halves of the wave: rows 0-1 and rows 2-3
    const int a10 = __builtin_amdgcn_readlane(m10, 0) + __builtin_amdgcn_readlane(m10, 16);
    const int b10 = __builtin_amdgcn_readlane(m10, 32) + __builtin_amdgcn_readlane(m10, 48);
    const int a01 = __builtin_amdgcn_readlane(m01, 0) + __builtin_amdgcn_readlane(m01, 16);
    const int b01 = __builtin_amdgcn_readlane(m01, 32) + __builtin_amdgcn_readlane(m01, 48);
    if (live && (threadIdx.x & 31) == 0) {
        const bool hi = (threadIdx.x & 32) != 0;
        const float s = lscale;
        vo_keypoint kp;
        kp.x = (float)cd.x * s;
        kp.y = (float)cd.y * s;
        kp.size = 31 * s;
        kp.angle = fast_atan2_deg((float)(hi ? b01 : a01), (float)(hi ? b10 : a10));
        kp.response = cd.harris;
        kp.octave = cd.level;
        kp.class_id = -1;
        out[(size_t)f * kp_cap + k] = kp;
    }
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur 7x7 sigma 2 (cv's 8-bit fixed-point separable path, BORDER_REFLECT_101): row pass
// in 32-bit ints, column pass (sum + 2^15) >> 16, saturate.  No LDS: a thread owns a column of four
// pixels, reads three aligned dwords per input row (the reflect-101 frame is real memory) and walks
// down GB_ROWS output rows two input rows at a time.  All levels of all frames are one launch
// (GaussGrid maps blockIdx.x to a level).  The taps (<= 55) are bytes and the row sums
// (<= 257 * 255 = 65535) are 16-bit, so the row pass is two v_dot4_u32_u8 per pixel on byte windows
// cut with v_alignbyte, and the column pass four v_dot2_u32_u16 per pixel on row sums packed in
// (even row, odd row) pairs: the sums are exact integers either way, the same as cv's.
// ---------------------------------------------------------------------------------------------
#define GB_ROWS 16

__global__ __launch_bounds__(256) void k_gauss7(const uint8_t* __restrict__ pyr, const PyrGeom* __restrict__ g,
                                                uint8_t* __restrict__ blur, Taps7 taps, GaussGrid gg) {
    const int f = blockIdx.y;
    int level = 0;
    for (int i = 1; i < gg.nlevels; i++)
        if ((int)blockIdx.x >= gg.b0[i]) level = i;
    const int local = blockIdx.x - gg.b0[level];
    const int bxi = local % gg.bx[level], byi = local / gg.bx[level];
    const int w = g->lv[level].w, h = g->lv[level].h, pitch = g->lv[level].pitch;
    const int x4 = 4 * (bxi * 64 + (threadIdx.x & 63));
    const int y0 = GB_ROWS * (byi * 4 + (threadIdx.x >> 6));
    if (x4 >= w || y0 >= h) return;
    const uint8_t* src = level_px(pyr, g, f, level) + x4;
    uint8_t* dst = level_px(blur, g, f, level) + x4;
    const uint32_t t0 = taps.t[0], t1 = taps.t[1], t2 = taps.t[2], t3 = taps.t[3], t4 = taps.t[4], t5 = taps.t[5],
                   t6 = taps.t[6];
    const uint32_t TA = t0 | (t1 << 8) | (t2 << 16) | (t3 << 24), TB = t4 | (t5 << 8) | (t6 << 16);
    // column taps for an even output row (input rows 2m .. 2m+6) and an odd one (2m+1 .. 2m+7)
    const uint32_t E0 = t0 | (t1 << 16), E1 = t2 | (t3 << 16), E2 = t4 | (t5 << 16), E3 = t6;
    const uint32_t O0 = t0 << 16, O1 = t1 | (t2 << 16), O2 = t3 | (t4 << 16), O3 = t5 | (t6 << 16);
    // the three dwords of input row r (12-byte window A B C), and the row sums of the four pixels from them;
    // the loads of the next two row pairs are issued before the sums of the current one are taken: a wave has
    // one short dependent step per pair otherwise and runs at memory latency
    auto row_load = [&](int r, uint32_t (&W)[3]) {
        const int y = min(y0 - 3 + r, h + 2);
        const uint32_t* row = reinterpret_cast<const uint32_t*>(src + (ptrdiff_t)y * pitch);
        W[0] = row[-1], W[1] = row[0], W[2] = row[1];
    };
    auto row_sums = [&](const uint32_t (&W)[3], uint32_t (&H)[4]) {
        const uint32_t A = W[0], B = W[1], C = W[2];
        // pixel j sums bytes 1+j .. 7+j of the 12-byte window A B C
        H[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 1), TA,
                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 1), TB, 0, false), false);
        H[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 2), TA,
                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 2), TB, 0, false), false);
        H[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 3), TA,
                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 3), TB, 0, false), false);
        H[3] = __builtin_amdgcn_udot4(B, TA, __builtin_amdgcn_udot4(C, TB, 0, false), false);
    };
    constexpr int NK = (GB_ROWS + 6) / 2, AHEAD = 2;
    uint32_t W[NK + AHEAD][2][3];
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {
        row_load(2 * k, W[k][0]);
        row_load(2 * k + 1, W[k][1]);
    }
    uint32_t P[4][4] = {};  // the last four (even, odd) pairs of row sums, per pixel
#pragma unroll
    for (int k = 0; k < NK; k++) {
        if (k + AHEAD < NK) {
            row_load(2 * (k + AHEAD), W[k + AHEAD][0]);
            row_load(2 * (k + AHEAD) + 1, W[k + AHEAD][1]);
        }
#pragma unroll
        for (int u = 0; u < 3; u++) pin_loaded(W[k][0][u]), pin_loaded(W[k][1][u]);
        uint32_t He[4], Ho[4];
        row_sums(W[k][0], He);
        row_sums(W[k][1], Ho);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            P[0][j] = P[1][j];
            P[1][j] = P[2][j];
            P[2][j] = P[3][j];
            P[3][j] = He[j] | (Ho[j] << 16);
        }
        if (k >= 3) {
            const int yo = y0 + 2 * (k - 3);
            uint32_t ae[4], ao[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                ae[j] = udot2(P[0][j], E0, udot2(P[1][j], E1, udot2(P[2][j], E2, udot2(P[3][j], E3, 1u << 15))));
                ao[j] = udot2(P[0][j], O0, udot2(P[1][j], O1, udot2(P[2][j], O2, udot2(P[3][j], O3, 1u << 15))));
            }
            // the taps sum to 257, so a sum can reach 257 << 16: the high halves of two sums side by side
            // (one byte permute), saturated to bytes by v_sat_pk_u8_i16, two of those joined by another permute
            auto pack4 = [](const uint32_t (&a)[4]) {
                uint32_t lo, hi;
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(lo) : "v"(__builtin_amdgcn_perm(a[1], a[0], 0x07060302u)));
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(hi) : "v"(__builtin_amdgcn_perm(a[3], a[2], 0x07060302u)));
                return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            };
            const uint32_t oe = pack4(ae), oo = pack4(ao);
            if (yo < h) *reinterpret_cast<uint32_t*>(dst + (ptrdiff_t)yo * pitch) = oe;
            if (yo + 1 < h) *reinterpret_cast<uint32_t*>(dst + (ptrdiff_t)(yo + 1) * pitch) = oo;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rBRIEF-256 (WTA_K = 2): 32 lanes per keypoint, lane b computes descriptor byte b (16 rotated
// pattern points).  cos/sin are taken in double and rounded to float, as cv does (once per keypoint,
// by the first lanes of the block).  The rotated pattern stays within 18 pixels of the centre
// (largest pattern norm 18.4), so the 37 x 37 window is first copied to LDS with aligned dword loads,
// three rows of ten dwords per step, and the 512 scattered byte reads per keypoint go to LDS.
// ---------------------------------------------------------------------------------------------
// (float)cos(a), (float)sin(a) of a double a in [0, 2 pi]: quadrant by two-part Cody-Waite reduction (k <= 4, so
// k * PIO2_HI is exact), fdlibm's kernel polynomials on |r| <= pi / 4, errors below one double ulp -- the
// float results differ from those of a correctly rounded double cos / sin only when that ulp straddles a
// float rounding boundary (2^-29 of all arguments).  ~45 double instructions instead of the library
// routine's ~200, which every block's first wave ran alone while the other three waited at the barrier.
__device__ __forceinline__ void sincos_2pi(double a, float& cf, float& sf) {
    const double kd = __builtin_rint(a * 6.36619772367581382433e-01);
    const int q = (int)kd;
    double r = __builtin_fma(-kd, 1.57079632673412561417e+00, a);
    r = __builtin_fma(-kd, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(z * r, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cs = 1.0 - __builtin_fma(z, 0.5, -(z * (z * pc)));
    const double c4 = (q & 1) ? sn : cs, s4 = (q & 1) ? cs : sn;
    cf = (float)(((q + 1) & 2) ? -c4 : c4);
    sf = (float)((q & 2) ? -s4 : s4);
}

__constant__ __attribute__((aligned(16))) int8_t c_orb_pattern[1024] = {
#include "orb_pattern.inc"
};

#define RB_R 18
#define RB_ROWS (2 * RB_R + 1)
#define RB_DW 10  // dwords per window row: 37 bytes at any byte offset 0..3
#define RB_KP 8   // keypoints per block

__global__ __launch_bounds__(256) void k_rbrief(const PyrGeom* __restrict__ g, const uint8_t* __restrict__ blur,
                                                const vo_keypoint* __restrict__ kps, const int* __restrict__ nkp,
                                                int kp_cap, uint8_t* __restrict__ desc) {
    __shared__ __attribute__((aligned(16))) float patf[1024];
    __shared__ uint32_t win[RB_KP][RB_ROWS * RB_DW];
    __shared__ float s_ca[RB_KP], s_sa[RB_KP];
    __shared__ PyrGeom s_g;
    const int tid = threadIdx.x;
    int bx, f;
    xcd_frame_block(bx, f);
    const int k0 = bx * RB_KP, n = nkp[f];
    if (k0 >= n) return;
    // Loads that depend on nothing else go out together: geometry, pattern, the slot's keypoint and (first
    // lanes) the angles; the window loads follow in one batch after the first barrier.
    const int slot = tid >> 5, b = tid & 31, k = k0 + slot;
    const bool live = k < n;
    const vo_keypoint kp = kps[(size_t)f * kp_cap + min(k, n - 1)];
    float angle = kps[(size_t)f * kp_cap + min(k0 + (tid & (RB_KP - 1)), n - 1)].angle;
    static_assert(sizeof(PyrGeom) % 4 == 0 && sizeof(PyrGeom) / 4 <= 256, "geometry copied by one pass");
    constexpr int GEOM_DW = (int)(sizeof(PyrGeom) / 4);
    const uint32_t gdw = reinterpret_cast<const uint32_t*>(g)[min(tid, GEOM_DW - 1)];
    const uint32_t q = reinterpret_cast<const uint32_t*>(c_orb_pattern)[tid];
    if (tid < GEOM_DW) reinterpret_cast<uint32_t*>(&s_g)[tid] = gdw;
    {
        float4 v;
        v.x = (float)(int8_t)(q & 0xff);
        v.y = (float)(int8_t)((q >> 8) & 0xff);
        v.z = (float)(int8_t)((q >> 16) & 0xff);
        v.w = (float)(int8_t)(q >> 24);
        // float4 tid = pattern points 2 tid, 2 tid + 1 = pair (tid & 7) of descriptor byte tid >> 3; stored
        // pair-major so that the 32 lanes of a group read consecutive 16-byte slots (byte-major, every lane's
        // slot started 128 bytes after its neighbour's: all in the same four banks, a 32-way conflict per read)
        reinterpret_cast<float4*>(patf)[(tid & 7) * 32 + (tid >> 3)] = v;
    }
    __syncthreads();
    const int l = kp.octave;
    const int pitch = s_g.lv[l].pitch, h = s_g.lv[l].h;
    const float scale = 1.f / s_g.lv[l].scale;
    const int cy = __float2int_rn(kp.y * scale), cx = __float2int_rn(kp.x * scale);
    const int xw = (cx - RB_R) & ~3, o = cx - RB_R - xw;
    const bool stage = live && b < 3 * RB_DW;
    const int r = b / RB_DW, c = b - r * RB_DW;
    // Thirteen rows per staging lane (rows r, r + 3, ...; the last only for r = 0).  A wave whose windows all lie
    // inside their level's padded frame (always, at edge thresholds >= 19) addresses them by one 32-bit offset
    // per lane from the frame's base and one addition per row; otherwise rows and columns are clamped to the
    // frame, so that a window leaving it reads defined memory.
    constexpr int NLD = (RB_ROWS + 2) / 3;
    static_assert(3 * (NLD - 1) == RB_ROWS - 1, "the last step holds row r = 0 only");
    const uint8_t* fb = blur + (size_t)f * s_g.pyr_bytes;
    const bool inside = cy - RB_R >= -VO_PADY && cy + RB_R <= h + VO_PADY - 1 && xw >= -VO_PADX &&
                        xw + 4 * RB_DW <= pitch - VO_PADX;
    uint32_t wd[NLD] = {};
    if (__builtin_amdgcn_ballot_w64(live && !inside) == 0) {
        uint32_t off = s_g.lv[l].offset + (uint32_t)((VO_PADY + cy - RB_R + r) * pitch + VO_PADX + xw + 4 * c);
        const uint32_t step = 3u * (uint32_t)pitch;
        if (stage) {
#pragma unroll
            for (int i = 0; i < NLD - 1; i++, off += step) wd[i] = *reinterpret_cast<const uint32_t*>(fb + off);
            if (r == 0) wd[NLD - 1] = *reinterpret_cast<const uint32_t*>(fb + off);
        }
    } else {
        const int gx = min(max(xw + 4 * c, -VO_PADX), pitch - VO_PADX - 4);
        const uint8_t* col = fb + s_g.lv[l].offset + (size_t)VO_PADY * pitch + VO_PADX + gx;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int row = r + 3 * i;
            const int gy = min(max(cy - RB_R + row, -VO_PADY), h + VO_PADY - 1);
            if (stage && row < RB_ROWS) wd[i] = *reinterpret_cast<const uint32_t*>(col + (ptrdiff_t)gy * pitch);
        }
    }
    if (tid < RB_KP) {  // overlaps the window loads
        angle *= (float)(3.14159265358979323846 / 180.f);
        sincos_2pi((double)angle, s_ca[tid], s_sa[tid]);
    }
#pragma unroll
    for (int i = 0; i < (RB_ROWS + 2) / 3; i++) {
        const int row = r + 3 * i;
        if (stage && row < RB_ROWS) win[slot][row * RB_DW + c] = wd[i];
    }
    __syncthreads();
    if (!live) return;
    const float ca = s_ca[slot], sa = s_sa[slot];
    // cvRound by the magic constant: y + 1.5 * 2^23 holds rint(y) in its low mantissa bits, so one 24-bit
    // multiply-add of the two bit patterns is row * 40 + column plus a constant that goes into the base
    typedef __attribute__((address_space(3))) const uint8_t lds_u8;
    constexpr float MAGIC = 12582912.f;
    constexpr uint32_t MAGIC_K = 0x4B400000u + (uint32_t)(4 * RB_DW) * 0x400000u;
    const uint32_t center = (uint32_t)(uintptr_t)(lds_u8*)(reinterpret_cast<const uint8_t*>(win[slot]) +
                                                           RB_R * (4 * RB_DW) + RB_R + o) - MAGIC_K;
    const float4* p = reinterpret_cast<const float4*>(patf) + b;
    int val = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        int v[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float4 pp = p[32 * t];
            const float2 pt = e ? make_float2(pp.z, pp.w) : make_float2(pp.x, pp.y);
            const float x = pt.x * ca - pt.y * sa;
            const float y = pt.x * sa + pt.y * ca;
            const int at = __mul24(__builtin_bit_cast(int, y + MAGIC), 4 * RB_DW) + __builtin_bit_cast(int, x + MAGIC);
            v[e] = *(lds_u8*)(uintptr_t)(center + (uint32_t)at);
        }
        val |= (v[0] < v[1]) << t;
    }
    desc[((size_t)f * kp_cap + k) * 32 + b] = (uint8_t)val;
}

// ---------------------------------------------------------------------------------------------
// k_ic_angle and k_rbrief in one kernel, for the detection path (the blurred pyramid is made before it).
// Both give a keypoint 32 lanes -- patch row / descriptor byte -- so the block keeps k_rbrief's shape and a
// group first does k_ic_angle's work: candidate -> nine patch dwords and nine mask dwords per lane, with the
// 37 x 37 window of the blurred level requested in the same batch (it depends on the position only, not on
// the angle), moments by dot products and DPP sums, the angle on the group's first lane.  What the separate
// kernels paid for twice -- count -> candidate -> geometry -> pixels, one wave lifetime each -- and the trip
// of the keypoint record through memory between them happen once.  Same arithmetic, same results.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_orient_rbrief(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                       const PyrGeom* __restrict__ g, const Cand* __restrict__ flat,
                                                       const int* __restrict__ nkp, int kp_cap,
                                                       vo_keypoint* __restrict__ out, uint8_t* __restrict__ desc) {
    __shared__ __attribute__((aligned(16))) float patf[1024];
    __shared__ uint32_t win[RB_KP][RB_ROWS * RB_DW];
    __shared__ float s_m01[RB_KP], s_m10[RB_KP], s_ca[RB_KP], s_sa[RB_KP];
    const int tid = threadIdx.x;
    int bx, f;
    xcd_frame_block(bx, f);
    const int k0 = bx * RB_KP;
    const int slot = tid >> 5, b = tid & 31;
    // The count and the group's candidate (flat[f][k]: k_sel_flatten's list, or the host selection's rows of a
    // flagged frame) are requested together; a slot at or beyond the count holds no candidate (stale memory):
    // it is replaced before anything is derived from it and loads no pixels.
    const int n = nkp[f];
    Cand cd = flat[(size_t)f * kp_cap + min(k0 + slot, kp_cap - 1)];
    if (k0 >= n) return;  // the whole block
    const int k = min(k0 + slot, n - 1);
    const bool live = k0 + slot < n;
    const uint32_t q = reinterpret_cast<const uint32_t*>(c_orb_pattern)[tid];
    if (!live) cd.x = cd.y = cd.level = 0;
    const int lg = cd.level;
    {
        float4 v;
        v.x = (float)(int8_t)(q & 0xff);
        v.y = (float)(int8_t)((q >> 8) & 0xff);
        v.z = (float)(int8_t)((q >> 16) & 0xff);
        v.w = (float)(int8_t)(q >> 24);
        // float4 tid = pattern points 2 tid, 2 tid + 1 = pair (tid & 7) of descriptor byte tid >> 3; stored
        // pair-major so that the 32 lanes of a group read consecutive 16-byte slots (byte-major, every lane's
        // slot started 128 bytes after its neighbour's: all in the same four banks, a 32-way conflict per read)
        reinterpret_cast<float4*>(patf)[(tid & 7) * 32 + (tid >> 3)] = v;
    }
    // a block's keypoints are neighbours in level order: nearly always one level per wave, whose geometry then
    // comes by scalar loads (a short hop) instead of a second round of vector loads
    int pitch, h;
    unsigned lofs;
    float lscale;
    const int lg0 = __builtin_amdgcn_readfirstlane(lg);
    if (__builtin_amdgcn_ballot_w64(lg != lg0) == 0) {
        pitch = g->lv[lg0].pitch, h = g->lv[lg0].h, lofs = g->lv[lg0].offset, lscale = g->lv[lg0].scale;
    } else {
        pitch = g->lv[lg].pitch, h = g->lv[lg].h, lofs = g->lv[lg].offset, lscale = g->lv[lg].scale;
    }
    const size_t fofs = (size_t)f * g->pyr_bytes;
    // --- orientation: the lane's patch row
    const int r = min(b, 30), v = r - 15;
    uint32_t dw[9], mk[9];
    int oi;
    {
        const int xw = (cd.x - 15) & ~3;
        oi = (cd.x - 15) - xw;
        // a slot without a candidate reads the first bytes of the frame's pyramid instead of a patch: no branch
        // around the loads (one per load would make each wait for the one before)
        const uint32_t* row = reinterpret_cast<const uint32_t*>(
            pyr + fofs + (live ? lofs + (size_t)__mul24(VO_PADY + cd.y + v, pitch) + VO_PADX + xw : (size_t)0));
        const uint32_t* msk = c_ic_masks.m[v < 0 ? -v : v][oi];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            dw[i] = row[i];
            mk[i] = msk[i];
        }
    }
    // --- description: the lane's rows of the blurred window, as in k_rbrief
    vo_keypoint kp;
    kp.x = (float)cd.x * lscale;
    kp.y = (float)cd.y * lscale;
    kp.size = 31 * lscale;
    kp.response = cd.harris;
    kp.octave = cd.level;
    kp.class_id = -1;
    const float scale = 1.f / lscale;
    const int cy = __float2int_rn(kp.y * scale), cx = __float2int_rn(kp.x * scale);
    const int xw = (cx - RB_R) & ~3, o = cx - RB_R - xw;
    const bool stage = live && b < 3 * RB_DW;
    const int rr = b / RB_DW, c = b - rr * RB_DW;
    constexpr int NLD = (RB_ROWS + 2) / 3;
    const uint8_t* fb = blur + fofs;
    const bool inside = cy - RB_R >= -VO_PADY && cy + RB_R <= h + VO_PADY - 1 && xw >= -VO_PADX &&
                        xw + 4 * RB_DW <= pitch - VO_PADX;
    uint32_t wd[NLD] = {};
    if (__builtin_amdgcn_ballot_w64(live && !inside) == 0) {
        uint32_t off = lofs + (uint32_t)((VO_PADY + cy - RB_R + rr) * pitch + VO_PADX + xw + 4 * c);
        const uint32_t step = 3u * (uint32_t)pitch;
        if (stage) {
#pragma unroll
            for (int i = 0; i < NLD - 1; i++, off += step) wd[i] = *reinterpret_cast<const uint32_t*>(fb + off);
            if (rr == 0) wd[NLD - 1] = *reinterpret_cast<const uint32_t*>(fb + off);
        }
    } else {
        const int gx = min(max(xw + 4 * c, -VO_PADX), pitch - VO_PADX - 4);
        const uint8_t* col = fb + lofs + (size_t)VO_PADY * pitch + VO_PADX + gx;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int row = rr + 3 * i;
            const int gy = min(max(cy - RB_R + row, -VO_PADY), h + VO_PADY - 1);
            if (stage && row < RB_ROWS) wd[i] = *reinterpret_cast<const uint32_t*>(col + (ptrdiff_t)gy * pitch);
        }
    }
    // --- moments and angle (k_ic_angle's arithmetic)
    {
        uint32_t sb = 0, sw = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t d = dw[i] & mk[i];
            sb = __builtin_amdgcn_udot4(d, 0x01010101u, sb, false);
            sw = __builtin_amdgcn_udot4(d, 0x03020100u + 0x04040404u * i, sw, false);
        }
        int m10 = (int)sw - __mul24(15 + oi, (int)sb), m01 = __mul24(v, (int)sb);
        if (!live || b == 31) m10 = m01 = 0;
        m10 = row16_sum(m10);
        m01 = row16_sum(m01);
        // halves of the wave: rows 0-1 and rows 2-3
        const int a10 = __builtin_amdgcn_readlane(m10, 0) + __builtin_amdgcn_readlane(m10, 16);
        const int b10 = __builtin_amdgcn_readlane(m10, 32) + __builtin_amdgcn_readlane(m10, 48);
        const int a01 = __builtin_amdgcn_readlane(m01, 0) + __builtin_amdgcn_readlane(m01, 16);
        const int b01 = __builtin_amdgcn_readlane(m01, 32) + __builtin_amdgcn_readlane(m01, 48);
        // the group's first lane stores the keypoint without its angle and hands the moments to the block's
        // first lanes, which take atan2 and cos / sin for all eight keypoints at once after the barrier
        if (b == 0) {
            const bool hi = (tid & 32) != 0;
            s_m01[slot] = (float)(hi ? b01 : a01);
            s_m10[slot] = (float)(hi ? b10 : a10);
            kp.angle = 0.f;
            if (live) out[(size_t)f * kp_cap + k] = kp;
        }
    }
#pragma unroll
    for (int i = 0; i < NLD; i++) {
        const int row = rr + 3 * i;
        if (stage && row < RB_ROWS) win[slot][row * RB_DW + c] = wd[i];
    }
    __syncthreads();
    if (tid < RB_KP) {
        float angle = fast_atan2_deg(s_m01[tid], s_m10[tid]);
        if (k0 + tid < n) out[(size_t)f * kp_cap + k0 + tid].angle = angle;
        angle *= (float)(3.14159265358979323846 / 180.f);
        sincos_2pi((double)angle, s_ca[tid], s_sa[tid]);
    }
    __syncthreads();
    if (!live) return;
    const float ca = s_ca[slot], sa = s_sa[slot];
    typedef __attribute__((address_space(3))) const uint8_t lds_u8;
    constexpr float MAGIC = 12582912.f;
    constexpr uint32_t MAGIC_K = 0x4B400000u + (uint32_t)(4 * RB_DW) * 0x400000u;
    const uint32_t center = (uint32_t)(uintptr_t)(lds_u8*)(reinterpret_cast<const uint8_t*>(win[slot]) +
                                                           RB_R * (4 * RB_DW) + RB_R + o) - MAGIC_K;
    const float4* p = reinterpret_cast<const float4*>(patf) + b;
    int val = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        int vv[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float4 pp = p[32 * t];
            const float2 pt = e ? make_float2(pp.z, pp.w) : make_float2(pp.x, pp.y);
            const float x = pt.x * ca - pt.y * sa;
            const float y = pt.x * sa + pt.y * ca;
            const int at = __mul24(__builtin_bit_cast(int, y + MAGIC), 4 * RB_DW) + __builtin_bit_cast(int, x + MAGIC);
            vv[e] = *(lds_u8*)(uintptr_t)(center + (uint32_t)at);
        }
        val |= (vv[0] < vv[1]) << t;
    }
    desc[((size_t)f * kp_cap + k) * 32 + b] = (uint8_t)val;
}

// The keypoints, descriptors and count of frame `from` of the chunk tables to frame 0: consecutive chunks of a
// sequence share a frame, and the context that ran the chunk before still holds what it computed for it.
__global__ __launch_bounds__(256) void k_carry_frame(vo_keypoint* __restrict__ kp, uint8_t* __restrict__ desc,
                                                     int* __restrict__ nkp, int kp_cap, int from) {
    const int n = min(nkp[from], kp_cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) nkp[0] = n;
    const uint32_t* ks = reinterpret_cast<const uint32_t*>(kp + (size_t)from * kp_cap);
    uint32_t* kd = reinterpret_cast<uint32_t*>(kp);
    const uint4* ds = reinterpret_cast<const uint4*>(desc + (size_t)from * kp_cap * 32);
    uint4* dd = reinterpret_cast<uint4*>(desc);
    static_assert(sizeof(vo_keypoint) == 28, "keypoints are copied as seven dwords");
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 7 * n; i += gridDim.x * 256) kd[i] = ks[i];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += gridDim.x * 256) dd[i] = ds[i];
}

void launch_carry_frame(hipStream_t st, vo_keypoint* kp, uint8_t* desc, int* nkp, int kp_cap, int from) {
    hipLaunchKernelGGL(k_carry_frame, dim3(16), dim3(256), 0, st, kp, desc, nkp, kp_cap, from);
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
void launch_pyr_level0(hipStream_t st, const uint8_t* img, size_t frame_stride, int stride, const PyrGeom* d_geom,
                       const PyrGeom& hg, uint8_t* pyr, int nframes) {
    (void)d_geom;
    const LevelGeom& L = hg.lv[0];
    const int nch = L.w - 15 > 0 ? (L.w - 16) / 16 + 1 : 0;  // whole 16-byte chunks inside [0, w)
    const int nint = cdiv(nch * (L.h + 6), 256 * L0_CHUNKS);   // blocks of the interior
    const unsigned magic = nch > 1 ? (unsigned)((0x100000000ull + (unsigned)nch - 1) / (unsigned)nch) : 0xffffffffu;
    const int end_int = nch * 16;
    const int nedge_blocks = cdiv((L.h + 6) * (1 + (L.w + 3 - end_int + 15) / 16), 256);
    dim3 grid(nedge_blocks + nint, 1, nframes);
    hipLaunchKernelGGL(k_pyr_level0, grid, dim3(256), 0, st, img, frame_stride, stride, L.w, L.h, L.pitch,
                       hg.pyr_bytes, L.offset + (unsigned)(VO_PADY * L.pitch + VO_PADX), nedge_blocks, magic, pyr);
}

void launch_pyr_resize(hipStream_t st, const PyrGeom* d_geom, const PyrGeom& hg, uint8_t* pyr, int level,
                       const uint32_t* xtab, const uint32_t* ytab, int ldw, int lrows, int nframes) {
    (void)d_geom;
    const LevelGeom &D = hg.lv[level], &S = hg.lv[level - 1];
    ResizeArgs a;
    a.dw = D.w, a.dh = D.h, a.dpitch = D.pitch;
    a.sw = S.w, a.sh = S.h, a.spitch = S.pitch;
    a.soff = S.offset + (unsigned)(VO_PADY * S.pitch + VO_PADX);
    a.doff = D.offset + (unsigned)(VO_PADY * D.pitch + VO_PADX);
    a.pyr_bytes = hg.pyr_bytes;
    a.ldw = ldw;
    a.magic = (65536u + (unsigned)ldw - 1) / (unsigned)ldw;
    a.ngroups = std::max(256 / ldw, 1);
    a.grows = cdiv(lrows, a.ngroups);
    dim3 grid(cdiv(D.w + 7, 256), cdiv(D.h + 6, PR_ROWS), nframes);
    const size_t lds = vo_resize_lds_bytes(ldw, lrows);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_resize),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_pyr_resize, grid, dim3(256), lds, st, pyr, a, xtab, ytab);
}

void launch_fast_nms(hipStream_t st, const PyrGeom* d_geom, const PyrGeom& hg, const uint8_t* pyr, uint8_t* score,
                     unsigned long long* masks, int threshold, int border, int nms, int nframes, int* flags) {
    if (hg.tiles_total == 0) {
        if (flags) (void)hipMemsetAsync(flags, 0, (size_t)(nframes + 1) * sizeof(int), st);
        return;
    }
    dim3 grid(hg.tiles_total, nframes);
    FastTiles ft = {};
    ft.nlevels = hg.nlevels;
    for (int i = 0; i < hg.nlevels; i++) ft.tile0[i] = hg.lv[i].tile0;
    hipLaunchKernelGGL(k_fast_nms, grid, dim3(256), 0, st, pyr, d_geom, score, masks, threshold, border, nms, ft,
                       flags);
}

void launch_row_scan(hipStream_t st, const PyrGeom* d_geom, const unsigned long long* masks, int* segbase,
                     int* lvlofs, int* flags, int nframes) {
    hipLaunchKernelGGL(k_row_scan, dim3(nframes), dim3(1024), 0, st, d_geom, masks, segbase, lvlofs, flags);
}

void launch_cand_write(hipStream_t st, const PyrGeom* d_geom, const PyrGeom& hg, const unsigned long long* masks,
                       const int* segbase, const uint8_t* score, int nms, Cand* cand, int max_cand, int nframes) {
    if (hg.segs_total == 0) return;
    dim3 grid(cdiv(hg.segs_total, 256), nframes);
    hipLaunchKernelGGL(k_cand_write, grid, dim3(256), 0, st, d_geom, masks, segbase, score, nms, cand, max_cand);
}

void launch_harris(hipStream_t st, const uint8_t* pyr, const PyrGeom* d_geom, const int* lvlofs, Cand* cand,
                   int max_cand, int grid_cand, const int* only_flagged, int nframes) {
    dim3 grid(cdiv(grid_cand, 256), nframes);
    hipLaunchKernelGGL(k_harris, grid, dim3(256), 0, st, pyr, d_geom, lvlofs, cand, max_cand, only_flagged);
}

void launch_orb_select(hipStream_t st, const uint8_t* pyr, const PyrGeom* d_geom, const PyrGeom& hg,
                       const unsigned long long* masks, const uint8_t* score, int nms, const Quotas& quota,
                       int sel_cap, Cand* selcand, int* selcnt, int* flags, int* nkp, int n_bound, int nframes,
                       Cand* flat, int kp_cap) {
    const size_t lds = (size_t)sel_cap * (sizeof(RI) + 2 * sizeof(uint16_t));
    const int sel_fs = hg.nlevels * sel_cap;  // selcand: [frame][level][sel_cap]
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_orb_select),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_orb_select, dim3(nframes, hg.nlevels), dim3(SEL_THREADS), lds, st, pyr, d_geom, masks, score, nms,
                       quota, sel_cap, selcand, sel_fs, selcnt, flags);
    if (flat)
        hipLaunchKernelGGL(k_sel_flatten, dim3(nframes + 1), dim3(256), 0, st, d_geom, selcnt, sel_cap, selcand,
                           sel_fs, flags, nkp, n_bound, nframes, flat, kp_cap);
    else
        hipLaunchKernelGGL(k_sel_count, dim3(cdiv(nframes + 1, 64)), dim3(64), 0, st, d_geom, selcnt, flags, nkp,
                           n_bound, nframes);
}

void launch_ic_angle(hipStream_t st, const uint8_t* pyr, const PyrGeom* d_geom, const Cand* sel, const int* flags,
                     int sel_fs, const Cand* selcand, const int* selcnt, int sel_cap, const int* nkp, int kp_cap,
                     int max_n, vo_keypoint* out, int nframes) {
    if (max_n <= 0) return;
    dim3 grid(cdiv(max_n, 8), nframes);
    hipLaunchKernelGGL(k_ic_angle, grid, dim3(256), 0, st, pyr, d_geom, sel, flags, sel_fs, selcand, selcnt, sel_cap,
                       nkp, kp_cap, out);
}

void launch_gauss7(hipStream_t st, const uint8_t* pyr, const PyrGeom* d_geom, const PyrGeom& hg, uint8_t* blur,
                   const Taps7& taps, int nlevels_used, int nframes) {
    GaussGrid gg;
    gg.nlevels = std::min(nlevels_used, hg.nlevels);
    int b = 0;
    for (int l = 0; l < gg.nlevels; l++) {
        gg.b0[l] = b;
        gg.bx[l] = cdiv(cdiv(hg.lv[l].w, 4), 64);
        b += gg.bx[l] * cdiv(cdiv(hg.lv[l].h, GB_ROWS), 4);
    }
    if (b == 0) return;
    hipLaunchKernelGGL(k_gauss7, dim3(b, nframes), dim3(256), 0, st, pyr, d_geom, blur, taps, gg);
}

void launch_rbrief(hipStream_t st, const PyrGeom* d_geom, const uint8_t* blur, const vo_keypoint* kps,
                   const int* nkp, int kp_cap, int max_n, uint8_t* desc, int nframes) {
    if (max_n <= 0) return;
    dim3 grid(cdiv(max_n, 8), nframes);
    hipLaunchKernelGGL(k_rbrief, grid, dim3(256), 0, st, d_geom, blur, kps, nkp, kp_cap, desc);
}

void launch_orient_rbrief(hipStream_t st, const uint8_t* pyr, const uint8_t* blur, const PyrGeom* d_geom,
                          const Cand* flat, const int* nkp, int kp_cap, int max_n, vo_keypoint* out, uint8_t* desc,
                          int nframes) {
    if (max_n <= 0) return;
    dim3 grid(cdiv(max_n, RB_KP), nframes);
    hipLaunchKernelGGL(k_orient_rbrief, grid, dim3(256), 0, st, pyr, blur, d_geom, flat, nkp, kp_cap, out, desc);
}
