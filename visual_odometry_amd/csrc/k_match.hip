// k_match.hip — brute-force Hamming 2-NN (BFMatcher(NORM_HAMMING).knnMatch(k=2)) and the
// reference's ratio / displacement filter, for gfx950.
//
// k_knn2: a block owns 64 queries of one pair for the whole train range (or one split of it); every
//   lane of every wave holds one of those queries in registers (8 or 16 dwords) and the block's four
//   waves each scan a quarter of the train rows.  A wave stages its own rows through a wave-private
//   LDS tile (no block barrier in the scan), every lane reads the same row (LDS broadcast), and the
//   running (best, second) are packed keys dist << 22 | index, so min / med3 on the key give the
//   (distance, lower-index-first) order of cv's insertion rule.  The four waves' pairs meet in LDS
//   once at the end and wave 0 writes the merged result: no partial table in HBM, no merge launch.
//   With `cross` the same launch also runs the swapped direction (train rows as queries), which
//   the mutual-consistency option of the filter needs.
// k_knn2_merge: only for one small problem spread over train splits to fill the chip (single-pair
//   entry point): lexicographic merge of the per-split keys.
// k_match_filter: ratio test + displacement gate (+ optional cross-check) + order-preserving compaction.
// Reference: src/matcher/matcher.cpp:17-38; SURVEY.md A.3.
#include "ctx.h"

#define KNN_IDX_MASK ((1u << VO_KEY_SHIFT) - 1u)
#define KNN_GROUP 64  // rows whose (distance << 6 | row) keys fit 16 bits: local top-2 by 16-bit min / max

// Issue costs measured on gfx950 (tools/micro/valu_table.hip, profiles/r2_valu_table.txt): v_xor / v_and /
// v_or / v_add_u32 / v_min_u16 / v_max_u16 run a wave64 in ~2.3 cycles, everything else the kernel needs
// (v_bcnt, v_lshl_or, 32-bit min / max / med3) in ~4.15.  Hence 16-bit keys inside a group of 64 rows.
static __device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// popcount(x) + acc in one instruction.  Written out because the optimizer otherwise rebalances the
// sum of eight popcounts into a tree (eight v_bcnt with a zero addend plus three v_add3).
static __device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
static __device__ __forceinline__ uint32_t bcnt0(uint32_t x) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}
static __device__ __forceinline__ uint32_t min_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ uint32_t max_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Train rows reach the lanes as SGPR operands: a wave reads its rows with scalar loads (one
// s_load_dwordx16 = 64 bytes = two 32-byte rows or one 64-byte row), so the scan loop holds no row in
// VGPRs, issues no LDS or vector-memory instruction, and the xor takes the row word straight from the
// scalar register file.  Scalar loads return out of order (only lgkmcnt(0) is a safe wait), so the
// loads are written as asm and waited for explicitly: the tuple loaded during one batch's arithmetic is
// waited for at the top of the next batch, just before the following tuple is requested.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
static __device__ __forceinline__ void sload16(u32x16& v, const uint32_t* p) {
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(v) : "s"(p));
}
static __device__ __forceinline__ void sload8(u32x8& v, const uint32_t* p) {
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(v) : "s"(p));
}
static __device__ __forceinline__ void swait(u32x16& a, u32x16& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b));
}

// Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), each XCD with its own L2.
// Renumber so that the 64-query tiles of one pair share an XCD: its train rows are then fetched into one
// L2 instead of eight (speed and traffic only; any bijection is correct).
static __device__ __forceinline__ void knn_block(int& bx, int& by, int& bz) {
    const int T = gridDim.x * gridDim.y, Z = gridDim.z, full = Z & ~7;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int rest;
    if (L < full * T) {
        bz = (L / (8 * T)) * 8 + (L & 7);
        rest = (L % (8 * T)) >> 3;
    } else {
        const int g = Z - full, l2 = L - full * T;
        bz = full + l2 % g;
        rest = l2 / g;
    }
    bx = rest % (int)gridDim.x;
    by = rest / (int)gridDim.x;
}

// W32: uint32 words per descriptor (8 or 16).  grid: x = 64-query tile, y = train split,
// z = pair (z >= npairs: the reverse direction of pair z - npairs).
// Per (query, row): W32 xor + W32 bcnt (the popcount instruction accumulates) + key pack + 16-bit max / min / min.
template <int W32>
__global__ __launch_bounds__(256) void k_knn2(KnnArgs a) {
    constexpr int RPT = 16 / W32;  // rows per 16-dword tuple
    constexpr int U = 2 * RPT;     // rows per batch (two tuples)
    __shared__ uint2 wres[4][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int bx, by, bz;
    knn_block(bx, by, bz);
    const bool rev = bz >= a.npairs;
    const int pair = rev ? bz - a.npairs : bz;
    int nq = a.d_n ? a.d_n[pair] : a.nq;
    int nt = a.d_n ? a.d_n[pair + 1] : a.nt;
    const uint32_t* qset = a.q + pair * a.q_stride;
    const uint32_t* tset = a.t + pair * a.t_stride;
    if (rev) {
        const int n = nq;
        nq = nt;
        nt = n;
        const uint32_t* s = qset;
        qset = tset;
        tset = s;
    }
    const int q0 = bx * 64;
    if (q0 >= nq) return;

    uint32_t qv[W32];
    {
        const uint4* Qp = reinterpret_cast<const uint4*>(qset);
        const int qr = min(q0 + lane, nq - 1);
#pragma unroll
        for (int k = 0; k < W32 / 4; k++) {
            const uint4 v = Qp[(size_t)qr * (W32 / 4) + k];
            qv[4 * k] = v.x;
            qv[4 * k + 1] = v.y;
            qv[4 * k + 2] = v.z;
            qv[4 * k + 3] = v.w;
        }
    }
    // rows of this split, then of this wave
    const int per_split = cdiv(nt, a.splits);
    const int sbeg = min(nt, by * per_split), send = min(nt, sbeg + per_split);
    const int per_wave = cdiv(send - sbeg, 4);
    const int wbeg = __builtin_amdgcn_readfirstlane(min(send, sbeg + wave * per_wave));
    const int wend = __builtin_amdgcn_readfirstlane(min(send, wbeg + per_wave));

    uint32_t k1 = VO_KEY_INVALID, k2 = VO_KEY_INVALID;  // global keys: distance << 22 | row
    uint32_t l1 = 0xffffu, l2 = 0xffffu;                // keys of the current group: distance << 6 | row - gbase
    // one row whose words are wave-uniform values (SGPRs) against the lane's query
    auto row = [&](const uint32_t (&w)[W32], int jj) {
        uint32_t d = bcnt0(qv[0] ^ w[0]);
#pragma unroll
        for (int k = 1; k < W32; k++) d = bcnt_acc(qv[k] ^ w[k], d);
        const uint32_t key = (d << 6) | (uint32_t)jj;
        const uint32_t hi = max_u16(key, l1);
        l1 = min_u16(key, l1);
        l2 = min_u16(hi, l2);
    };
    auto tuple_rows = [&](const u32x16& t, int jj) {
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            uint32_t w[W32];
#pragma unroll
            for (int k = 0; k < W32; k++) w[k] = t[r * W32 + k];
            row(w, jj + r);
        }
    };
    // fold the group's two keys into the global pair and start a new group
    auto fold = [&](int gbase) {
        const uint32_t g1 = ((l1 >> 6) << VO_KEY_SHIFT) | (uint32_t)(gbase + (int)(l1 & 63u));
        const uint32_t g2 = ((l2 >> 6) << VO_KEY_SHIFT) | (uint32_t)(gbase + (int)(l2 & 63u));
        const uint32_t e1 = l1 == 0xffffu ? VO_KEY_INVALID : g1, e2 = l2 == 0xffffu ? VO_KEY_INVALID : g2;
        k2 = med3_u32(e1, k1, k2);
        k1 = min(e1, k1);
        k2 = med3_u32(e2, k1, k2);
        k1 = min(e2, k1);
        l1 = l2 = 0xffffu;
    };

    const int nb = (wend - wbeg) / U;  // full batches
    int r = wbeg, gbase = wbeg, jj = 0;
    if (nb > 0) {
        u32x16 A, B, C, D;
        sload16(A, tset + (size_t)r * W32);
        sload16(B, tset + (size_t)(r + RPT) * W32);
        // The SIMD's arbiter serves its oldest wave first, so equal waves finish one after the other and the
        // SIMD ends its launch with one or two waves left, which cannot fill the VALU.  A wave therefore lowers
        // its own priority as it advances: the waves of a SIMD stay together and finish together.
        const int pb1 = (nb / 2) & ~1, pb2 = (nb * 3 / 4) & ~1, pb3 = (nb * 9 / 10) & ~1;
        __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (int b = 0; b < nb; b += 2) {
            if (b == pb1) __builtin_amdgcn_s_setprio(2);
            if (b == pb2) __builtin_amdgcn_s_setprio(1);
            if (b == pb3) __builtin_amdgcn_s_setprio(0);
            // (A, B) requested one batch ago: wait, then request (C, D) for the batch after this one
            swait(A, B);
            const int r1 = min(r + U, wend - U);  // the last trip may have no second batch: reload in bounds
            sload16(C, tset + (size_t)r1 * W32);
            sload16(D, tset + (size_t)(r1 + RPT) * W32);
            tuple_rows(A, jj);
            tuple_rows(B, jj + RPT);
            jj += U;
            if (jj == KNN_GROUP) {
                fold(gbase);
                gbase += KNN_GROUP;
                jj = 0;
            }
            swait(C, D);
            const int r2 = min(r + 2 * U, wend - U);
            sload16(A, tset + (size_t)r2 * W32);
            sload16(B, tset + (size_t)(r2 + RPT) * W32);
            if (b + 1 < nb) {
                tuple_rows(C, jj);
                tuple_rows(D, jj + RPT);
                jj += U;
                if (jj == KNN_GROUP) {
                    fold(gbase);
                    gbase += KNN_GROUP;
                    jj = 0;
                }
            }
            r += 2 * U;
        }
        swait(A, B);  // nothing of this wave's may stay in flight into the tail's own loads
        __builtin_amdgcn_s_setprio(0);
        r = wbeg + nb * U;
    }
    // the last rows (fewer than a batch), one scalar load each
#pragma unroll 1
    for (; r < wend; r++) {
        uint32_t w[W32];
        if (W32 == 8) {
            u32x8 t;
            sload8(t, tset + (size_t)r * W32);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t));
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = t[k];
        } else {
            u32x16 t;
            sload16(t, tset + (size_t)r * W32);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t));
#pragma unroll
            for (int k = 0; k < W32; k++) w[k] = t[k];
        }
        row(w, jj);
        if (++jj == KNN_GROUP) {
            fold(gbase);
            gbase += KNN_GROUP;
            jj = 0;
        }
    }
    fold(gbase);

    wres[wave][lane] = make_uint2(k1, k2);
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < 4; w++) {
        const uint2 p = wres[w][lane];
        k2 = med3_u32(p.x, k1, k2);
        k1 = min(p.x, k1);
        k2 = med3_u32(p.y, k1, k2);
        k1 = min(p.y, k1);
    }
    const int qi = q0 + lane;
    if (qi >= nq) return;
    const uint2 out = make_uint2(k1, k2);
    if (a.splits > 1)
        a.part[((size_t)bz * a.splits + by) * a.nq_cap + qi] = out;
    else if (rev)
        a.rev[(size_t)pair * a.nq_cap + qi] = out;
    else
        a.knn[(size_t)pair * a.nq_cap + qi] = out;
}

// ---- 32-byte descriptors on the matrix cores -----------------------------------------------------------
// Hamming distance as an exact integer dot product: with train bits t_k in {0, 1} and query bits mapped to
// s_k = 1 - 2 q_k in {+1, -1},  sum_k t_k s_k = popcount(t) - 2 <t, q>  =  hamming(q, t) - popcount(q).
// popcount(q) is a constant of the query, so the order of a query's candidates — distance first, lower
// train row on ties — is the order of that sum.  v_mfma_i32_32x32x32_i8 forms it for 32 train rows x 32
// queries x 32 bit positions per instruction, 8 instructions per 256-bit tile, in integer arithmetic:
// the result is the same number the popcount kernel produces, not an approximation of it.
//
// Operands (lane = 32 h + c).  The sum over k does not care which byte of which instruction carries which
// bit as long as A and B agree, nor how a product of 64 is split between its factors.  So a lane takes the
// four words 4h .. 4h+3 of its row, and word t feeds instructions 2t and 2t+1: operand dword j = 0..7 is
//   A (train row c):   w & (0x01010101 << j)            one v_and: bytes hold 0 or 2^j   (j = 7: (w >> 1) & 0x40404040)
//   B (query c):       +-(64 >> j) per byte, minus where the bit is set                   (j = 7: +-1)
// every product is 0 or +-64, the accumulator is 64 * (hamming - popcount(q)) + C.
// D: lane holds query c, register g <-> train row (g & 3) + 8 (g >> 2) + 4 h of the tile.  One lane sees 16
// rows of ONE query per tile, so the running top two stay lane-private as in k_knn2; the C operand of the
// first instruction is the row's position in its group of 64 rows ((g & 3) + 8 (g >> 2) + 32 * odd tile),
// which makes the accumulator itself the 16-bit key `value << 6 | position` (signed): three 16-bit min / max
// per candidate and nothing else.  The bit 4 h of the position is the lane's and is added when a group's two
// keys are folded into the 32-bit (distance << 22 | row) pair.
// The loop is software-pipelined over two accumulator sets: the 16 matrix instructions of one tile issue
// among the selection of the previous tile's 32 x 16 candidates and the 18 v_and of the next operands.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define KNN_MF_NONE ((short)0x7fff)  // no candidate: above every key (|value << 6| <= 16384)

// operand dwords 4u .. 4u+3 of a train word
static __device__ __forceinline__ i32x4 knn_mf_a(uint32_t w, int u) {
    i32x4 A;
    if (u == 0) {
        A[0] = (int)(w & 0x01010101u);
        A[1] = (int)(w & 0x02020202u);
        A[2] = (int)(w & 0x04040404u);
        A[3] = (int)(w & 0x08080808u);
    } else {
        A[0] = (int)(w & 0x10101010u);
        A[1] = (int)(w & 0x20202020u);
        A[2] = (int)(w & 0x40404040u);
        A[3] = (int)((w >> 1) & 0x40404040u);
    }
    return A;
}
// operand dword j of a query word: bytes +s where the bit is clear, -s where it is set, s = 64 >> min(j, 6)
static __device__ __forceinline__ int knn_mf_b(uint32_t w, int j) {
    const int p = j < 7 ? j : 6;
    const uint32_t t = ((j < 7 ? w : w >> 1) >> p) & 0x01010101u;
    const uint32_t s4 = (64u >> p) * 0x01010101u;
    return (int)((s4 ^ ((t << 8) - t)) + t);  // two's complement per byte; s >= 1, so no carry leaves a byte
}

// Same grid, arguments and results as k_knn2<8>.  A block owns 64 queries (two 32-query operand tiles held
// in registers by every wave), its four waves each scan a quarter of the train rows in tiles of 32.
__global__ __launch_bounds__(256) void k_knn2_mfma(KnnArgs a) {
    __shared__ uint2 wres[4][2][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;
    int bx, by, bz;
    knn_block(bx, by, bz);
    const bool rev = bz >= a.npairs;
    const int pair = rev ? bz - a.npairs : bz;
    int nq = a.d_n ? a.d_n[pair] : a.nq;
    int nt = a.d_n ? a.d_n[pair + 1] : a.nt;
    const uint32_t* qset = a.q + pair * a.q_stride;
    const uint32_t* tset = a.t + pair * a.t_stride;
    if (rev) {
        const int n = nq;
        nq = nt;
        nt = n;
        const uint32_t* s = qset;
        qset = tset;
        tset = s;
    }
    const int q0 = bx * 64;
    if (q0 >= nq) return;

    // query operands and the query's own popcount (the other half of the row is in lane ^ 32)
    i32x4 B[2][8];
    int pq[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const uint4 u = reinterpret_cast<const uint4*>(qset)[(size_t)min(q0 + 32 * n + c, nq - 1) * 2 + h];
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        int p = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            p += __popc(w[t]);
#pragma unroll
            for (int j = 0; j < 8; j++) B[n][2 * t + (j >> 2)][j & 3] = knn_mf_b(w[t], j);
        }
        pq[n] = p + __shfl_xor(p, 32);
    }

    // rows of this split, then of this wave (whole groups of 64 rows for all waves but the last)
    const int per_split = cdiv(nt, a.splits);
    const int sbeg = min(nt, by * per_split), send = min(nt, sbeg + per_split);
    const int per_wave = (cdiv(send - sbeg, 4) + KNN_GROUP - 1) & ~(KNN_GROUP - 1);
    const int wbeg = __builtin_amdgcn_readfirstlane(min(send, sbeg + wave * per_wave));
    const int wend = __builtin_amdgcn_readfirstlane(min(send, wbeg + per_wave));

    uint32_t k1[2] = {VO_KEY_INVALID, VO_KEY_INVALID}, k2[2] = {VO_KEY_INVALID, VO_KEY_INVALID};
    short l1[2] = {KNN_MF_NONE, KNN_MF_NONE}, l2[2] = {KNN_MF_NONE, KNN_MF_NONE};
    i32x16 C0, C1;  // positions of the 16 rows a lane sees in the even / odd tile of a group (without 4 h)
#pragma unroll
    for (int g = 0; g < 16; g++) {
        C0[g] = (g & 3) + 8 * (g >> 2);
        C1[g] = C0[g] + 32;
    }
    // keep the constants in registers: as immediates they would be sixteen v_mov per tile
    asm volatile("" : "+v"(C0), "+v"(C1));
    const uint4* T4 = reinterpret_cast<const uint4*>(tset);

    auto chain = [&](const uint4& rw, const i32x16& cinit, i32x16& a0, i32x16& a1) {
        const uint32_t w[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const i32x4 A = knn_mf_a(w[s >> 1], s & 1);
            a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[0][s], s == 0 ? cinit : a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[1][s], s == 0 ? cinit : a1, 0, 0, 0);
        }
    };
    auto take = [&](int n, short key) {
        const short hi = key > l1[n] ? key : l1[n];
        l1[n] = key < l1[n] ? key : l1[n];
        l2[n] = hi < l2[n] ? hi : l2[n];
        // two-operand 16-bit min / max issue in ~2.3 cycles, the three-operand forms the compiler would merge
        // successive updates into (v_min3_i16) in ~8 (tools/micro/mfma_valu.hip): keep the updates apart
        asm("" : "+v"(l1[n]), "+v"(l2[n]));
    };
    auto select = [&](const i32x16& a0, const i32x16& a1) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            take(0, (short)a0[g]);
            take(1, (short)a1[g]);
        }
    };
    // rows beyond `valid` (the tile's last rows past the wave's range) do not compete
    auto select_masked = [&](const i32x16& a0, const i32x16& a1, int valid) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const bool in = (g & 3) + 8 * (g >> 2) + 4 * h < valid;
            take(0, in ? (short)a0[g] : KNN_MF_NONE);
            take(1, in ? (short)a1[g] : KNN_MF_NONE);
        }
    };
    // fold the group's two keys of either query into the global pairs and start a new group
    auto fold = [&](int gbase) {
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const short l = j == 0 ? l1[n] : l2[n];
                const int val = (int)l >> 6;  // hamming - popcount(query)
                const uint32_t gk =
                    ((uint32_t)(val + pq[n]) << VO_KEY_SHIFT) | (uint32_t)(gbase + ((int)l & 63) + 4 * h);
                const uint32_t e = l == KNN_MF_NONE ? VO_KEY_INVALID : gk;
                k2[n] = med3_u32(e, k1[n], k2[n]);
                k1[n] = min(e, k1[n]);
            }
            l1[n] = l2[n] = KNN_MF_NONE;
        }
    };
    auto load_rows = [&](int r) { return T4[(size_t)min(r + c, wend - 1) * 2 + h]; };

    int r = wbeg;
    const int ngroups = (wend - wbeg) / KNN_GROUP;
    if (ngroups > 0) {
        i32x16 aE0, aE1, aO0, aO1;
#pragma unroll
        for (int g = 0; g < 16; g++) aO0[g] = aO1[g] = 0x7fff;  // nothing to select on the first trip
        uint4 we = load_rows(r), wo = load_rows(r + 32);
#pragma unroll 1
        for (int g = 0; g < ngroups; g++) {
            const int rn = min(r + KNN_GROUP, wend - KNN_GROUP);  // last trip: reload in bounds, unused
            const uint4 ne = load_rows(rn), no = load_rows(rn + 32);
            chain(we, C0, aE0, aE1);  // even tile of this group ...
            select(aO0, aO1);         // ... among the selection of the previous group's odd tile
            fold(r - KNN_GROUP);      // previous group complete (first trip: no keys, nothing changes)
            chain(wo, C1, aO0, aO1);
            select(aE0, aE1);
#pragma unroll
            for (int i = 0; i < 32; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one matrix instruction
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);  // its share of the vector instructions
            }
            we = ne;
            wo = no;
            r += KNN_GROUP;
        }
        select(aO0, aO1);
        fold(r - KNN_GROUP);
    }
    // the last rows (fewer than a group): one tile at a time
#pragma unroll 1
    for (; r < wend; r += 32) {
        i32x16 a0, a1;
        chain(load_rows(r), C0, a0, a1);
        select_masked(a0, a1, wend - r);
        fold(r);
    }

    wres[wave][0][lane] = make_uint2(k1[0], k2[0]);
    wres[wave][1][lane] = make_uint2(k1[1], k2[1]);
    __syncthreads();
    if (wave != 0) return;
    uint32_t m1 = VO_KEY_INVALID, m2 = VO_KEY_INVALID;
#pragma unroll
    for (int w = 0; w < 4; w++) {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const uint2 p = wres[w][h][c + 32 * hh];  // this lane's query: operand tile h, column c
            m2 = med3_u32(p.x, m1, m2);
            m1 = min(p.x, m1);
            m2 = med3_u32(p.y, m1, m2);
            m1 = min(p.y, m1);
        }
    }
    const int qi = q0 + lane;
    if (qi >= nq) return;
    const uint2 out = make_uint2(m1, m2);
    if (a.splits > 1)
        a.part[((size_t)bz * a.splits + by) * a.nq_cap + qi] = out;
    else if (rev)
        a.rev[(size_t)pair * a.nq_cap + qi] = out;
    else
        a.knn[(size_t)pair * a.nq_cap + qi] = out;
}

// splits > 1 only: grid y = pair (+ npairs for the reverse direction)
__global__ __launch_bounds__(256) void k_knn2_merge(KnnArgs a) {
    const bool rev = (int)blockIdx.y >= a.npairs;
    const int pair = rev ? blockIdx.y - a.npairs : blockIdx.y;
    const int nq = a.d_n ? a.d_n[pair + (rev ? 1 : 0)] : (rev ? a.nt : a.nq);
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t k1 = VO_KEY_INVALID, k2 = VO_KEY_INVALID;
    for (int s = 0; s < a.splits; s++) {
        const uint2 p = a.part[((size_t)blockIdx.y * a.splits + s) * a.nq_cap + qi];
        k2 = med3_u32(p.x, k1, k2);
        k1 = min(p.x, k1);
        k2 = med3_u32(p.y, k1, k2);
        k1 = min(p.y, k1);
    }
    (rev ? a.rev : a.knn)[(size_t)pair * a.nq_cap + qi] = make_uint2(k1, k2);
}

void launch_knn2(hipStream_t st, const KnnArgs& a) {
    const int nmax = a.cross ? (a.nq > a.nt ? a.nq : a.nt) : a.nq;
    if (nmax <= 0 || a.npairs <= 0) return;
    const int dirs = a.cross ? 2 : 1;
    dim3 grid(cdiv(nmax, 64), a.splits, a.npairs * dirs);
    // 32-byte rows go to the matrix-core kernel; VO_OPT_KNN_MFMA = 0 keeps them on the popcount kernel, which
    // 64-byte rows always use
    if (a.words == 8 && a.mfma) hipLaunchKernelGGL(k_knn2_mfma, grid, dim3(256), 0, st, a);
    else if (a.words == 8) hipLaunchKernelGGL(k_knn2<8>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_knn2<16>, grid, dim3(256), 0, st, a);
}

void launch_knn2_merge(hipStream_t st, const KnnArgs& a) {
    const int nmax = a.cross ? (a.nq > a.nt ? a.nq : a.nt) : a.nq;
    if (nmax <= 0 || a.npairs <= 0 || a.splits <= 1) return;
    hipLaunchKernelGGL(k_knn2_merge, dim3(cdiv(nmax, 256), a.npairs * (a.cross ? 2 : 1)), dim3(256), 0, st, a);
}

// One block per pair; queries processed in chunks of 1024 in index order.
__global__ __launch_bounds__(1024) void k_match_filter(FilterArgs a) {
    __shared__ int wave_cnt[16];
    __shared__ int running;
    const int pair = blockIdx.x;
    const int nq = a.d_n ? a.d_n[pair] : a.nq;
    const uint2* knn = a.knn + (size_t)pair * a.nq_cap;
    const uint2* rev = a.rev ? a.rev + (size_t)pair * a.nq_cap : nullptr;
    const vo_keypoint* kq = a.kq + pair * a.kp_stride;
    const vo_keypoint* kt = a.kt + pair * a.kp_stride;
    GoodMatch* good = a.good + (size_t)pair * a.nq_cap;
    float2* p0 = a.p0 + (size_t)pair * a.nq_cap;
    float2* p1 = a.p1 + (size_t)pair * a.nq_cap;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024) {
        const int qi = q0 + threadIdx.x;
        bool keep = false;
        GoodMatch gm = {0, 0, 0, 0.f};
        float2 a0 = make_float2(0.f, 0.f), a1 = a0;
        if (qi < nq) {
            const uint2 k = knn[qi];
            // rows with fewer than two neighbours: undefined behaviour in the reference, skipped
            if (k.x != VO_KEY_INVALID && k.y != VO_KEY_INVALID) {
                const float d0 = (float)(k.x >> VO_KEY_SHIFT), d1 = (float)(k.y >> VO_KEY_SHIFT);
                if (d0 < a.ratio * d1) {
                    const int ti = (int)(k.x & KNN_IDX_MASK);
                    // cross-check (off in the reference, matcher.cpp:10): the train row's own nearest
                    // query, under the same tie rule, must be this query
                    const bool mutual = !rev || (int)(rev[ti].x & KNN_IDX_MASK) == qi;
                    a0 = make_float2(kq[qi].x, kq[qi].y);
                    a1 = make_float2(kt[ti].x, kt[ti].y);
                    const float dx = a0.x - a1.x, dy = a0.y - a1.y;
                    const double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
                    if (mutual && nrm < (double)a.max_disp) {
                        keep = true;
                        gm.queryIdx = qi;
                        gm.trainIdx = ti;
                        gm.distance = d0;
                    }
                }
            }
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int before = running;
        for (int i = 0; i < wv; i++) before += wave_cnt[i];
        if (keep) {
            const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
            good[pos] = gm;
            p0[pos] = a0;
            p1[pos] = a1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = running;
            for (int i = 0; i < 16; i++) s += wave_cnt[i];
            running = s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.ngood[pair] = running;
}

void launch_match_filter(hipStream_t st, const FilterArgs& a) {
    if (a.npairs <= 0) return;
    hipLaunchKernelGGL(k_match_filter, dim3(a.npairs), dim3(1024), 0, st, a);
}

// ---------------------------------------------------------------------------------------------
// vo_knn2_hamming_static: the 2-NN of one descriptor pair for callers that cannot allocate (HIP graph
// capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
// Device pointers in and out; the per-split partial keys and the merged keys, which the ordinary entry
// point keeps in the context's grow-only workspace, live in the file-scope array below, so the call
// makes no hipMalloc / hipFree and no allocating library call.  Every GPU holds its own copy of the array
// at its own address: knn_static_init() (called from vo_ctx_create with that device current) asks for it
// once per device with hipGetSymbolAddress; host code only ever uses that pointer, never the array's name.
// All calls on one GPU share the array: calls that overlap on one GPU race (one context, or calls
// serialised by the caller).  Nothing needs clearing between calls: every word a call reads was written
// by that call (k_knn2 writes all nq rows of every split, also for an empty train range).
// ---------------------------------------------------------------------------------------------
#define KNN_STATIC_MAX_SPLITS 12
static __device__ __align__(256) uint2 scratch[(size_t)(KNN_STATIC_MAX_SPLITS + 1) * VO_KNN_STATIC_MAX];
#define KNN_STATIC_MAX_DEVICES 64
static uint2* g_scratch_ptr[KNN_STATIC_MAX_DEVICES];  // per device: address of that device's copy

int knn_static_init(int device) {  // the caller has made `device` current
    if (device < 0 || device >= KNN_STATIC_MAX_DEVICES) return VO_ERR_INVALID;
    if (g_scratch_ptr[device]) return VO_OK;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return VO_ERR_DEVICE;
    g_scratch_ptr[device] = static_cast<uint2*>(p);
    return VO_OK;
}

__global__ __launch_bounds__(256) void k_keys_to_dmatch(const uint2* keys, int nq, vo_dmatch* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const uint2 k = keys[i];
    const uint32_t key[2] = {k.x, k.y};
#pragma unroll
    for (int j = 0; j < 2; j++) {
        vo_dmatch m;
        m.queryIdx = i;
        m.imgIdx = 0;
        m.trainIdx = key[j] != VO_KEY_INVALID ? (int)(key[j] & KNN_IDX_MASK) : -1;
        m.distance = key[j] != VO_KEY_INVALID ? (float)(key[j] >> VO_KEY_SHIFT) : 3.402823466e+38F;
        out[(size_t)i * 2 + j] = m;
    }
}

int knn_splits(int nq, int nt, int npairs, int dirs, bool mfma);  // vo_match.cpp

extern "C" int vo_knn2_hamming_static(vo_ctx* c, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                                      int desc_bytes, vo_dmatch* d_out) {
    if (!c) return VO_ERR_INVALID;
    if (nq < 0 || nt < 0 || (desc_bytes != 32 && desc_bytes != 64) || (nq > 0 && (!d_q || !d_out)) ||
        (nt > 0 && !d_t) || ((uintptr_t)d_q & 15) || ((uintptr_t)d_t & 15) || ((uintptr_t)d_out & 3))
        return c->fail(VO_ERR_INVALID, "vo_knn2_hamming_static: bad argument (rows of exactly 32 or 64 bytes, "
                                       "16-byte aligned device pointers)");
    if (nq > VO_KNN_STATIC_MAX || nt > VO_KNN_STATIC_MAX)
        return c->fail(VO_ERR_CAPACITY, "vo_knn2_hamming_static: %d x %d exceeds the static workspace (%d rows a side)",
                       nq, nt, VO_KNN_STATIC_MAX);
    VO_HIP(c, hipSetDevice(c->device));
    uint2* ws = c->device < KNN_STATIC_MAX_DEVICES ? g_scratch_ptr[c->device] : nullptr;
    if (!ws) return c->fail(VO_ERR_DEVICE, "vo_knn2_hamming_static: no static workspace on device %d", c->device);
    if (nq == 0) return VO_OK;
    KnnArgs a;
    a.q = reinterpret_cast<const uint32_t*>(d_q);
    a.t = reinterpret_cast<const uint32_t*>(d_t);
    a.q_stride = a.t_stride = 0;
    a.d_n = nullptr;
    a.nq = nq;
    a.nt = nt;
    a.words = desc_bytes / 4;
    a.splits = knn_splits(nq, nt, 1, 1, desc_bytes == 32 && c->opt_knn_mfma);
    if (a.splits > KNN_STATIC_MAX_SPLITS) a.splits = KNN_STATIC_MAX_SPLITS;
    a.cross = 0;
    a.knn = ws;                         // [VO_KNN_STATIC_MAX] merged keys
    a.part = ws + VO_KNN_STATIC_MAX;    // [splits][VO_KNN_STATIC_MAX] partial keys
    a.rev = nullptr;
    a.nq_cap = VO_KNN_STATIC_MAX;
    a.npairs = 1;
    a.mfma = c->opt_knn_mfma;
    launch_knn2(c->stream, a);
    launch_knn2_merge(c->stream, a);
    hipLaunchKernelGGL(k_keys_to_dmatch, dim3(cdiv(nq, 256)), dim3(256), 0, c->stream, ws, nq, d_out);
    return vo_check_launch(c, "knn2 (static workspace)");
}
