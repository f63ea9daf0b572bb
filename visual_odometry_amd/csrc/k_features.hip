// k_features.hip — gfx950 kernels for FAST-9/16 and ORB (pyramid, FAST score, NMS + ordered
// compaction + Harris, IC angle, 7x7 blur, rBRIEF).  Integer pipeline throughout; the few float
// expressions (Harris epilogue, fastAtan2, pattern rotation) follow cv's operation order and
// the file is compiled with -ffp-contract=off so nothing is fused.
// Behaviour follows SURVEY.md A.1/A.2; reference call sites: src/features/feature_detector.cpp:4-5,15,24,28,41.
#include "kernels.h"

namespace {

__device__ __forceinline__ const uint8_t* level_base(const FrameSrc& s, const PyrGeom* g, int f, int l,
                                                     int& pitch) {
    if (l == 0) {
        pitch = s.stride;
        return s.img + (size_t)f * s.frame_stride;
    }
    pitch = g->lv[l].pitch;
    return s.pyr + (size_t)f * g->pyr_bytes + g->lv[l].offset;
}

__device__ __forceinline__ int find_level_by_tile(const PyrGeom* g, int tile, int& local) {
    int l = 0;
    for (int i = 1; i < g->nlevels; i++)
        if (tile >= g->tile_start[i]) l = i;
    local = tile - g->tile_start[l];
    return l;
}

__device__ __forceinline__ int find_level_by_row(const PyrGeom* g, int row) {
    int l = 0;
    for (int i = 1; i < g->nlevels; i++)
        if (row >= g->lv[i].row0) l = i;
    return l;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Pyramid level l from level l-1: cv::resize(INTER_LINEAR_EXACT), 8.8 fixed-point coefficients
// from host tables (xtab/ytab entries: ofs << 16 | c1, c0 = 256 - c1).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_resize(FrameSrc src, const PyrGeom* __restrict__ g,
                                                    uint8_t* __restrict__ pyr, int level,
                                                    const uint32_t* __restrict__ xtab,
                                                    const uint32_t* __restrict__ ytab) {
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dw = g->lv[level].w, dh = g->lv[level].h;
    if (x >= dw || y >= dh) return;
    const int sw = g->lv[level - 1].w, sh = g->lv[level - 1].h;
    int spitch;
    const uint8_t* s = level_base(src, g, f, level - 1, spitch);
    const uint32_t tx = xtab[x], ty = ytab[y];
    const int ox = tx >> 16, cx1 = tx & 0xffff, cx0 = 256 - cx1;
    const int oy = ty >> 16, cy1 = ty & 0xffff, cy0 = 256 - cy1;
    const int ox1 = min(ox + 1, sw - 1), oy1 = min(oy + 1, sh - 1);
    const uint8_t* r0 = s + (size_t)oy * spitch;
    const uint8_t* r1 = s + (size_t)oy1 * spitch;
    const uint32_t h0 = cx0 * r0[ox] + cx1 * r0[ox1];
    const uint32_t h1 = cx0 * r1[ox] + cx1 * r1[ox1];
    const uint32_t v = (cy0 * h0 + cy1 * h1 + 32768u) >> 16;
    pyr[(size_t)f * g->pyr_bytes + g->lv[level].offset + (size_t)y * g->lv[level].pitch + x] =
        (uint8_t)min(v, 255u);
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16 corner test + cornerScore<16> for every pixel of every level.  One block = a
// 64 x 16 pixel tile staged in LDS with a 3-pixel halo.  Output map: 0 = not a corner,
// score + 1 otherwise (score <= 254).
// ---------------------------------------------------------------------------------------------
#define FT_W 64
#define FT_H 16
__global__ __launch_bounds__(256) void k_fast_score(FrameSrc src, const PyrGeom* __restrict__ g,
                                                    uint8_t* __restrict__ score, int threshold) {
    __shared__ uint8_t tile[FT_H + 6][FT_W + 8];
    const int f = blockIdx.y;
    int local;
    const int l = find_level_by_tile(g, blockIdx.x, local);
    const int w = g->lv[l].w, h = g->lv[l].h;
    const int tiles_x = (w + FT_W - 1) / FT_W;
    const int tx0 = (local % tiles_x) * FT_W, ty0 = (local / tiles_x) * FT_H;
    int pitch;
    const uint8_t* img = level_base(src, g, f, l, pitch);

    for (int i = threadIdx.x; i < (FT_H + 6) * (FT_W + 6); i += 256) {
        const int r = i / (FT_W + 6), c = i - r * (FT_W + 6);
        const int gx = min(max(tx0 + c - 3, 0), w - 1), gy = min(max(ty0 + r - 3, 0), h - 1);
        tile[r][c] = img[(size_t)gy * pitch + gx];
    }
    __syncthreads();

    const int lx = threadIdx.x & 63;
    const int x = tx0 + lx;
    uint8_t* out = score + (size_t)f * g->pyr_bytes + g->lv[l].offset;
    const int opitch = g->lv[l].pitch;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ly = (threadIdx.x >> 6) * 4 + k;
        const int y = ty0 + ly;
        if (x >= w || y >= h) continue;
        int res = 0;
        if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
            const uint8_t* c = &tile[ly + 3][lx + 3];
            const int v = c[0];
            int ring[16];
            ring[0] = c[3 * (FT_W + 8) + 0];
            ring[1] = c[3 * (FT_W + 8) + 1];
            ring[2] = c[2 * (FT_W + 8) + 2];
            ring[3] = c[1 * (FT_W + 8) + 3];
            ring[4] = c[3];
            ring[5] = c[-1 * (FT_W + 8) + 3];
            ring[6] = c[-2 * (FT_W + 8) + 2];
            ring[7] = c[-3 * (FT_W + 8) + 1];
            ring[8] = c[-3 * (FT_W + 8) + 0];
            ring[9] = c[-3 * (FT_W + 8) - 1];
            ring[10] = c[-2 * (FT_W + 8) - 2];
            ring[11] = c[-1 * (FT_W + 8) - 3];
            ring[12] = c[-3];
            ring[13] = c[1 * (FT_W + 8) - 3];
            ring[14] = c[2 * (FT_W + 8) - 2];
            ring[15] = c[3 * (FT_W + 8) - 1];
            uint32_t dark = 0, bright = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                dark |= (uint32_t)(ring[i] < v - threshold) << i;
                bright |= (uint32_t)(ring[i] > v + threshold) << i;
            }
            // >= 9 contiguous set bits on the 16-ring
            uint32_t xd = dark | (dark << 16), xb = bright | (bright << 16);
            uint32_t rd = xd & (xd >> 1), rb = xb & (xb >> 1);
            rd &= rd >> 2; rb &= rb >> 2;
            rd &= rd >> 4; rb &= rb >> 4;
            rd &= xd >> 8; rb &= xb >> 8;
            if (rd | rb) {
                // cornerScore<16>: max(threshold, max over the 16 arcs of min(d), of min(-d)) - 1
                int d[16];
#pragma unroll
                for (int i = 0; i < 16; i++) d[i] = v - ring[i];
                int best = threshold;
                {
                    int m2[16], m4[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) m2[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
                    for (int i = 0; i < 16; i++) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        int m8 = min(m4[i], m4[(i + 4) & 15]);
                        best = max(best, min(m8, d[(i + 8) & 15]));
                    }
                }
                {
                    int m2[16], m4[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) m2[i] = max(d[i], d[(i + 1) & 15]);
#pragma unroll
                    for (int i = 0; i < 16; i++) m4[i] = max(m2[i], m2[(i + 2) & 15]);
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        int m8 = max(m4[i], m4[(i + 4) & 15]);
                        best = max(best, -max(m8, d[(i + 8) & 15]));
                    }
                }
                res = best - 1 + 1;
            }
        }
        out[(size_t)y * opitch + x] = (uint8_t)res;
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 strict-greater NMS + border filter on the score map; one wave per image row.
// WRITE=false: per-row keypoint counts.  WRITE=true: ordered (x ascending) compaction into the
// frame's candidate list at rowofs[row], with the Harris response (7x7 block, k=0.04f) when
// `harris` is set.  Row-major order is what cv's FAST emits.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float harris_response(const uint8_t* img, int pitch, int x0, int y0) {
    const int blockSize = 7, r = 3;
    const float harris_k = 0.04f;
    const float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const uint8_t* p00 = img + (size_t)(y0 - r) * pitch + (x0 - r);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; i++)
        for (int j = 0; j < blockSize; j++) {
            const uint8_t* p = p00 + i * pitch + j;
            int Ix = (p[1] - p[-1]) * 2 + (p[-pitch + 1] - p[-pitch - 1]) + (p[pitch + 1] - p[pitch - 1]);
            int Iy = (p[pitch] - p[-pitch]) * 2 + (p[pitch - 1] - p[-pitch - 1]) + (p[pitch + 1] - p[-pitch + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    return (fa * fb - fc * fc - harris_k * (fa + fb) * (fa + fb)) * scale_sq_sq;
}

template <bool WRITE>
__global__ __launch_bounds__(64) void k_nms_rows(FrameSrc src, const PyrGeom* __restrict__ g,
                                                 const uint8_t* __restrict__ score, int border, int nms,
                                                 int harris, int* __restrict__ rowcnt,
                                                 const int* __restrict__ rowofs, Cand* __restrict__ cand,
                                                 int max_cand) {
    const int f = blockIdx.y, row = blockIdx.x;
    const int lane = threadIdx.x;
    const int l = find_level_by_row(g, row);
    const int w = g->lv[l].w, h = g->lv[l].h, pitch = g->lv[l].pitch;
    const int y = row - g->lv[l].row0;
    const int b = max(border, 3);
    int total = 0;
    const bool level_ok = border == 0 || (w > 2 * border && h > 2 * border);
    if (level_ok && y >= b && y < h - b) {
        const uint8_t* S = score + (size_t)f * g->pyr_bytes + g->lv[l].offset + (size_t)y * pitch;
        int ipitch = 0;
        const uint8_t* img = nullptr;
        int base = 0;
        if (WRITE) {
            img = level_base(src, g, f, l, ipitch);
            base = rowofs[(size_t)f * (g->rows_total + 1) + row];
        }
        for (int x0 = 0; x0 < w; x0 += 64) {
            const int x = x0 + lane;
            bool kp = false;
            int s = 0;
            if (x >= b && x < w - b) {
                const int c = S[x];
                if (c) {
                    s = c - 1;
                    if (nms) {
                        auto sc = [&](int dy, int dx) {
                            int v = S[dy * pitch + x + dx];
                            return v ? v - 1 : 0;
                        };
                        kp = s > sc(0, -1) && s > sc(0, 1) && s > sc(-1, -1) && s > sc(-1, 0) &&
                             s > sc(-1, 1) && s > sc(1, -1) && s > sc(1, 0) && s > sc(1, 1);
                    } else {
                        kp = true;
                        s = 0;
                    }
                }
            }
            const unsigned long long m = __ballot(kp);
            if (WRITE && kp) {
                const int pos = base + total + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < max_cand) {
                    Cand cd;
                    cd.x = (int16_t)x;
                    cd.y = (int16_t)y;
                    cd.level = (int16_t)l;
                    cd.score = (int16_t)s;
                    cd.harris = 0.f;  // filled by k_harris (one thread per candidate)
                    cd.pad = 0;
                    cand[(size_t)f * max_cand + pos] = cd;
                }
            }
            total += __popcll(m);
        }
    }
    if (!WRITE && lane == 0) rowcnt[(size_t)f * g->rows_total + row] = total;
}

// Harris response of every candidate (ORB): one thread per candidate, dense over the list.
__global__ __launch_bounds__(256) void k_harris(FrameSrc src, const PyrGeom* __restrict__ g,
                                                const int* __restrict__ lvlofs, Cand* __restrict__ cand,
                                                int max_cand) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = min(lvlofs[(size_t)f * (VO_ORB_MAX_LEVELS + 1) + g->nlevels], max_cand);
    if (i >= total) return;
    Cand* c = cand + (size_t)f * max_cand + i;
    int pitch;
    const uint8_t* img = level_base(src, g, f, c->level, pitch);
    c->harris = harris_response(img, pitch, c->x, c->y);
}

// Exclusive scan of the per-row counts of one frame (one block per frame) -> rowofs[rows+1],
// plus the per-level start offsets lvlofs[nlevels+1].
__global__ __launch_bounds__(1024) void k_row_scan(const PyrGeom* __restrict__ g,
                                                   const int* __restrict__ rowcnt,
                                                   int* __restrict__ rowofs, int* __restrict__ lvlofs) {
    __shared__ int part[1024];
    const int f = blockIdx.x, t = threadIdx.x;
    const int rows = g->rows_total;
    const int per = (rows + 1023) / 1024;
    const int* cnt = rowcnt + (size_t)f * rows;
    int* ofs = rowofs + (size_t)f * (rows + 1);
    int sum = 0;
    for (int i = 0; i < per; i++) {
        int r = t * per + i;
        if (r < rows) sum += cnt[r];
    }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = 0; i < per; i++) {
        int r = t * per + i;
        if (r < rows) {
            ofs[r] = run;
            run += cnt[r];
        }
    }
    if (t == 1023) ofs[rows] = part[1023];
    __syncthreads();
    __threadfence_block();
    if (t <= g->nlevels) {
        const int r = t < g->nlevels ? g->lv[t].row0 : rows;
        lvlofs[(size_t)f * (VO_ORB_MAX_LEVELS + 1) + t] = ofs[r];
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

__global__ __launch_bounds__(256) void k_ic_angle(FrameSrc src, const PyrGeom* __restrict__ g,
                                                  const Cand* __restrict__ sel, const int* __restrict__ nkp,
                                                  int kp_cap, vo_keypoint* __restrict__ out) {
    const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    const int f = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= nkp[f]) return;
    const Cand cd = sel[(size_t)f * kp_cap + k];
    int pitch;
    const uint8_t* img = level_base(src, g, f, cd.level, pitch);
    const uint8_t* center = img + (size_t)cd.y * pitch + cd.x;
    int m01 = 0, m10 = 0;
    for (int idx = lane; idx < 31 * 31; idx += 64) {
        const int v = idx / 31 - 15, u = idx % 31 - 15;
        const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
        if (au <= umax[av]) {
            const int val = center[v * pitch + u];
            m10 += u * val;
            m01 += v * val;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    if (lane == 0) {
        const float s = g->lv[cd.level].scale;
        vo_keypoint kp;
        kp.x = (float)cd.x * s;
        kp.y = (float)cd.y * s;
        kp.size = 31 * s;
        kp.angle = fast_atan2_deg((float)m01, (float)m10);
        kp.response = cd.harris;
        kp.octave = cd.level;
        kp.class_id = -1;
        out[(size_t)f * kp_cap + k] = kp;
    }
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur 7x7 sigma 2 (cv's 8-bit fixed-point separable path, BORDER_REFLECT_101):
// row pass in 32-bit ints (fits 16), column pass (sum + 2^15) >> 16, saturate.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gauss7(FrameSrc src, const PyrGeom* __restrict__ g,
                                                uint8_t* __restrict__ blur, Taps7 taps, int nlevels_used) {
    __shared__ uint8_t raw[FT_H + 6][FT_W + 8];
    __shared__ uint16_t hz[FT_H + 6][FT_W];
    const int f = blockIdx.y;
    int local;
    const int l = find_level_by_tile(g, blockIdx.x, local);
    if (l >= nlevels_used) return;
    const int w = g->lv[l].w, h = g->lv[l].h;
    const int tiles_x = (w + FT_W - 1) / FT_W;
    const int tx0 = (local % tiles_x) * FT_W, ty0 = (local / tiles_x) * FT_H;
    int pitch;
    const uint8_t* img = level_base(src, g, f, l, pitch);
    for (int i = threadIdx.x; i < (FT_H + 6) * (FT_W + 6); i += 256) {
        const int r = i / (FT_W + 6), c = i - r * (FT_W + 6);
        const int gx = reflect101(tx0 + c - 3, w), gy = reflect101(ty0 + r - 3, h);
        raw[r][c] = img[(size_t)gy * pitch + gx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (FT_H + 6) * FT_W; i += 256) {
        const int r = i / FT_W, c = i - r * FT_W;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += taps.t[k] * raw[r][c + k];
        hz[r][c] = (uint16_t)acc;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, x = tx0 + lx;
    uint8_t* out = blur + (size_t)f * g->pyr_bytes + g->lv[l].offset;
    const int opitch = g->lv[l].pitch;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ly = (threadIdx.x >> 6) * 4 + k, y = ty0 + ly;
        if (x >= w || y >= h) continue;
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) acc += taps.t[j] * (int)hz[ly + j][lx];
        const int v = (acc + (1 << 15)) >> 16;
        out[(size_t)y * opitch + x] = (uint8_t)min(max(v, 0), 255);
    }
}

// ---------------------------------------------------------------------------------------------
// rBRIEF-256 (WTA_K = 2): 32 lanes per keypoint, lane b computes descriptor byte b (16 rotated
// pattern points).  cos/sin are taken in double and rounded to float, as cv does.
// ---------------------------------------------------------------------------------------------
__constant__ int8_t c_orb_pattern[1024] = {
#include "orb_pattern.inc"
};

__global__ __launch_bounds__(256) void k_rbrief(const PyrGeom* __restrict__ g, const uint8_t* __restrict__ blur,
                                                const vo_keypoint* __restrict__ kps, const int* __restrict__ nkp,
                                                int kp_cap, uint8_t* __restrict__ desc) {
    __shared__ int8_t pat[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) pat[i] = c_orb_pattern[i];
    __syncthreads();
    const int f = blockIdx.y;
    const int k = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int b = threadIdx.x & 31;
    if (k >= nkp[f]) return;
    const vo_keypoint kp = kps[(size_t)f * kp_cap + k];
    const int l = kp.octave;
    const int pitch = g->lv[l].pitch;
    const float scale = 1.f / g->lv[l].scale;
    float angle = kp.angle;
    angle *= (float)(3.14159265358979323846 / 180.f);
    const float ca = (float)cos((double)angle), sa = (float)sin((double)angle);
    const int cy = __float2int_rn(kp.y * scale), cx = __float2int_rn(kp.x * scale);
    const uint8_t* center = blur + (size_t)f * g->pyr_bytes + g->lv[l].offset + (size_t)cy * pitch + cx;
    const int8_t* p = pat + b * 32;
    int val = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        int v[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float px = (float)p[(2 * t + e) * 2], py = (float)p[(2 * t + e) * 2 + 1];
            const float x = px * ca - py * sa;
            const float y = px * sa + py * ca;
            v[e] = center[__float2int_rn(y) * pitch + __float2int_rn(x)];
        }
        val |= (v[0] < v[1]) << t;
    }
    desc[((size_t)f * kp_cap + k) * 32 + b] = (uint8_t)val;
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
void launch_pyr_resize(hipStream_t st, const FrameSrc& src, const PyrGeom* d_geom, const PyrGeom& hg,
                       uint8_t* pyr, int level, const uint32_t* xtab, const uint32_t* ytab, int nframes) {
    dim3 grid(cdiv(hg.lv[level].w, 64), cdiv(hg.lv[level].h, 4), nframes);
    hipLaunchKernelGGL(k_pyr_resize, grid, dim3(256), 0, st, src, d_geom, pyr, level, xtab, ytab);
}

void launch_fast_score(hipStream_t st, const FrameSrc& src, const PyrGeom* d_geom, const PyrGeom& hg,
                       uint8_t* score, int threshold, int nframes) {
    dim3 grid(hg.tiles_total, nframes);
    hipLaunchKernelGGL(k_fast_score, grid, dim3(256), 0, st, src, d_geom, score, threshold);
}

void launch_nms_count(hipStream_t st, const FrameSrc& src, const PyrGeom* d_geom, const PyrGeom& hg,
                      const uint8_t* score, int border, int nms, int* rowcnt, int nframes) {
    dim3 grid(hg.rows_total, nframes);
    hipLaunchKernelGGL(k_nms_rows<false>, grid, dim3(64), 0, st, src, d_geom, score, border, nms, 0,
                       rowcnt, (const int*)nullptr, (Cand*)nullptr, 0);
}

void launch_row_scan(hipStream_t st, const PyrGeom* d_geom, const int* rowcnt, int* rowofs, int* lvlofs,
                     int nframes) {
    hipLaunchKernelGGL(k_row_scan, dim3(nframes), dim3(1024), 0, st, d_geom, rowcnt, rowofs, lvlofs);
}

void launch_nms_write(hipStream_t st, const FrameSrc& src, const PyrGeom* d_geom, const PyrGeom& hg,
                      const uint8_t* score, int border, int nms, int harris, const int* rowofs,
                      Cand* cand, int max_cand, int nframes) {
    dim3 grid(hg.rows_total, nframes);
    hipLaunchKernelGGL(k_nms_rows<true>, grid, dim3(64), 0, st, src, d_geom, score, border, nms, harris,
                       (int*)nullptr, rowofs, cand, max_cand);
}

void launch_harris(hipStream_t st, const FrameSrc& src, const PyrGeom* d_geom, const int* lvlofs, Cand* cand,
                   int max_cand, int grid_cand, int nframes) {
    dim3 grid(cdiv(grid_cand, 256), nframes);
    hipLaunchKernelGGL(k_harris, grid, dim3(256), 0, st, src, d_geom, lvlofs, cand, max_cand);
}

void launch_ic_angle(hipStream_t st, const FrameSrc& src, const PyrGeom* d_geom, const Cand* sel,
                     const int* nkp, int kp_cap, int max_n, vo_keypoint* out, int nframes) {
    if (max_n <= 0) return;
    dim3 grid(cdiv(max_n, 4), nframes);
    hipLaunchKernelGGL(k_ic_angle, grid, dim3(256), 0, st, src, d_geom, sel, nkp, kp_cap, out);
}

void launch_gauss7(hipStream_t st, const FrameSrc& src, const PyrGeom* d_geom, const PyrGeom& hg,
                   uint8_t* blur, const Taps7& taps, int nlevels_used, int nframes) {
    int tiles = nlevels_used >= hg.nlevels ? hg.tiles_total : hg.tile_start[nlevels_used];
    dim3 grid(tiles, nframes);
    hipLaunchKernelGGL(k_gauss7, grid, dim3(256), 0, st, src, d_geom, blur, taps, nlevels_used);
}

void launch_rbrief(hipStream_t st, const PyrGeom* d_geom, const uint8_t* blur, const vo_keypoint* kps,
                   const int* nkp, int kp_cap, int max_n, uint8_t* desc, int nframes) {
    if (max_n <= 0) return;
    dim3 grid(cdiv(max_n, 8), nframes);
    hipLaunchKernelGGL(k_rbrief, grid, dim3(256), 0, st, d_geom, blur, kps, nkp, kp_cap, desc);
}
