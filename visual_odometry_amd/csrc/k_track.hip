// k_track.hip — gfx950 kernels for cv::SparsePyrLKOpticalFlow::calc (pyramidal Lucas-Kanade,
// cv 3.4 video/src/lkpyramid.cpp) as FeatureTracker::trackPoints calls it
// (reference: src/features/feature_tracker.cpp:6,13).
//
//   k_lk_level0    caller's image -> level 0 of the tracker pyramid, with a reflect-101 frame
//   k_lk_pyrdown   level l from level l-1: cv::pyrDown (5x5 binomial, (sum + 128) >> 8), frame included
//   k_lk_scharr    calcSharrDeriv: (dx, dy) shorts per pixel; the frame of the derivative image is
//                  zero (copyMakeBorder BORDER_CONSTANT) and is never written after allocation
//   k_lk_track     one wavefront per point, all levels in one launch: 14-bit fixed-point bilinear
//                  window of I and its derivatives into LDS (shorts), exact integer sums of the
//                  products reduced across the wave (the sums do not depend on the order, so the
//                  result equals the sequential CPU loop bit for bit), then <= 30 iterations that
//                  resample J, all in cv's float expression order (-ffp-contract=off)
// Every level lives in one packed buffer per image with a frame of TK_PAD pixels on all sides
// (>= the window, so a window that hangs over the edge reads real memory, as cv's bordered
// pyramid does).
#include "kernels.h"

namespace {

__device__ __forceinline__ int tk_reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__device__ __forceinline__ const uint8_t* tk_px(const uint8_t* buf, const TrackLevel& L) {
    return buf + L.offset + (size_t)TK_PAD * L.pitch + TK_PAD;
}
__device__ __forceinline__ uint8_t* tk_px(uint8_t* buf, const TrackLevel& L) {
    return buf + L.offset + (size_t)TK_PAD * L.pitch + TK_PAD;
}

__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

}  // namespace

// one thread per aligned dword of the padded level
// blockIdx.z: image of a batch (images img_stride apart, pyramids g.bytes apart)
__global__ __launch_bounds__(256) void k_lk_level0(const uint8_t* __restrict__ img, size_t img_stride, int stride,
                                                   TrackGeom g, uint8_t* __restrict__ pyr) {
    img += (size_t)blockIdx.z * img_stride;
    pyr += (size_t)blockIdx.z * g.bytes;
    const TrackLevel L = g.lv[0];
    const int x4 = -TK_PAD + 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = -TK_PAD + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= L.w + TK_PAD || y >= L.h + TK_PAD) return;
    const uint8_t* src = img + (size_t)tk_reflect101(y, L.h) * stride;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) v |= (uint32_t)src[tk_reflect101(x4 + j, L.w)] << (8 * j);
    *reinterpret_cast<uint32_t*>(tk_px(pyr, L) + (ptrdiff_t)y * L.pitch + x4) = v;
}

// The source level carries its reflect-101 frame, so source indices 2x-2 .. 2x+2 need no border
// logic; frame pixels of the new level are the values at the reflected destination coordinate.
__global__ __launch_bounds__(256) void k_lk_pyrdown(TrackGeom g, uint8_t* __restrict__ pyr, int level) {
    pyr += (size_t)blockIdx.z * g.bytes;
    const TrackLevel D = g.lv[level], S = g.lv[level - 1];
    const int x4 = -TK_PAD + 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = -TK_PAD + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= D.w + TK_PAD || y >= D.h + TK_PAD) return;
    const uint8_t* s = tk_px((const uint8_t*)pyr, S);
    const int ry = tk_reflect101(y, D.h);
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int rx = tk_reflect101(x4 + j, D.w);
        int sum = 0;
#pragma unroll
        for (int ky = 0; ky < 5; ky++) {
            const uint8_t* r = s + (ptrdiff_t)(2 * ry + ky - 2) * S.pitch + 2 * rx;
            const int row = r[-2] + r[2] + 4 * (r[-1] + r[1]) + 6 * r[0];
            sum += (ky == 0 || ky == 4 ? 1 : (ky == 2 ? 6 : 4)) * row;
        }
        out |= (uint32_t)((sum + 128) >> 8) << (8 * j);
    }
    *reinterpret_cast<uint32_t*>(tk_px(pyr, D) + (ptrdiff_t)y * D.pitch + x4) = out;
}

// interior pixels only; reads the reflect-101 frame of the image for the edge rows / columns
// blockIdx.z = image * nlevels + level: every level of every image in one launch (the grid covers level 0;
// the blocks of a smaller level that lie outside it leave at once)
__global__ __launch_bounds__(256) void k_lk_scharr(TrackGeom g, const uint8_t* __restrict__ pyr,
                                                   short2* __restrict__ deriv) {
    const int im = blockIdx.z / g.nlevels, level = blockIdx.z - im * g.nlevels;
    pyr += (size_t)im * g.bytes;
    deriv += (size_t)im * g.bytes;
    const TrackLevel L = g.lv[level];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= L.w || y >= L.h) return;
    const uint8_t* p = tk_px(pyr, L) + (ptrdiff_t)y * L.pitch + x;
    const int pitch = L.pitch;
    int t0[3], t1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int a = p[-pitch + k - 1], b = p[k - 1], c = p[pitch + k - 1];
        t0[k] = (a + c) * 3 + b * 10;
        t1[k] = c - a;
    }
    const size_t o = L.offset + (size_t)(y + TK_PAD) * L.pitch + TK_PAD + x;
    deriv[o] = make_short2((short)(t0[2] - t0[0]), (short)((t1[2] + t1[0]) * 3 + t1[1] * 10));
}

#define TK_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

// The tracker kernels take a batch of point sets: blockIdx.y = image pair of the batch, whose pyramids and
// derivative image lie g.bytes apart and whose points lie b.pts_stride apart; b.n[pair] points each.
#define LK_BATCH_PROLOGUE                                          \
    const int bi = blockIdx.y, n = b.n[bi];                        \
    pyrI += (size_t)bi * g.bytes;                                  \
    pyrJ += (size_t)bi * g.bytes;                                  \
    deriv += (size_t)bi * g.bytes;                                 \
    prev_pts += (size_t)bi * b.pts_stride;                         \
    next_pts += (size_t)bi * b.pts_stride;                         \
    status += (size_t)bi * b.pts_stride;

__global__ __launch_bounds__(256) void k_lk_track(TrackGeom g, const uint8_t* __restrict__ pyrI,
                                                  const short2* __restrict__ deriv,
                                                  const uint8_t* __restrict__ pyrJ,
                                                  const float2* __restrict__ prev_pts, float2* __restrict__ next_pts,
                                                  uint8_t* __restrict__ status, LkBatch b, LkArgs a) {
    LK_BATCH_PROLOGUE
    extern __shared__ int16_t lk_lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wv;
    if (pt >= n) return;  // a whole wave leaves; the kernel has no block-wide barrier
    const int ww = a.win_w, wh = a.win_h, area = ww * wh;
    int16_t* Iw = lk_lds + (size_t)wv * area * 3;
    int16_t* Dx = Iw + area;
    int16_t* Dy = Dx + area;
    const float2 p = prev_pts[pt];
    const float half_x = (ww - 1) * 0.5f, half_y = (wh - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const long long cn = a.channels;
    float2 np = make_float2(0.f, 0.f);
    bool ok = true;
    for (int level = g.nlevels - 1; level >= 0; level--) {
        const TrackLevel L = g.lv[level];
        const float sc = __int_as_float((127 - level) << 23);  // (float)(1. / (1 << level))
        float px = p.x * sc, py = p.y * sc;
        if (level == g.nlevels - 1) np = make_float2(px, py);
        else np = make_float2(np.x * 2.f, np.y * 2.f);
        px -= half_x;
        py -= half_y;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -ww || ipx >= L.w || ipy < -wh || ipy >= L.h) {
            if (level == 0) ok = false;
            continue;
        }
        float fa = px - ipx, fb = py - ipy;
        int iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
        int iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
        int iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        const int pitch = L.pitch;
        {
            const uint8_t* I0 = tk_px(pyrI, L) + (ptrdiff_t)ipy * pitch + ipx;
            const short2* D0 = deriv + L.offset + (size_t)(ipy + TK_PAD) * pitch + TK_PAD + ipx;
            int sA11 = 0, sA12 = 0, sA22 = 0;  // per lane: <= 16 terms below 2^25
            for (int i = lane; i < area; i += 64) {
                const int y = i / ww, x = i - y * ww;
                const uint8_t* s = I0 + y * pitch + x;
                const short2* d = D0 + y * pitch + x;
                const int ival = TK_DESCALE(s[0] * iw00 + s[1] * iw01 + s[pitch] * iw10 + s[pitch + 1] * iw11, W_BITS - 5);
                const short2 d00 = d[0], d01 = d[1], d10 = d[pitch], d11 = d[pitch + 1];
                const int ixval = TK_DESCALE(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, W_BITS);
                const int iyval = TK_DESCALE(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, W_BITS);
                Iw[i] = (int16_t)ival;
                Dx[i] = (int16_t)ixval;
                Dy[i] = (int16_t)iyval;
                sA11 += ixval * ixval;
                sA12 += ixval * iyval;
                sA22 += iyval * iyval;
            }
            const long long iA11 = wave_sum((long long)sA11), iA12 = wave_sum((long long)sA12),
                            iA22 = wave_sum((long long)sA22);
            const float A11 = __ll2float_rn(iA11 * cn) * FLT_SCALE, A12 = __ll2float_rn(iA12 * cn) * FLT_SCALE,
                        A22 = __ll2float_rn(iA22 * cn) * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig =
                (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
            if (minEig < a.min_eig || D < 1.1920928955078125e-07F) {
                if (level == 0) ok = false;
                continue;
            }
            D = 1.f / D;
            const uint8_t* Jb = tk_px(pyrJ, L);
            float nx = np.x - half_x, ny = np.y - half_y;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < a.max_count; j++) {
                const int inx = (int)floorf(nx), iny = (int)floorf(ny);
                if (inx < -ww || inx >= L.w || iny < -wh || iny >= L.h) {
                    if (level == 0) ok = false;
                    break;
                }
                fa = nx - inx;
                fb = ny - iny;
                iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
                iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
                iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                const uint8_t* J0 = Jb + (ptrdiff_t)iny * pitch + inx;
                int sb1 = 0, sb2 = 0;  // per lane: <= 16 terms below 2^26
                for (int i = lane; i < area; i += 64) {
                    const int y = i / ww, x = i - y * ww;
                    const uint8_t* s = J0 + y * pitch + x;
                    const int diff =
                        TK_DESCALE(s[0] * iw00 + s[1] * iw01 + s[pitch] * iw10 + s[pitch + 1] * iw11, W_BITS - 5) - Iw[i];
                    sb1 += diff * Dx[i];
                    sb2 += diff * Dy[i];
                }
                const long long ib1 = wave_sum((long long)sb1), ib2 = wave_sum((long long)sb2);
                const float b1 = __ll2float_rn(ib1 * cn) * FLT_SCALE, b2 = __ll2float_rn(ib2 * cn) * FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
                nx += dx;
                ny += dy;
                np = make_float2(nx + half_x, ny + half_y);
                if ((double)dx * dx + (double)dy * dy <= a.eps2) break;
                if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                    np.x -= dx * 0.5f;
                    np.y -= dy * 0.5f;
                    break;
                }
                pdx = dx;
                pdy = dy;
            }
        }
    }
    if (lane == 0) {
        next_pts[pt] = np;
        status[pt] = ok ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// The same tracker for windows of at most 64 x LK_K pixels (the default 21 x 21 needs 7 per lane): a lane
// keeps its window samples of I, Ix, Iy in registers (no LDS), the pixel offsets of its items are
// computed once per level instead of a division per sample, all loads of a pass are issued before the
// first is used, products are 24-bit multiplies, and the wave sums are two 32-bit DPP reductions (low and
// high halves of the per-lane sums) instead of 64-bit shuffles.  Integer sums and float expressions are
// those of the generic kernel, so the results are identical.
// ---------------------------------------------------------------------------------------------
#define LK_K 8

namespace {

__device__ __forceinline__ int lk_row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);  // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);  // row_ror:1
    return v;
}

// exact sum over the wave of per-lane int32 values (the total may need more than 32 bits)
__device__ __forceinline__ long long lk_wave_sum(int v) {
    const int lo = lk_row16_sum(v & 0xffff), hi = lk_row16_sum(v >> 16);
    const int slo = __builtin_amdgcn_readlane(lo, 0) + __builtin_amdgcn_readlane(lo, 16) +
                    __builtin_amdgcn_readlane(lo, 32) + __builtin_amdgcn_readlane(lo, 48);
    const int shi = __builtin_amdgcn_readlane(hi, 0) + __builtin_amdgcn_readlane(hi, 16) +
                    __builtin_amdgcn_readlane(hi, 32) + __builtin_amdgcn_readlane(hi, 48);
    return (long long)shi * 65536 + slo;
}

}  // namespace

__global__ __launch_bounds__(256) void k_lk_track_small(TrackGeom g, const uint8_t* __restrict__ pyrI,
                                                        const short2* __restrict__ deriv,
                                                        const uint8_t* __restrict__ pyrJ,
                                                        const float2* __restrict__ prev_pts,
                                                        float2* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                        LkBatch b, LkArgs a) {
    LK_BATCH_PROLOGUE
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wv;
    if (pt >= n) return;  // a whole wave leaves; the kernel has no block-wide barrier
    const int ww = a.win_w, wh = a.win_h, area = ww * wh;
    int iy[LK_K], ix[LK_K];
    bool live[LK_K];
#pragma unroll
    for (int k = 0; k < LK_K; k++) {
        const int i = lane + 64 * k;
        live[k] = i < area;
        iy[k] = live[k] ? i / ww : 0;
        ix[k] = live[k] ? i - iy[k] * ww : 0;
    }
    const float2 p = prev_pts[pt];
    const float half_x = (ww - 1) * 0.5f, half_y = (wh - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const long long cn = a.channels;
    float2 np = make_float2(0.f, 0.f);
    bool ok = true;
    for (int level = g.nlevels - 1; level >= 0; level--) {
        const TrackLevel L = g.lv[level];
        const float sc = __int_as_float((127 - level) << 23);  // (float)(1. / (1 << level))
        float px = p.x * sc, py = p.y * sc;
        if (level == g.nlevels - 1) np = make_float2(px, py);
        else np = make_float2(np.x * 2.f, np.y * 2.f);
        px -= half_x;
        py -= half_y;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -ww || ipx >= L.w || ipy < -wh || ipy >= L.h) {
            if (level == 0) ok = false;
            continue;
        }
        float fa = px - ipx, fb = py - ipy;
        int iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
        int iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
        int iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        const int pitch = L.pitch;
        int off[LK_K];
#pragma unroll
        for (int k = 0; k < LK_K; k++) off[k] = __mul24(iy[k], pitch) + ix[k];
        int Iw[LK_K], Dx[LK_K], Dy[LK_K];
        int sA11 = 0, sA12 = 0, sA22 = 0;  // per lane: <= LK_K terms below 2^25
        {
            const uint8_t* I0 = tk_px(pyrI, L) + (ptrdiff_t)ipy * pitch + ipx;
            const short2* D0 = deriv + L.offset + (size_t)(ipy + TK_PAD) * pitch + TK_PAD + ipx;
            int sv[LK_K][4];
            short2 dv[LK_K][4];
#pragma unroll
            for (int k = 0; k < LK_K; k++) {
                const uint8_t* s = I0 + off[k];
                const short2* d = D0 + off[k];
                sv[k][0] = s[0], sv[k][1] = s[1], sv[k][2] = s[pitch], sv[k][3] = s[pitch + 1];
                dv[k][0] = d[0], dv[k][1] = d[1], dv[k][2] = d[pitch], dv[k][3] = d[pitch + 1];
            }
#pragma unroll
            for (int k = 0; k < LK_K; k++) {
                const int ival = TK_DESCALE(__mul24(sv[k][0], iw00) + __mul24(sv[k][1], iw01) +
                                                __mul24(sv[k][2], iw10) + __mul24(sv[k][3], iw11),
                                            W_BITS - 5);
                int ixval = TK_DESCALE(__mul24((int)dv[k][0].x, iw00) + __mul24((int)dv[k][1].x, iw01) +
                                           __mul24((int)dv[k][2].x, iw10) + __mul24((int)dv[k][3].x, iw11),
                                       W_BITS);
                int iyval = TK_DESCALE(__mul24((int)dv[k][0].y, iw00) + __mul24((int)dv[k][1].y, iw01) +
                                           __mul24((int)dv[k][2].y, iw10) + __mul24((int)dv[k][3].y, iw11),
                                       W_BITS);
                if (!live[k]) ixval = iyval = 0;  // items past the window add nothing anywhere below
                Iw[k] = ival;
                Dx[k] = ixval;
                Dy[k] = iyval;
                sA11 += __mul24(ixval, ixval);
                sA12 += __mul24(ixval, iyval);
                sA22 += __mul24(iyval, iyval);
            }
        }
        const long long iA11 = lk_wave_sum(sA11), iA12 = lk_wave_sum(sA12), iA22 = lk_wave_sum(sA22);
        const float A11 = __ll2float_rn(iA11 * cn) * FLT_SCALE, A12 = __ll2float_rn(iA12 * cn) * FLT_SCALE,
                    A22 = __ll2float_rn(iA22 * cn) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
        if (minEig < a.min_eig || D < 1.1920928955078125e-07F) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;
        const uint8_t* Jb = tk_px(pyrJ, L);
        float nx = np.x - half_x, ny = np.y - half_y;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -ww || inx >= L.w || iny < -wh || iny >= L.h) {
                if (level == 0) ok = false;
                break;
            }
            fa = nx - inx;
            fb = ny - iny;
            iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
            iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
            iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            const uint8_t* J0 = Jb + (ptrdiff_t)iny * pitch + inx;
            int sv[LK_K][4];
#pragma unroll
            for (int k = 0; k < LK_K; k++) {
                const uint8_t* s = J0 + off[k];
                sv[k][0] = s[0], sv[k][1] = s[1], sv[k][2] = s[pitch], sv[k][3] = s[pitch + 1];
            }
            int sb1 = 0, sb2 = 0;  // per lane: <= LK_K terms below 2^26
#pragma unroll
            for (int k = 0; k < LK_K; k++) {
                const int diff = TK_DESCALE(__mul24(sv[k][0], iw00) + __mul24(sv[k][1], iw01) +
                                                __mul24(sv[k][2], iw10) + __mul24(sv[k][3], iw11),
                                            W_BITS - 5) -
                                 Iw[k];
                sb1 += __mul24(diff, Dx[k]);
                sb2 += __mul24(diff, Dy[k]);
            }
            const long long ib1 = lk_wave_sum(sb1), ib2 = lk_wave_sum(sb2);
            const float b1 = __ll2float_rn(ib1 * cn) * FLT_SCALE, b2 = __ll2float_rn(ib2 * cn) * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            nx += dx;
            ny += dy;
            np = make_float2(nx + half_x, ny + half_y);
            if ((double)dx * dx + (double)dy * dy <= a.eps2) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                np.x -= dx * 0.5f;
                np.y -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
        }
    }
    if (lane == 0) {
        next_pts[pt] = np;
        status[pt] = ok ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// cv's default 21 x 21 window (what the reference tracks with, src/features/feature_tracker.cpp:6): lane 3 r + s
// owns the seven pixels 7 s .. 7 s + 6 of window row r.  Their 2 x 2 bilinear taps are the bytes 7 s .. 7 s + 7 of
// rows r and r + 1, i.e. TWO unaligned 8-byte loads per resampling of J where k_lk_track_small issues 28 byte
// loads (its items are 64 pixels apart, so no two share a load); the derivative samples of the I window are
// two 16-byte loads per row.  That kernel is bound by the number of load instructions the texture path takes
// per iteration (3 000 waves x 4 levels x up to 30 iterations x 32 loads), not by their bytes.  Same integer
// sums (any order gives the same total), same float expressions: identical results.
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ uint2 lk_load8(const uint8_t* p) {  // 8 bytes at any alignment
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ int lk_byte(const uint2& v, int j) {  // byte j (0..7), j known at compile time
    return (int)(((j < 4 ? v.x : v.y) >> (8 * (j & 3))) & 0xffu);
}
}  // namespace

__global__ __launch_bounds__(256) void k_lk_track_21(TrackGeom g, const uint8_t* __restrict__ pyrI,
                                                     const short2* __restrict__ deriv,
                                                     const uint8_t* __restrict__ pyrJ,
                                                     const float2* __restrict__ prev_pts,
                                                     float2* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                     LkBatch b, LkArgs a) {
    LK_BATCH_PROLOGUE
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wv;
    if (pt >= n) return;  // a whole wave leaves; the kernel has no block-wide barrier
    constexpr int ww = 21, wh = 21, NP = 7;
    const bool live = lane < 63;
    const int r = live ? (lane * 21846) >> 16 : 0;  // lane / 3 for lane < 64
    const int c0 = live ? NP * (lane - 3 * r) : 0;
    const float2 p = prev_pts[pt];
    const float half_x = (ww - 1) * 0.5f, half_y = (wh - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const long long cn = a.channels;
    float2 np = make_float2(0.f, 0.f);
    bool ok = true;
    for (int level = g.nlevels - 1; level >= 0; level--) {
        const TrackLevel L = g.lv[level];
        const float sc = __int_as_float((127 - level) << 23);  // (float)(1. / (1 << level))
        float px = p.x * sc, py = p.y * sc;
        if (level == g.nlevels - 1) np = make_float2(px, py);
        else np = make_float2(np.x * 2.f, np.y * 2.f);
        px -= half_x;
        py -= half_y;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -ww || ipx >= L.w || ipy < -wh || ipy >= L.h) {
            if (level == 0) ok = false;
            continue;
        }
        float fa = px - ipx, fb = py - ipy;
        int iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
        int iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
        int iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        const int pitch = L.pitch;
        const int off = __mul24(r, pitch) + c0;  // the lane's first pixel inside a window
        int Iw[NP], Dx[NP], Dy[NP];
        int sA11 = 0, sA12 = 0, sA22 = 0;  // per lane: NP terms below 2^25
        {
            const uint8_t* I0 = tk_px(pyrI, L) + (ptrdiff_t)ipy * pitch + ipx + off;
            const short2* D0 = deriv + L.offset + (size_t)(ipy + TK_PAD) * pitch + TK_PAD + ipx + off;
            const uint2 s0 = lk_load8(I0), s1 = lk_load8(I0 + pitch);
            uint4 d0[2], d1[2];  // (dx, dy) of the eight pixels of row r / row r + 1
            __builtin_memcpy(&d0[0], D0, 16);
            __builtin_memcpy(&d0[1], D0 + 4, 16);
            __builtin_memcpy(&d1[0], D0 + pitch, 16);
            __builtin_memcpy(&d1[1], D0 + pitch + 4, 16);
            const uint32_t* e0 = reinterpret_cast<const uint32_t*>(d0);
            const uint32_t* e1 = reinterpret_cast<const uint32_t*>(d1);
#pragma unroll
            for (int j = 0; j < NP; j++) {
                const int ival = TK_DESCALE(__mul24(lk_byte(s0, j), iw00) + __mul24(lk_byte(s0, j + 1), iw01) +
                                                __mul24(lk_byte(s1, j), iw10) + __mul24(lk_byte(s1, j + 1), iw11),
                                            W_BITS - 5);
                const int x00 = (int)(short)(e0[j] & 0xffff), y00 = (int)e0[j] >> 16;
                const int x01 = (int)(short)(e0[j + 1] & 0xffff), y01 = (int)e0[j + 1] >> 16;
                const int x10 = (int)(short)(e1[j] & 0xffff), y10 = (int)e1[j] >> 16;
                const int x11 = (int)(short)(e1[j + 1] & 0xffff), y11 = (int)e1[j + 1] >> 16;
                int ixval = TK_DESCALE(__mul24(x00, iw00) + __mul24(x01, iw01) + __mul24(x10, iw10) + __mul24(x11, iw11),
                                       W_BITS);
                int iyval = TK_DESCALE(__mul24(y00, iw00) + __mul24(y01, iw01) + __mul24(y10, iw10) + __mul24(y11, iw11),
                                       W_BITS);
                if (!live) ixval = iyval = 0;  // the idle lane adds nothing anywhere below
                Iw[j] = ival;
                Dx[j] = ixval;
                Dy[j] = iyval;
                sA11 += __mul24(ixval, ixval);
                sA12 += __mul24(ixval, iyval);
                sA22 += __mul24(iyval, iyval);
            }
        }
        const long long iA11 = lk_wave_sum(sA11), iA12 = lk_wave_sum(sA12), iA22 = lk_wave_sum(sA22);
        const float A11 = __ll2float_rn(iA11 * cn) * FLT_SCALE, A12 = __ll2float_rn(iA12 * cn) * FLT_SCALE,
                    A22 = __ll2float_rn(iA22 * cn) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * ww * wh);
        if (minEig < a.min_eig || D < 1.1920928955078125e-07F) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;
        const uint8_t* Jb = tk_px(pyrJ, L) + off;
        float nx = np.x - half_x, ny = np.y - half_y;
        float pdx = 0.f, pdy = 0.f;
        for (int it = 0; it < a.max_count; it++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -ww || inx >= L.w || iny < -wh || iny >= L.h) {
                if (level == 0) ok = false;
                break;
            }
            fa = nx - inx;
            fb = ny - iny;
            iw00 = __float2int_rn((1.f - fa) * (1.f - fb) * (1 << W_BITS));
            iw01 = __float2int_rn(fa * (1.f - fb) * (1 << W_BITS));
            iw10 = __float2int_rn((1.f - fa) * fb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            const uint8_t* J0 = Jb + (ptrdiff_t)iny * pitch + inx;
            const uint2 s0 = lk_load8(J0), s1 = lk_load8(J0 + pitch);
            int sb1 = 0, sb2 = 0;  // per lane: NP terms below 2^26
#pragma unroll
            for (int j = 0; j < NP; j++) {
                const int diff = TK_DESCALE(__mul24(lk_byte(s0, j), iw00) + __mul24(lk_byte(s0, j + 1), iw01) +
                                                __mul24(lk_byte(s1, j), iw10) + __mul24(lk_byte(s1, j + 1), iw11),
                                            W_BITS - 5) -
                                 Iw[j];
                sb1 += __mul24(diff, Dx[j]);
                sb2 += __mul24(diff, Dy[j]);
            }
            const long long ib1 = lk_wave_sum(sb1), ib2 = lk_wave_sum(sb2);
            const float b1 = __ll2float_rn(ib1 * cn) * FLT_SCALE, b2 = __ll2float_rn(ib2 * cn) * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            nx += dx;
            ny += dy;
            np = make_float2(nx + half_x, ny + half_y);
            if ((double)dx * dx + (double)dy * dy <= a.eps2) break;
            if (it > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                np.x -= dx * 0.5f;
                np.y -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
        }
    }
    if (lane == 0) {
        next_pts[pt] = np;
        status[pt] = ok ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
void launch_lk_pyramid(hipStream_t st, const uint8_t* img, size_t img_stride, int stride, const TrackGeom& g,
                       uint8_t* pyr, int nimg) {
    {
        const TrackLevel& L = g.lv[0];
        dim3 grid(cdiv(cdiv(L.w + 2 * TK_PAD, 4), 64), cdiv(L.h + 2 * TK_PAD, 4), nimg);
        hipLaunchKernelGGL(k_lk_level0, grid, dim3(256), 0, st, img, img_stride, stride, g, pyr);
    }
    for (int l = 1; l < g.nlevels; l++) {
        const TrackLevel& L = g.lv[l];
        dim3 grid(cdiv(cdiv(L.w + 2 * TK_PAD, 4), 64), cdiv(L.h + 2 * TK_PAD, 4), nimg);
        hipLaunchKernelGGL(k_lk_pyrdown, grid, dim3(256), 0, st, g, pyr, l);
    }
}

void launch_lk_scharr(hipStream_t st, const TrackGeom& g, const uint8_t* pyr, short2* deriv, int nimg) {
    const TrackLevel& L = g.lv[0];
    dim3 grid(cdiv(L.w, 64), cdiv(L.h, 4), nimg * g.nlevels);
    hipLaunchKernelGGL(k_lk_scharr, grid, dim3(256), 0, st, g, pyr, deriv);
}

void launch_lk_track(hipStream_t st, const TrackGeom& g, const uint8_t* pyrI, const short2* deriv,
                     const uint8_t* pyrJ, const float2* prev_pts, float2* next_pts, uint8_t* status,
                     const LkBatch& b, const LkArgs& a) {
    int nmax = 0;
    for (int i = 0; i < b.count; i++) nmax = b.n[i] > nmax ? b.n[i] : nmax;
    if (nmax <= 0 || b.count <= 0) return;
    const dim3 grid(cdiv(nmax, 4), b.count);
    // a.generic (VO_OPT_LK_GENERIC): the 21 x 21 window through the kernels every other window size uses -- the
    // parity tests compare the two
    if (a.win_w == 21 && a.win_h == 21 && !a.generic) {
        hipLaunchKernelGGL(k_lk_track_21, grid, dim3(256), 0, st, g, pyrI, deriv, pyrJ, prev_pts, next_pts, status, b, a);
        return;
    }
    if (a.win_w * a.win_h <= 64 * LK_K) {
        hipLaunchKernelGGL(k_lk_track_small, grid, dim3(256), 0, st, g, pyrI, deriv, pyrJ, prev_pts, next_pts, status,
                           b, a);
        return;
    }
    const size_t lds = (size_t)4 * a.win_w * a.win_h * 3 * sizeof(int16_t);
    hipLaunchKernelGGL(k_lk_track, grid, dim3(256), lds, st, g, pyrI, deriv, pyrJ, prev_pts, next_pts, status, b, a);
}
