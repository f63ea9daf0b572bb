// k_akaze.hip — gfx950 kernels for cv::AKAZE (MLDB-486), SURVEY.md 8f row 4.
// Reference call site: src/features/feature_detector.cpp:7,43-45 (from src/sfm/bundle_adjustment.cpp:41).
//
// A float pipeline: every expression keeps the operation order of the CPU form (the file is built with
// -ffp-contract=off and correctly rounded division / sqrt), so the planes are the same floats as a scalar
// x86 evaluation.  All planes are tight (pitch = width) float images in one arena; a pixel is a thread.
// The non-linear scale space is bandwidth-trivial at these sizes (0.47 M pixels at level 0, a quarter per
// octave) and launch-bound: ~300 short launches per image, replayed as one hipGraph (vo_akaze.cpp).  (A FED
// cycle of a coarse level in ONE workgroup with Lt in LDS was tried to save its 8-29 launches: 0.4 ms per level
// on one CU against ~5 us per step across the chip. Removed.)  What is order-dependent stays sequential where
// the order is defined: the 109-sample orientation sums run one window per lane (each lane adds its samples
// in index order), the MLDB cell means one cell per lane (samples in scan order).
#include <algorithm>
#include <cfloat>
#include <cmath>

#include "kernels.h"
#include "akaze_types.h"

namespace {

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// cv::fastAtan2, degrees in [0, 360): the same polynomial as k_features.hip's IC angle
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI),
                p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

}  // namespace

__global__ __launch_bounds__(256) void k_ak_cvt(const uint8_t* __restrict__ gray, int w, int h,
                                                float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    out[i] = (float)gray[i] * (1.0f / 255.0f);
}

// One pass of cv::GaussianBlur (BORDER_REPLICATE): centre + sum_j k[j] (left + right).  dir 0: along x.
__global__ __launch_bounds__(256) void k_ak_gauss(const float* __restrict__ src, float* __restrict__ dst, int w, int h,
                                                  AkTaps t, int dir) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    float s = t.k[0] * src[i];
    if (dir == 0) {
        for (int j = 1; j <= t.r; j++) s += t.k[j] * (src[y * w + max(x - j, 0)] + src[y * w + min(x + j, w - 1)]);
    } else {
        for (int j = 1; j <= t.r; j++) s += t.k[j] * (src[max(y - j, 0) * w + x] + src[min(y + j, h - 1) * w + x]);
    }
    dst[i] = s;
}

// The separable three-tap derivative filters (taps at distance `scale`, BORDER_REFLECT_101), both passes in one
// kernel: the row pass value T of the three rows a pixel needs is recomputed instead of stored, with the same
// operations in the same order as the two-pass form.
//   d/dx:  T(y, x) = S(y, x + s) - S(y, x - s);                out = wnorm T(y, x) + norm (T(y - s, x) + T(y + s, x))
//   d/dy:  T(y, x) = wnorm S(y, x) + norm (S(y, x - s) + S(y, x + s));   out = T(y + s, x) - T(y - s, x)
struct Tap3 {
    int xm, xp, ym, yp;  // reflected neighbour coordinates
};
__device__ __forceinline__ Tap3 tap3(int x, int y, int w, int h, int s) {
    return {reflect101(x - s, w), reflect101(x + s, w), reflect101(y - s, h), reflect101(y + s, h)};
}
__device__ __forceinline__ float ddx(const float* __restrict__ S, int w, int x, int y, const Tap3& t, float norm,
                                     float wnorm) {
    const float tc = S[y * w + t.xp] - S[y * w + t.xm];
    const float ta = S[t.ym * w + t.xp] - S[t.ym * w + t.xm];
    const float tb = S[t.yp * w + t.xp] - S[t.yp * w + t.xm];
    return wnorm * tc + norm * (ta + tb);
}
__device__ __forceinline__ float ddy(const float* __restrict__ S, int w, int x, int y, const Tap3& t, float norm,
                                     float wnorm) {
    const float ta = wnorm * S[t.ym * w + x] + norm * (S[t.ym * w + t.xm] + S[t.ym * w + t.xp]);
    const float tb = wnorm * S[t.yp * w + x] + norm * (S[t.yp * w + t.xm] + S[t.yp * w + t.xp]);
    return tb - ta;
}

// Lx and Ly of one plane (the multiscale derivatives kept for the descriptors)
__global__ __launch_bounds__(256) void k_ak_grad(const float* __restrict__ S, float* __restrict__ Lx,
                                                 float* __restrict__ Ly, int w, int h, int scale, float norm,
                                                 float wnorm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    const Tap3 t = tap3(x, y, w, h, scale);
    Lx[i] = ddx(S, w, x, y, t, norm, wnorm);
    Ly[i] = ddy(S, w, x, y, t, norm, wnorm);
}

// Perona-Malik g2 conductivity from the un-normalised Scharr gradient (3 10 3) of the smoothed level
__global__ __launch_bounds__(256) void k_ak_cond(const float* __restrict__ S, int w, int h,
                                                 const AkContrast* __restrict__ kc, int octave,
                                                 float* __restrict__ flow) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    const Tap3 t = tap3(x, y, w, h, 1);
    const float lx = ddx(S, w, x, y, t, 3.f, 10.f), ly = ddy(S, w, x, y, t, 3.f, 10.f);
    const float k = kc->k[octave];
    const float inv_k = 1.0f / (k * k);
    flow[i] = 1.0f / (1.0f + inv_k * (lx * lx + ly * ly));
}

// Hessian determinant from the first derivatives: Lxx = d/dx Lx, Lxy = d/dy Lx, Lyy = d/dy Ly
__global__ __launch_bounds__(256) void k_ak_hessdet(const float* __restrict__ Lx, const float* __restrict__ Ly, int w,
                                                    int h, int scale, float norm, float wnorm, float s4,
                                                    float* __restrict__ Ldet) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    const Tap3 t = tap3(x, y, w, h, scale);
    const float lxx = ddx(Lx, w, x, y, t, norm, wnorm);
    const float lxy = ddy(Lx, w, x, y, t, norm, wnorm);
    const float lyy = ddy(Ly, w, x, y, t, norm, wnorm);
    Ldet[i] = (lxx * lyy - lxy * lxy) * s4;
}

// compute_kcontrast, step 1: the largest gradient magnitude of the interior (floats >= 0 order like their bits)
__global__ __launch_bounds__(256) void k_ak_hmax(const float* __restrict__ Lx, const float* __restrict__ Ly, int w, int h,
                                                 AkContrast* __restrict__ kc) {
    __shared__ float part[4];
    float v = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        if (y >= 1 && y < h - 1 && x >= 1 && x < w - 1) v = fmaxf(v, sqrtf(Lx[i] * Lx[i] + Ly[i] * Ly[i]));
    }
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (v > 0.f) atomicMax(&kc->hmax_bits, __float_as_uint(v));
    }
}

// step 2: histogram of the magnitudes scaled to [0, nbins - 1] (integer counts: any order gives the same table)
__global__ __launch_bounds__(256) void k_ak_hist(const float* __restrict__ Lx, const float* __restrict__ Ly, int w, int h,
                                                 int nbins, AkContrast* __restrict__ kc) {
    __shared__ unsigned int hist[AK_MAX_BINS];
    for (int k = threadIdx.x; k < nbins; k += 256) hist[k] = 0;
    __syncthreads();
    const float hmax = __uint_as_float(kc->hmax_bits);
    if (hmax == 0.0f) return;  // uniform: every thread of the grid
    const float sc = (float)(nbins - 1) / hmax;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        if (y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {
            const float d = sqrtf(Lx[i] * Lx[i] + Ly[i] * Ly[i]);
            atomicAdd(&hist[(int)(d * sc)], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += 256)
        if (hist[k]) atomicAdd(&kc->hist[k], hist[k]);
}

// step 3: the percentile, and the contrast of every octave (x 0.75 per octave, multiplied one at a time)
__global__ void k_ak_kcontrast(int total, float perc, int nbins, int n_octaves, AkContrast* __restrict__ kc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float hmax = __uint_as_float(kc->hmax_bits);
    float k = 0.03f;
    if (hmax != 0.0f) {
        const int nthreshold = (int)((float)(total - (int)kc->hist[0]) * perc);
        int nelements = 0;
        for (int b = 1; b < nbins; b++) {
            if (nelements >= nthreshold) {
                k = hmax * (float)b / (float)nbins;
                break;
            }
            nelements += (int)kc->hist[b];
        }
    }
    for (int o = 0; o < n_octaves; o++) {
        kc->k[o] = k;
        k *= 0.75f;
    }
}

// one explicit diffusion step: out = Lt + step * div(c grad Lt), zero-flux borders
__global__ __launch_bounds__(256) void k_ak_nld(const float* __restrict__ Lt, const float* __restrict__ c,
                                                float* __restrict__ out, int w, int h, float step) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    const int xp = max(x - 1, 0), xn = min(x + 1, w - 1), yp = max(y - 1, 0), yn = min(y + 1, h - 1);
    const float l = Lt[i], cc = c[i];
    const float xpos = (cc + c[y * w + xn]) * (Lt[y * w + xn] - l);
    const float xneg = (c[y * w + xp] + cc) * (l - Lt[y * w + xp]);
    const float ypos = (cc + c[yn * w + x]) * (Lt[yn * w + x] - l);
    const float yneg = (c[yp * w + x] + cc) * (l - Lt[yp * w + x]);
    out[i] = l + step * (xpos - xneg + ypos - yneg);
}

// cv::resize(INTER_AREA), general path: tables from the host (first[d] .. first[d + 1]: entries of destination d)
__global__ __launch_bounds__(256) void k_ak_area(const float* __restrict__ src, int sw, float* __restrict__ dst, int dw,
                                                 int dh, const int* __restrict__ xfirst, const int* __restrict__ xsi,
                                                 const float* __restrict__ xalpha, const int* __restrict__ yfirst,
                                                 const int* __restrict__ ysi, const float* __restrict__ yalpha) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dw * dh) return;
    const int y = i / dw, x = i - y * dw;
    const int b0 = yfirst[y], b1 = yfirst[y + 1], a0 = xfirst[x], a1 = xfirst[x + 1];
    float sum = 0.f;
    for (int b = b0; b < b1; b++) {
        const float* row = src + (size_t)ysi[b] * sw;
        float buf = 0.f;
        for (int a = a0; a < a1; a++) buf += row[xsi[a]] * xalpha[a];
        if (b == b0) sum = yalpha[b] * buf;
        else sum += yalpha[b] * buf;
    }
    dst[i] = sum;
}

// 3 x 3 strict maxima above the threshold, inside the level's border: appended (in any order; the host sorts by
// position) with the nine determinants the sub-pixel fit needs
__global__ __launch_bounds__(256) void k_ak_extrema(const float* __restrict__ L, int w, int h, int border, float thr,
                                                    int level_at, AkExt* __restrict__ list,
                                                    unsigned* __restrict__ count, unsigned cap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    if (!(border + 1 < h && y >= border && y < h - border && x >= border && x < w - border)) return;
    const float v = L[i];
    if (!(v > thr)) return;
    const float d0 = L[i - w - 1], d1 = L[i - w], d2 = L[i - w + 1], d3 = L[i - 1], d5 = L[i + 1], d6 = L[i + w - 1],
                d7 = L[i + w], d8 = L[i + w + 1];
    if (v > d3 && v > d5 && v > d0 && v > d1 && v > d2 && v > d6 && v > d7 && v > d8) {
        const unsigned k = atomicAdd(count, 1u);
        if (k < cap) {
            AkExt e;
            e.at = level_at + i;
            e.d[0] = d0; e.d[1] = d1; e.d[2] = d2; e.d[3] = d3; e.d[4] = v; e.d[5] = d5; e.d[6] = d6; e.d[7] = d7; e.d[8] = d8;
            list[k] = e;
        }
    }
}

// Compute_Main_Orientation: one wave per keypoint.  Lanes gather the 109 weighted samples; lane = window for the
// 42 sliding windows, each adding its samples in index order; the first maximum wins (strict >).
__global__ __launch_bounds__(64) void k_ak_orient(vo_keypoint* __restrict__ kps, int n, AkLevels lv,
                                                  const float* __restrict__ arena) {
    __shared__ float resX[112], resY[112], Ang[112];
    const int kpi = blockIdx.x, lane = threadIdx.x;
    if (kpi >= n) return;
    vo_keypoint kp = kps[kpi];
    const AkLevel e = lv.l[kp.class_id];
    const float* Lx = arena + e.lx;
    const float* Ly = arena + e.ly;
    const int s = __float2int_rn(0.5f * kp.size / e.ratio);
    const float xf = kp.x / e.ratio, yf = kp.y / e.ratio;
    const float deg2rad = (float)(M_PI / 180.0);
    // sample k of the scan i = -6..6, j = -6..6 with i^2 + j^2 < 36 (table built by the host in that order)
    for (int k = lane; k < 109; k += 64) {
        const int i = lv.oi[k], j = lv.oj[k];
        const int iy = min(max(__float2int_rn(yf + (float)(j * s)), 0), e.h - 1);
        const int ix = min(max(__float2int_rn(xf + (float)(i * s)), 0), e.w - 1);
        const float gw = lv.gauss25[abs(i)][abs(j)];
        const float rx = gw * Lx[iy * e.w + ix], ry = gw * Ly[iy * e.w + ix];
        resX[k] = rx;
        resY[k] = ry;
        Ang[k] = fast_atan2_deg(ry, rx) * deg2rad;
    }
    __syncthreads();
    float ang1 = 0.f;
    for (int k = 0; k < lane; k++) ang1 += 0.15f;
    float val = -1.f, angle = 0.f;
    if (ang1 < (float)(2.0 * M_PI)) {
        const float ang2 = (ang1 + (float)(M_PI / 3.0) > (float)(2.0 * M_PI) ? ang1 - (float)(5.0 * M_PI / 3.0)
                                                                           : ang1 + (float)(M_PI / 3.0));
        float sumX = 0.f, sumY = 0.f;
        for (int k = 0; k < 109; ++k) {
            const float ang = Ang[k];
            if (ang1 < ang2 && ang1 < ang && ang < ang2) {
                sumX += resX[k];
                sumY += resY[k];
            } else if (ang2 < ang1 && ((ang > 0 && ang < ang2) || (ang > ang1 && ang < (float)(2.0 * M_PI)))) {
                sumX += resX[k];
                sumY += resY[k];
            }
        }
        val = sumX * sumX + sumY * sumY;
        angle = fast_atan2_deg(sumY, sumX);
    }
    // sequential rule: a window replaces the best so far only if strictly larger, starting from max = 0
    float bv = val;
    int bl = lane;
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(bv, m);
        const int ol = __shfl_xor(bl, m);
        if (ov > bv || (ov == bv && ol < bl)) {
            bv = ov;
            bl = ol;
        }
    }
    const float best = __shfl(angle, bl);
    if (lane == 0) kps[kpi].angle = bv > 0.f ? best : 0.f;
}

// MLDB full descriptor: one wave per keypoint.  Lane c < 29 owns one cell of the 2x2 / 3x3 / 4x4 grids and adds its
// samples in scan order; lane b < 61 then packs byte b from the comparison table.
__global__ __launch_bounds__(64) void k_ak_mldb(const vo_keypoint* __restrict__ kps, int n, AkLevels lv,
                                                const float* __restrict__ arena, const AkPair* __restrict__ pairs,
                                                int pattern, int chan, uint8_t* __restrict__ desc) {
    __shared__ int vals[29 * 3];
    const int kpi = blockIdx.x, lane = threadIdx.x;
    if (kpi >= n) return;
    const vo_keypoint kp = kps[kpi];
    const AkLevel e = lv.l[kp.class_id];
    const float* Lt = arena + e.lt;
    const float* Lx = arena + e.lx;
    const float* Ly = arena + e.ly;
    const float scale = (float)__float2int_rn(0.5f * kp.size / e.ratio);
    const float xf = kp.x / e.ratio, yf = kp.y / e.ratio;
    const float angle = kp.angle * (float)M_PI / 180.f;
    const float co = (float)cos((double)angle), si = (float)sin((double)angle);
    if (lane < 29) {
        const int z = lane < 4 ? 0 : (lane < 13 ? 1 : 2);
        const int g = z + 2, c = lane - (z == 0 ? 0 : (z == 1 ? 4 : 13));
        const int step = z == 0 ? pattern : (z == 1 ? (pattern * 2 + 2) / 3 : (pattern + 1) / 2);
        const int i0 = -pattern + (c / g) * step, j0 = -pattern + (c % g) * step;
        float di = 0.f, dx = 0.f, dy = 0.f;
        int ns = 0;
        for (int k = i0; k < i0 + step; k++)
            for (int l = j0; l < j0 + step; l++) {
                const float sy = yf + ((float)l * co * scale + (float)k * si * scale);
                const float sx = xf + (-(float)l * si * scale + (float)k * co * scale);
                const int y1 = __float2int_rn(sy), x1 = __float2int_rn(sx);
                if (y1 < 0 || y1 >= e.h || x1 < 0 || x1 >= e.w) continue;
                const int at = y1 * e.w + x1;
                di += Lt[at];
                if (chan > 1) {
                    const float rx = Lx[at], ry = Ly[at];
                    if (chan == 2) {
                        dx += sqrtf(rx * rx + ry * ry);
                    } else {
                        dx += -rx * si + ry * co;
                        dy += rx * co + ry * si;
                    }
                }
                ns++;
            }
        if (ns > 0) {
            const float inv = 1.0f / (float)ns;
            di *= inv;
            dx *= inv;
            dy *= inv;
        }
        // CV_TOGGLE_FLT: integer order of floats
        const int a = __float_as_int(di), b = __float_as_int(dx), d = __float_as_int(dy);
        vals[lane * 3] = a < 0 ? (a ^ 0x7fffffff) : a;
        vals[lane * 3 + 1] = b < 0 ? (b ^ 0x7fffffff) : b;
        vals[lane * 3 + 2] = d < 0 ? (d ^ 0x7fffffff) : d;
    }
    __syncthreads();
    const int nbits = lv.nbits;
    if (lane < AK_DESC_BYTES) {  // bytes beyond the last bit (fewer channels) are zero, as cv clears the row
        unsigned byte = 0;
        for (int bit = 0; bit < 8; bit++) {
            const int dpos = lane * 8 + bit;
            if (dpos < nbits) {
                const AkPair p = pairs[dpos];
                if (vals[p.a] > vals[p.b]) byte |= 1u << bit;
            }
        }
        desc[(size_t)kpi * AK_DESC_BYTES + lane] = (uint8_t)byte;
    }
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
static inline dim3 grid1(int n) { return dim3((n + 255) / 256); }

void ak_launch_cvt(hipStream_t st, const uint8_t* gray, int w, int h, float* out) {
    hipLaunchKernelGGL(k_ak_cvt, grid1(w * h), dim3(256), 0, st, gray, w, h, out);
}
void ak_launch_gauss(hipStream_t st, const float* src, float* tmp, float* dst, int w, int h, const AkTaps& t) {
    hipLaunchKernelGGL(k_ak_gauss, grid1(w * h), dim3(256), 0, st, src, tmp, w, h, t, 0);
    hipLaunchKernelGGL(k_ak_gauss, grid1(w * h), dim3(256), 0, st, tmp, dst, w, h, t, 1);
}
void ak_launch_grad(hipStream_t st, const float* S, float* Lx, float* Ly, int w, int h, int scale, float norm,
                    float wnorm) {
    hipLaunchKernelGGL(k_ak_grad, grid1(w * h), dim3(256), 0, st, S, Lx, Ly, w, h, scale, norm, wnorm);
}
void ak_launch_cond(hipStream_t st, const float* S, int w, int h, const AkContrast* kc, int octave, float* flow) {
    hipLaunchKernelGGL(k_ak_cond, grid1(w * h), dim3(256), 0, st, S, w, h, kc, octave, flow);
}
void ak_launch_hessdet(hipStream_t st, const float* Lx, const float* Ly, int w, int h, int scale, float norm,
                       float wnorm, float s4, float* Ldet) {
    hipLaunchKernelGGL(k_ak_hessdet, grid1(w * h), dim3(256), 0, st, Lx, Ly, w, h, scale, norm, wnorm, s4, Ldet);
}
void ak_launch_kcontrast(hipStream_t st, const float* Lx, const float* Ly, int w, int h, float perc, int nbins,
                         int n_octaves, AkContrast* kc) {
    (void)hipMemsetAsync(kc, 0, sizeof(AkContrast), st);
    const dim3 g(std::min((w * h + 255) / 256, 512));
    hipLaunchKernelGGL(k_ak_hmax, g, dim3(256), 0, st, Lx, Ly, w, h, kc);
    hipLaunchKernelGGL(k_ak_hist, g, dim3(256), 0, st, Lx, Ly, w, h, nbins, kc);
    hipLaunchKernelGGL(k_ak_kcontrast, dim3(1), dim3(64), 0, st, (w - 2) * (h - 2), perc, nbins, n_octaves, kc);
}
void ak_launch_nld(hipStream_t st, const float* Lt, const float* flow, float* out, int w, int h, float step) {
    hipLaunchKernelGGL(k_ak_nld, grid1(w * h), dim3(256), 0, st, Lt, flow, out, w, h, step);
}
void ak_launch_area(hipStream_t st, const float* src, int sw, float* dst, int dw, int dh, const int* xfirst,
                    const int* xsi, const float* xalpha, const int* yfirst, const int* ysi, const float* yalpha) {
    hipLaunchKernelGGL(k_ak_area, grid1(dw * dh), dim3(256), 0, st, src, sw, dst, dw, dh, xfirst, xsi, xalpha, yfirst,
                       ysi, yalpha);
}
void ak_launch_extrema(hipStream_t st, const float* L, int w, int h, int border, float thr, int level_at, AkExt* list,
                       unsigned* count, unsigned cap) {
    hipLaunchKernelGGL(k_ak_extrema, grid1(w * h), dim3(256), 0, st, L, w, h, border, thr, level_at, list, count, cap);
}
void ak_launch_describe(hipStream_t st, vo_keypoint* kps, int n, const AkLevels& lv, const float* arena,
                        const AkPair* pairs, int pattern, int chan, uint8_t* desc) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ak_orient, dim3(n), dim3(64), 0, st, kps, n, lv, arena);
    hipLaunchKernelGGL(k_ak_mldb, dim3(n), dim3(64), 0, st, kps, n, lv, arena, pairs, pattern, chan, desc);
}
