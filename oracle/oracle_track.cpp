// oracle_track.cpp — TEST INFRASTRUCTURE ONLY (see vo_oracle.h).
//
// CPU restatement of cv::SparsePyrLKOpticalFlow::create()->calc(prev, next, prevPts, nextPts,
// status) with the default parameters the reference uses (winSize 21x21, maxLevel 3, criteria
// COUNT+EPS 30 / 0.01, flags 0, minEigThreshold 1e-4), reference:
// src/features/feature_tracker.cpp:6,13, and of removePoints(), reference:
// src/features/utils.h:9-58.  Follows OpenCV 3.4.x video/src/lkpyramid.cpp and imgproc's pyrDown
// (restated from memory, parity unpinned):
//   * buildOpticalFlowPyramid: level 0 = the image, level l = pyrDown(level l-1) (5x5 binomial,
//     (sum + 128) >> 8, BORDER_REFLECT_101), size (w+1)/2 x (h+1)/2; every level carries a
//     winSize-wide BORDER_REFLECT_101 frame; levels stop at maxLevel or when a level would not be
//     larger than the window;
//   * calcSharrDeriv: 3/10/3 Scharr derivatives as shorts, reflect-101 inside the level, and a
//     constant-zero frame around the derivative image;
//   * LKTrackerInvoker: 14-bit fixed-point bilinear window extraction, integer products.
// Accumulators: cv keeps `acctype` sums of the integer products; its NEON build uses int64 (exact),
// its SSE2 build four float lanes.  The restatement uses the exact integer sums, which do not
// depend on the summation order (the GPU path reduces them across a wavefront); an x86 build of
// cv can differ from this in the last bits of A11..b2.
// Channels: the reference's CPU build hands calc() the 8UC3 image cv::imread returns for a gray
// PNG (src/main.cpp:61, vo_frame.h:37-44), so every sum runs over three identical channels;
// `channels` = 3 multiplies the exact integer sums by three, which is the same thing.
#include <numeric>

#include "oracle_common.h"

namespace {

using namespace vo_oracle;

inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

struct Level {
    int w = 0, h = 0, pad = 0, pitch = 0;
    std::vector<uint8_t> img;    // (h + 2 pad) x pitch, reflect-101 frame
    std::vector<int16_t> deriv;  // (h + 2 pad) x pitch x 2, zero frame
    const uint8_t* px(int x, int y) const { return &img[(size_t)(y + pad) * pitch + x + pad]; }
    const int16_t* d(int x, int y) const { return &deriv[((size_t)(y + pad) * pitch + x + pad) * 2]; }
};

void fill_border(Level& L) {
    for (int y = -L.pad; y < L.h + L.pad; y++)
        for (int x = -L.pad; x < L.w + L.pad; x++)
            if (x < 0 || x >= L.w || y < 0 || y >= L.h)
                L.img[(size_t)(y + L.pad) * L.pitch + x + L.pad] = *L.px(reflect101(x, L.w), reflect101(y, L.h));
}

void alloc_level(Level& L, int w, int h, int pad) {
    L.w = w;
    L.h = h;
    L.pad = pad;
    L.pitch = w + 2 * pad;
    L.img.assign((size_t)(h + 2 * pad) * L.pitch, 0);
}

// cv::pyrDown, 8-bit, one channel, BORDER_REFLECT_101
void pyr_down(const Level& S, Level& D) {
    static const int wt[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < D.h; y++)
        for (int x = 0; x < D.w; x++) {
            int sum = 0;
            for (int ky = 0; ky < 5; ky++) {
                const int sy = reflect101(2 * y + ky - 2, S.h);
                int row = 0;
                for (int kx = 0; kx < 5; kx++) row += wt[kx] * *S.px(reflect101(2 * x + kx - 2, S.w), sy);
                sum += wt[ky] * row;
            }
            D.img[(size_t)(y + D.pad) * D.pitch + x + D.pad] = (uint8_t)((sum + 128) >> 8);
        }
}

// calcSharrDeriv + copyMakeBorder(BORDER_CONSTANT)
void scharr(Level& L) {
    L.deriv.assign((size_t)(L.h + 2 * L.pad) * L.pitch * 2, 0);
    const int w = L.w, h = L.h;
    std::vector<int> t0(w + 2), t1(w + 2);
    for (int y = 0; y < h; y++) {
        const int ya = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yb = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
        for (int x = 0; x < w; x++) {
            const int a = *L.px(x, ya), b = *L.px(x, y), c = *L.px(x, yb);
            t0[x + 1] = (a + c) * 3 + b * 10;
            t1[x + 1] = c - a;
        }
        const int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
        t0[0] = t0[x0 + 1];
        t1[0] = t1[x0 + 1];
        t0[w + 1] = t0[x1 + 1];
        t1[w + 1] = t1[x1 + 1];
        int16_t* d = &L.deriv[((size_t)(y + L.pad) * L.pitch + L.pad) * 2];
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
}

int build_pyramid(const uint8_t* img, int w, int h, int stride, int win_w, int win_h, int max_level,
                  std::vector<Level>& pyr) {
    const int pad = std::max(win_w, win_h);
    pyr.clear();
    int lw = w, lh = h;
    for (int level = 0; level <= max_level; level++) {
        pyr.emplace_back();
        Level& L = pyr.back();
        alloc_level(L, lw, lh, pad);
        if (level == 0) {
            for (int y = 0; y < h; y++) memcpy(&L.img[(size_t)(y + pad) * L.pitch + pad], img + (size_t)y * stride, w);
        } else {
            pyr_down(pyr[level - 1], L);
        }
        fill_border(L);
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
        if (lw <= win_w || lh <= win_h) return level;
    }
    return max_level;
}

#define CV_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

}  // namespace

extern "C" {

void vo_oracle_lk_params_default(vo_lk_params* p) {
    p->win_w = 21;
    p->win_h = 21;
    p->max_level = 3;
    p->max_count = 30;
    p->epsilon = 0.01;
    p->min_eig_threshold = 1e-4;
    p->channels = 3;
}

int vo_oracle_lk_levels(int w, int h, const vo_lk_params* prm, int* lw, int* lh) {
    int n = 0, cw = w, ch = h;
    for (int level = 0; level <= prm->max_level; level++) {
        lw[n] = cw;
        lh[n] = ch;
        n++;
        cw = (cw + 1) / 2;
        ch = (ch + 1) / 2;
        if (cw <= prm->win_w || ch <= prm->win_h) break;
    }
    return n;
}

int vo_oracle_lk_level(const uint8_t* img, int w, int h, int stride, const vo_lk_params* prm, int level,
                       uint8_t* out_img, int16_t* out_deriv) {
    std::vector<Level> pyr;
    const int top = build_pyramid(img, w, h, stride, prm->win_w, prm->win_h, prm->max_level, pyr);
    if (level < 0 || level > top) return VO_ERR_INVALID;
    Level& L = pyr[level];
    scharr(L);
    for (int y = 0; y < L.h; y++)
        for (int x = 0; x < L.w; x++) {
            if (out_img) out_img[(size_t)y * L.w + x] = *L.px(x, y);
            if (out_deriv) {
                out_deriv[((size_t)y * L.w + x) * 2] = L.d(x, y)[0];
                out_deriv[((size_t)y * L.w + x) * 2 + 1] = L.d(x, y)[1];
            }
        }
    return VO_OK;
}

int vo_oracle_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h, int stride, const float* prev_pts,
                       int n, const vo_lk_params* prm, float* next_pts, uint8_t* status) {
    if (!prev || !next || !prm || n < 0 || (n > 0 && (!prev_pts || !next_pts || !status))) return VO_ERR_INVALID;
    const int ww = prm->win_w, wh = prm->win_h;
    if (prm->max_level < 0 || ww <= 2 || wh <= 2 || (prm->channels != 1 && prm->channels != 3)) return VO_ERR_INVALID;
    std::vector<Level> P, N;
    const int max_level = build_pyramid(prev, w, h, stride, ww, wh, prm->max_level, P);
    build_pyramid(next, w, h, stride, ww, wh, prm->max_level, N);
    const int max_count = std::min(std::max(prm->max_count, 0), 100);
    double eps = std::min(std::max(prm->epsilon, 0.), 10.);
    eps *= eps;
    const long long cn = prm->channels;
    for (int i = 0; i < n; i++) status[i] = 1;
    const float half_x = (ww - 1) * 0.5f, half_y = (wh - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    std::vector<int16_t> Iw((size_t)ww * wh), Dw((size_t)ww * wh * 2);
    for (int level = max_level; level >= 0; level--) {
        Level& I = P[level];
        const Level& J = N[level];
        scharr(I);
        for (int pt = 0; pt < n; pt++) {
            const float sc = (float)(1. / (1 << level));
            float px = prev_pts[2 * pt] * sc, py = prev_pts[2 * pt + 1] * sc;
            float nx, ny;
            if (level == max_level) {
                nx = px;
                ny = py;
            } else {
                nx = next_pts[2 * pt] * 2.f;
                ny = next_pts[2 * pt + 1] * 2.f;
            }
            next_pts[2 * pt] = nx;
            next_pts[2 * pt + 1] = ny;
            px -= half_x;
            py -= half_y;
            const int ipx = cvFloor(px), ipy = cvFloor(py);
            if (ipx < -ww || ipx >= I.w || ipy < -wh || ipy >= I.h) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = cvRound((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cvRound(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cvRound((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            long long iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < wh; y++) {
                const uint8_t* s0 = I.px(ipx, ipy + y);
                const uint8_t* s1 = I.px(ipx, ipy + y + 1);
                const int16_t* d0 = I.d(ipx, ipy + y);
                const int16_t* d1 = I.d(ipx, ipy + y + 1);
                for (int x = 0; x < ww; x++) {
                    const int ival = CV_DESCALE(s0[x] * iw00 + s0[x + 1] * iw01 + s1[x] * iw10 + s1[x + 1] * iw11, W_BITS - 5);
                    const int ixval = CV_DESCALE(d0[2 * x] * iw00 + d0[2 * x + 2] * iw01 + d1[2 * x] * iw10 + d1[2 * x + 2] * iw11, W_BITS);
                    const int iyval = CV_DESCALE(d0[2 * x + 1] * iw00 + d0[2 * x + 3] * iw01 + d1[2 * x + 1] * iw10 + d1[2 * x + 3] * iw11, W_BITS);
                    Iw[(size_t)y * ww + x] = (int16_t)ival;
                    Dw[((size_t)y * ww + x) * 2] = (int16_t)ixval;
                    Dw[((size_t)y * ww + x) * 2 + 1] = (int16_t)iyval;
                    iA11 += ixval * ixval;
                    iA12 += ixval * iyval;
                    iA22 += iyval * iyval;
                }
            }
            const float A11 = (float)(iA11 * cn) * FLT_SCALE, A12 = (float)(iA12 * cn) * FLT_SCALE,
                        A22 = (float)(iA22 * cn) * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * ww * wh);
            if (minEig < (float)prm->min_eig_threshold || D < 1.1920928955078125e-07F) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half_x;
            ny -= half_y;
            float pdx = 0, pdy = 0;
            for (int j = 0; j < max_count; j++) {
                const int inx = cvFloor(nx), iny = cvFloor(ny);
                if (inx < -ww || inx >= J.w || iny < -wh || iny >= J.h) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nx - inx;
                b = ny - iny;
                iw00 = cvRound((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cvRound(a * (1.f - b) * (1 << W_BITS));
                iw10 = cvRound((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                long long ib1 = 0, ib2 = 0;
                for (int y = 0; y < wh; y++) {
                    const uint8_t* j0 = J.px(inx, iny + y);
                    const uint8_t* j1 = J.px(inx, iny + y + 1);
                    for (int x = 0; x < ww; x++) {
                        const int diff = CV_DESCALE(j0[x] * iw00 + j0[x + 1] * iw01 + j1[x] * iw10 + j1[x + 1] * iw11, W_BITS - 5) -
                                         Iw[(size_t)y * ww + x];
                        ib1 += diff * Dw[((size_t)y * ww + x) * 2];
                        ib2 += diff * Dw[((size_t)y * ww + x) * 2 + 1];
                    }
                }
                const float b1 = (float)(ib1 * cn) * FLT_SCALE, b2 = (float)(ib2 * cn) * FLT_SCALE;
                const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                nx += dx;
                ny += dy;
                next_pts[2 * pt] = nx + half_x;
                next_pts[2 * pt + 1] = ny + half_y;
                if ((double)dx * dx + (double)dy * dy <= eps) break;
                if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
                    next_pts[2 * pt] -= dx * 0.5f;
                    next_pts[2 * pt + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx;
                pdy = dy;
            }
        }
    }
    return VO_OK;
}

/* removePoints(vo0, vo1, &status): the erase loop runs from the back, so the decision for index i
 * only ever sees original data at i and the result is a stable compaction; `dists` is sorted first
 * and then erased by ORIGINAL index (the reference's quirk), the return value is the middle
 * element of what is left, truncated to int by the function's return type. */
int vo_oracle_remove_points(float* p0, float* p1, int n, uint8_t* status, double* pts3d, int n3d, uint8_t* mask,
                            int nmask, int* n_out, int* n3d_out, int* nmask_out, int* ret) {
    std::vector<float> dists(n);
    for (int i = 0; i < n; i++) {
        const float dx = p0[2 * i] - p1[2 * i], dy = p0[2 * i + 1] - p1[2 * i + 1];
        dists[i] = (float)std::sqrt((double)dx * dx + (double)dy * dy);
    }
    std::vector<float> orig = dists;
    std::sort(dists.begin(), dists.end());
    const double sum = std::accumulate(dists.begin(), dists.end(), 0.0);
    const double mean = sum / dists.size();
    const double sq_sum = std::inner_product(dists.begin(), dists.end(), dists.begin(), 0.0);
    const double stdev = std::sqrt(sq_sum / dists.size() - mean * mean);
    int k = 0, k3 = 0, km = 0;
    std::vector<float> kept;
    for (int i = 0; i < n; i++) {
        const float x = p1[2 * i], y = p1[2 * i + 1];
        const float dist = orig[i];
        const bool drop = status[i] == 0 || x < 0 || y < 0 || std::abs(dist - mean) > stdev * 2;
        if (drop) {
            if (x < 0 || y < 0) status[i] = 0;
            continue;
        }
        p0[2 * k] = p0[2 * i];
        p0[2 * k + 1] = p0[2 * i + 1];
        p1[2 * k] = x;
        p1[2 * k + 1] = y;
        k++;
        kept.push_back(dists[i]);
        if (i < n3d) {
            for (int c = 0; c < 3; c++) pts3d[3 * k3 + c] = pts3d[3 * i + c];
            k3++;
        }
        if (i < nmask) mask[km++] = mask[i];
    }
    for (int i = n; i < n3d; i++, k3++)
        for (int c = 0; c < 3; c++) pts3d[3 * k3 + c] = pts3d[3 * i + c];
    for (int i = n; i < nmask; i++) mask[km++] = mask[i];
    *n_out = k;
    if (n3d_out) *n3d_out = k3;
    if (nmask_out) *nmask_out = km;
    *ret = kept.empty() ? 0 : (int)kept[kept.size() / 2];
    return VO_OK;
}

}  // extern "C"
