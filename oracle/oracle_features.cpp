// oracle_features.cpp — TEST INFRASTRUCTURE ONLY (see vo_oracle.h; parity unpinned).
// CPU restatement of the OpenCV 3.4.x feature code the reference calls:
//   FAST-9/16 + NMS      (features2d/src/fast.cpp)            SURVEY.md A.1
//   ORB detectAndCompute (features2d/src/orb.cpp)              SURVEY.md A.2
//   resize INTER_LINEAR_EXACT 8u (imgproc/src/resize.cpp), GaussianBlur 7x7 8u (imgproc/src/smooth.cpp)
// Reference call sites: src/features/feature_detector.cpp:4-5,15,24,28,41.
// Build with -ffp-contract=off: every float expression below is evaluated op by op.
#include "oracle_common.h"

#include <cfloat>

namespace vo_oracle {

// ---------------------------------------------------------------------------------------
// FAST-9/16
// ---------------------------------------------------------------------------------------
static const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// cornerScore<16>: largest threshold for which the pixel is still a 9-of-16 corner.
static int corner_score16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

static bool is_corner16(const uint8_t* ptr, const int* pixel, int t) {
    int v = ptr[0];
    uint32_t dark = 0, bright = 0;
    for (int k = 0; k < 16; k++) {
        int p = ptr[pixel[k]];
        if (p < v - t) dark |= 1u << k;
        if (p > v + t) bright |= 1u << k;
    }
    auto run9 = [](uint32_t m) {
        uint32_t x = m | (m << 16);
        uint32_t r = x & (x >> 1);
        r = r & (r >> 2);
        r = r & (r >> 4);
        r = r & (x >> 8);
        return r != 0;
    };
    return run9(dark) || run9(bright);
}

struct FastPt {
    int x, y, score;
};

#ifndef VO_CPU_BASELINE  // cpu_baseline/ replaces this definition by an optimised one with equal results
static void fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, bool nms,
                     std::vector<FastPt>& out) {
    out.clear();
    if (w < 7 || h < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    int pixel[25];
    for (int k = 0; k < 25; k++) pixel[k] = kRingDx[k % 16] + kRingDy[k % 16] * stride;
    // score map: 0 for non-corners (cv keeps three rolling rows; a full map is equivalent)
    std::vector<uint8_t> score((size_t)w * h, 0);
    std::vector<uint8_t> corner((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (is_corner16(p, pixel, threshold)) {
                corner[(size_t)y * w + x] = 1;
                if (nms) score[(size_t)y * w + x] = (uint8_t)corner_score16(p, pixel, threshold);
            }
        }
    }
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            size_t i = (size_t)y * w + x;
            if (!corner[i]) continue;
            int s = score[i];
            if (!nms) {
                out.push_back({x, y, 0});  // cv pushes the rolling-buffer score, which is 0 without nms
                continue;
            }
            if (s > score[i - 1] && s > score[i + 1] && s > score[i - w - 1] && s > score[i - w] &&
                s > score[i - w + 1] && s > score[i + w - 1] && s > score[i + w] &&
                s > score[i + w + 1])
                out.push_back({x, y, s});
        }
    }
}

#endif

// ---------------------------------------------------------------------------------------
// resize INTER_LINEAR_EXACT, 8-bit: ufixedpoint16 (8 fractional bits) coefficients,
// horizontal pass exact in 16 bits, vertical pass (sum + 2^15) >> 16.
// ---------------------------------------------------------------------------------------
struct LinCoef {
    int ofs;
    int c0, c1;  // 8.8 fixed point; c0 + c1 == 256
};

static void linear_coeffs(int ssize, int dsize, std::vector<LinCoef>& co) {
    co.resize(dsize);
    double inv_scale = (double)dsize / ssize;
    double scale = 1.0 / inv_scale;
    for (int v = 0; v < dsize; v++) {
        double fval = scale * ((double)v + 0.5) - 0.5;
        int ival = cvFloor(fval);
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                int c1 = cvRound((fval - (double)ival) * 256.0);
                co[v] = {ival, 256 - c1, c1};
            } else {
                co[v] = {ssize - 1, 256, 0};  // past the end: last pixel
            }
        } else {
            co[v] = {0, 256, 0};  // before the start: first pixel
        }
    }
}

static void resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst,
                                int dw, int dh, int dstride) {
    std::vector<LinCoef> cx, cy;
    linear_coeffs(sw, dw, cx);
    linear_coeffs(sh, dh, cy);
    std::vector<uint16_t> h0(dw), h1(dw);
    auto hline = [&](const uint8_t* row, std::vector<uint16_t>& out) {
        for (int x = 0; x < dw; x++) {
            const LinCoef& c = cx[x];
            int s0 = row[c.ofs];
            int s1 = row[std::min(c.ofs + 1, sw - 1)];
            out[x] = (uint16_t)(c.c0 * s0 + c.c1 * s1);
        }
    };
    for (int y = 0; y < dh; y++) {
        const LinCoef& c = cy[y];
        hline(src + (size_t)c.ofs * sstride, h0);
        hline(src + (size_t)std::min(c.ofs + 1, sh - 1) * sstride, h1);
        for (int x = 0; x < dw; x++) {
            uint32_t v = (uint32_t)c.c0 * h0[x] + (uint32_t)c.c1 * h1[x];
            uint32_t r = (v + 32768u) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)std::min(r, 255u);
        }
    }
}

// ---------------------------------------------------------------------------------------
// GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101, 8-bit fixed-point separable filter
// (taps x256 = cvRound(getGaussianKernel(7,2)*256), row pass int32, column pass >>16).
// ---------------------------------------------------------------------------------------
static void gauss7_taps(int taps[7]) {
    // cv::getGaussianKernel(7, 2, CV_32F)
    const int n = 7;
    const double sigma = 2.0;
    double scale2X = -0.5 / (sigma * sigma);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        taps[i] = cvRound((double)cf[i] * 256.0);  // Mat::convertTo(CV_32S, 1<<8)
    }
}

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

#ifndef VO_CPU_BASELINE  // see fast9_16
static void gauss7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    int taps[7];
    gauss7_taps(taps);
    std::vector<int32_t> rows((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int32_t acc = 0;
            for (int k = -3; k <= 3; k++) acc += taps[k + 3] * (int32_t)s[reflect101(x + k, w)];
            rows[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int32_t acc = 0;
            for (int k = -3; k <= 3; k++)
                acc += taps[k + 3] * rows[(size_t)reflect101(y + k, h) * w + x];
            int32_t v = (acc + (1 << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

#else
#include "../cpu_baseline/cpu_kernels.inc"
#endif

// ---------------------------------------------------------------------------------------
// ORB
// ---------------------------------------------------------------------------------------
static const int8_t kBitPattern31[256 * 4] = {
#include "orb_pattern.inc"
};

struct OrbLayout {
    int nlevels;
    int lw[VO_ORB_MAX_LEVELS], lh[VO_ORB_MAX_LEVELS], quota[VO_ORB_MAX_LEVELS];
    float scale[VO_ORB_MAX_LEVELS];
};

static bool orb_params_ok(const vo_orb_params* p) {
    return p && p->nfeatures >= 0 && p->nlevels >= 1 && p->nlevels <= VO_ORB_MAX_LEVELS &&
           p->first_level == 0 && p->wta_k == 2 && p->score_type == 0 && p->patch_size == 31 &&
           p->scale_factor > 1.0f && p->edge_threshold >= 0;
}

static void orb_layout(int w, int h, const vo_orb_params* p, OrbLayout& L) {
    L.nlevels = p->nlevels;
    double scaleFactor = (double)p->scale_factor;  // cv stores the float argument in a double
    for (int l = 0; l < L.nlevels; l++) {
        float s = (float)std::pow(scaleFactor, (double)l);
        L.scale[l] = s;
        L.lw[l] = cvRound(w / s);
        L.lh[l] = cvRound(h / s);
    }
    float factor = (float)(1.0 / scaleFactor);
    float ndesired =
        p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L.nlevels));
    int sum = 0;
    for (int l = 0; l < L.nlevels - 1; l++) {
        L.quota[l] = cvRound(ndesired);
        sum += L.quota[l];
        ndesired *= factor;
    }
    L.quota[L.nlevels - 1] = std::max(p->nfeatures - sum, 0);
}

struct Level {
    int w, h;
    std::vector<uint8_t> px;  // tight, stride == w
};

static void build_pyramid(const uint8_t* gray, int w, int h, int stride, const OrbLayout& L,
                          int nlevels, std::vector<Level>& pyr) {
    pyr.resize(nlevels);
    pyr[0].w = w;
    pyr[0].h = h;
    pyr[0].px.resize((size_t)w * h);
    for (int y = 0; y < h; y++) memcpy(&pyr[0].px[(size_t)y * w], gray + (size_t)y * stride, w);
    for (int l = 1; l < nlevels; l++) {
        pyr[l].w = L.lw[l];
        pyr[l].h = L.lh[l];
        pyr[l].px.resize((size_t)L.lw[l] * L.lh[l]);
        // each level is resized from the previous level, not from level 0
        resize_linear_exact(pyr[l - 1].px.data(), pyr[l - 1].w, pyr[l - 1].h, pyr[l - 1].w,
                            pyr[l].px.data(), pyr[l].w, pyr[l].h, pyr[l].w);
    }
}

// KeyPointsFilter::retainBest: nth_element, threshold = kp[n-1].response, partition the tail on
// response >= threshold.  The permutation is libstdc++'s.  Where nth_element splits is VERSION-SENSITIVE and
// therefore a named option (vo_oracle_set_retain_best_split): OpenCV 3.4.2 - the release the reference pins,
// README.md:23-25 - calls nth_element(begin, begin + n, end), so kp[n-1] is merely one of the n best, not
// the n-th; later OpenCV releases call nth_element(begin, begin + n - 1, end).  The two leave different
// permutations (and, on ties at the threshold, can keep different sets), so every keypoint index differs.
// Default and product path: the 3.4.2 form.  tests/test_cv2_pin.py picks the form from cv2.__version__.
static int g_retain_split = VO_ORACLE_RETAIN_CV342;
extern "C" void vo_oracle_set_retain_best_split(int mode) {
    g_retain_split = mode == VO_ORACLE_RETAIN_LATER ? VO_ORACLE_RETAIN_LATER : VO_ORACLE_RETAIN_CV342;
}
extern "C" int vo_oracle_get_retain_best_split(void) { return g_retain_split; }
struct RespGreater {
    bool operator()(const vo_keypoint& a, const vo_keypoint& b) const {
        return a.response > b.response;
    }
};
static void retain_best(std::vector<vo_keypoint>& kps, int n_points) {
    if (n_points >= 0 && kps.size() > (size_t)n_points) {
        if (n_points == 0) {
            kps.clear();
            return;
        }
        std::nth_element(kps.begin(), kps.begin() + n_points - (g_retain_split == VO_ORACLE_RETAIN_LATER ? 1 : 0),
                         kps.end(), RespGreater());
        float ambiguous = kps[n_points - 1].response;
        auto new_end = std::partition(kps.begin() + n_points, kps.end(),
                                      [ambiguous](const vo_keypoint& k) { return k.response >= ambiguous; });
        kps.resize(new_end - kps.begin());
    }
}

// HarrisResponses(img, ..., blockSize 7, k 0.04f)
static float harris_response(const Level& lv, int x0, int y0) {
    const int blockSize = 7;
    const float harris_k = 0.04f;
    const int r = blockSize / 2;
    const int step = lv.w;
    float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    const uint8_t* ptr00 = lv.px.data() + (size_t)(y0 - r) * step + (x0 - r);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; i++)
        for (int j = 0; j < blockSize; j++) {
            const uint8_t* ptr = ptr00 + i * step + j;
            int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) +
                     (ptr[step + 1] - ptr[step - 1]);
            int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) +
                     (ptr[step + 1] - ptr[-step + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

static void compute_umax(int umax[17]) {
    const int halfPatchSize = 15;
    int v, v0, vmax = cvFloor(halfPatchSize * std::sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(halfPatchSize * std::sqrt(2.f) / 2);
    for (v = 0; v <= 16; v++) umax[v] = 0;
    for (v = 0; v <= vmax; ++v)
        umax[v] = cvRound(std::sqrt((double)halfPatchSize * halfPatchSize - v * v));
    for (v = halfPatchSize, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// cv::fastAtan2 (scalar path of core/src/mathfuncs_core), degrees in [0,360)
static float fast_atan2(float y, float x) {
    const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static float ic_angle(const Level& lv, int x0, int y0, const int* umax) {
    const int half_k = 15;
    const int step = lv.w;
    const uint8_t* center = lv.px.data() + (size_t)y0 * step + x0;
    int m_01 = 0, m_10 = 0;
    for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
    for (int v = 1; v <= half_k; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// computeOrbDescriptors, WTA_K = 2
static void orb_descriptor(const Level& blurred, float layer_scale, const vo_keypoint& kpt,
                           uint8_t* desc) {
    const int step = blurred.w;
    float scale = 1.f / layer_scale;
    float angle = kpt.angle;
    angle *= (float)(M_PI / 180.f);
    float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
    int cy = cvRound(kpt.y * scale), cx = cvRound(kpt.x * scale);
    const uint8_t* center = blurred.px.data() + (size_t)cy * step + cx;
    const int8_t* pattern = kBitPattern31;
    auto get_value = [&](int idx) -> int {
        float px = (float)pattern[idx * 2], py = (float)pattern[idx * 2 + 1];
        float x = px * a - py * b;
        float y = px * b + py * a;
        int ix = cvRound(x), iy = cvRound(y);
        return center[iy * step + ix];
    };
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int t0 = get_value(2 * k), t1 = get_value(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

static void level_candidates(const Level& lv, const vo_orb_params* p, std::vector<vo_keypoint>& kps) {
    std::vector<FastPt> fp;
    fast9_16(lv.px.data(), lv.w, lv.h, lv.w, p->fast_threshold, true, fp);
    kps.clear();
    const int b = p->edge_threshold;
    for (const FastPt& f : fp) {
        // KeyPointsFilter::runByImageBorder: Rect(b, b, w-2b, h-2b).contains(pt)
        if (lv.w <= 2 * b || lv.h <= 2 * b) {
            // cv: a non-positive-size rect clears everything (borderSize > 0 and too small image)
            continue;
        }
        if (f.x >= b && f.x < lv.w - b && f.y >= b && f.y < lv.h - b) {
            vo_keypoint k;
            k.x = (float)f.x;
            k.y = (float)f.y;
            k.size = 7.f;
            k.angle = -1.f;
            k.response = (float)f.score;
            k.octave = 0;
            k.class_id = -1;
            kps.push_back(k);
        }
    }
}

static int orb_detect_compute(const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                              std::vector<vo_keypoint>& out_kps, std::vector<uint8_t>& out_desc) {
    OrbLayout L;
    orb_layout(w, h, p, L);
    std::vector<Level> pyr;
    build_pyramid(gray, w, h, stride, L, L.nlevels, pyr);
    int umax[17];
    compute_umax(umax);

    std::vector<vo_keypoint> all;
    std::vector<int> counters(L.nlevels);
    for (int l = 0; l < L.nlevels; l++) {
        std::vector<vo_keypoint> kps;
        level_candidates(pyr[l], p, kps);
        retain_best(kps, 2 * L.quota[l]);  // HARRIS_SCORE: keep twice the quota by FAST score
        counters[l] = (int)kps.size();
        for (auto& k : kps) {
            k.octave = l;
            k.size = 31 * L.scale[l];
        }
        all.insert(all.end(), kps.begin(), kps.end());
    }
    // Harris responses, then cull to the quota per level
    for (auto& k : all) k.response = harris_response(pyr[k.octave], cvRound(k.x), cvRound(k.y));
    std::vector<vo_keypoint> fin;
    int offset = 0;
    for (int l = 0; l < L.nlevels; l++) {
        std::vector<vo_keypoint> kps(all.begin() + offset, all.begin() + offset + counters[l]);
        offset += counters[l];
        retain_best(kps, L.quota[l]);
        fin.insert(fin.end(), kps.begin(), kps.end());
    }
    for (auto& k : fin) k.angle = ic_angle(pyr[k.octave], cvRound(k.x), cvRound(k.y), umax);
    for (auto& k : fin) {
        float s = L.scale[k.octave];
        k.x *= s;
        k.y *= s;
    }
    // descriptors on the blurred pyramid
    std::vector<Level> blur(L.nlevels);
    for (int l = 0; l < L.nlevels; l++) {
        blur[l].w = pyr[l].w;
        blur[l].h = pyr[l].h;
        blur[l].px.resize(pyr[l].px.size());
        gauss7(pyr[l].px.data(), pyr[l].w, pyr[l].h, pyr[l].w, blur[l].px.data(), pyr[l].w);
    }
    out_kps = fin;
    out_desc.assign(fin.size() * 32, 0);
    for (size_t i = 0; i < fin.size(); i++)
        orb_descriptor(blur[fin[i].octave], L.scale[fin[i].octave], fin[i], &out_desc[i * 32]);
    return 0;
}

}  // namespace vo_oracle

using namespace vo_oracle;

extern "C" {

int vo_oracle_fast_detect(const uint8_t* gray, int w, int h, int stride, int threshold, int nms,
                          vo_keypoint* out, int cap, int* n) {
    if (!gray || !n || w <= 0 || h <= 0 || stride < w) return VO_ERR_INVALID;
    std::vector<FastPt> fp;
    fast9_16(gray, w, h, stride, threshold, nms != 0, fp);
    *n = (int)fp.size();
    for (size_t i = 0; i < fp.size() && (int)i < cap; i++) {
        vo_keypoint k;
        k.x = (float)fp[i].x;
        k.y = (float)fp[i].y;
        k.size = 7.f;
        k.angle = -1.f;
        k.response = (float)fp[i].score;
        k.octave = 0;
        k.class_id = -1;
        out[i] = k;
    }
    return (int)fp.size() > cap ? VO_ERR_CAPACITY : VO_OK;
}

int vo_oracle_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst,
                                  int dw, int dh, int dstride) {
    if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return VO_ERR_INVALID;
    resize_linear_exact(src, sw, sh, sstride, dst, dw, dh, dstride);
    return VO_OK;
}

int vo_oracle_gauss7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    if (!src || !dst || w <= 0 || h <= 0) return VO_ERR_INVALID;
    gauss7(src, w, h, sstride, dst, dstride);
    return VO_OK;
}

void vo_oracle_orb_params_default(vo_orb_params* p, int nfeatures) {
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

int vo_oracle_orb_layout(int w, int h, const vo_orb_params* p, int* lw, int* lh, float* lscale,
                         int* quota) {
    if (!orb_params_ok(p)) return -1;
    OrbLayout L;
    orb_layout(w, h, p, L);
    for (int l = 0; l < L.nlevels; l++) {
        if (lw) lw[l] = L.lw[l];
        if (lh) lh[l] = L.lh[l];
        if (lscale) lscale[l] = L.scale[l];
        if (quota) quota[l] = L.quota[l];
    }
    return L.nlevels;
}

int vo_oracle_orb_level(const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                        int level, int blurred, uint8_t* out) {
    if (!orb_params_ok(p) || level < 0 || level >= p->nlevels) return VO_ERR_INVALID;
    OrbLayout L;
    orb_layout(w, h, p, L);
    std::vector<Level> pyr;
    build_pyramid(gray, w, h, stride, L, level + 1, pyr);
    const Level& lv = pyr[level];
    if (blurred) gauss7(lv.px.data(), lv.w, lv.h, lv.w, out, lv.w);
    else memcpy(out, lv.px.data(), lv.px.size());
    return VO_OK;
}

int vo_oracle_orb_candidates(const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                             int level, int* xs, int* ys, float* fast_score, float* harris, int cap,
                             int* n) {
    if (!orb_params_ok(p) || level < 0 || level >= p->nlevels) return VO_ERR_INVALID;
    OrbLayout L;
    orb_layout(w, h, p, L);
    std::vector<Level> pyr;
    build_pyramid(gray, w, h, stride, L, level + 1, pyr);
    std::vector<vo_keypoint> kps;
    level_candidates(pyr[level], p, kps);
    *n = (int)kps.size();
    for (size_t i = 0; i < kps.size() && (int)i < cap; i++) {
        xs[i] = (int)kps[i].x;
        ys[i] = (int)kps[i].y;
        fast_score[i] = kps[i].response;
        harris[i] = harris_response(pyr[level], xs[i], ys[i]);
    }
    return (int)kps.size() > cap ? VO_ERR_CAPACITY : VO_OK;
}

int vo_oracle_orb_detect_compute(const uint8_t* gray, int w, int h, int stride,
                                 const vo_orb_params* p, vo_keypoint* kp, uint8_t* desc, int cap,
                                 int* n) {
    if (!gray || !orb_params_ok(p) || !n) return VO_ERR_INVALID;
    std::vector<vo_keypoint> kps;
    std::vector<uint8_t> d;
    orb_detect_compute(gray, w, h, stride, p, kps, d);
    *n = (int)kps.size();
    if ((int)kps.size() > cap) return VO_ERR_CAPACITY;
    if (kp) memcpy(kp, kps.data(), kps.size() * sizeof(vo_keypoint));
    if (desc) memcpy(desc, d.data(), d.size());
    return VO_OK;
}

int vo_oracle_orb_compute(const uint8_t* gray, int w, int h, int stride, const vo_orb_params* p,
                          vo_keypoint* kp, int* n_inout, uint8_t* desc) {
    if (!gray || !orb_params_ok(p) || !n_inout || !kp) return VO_ERR_INVALID;
    int n = *n_inout;
    // KeyPointsFilter::runByImageBorder(keypoints, image.size(), edgeThreshold) on level-0 coords
    std::vector<vo_keypoint> kps;
    const int b = p->edge_threshold;
    for (int i = 0; i < n; i++) {
        const vo_keypoint& k = kp[i];
        if (w > 2 * b && h > 2 * b && k.x >= (float)b && k.x < (float)(w - b) && k.y >= (float)b &&
            k.y < (float)(h - b))
            kps.push_back(k);
    }
    int nlevels = 0;
    for (auto& k : kps) {
        if (k.octave < 0 || k.octave >= VO_ORB_MAX_LEVELS) return VO_ERR_INVALID;
        nlevels = std::max(nlevels, k.octave);
    }
    nlevels++;
    // group by level, stable
    std::vector<vo_keypoint> sorted;
    for (int l = 0; l < nlevels; l++)
        for (auto& k : kps)
            if (k.octave == l) sorted.push_back(k);
    vo_orb_params pp = *p;
    pp.nlevels = std::max(pp.nlevels, nlevels);
    OrbLayout L;
    orb_layout(w, h, &pp, L);
    std::vector<Level> pyr;
    build_pyramid(gray, w, h, stride, L, nlevels, pyr);
    std::vector<Level> blur(nlevels);
    for (int l = 0; l < nlevels; l++) {
        blur[l].w = pyr[l].w;
        blur[l].h = pyr[l].h;
        blur[l].px.resize(pyr[l].px.size());
        gauss7(pyr[l].px.data(), pyr[l].w, pyr[l].h, pyr[l].w, blur[l].px.data(), pyr[l].w);
    }
    for (size_t i = 0; i < sorted.size(); i++) {
        kp[i] = sorted[i];
        orb_descriptor(blur[sorted[i].octave], L.scale[sorted[i].octave], sorted[i], desc + i * 32);
    }
    *n_inout = (int)sorted.size();
    return VO_OK;
}

}  // extern "C"
