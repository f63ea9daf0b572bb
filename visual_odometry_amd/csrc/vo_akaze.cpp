// vo_akaze.cpp — host side of cv::AKAZE::create()->detectAndCompute (MLDB, 486 bits), SURVEY.md 8f row 4.
// Replaces FeatureDetector::detectComputeAKAZE (reference: src/features/feature_detector.cpp:7,43-45), whose only
// caller is BundleAdjustment::addKeyFrame (src/sfm/bundle_adjustment.cpp:41): a key-frame-rate operation.
//
// Two device phases around one short host step:
//   1. device  non-linear scale space (FED cycles of Perona-Malik g2 diffusion), multiscale Scharr derivatives,
//              Hessian determinant, 3 x 3 extrema masks of every level                       (k_akaze.hip)
//   2. host    cross-level duplicate suppression and sub-pixel refinement of the few thousand extrema.  The rule is
//              sequential by definition (a keypoint removes the FIRST weaker neighbour a scan of the adjacent level
//              finds, and what one keypoint removes changes what the next one finds), so it runs in the order cv
//              runs it, on the list of maxima the device appended (position + 3 x 3 determinants: ~0.2 MB)
//   3. device  main orientation and the MLDB descriptor of every keypoint
// Every float expression is evaluated in the order of the scalar CPU form (-ffp-contract=off).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "ctx.h"

namespace {

struct AkGeom {  // one evolution level
    int w, h, octave, sigma_size, border;
    float esigma, etime, ratio;
    size_t at;  // pixel offset of the level inside a per-plane section
};

int round_half_even(float v) { return (int)std::lrintf(v); }
int ceil_int(double v) {
    const int i = (int)v;
    return i + (i < v);
}
int floor_int(double v) {
    const int i = (int)v;
    return i - (i > v);
}

bool ak_params_ok(const vo_akaze_params* p) {
    if (!p || p->n_octaves < 1 || p->n_octaves > AK_MAX_OCTAVES || p->n_sublevels < 1 ||
        p->n_octaves * p->n_sublevels > AK_MAX_LEVELS || p->channels < 1 || p->channels > 3 || p->pattern_size < 4 ||
        p->pattern_size > 16 || (p->pattern_size & 1) || p->kcontrast_nbins < 3 || p->kcontrast_nbins > AK_MAX_BINS ||
        !(p->soffset > 0.8f) || !(p->soffset <= 3.2f) || !(p->derivative_factor > 0) || !(p->threshold >= 0) ||
        !(p->kcontrast_percentile > 0 && p->kcontrast_percentile < 1))
        return false;
    // the three grids must be 2 x 2, 3 x 3 and 4 x 4 (true for cv's default 10 and most even sizes)
    const int ps = p->pattern_size, s1 = (ps * 2 + 2) / 3, s2 = (ps + 1) / 2;
    return (2 * ps + s1 - 1) / s1 == 3 && (2 * ps + s2 - 1) / s2 == 4;
}

// AKAZEFeatures::Allocate_Memory_Evolution
std::vector<AkGeom> ak_layout(int w, int h, const vo_akaze_params& p, size_t* total) {
    std::vector<AkGeom> ev;
    const float smax = 10.0f * std::sqrt(2.0f);
    size_t at = 0;
    for (int o = 0, power = 1; o < p.n_octaves; o++, power *= 2) {
        const float rfactor = 1.0f / (float)power;
        const int lh = (int)((float)h * rfactor), lw = (int)((float)w * rfactor);
        if ((lw < 80 || lh < 40) && o != 0) break;
        for (int j = 0; j < p.n_sublevels; j++) {
            AkGeom g;
            g.w = lw;
            g.h = lh;
            g.octave = o;
            g.esigma = p.soffset * std::pow(2.f, (float)j / (float)p.n_sublevels + (float)o);
            g.sigma_size = round_half_even(g.esigma * p.derivative_factor / (float)power);
            g.etime = 0.5f * (g.esigma * g.esigma);
            g.ratio = (float)power;
            g.border = round_half_even(smax * (float)g.sigma_size) + 1;
            g.at = at;
            at += (size_t)lw * lh;
            ev.push_back(g);
        }
    }
    *total = at;
    return ev;
}

// fed.cpp: step sizes of one FED cycle of stopping time T (tau_max 0.25, kappa reordering)
bool is_prime(int n) {
    if (n < 2) return false;
    for (int d = 2; d * d <= n; d++)
        if (n % d == 0) return false;
    return true;
}
std::vector<float> fed_steps(float T) {
    const float tau_max = 0.25f;
    const int n = ceil_int(std::sqrt(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f);
    std::vector<float> tau;
    if (n <= 0) return tau;
    const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
    const float c = 1.0f / (4.0f * (float)n + 2.0f), d = scale * tau_max / 2.0f;
    std::vector<float> plain((size_t)n);
    for (int k = 0; k < n; ++k) {
        const float hh = std::cos((float)M_PI * (2.0f * (float)k + 1.0f) * c);
        plain[(size_t)k] = d / (hh * hh);
    }
    const int kappa = n / 2;
    int prime = n + 1;
    while (!is_prime(prime)) prime++;
    tau.resize((size_t)n);
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[(size_t)l] = plain[(size_t)index];
    }
    return tau;
}

// cv::getGaussianKernel(n, sigma, CV_32F), upper half
AkTaps gauss_taps(int n, float sigma) {
    AkTaps t;
    t.r = n / 2;
    float k[17];
    const double s2 = -0.5 / ((double)sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        k[i] = (float)std::exp(s2 * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int j = 0; j <= t.r; j++) t.k[j] = (float)(k[t.r + j] * sum);
    return t;
}

// computeResizeAreaTab (imgproc/resize.cpp) of one axis
struct AreaAxis {
    std::vector<int> first, si;
    std::vector<float> alpha;
};
AreaAxis area_axis(int ssize, int dsize) {
    AreaAxis t;
    const double scale = (double)ssize / dsize;
    for (int d = 0; d < dsize; d++) {
        t.first.push_back((int)t.si.size());
        const double f1 = d * scale, f2 = f1 + scale, cell = std::min(scale, ssize - f1);
        int s1 = ceil_int(f1), s2 = std::min(floor_int(f2), ssize - 1);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) {
            t.si.push_back(s1 - 1);
            t.alpha.push_back((float)((s1 - f1) / cell));
        }
        for (int s = s1; s < s2; s++) {
            t.si.push_back(s);
            t.alpha.push_back((float)(1.0 / cell));
        }
        if (f2 - s2 > 1e-3) {
            t.si.push_back(s2);
            t.alpha.push_back((float)(std::min(std::min(f2 - s2, 1.), cell) / cell));
        }
    }
    t.first.push_back((int)t.si.size());
    return t;
}

struct HostLevel {  // a level on the host: a dense mask (rebuilt from the device's list) and its maxima, row-major
    const AkGeom* g;
    uint8_t* mask;
    const AkExt* ext;  // sorted by position; ext[k].at - g->at is the pixel index inside the level
    int n;
    int pos(int k) const { return ext[k].at - (int)g->at; }
    float det_at(int p) const {  // determinant at a position that holds a maximum
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (pos(mid) < p) lo = mid + 1;
            else hi = mid;
        }
        return ext[lo].d[4];
    }
};

// first keypoint of `lv` (row-major scan of the window) within `radius` of (x, y); -1 if none
long first_neighbor(const HostLevel& lv, int x, int y, int radius) {
    const int w = lv.g->w, h = lv.g->h;
    const int j0 = std::max(x - radius, 0), j1 = std::min(x + radius, w);
    for (int i = std::max(y - radius, 0); i < std::min(y + radius, h); i++) {
        const uint8_t* row = lv.mask + (size_t)i * w;
        for (int j = j0; j < j1;) {
            if (j + 8 <= j1) {  // the masks are almost empty: eight bytes at a time
                uint64_t v;
                memcpy(&v, row + j, 8);
                if (!v) {
                    j += 8;
                    continue;
                }
            }
            if (row[j]) {
                const int dx = j - x, dy = i - y;
                if (dx * dx + dy * dy <= radius * radius) return (long)i * w + j;
            }
            j++;
        }
    }
    return -1;
}

// Find_Scale_Space_Extrema after the per-level maxima: upwards pass (a keypoint removes a weaker neighbour one
// level down), then downwards from the top (a weaker neighbour one level up).  A level's list holds its maxima
// as the device found them; an entry whose mask byte has been cleared since is skipped, which is what a scan of
// the mask itself would do.
void suppress_duplicates(std::vector<HostLevel>& lv) {
    for (size_t i = 1; i < lv.size(); i++) {
        const AkGeom &e = *lv[i].g, &ep = *lv[i - 1].g;
        const int dr = (int)e.ratio / (int)ep.ratio, radius = e.sigma_size * dr;
        for (int k = 0; k < lv[i].n; k++) {
            const int p = lv[i].pos(k);
            if (!lv[i].mask[p]) continue;
            const int y = p / e.w, x = p - y * e.w;
            const long at = first_neighbor(lv[i - 1], x * dr, y * dr, radius);
            if (at >= 0 && lv[i].ext[k].d[4] > lv[i - 1].det_at((int)at)) lv[i - 1].mask[at] = 0;
        }
    }
    for (size_t i = lv.size() - 1; i-- > 0;) {
        const AkGeom &e = *lv[i].g, &en = *lv[i + 1].g;
        const int dr = (int)en.ratio / (int)e.ratio, radius = en.sigma_size;
        for (int k = 0; k < lv[i].n; k++) {
            const int p = lv[i].pos(k);
            if (!lv[i].mask[p]) continue;
            const int y = p / e.w, x = p - y * e.w;
            const long at = first_neighbor(lv[i + 1], x / dr, y / dr, radius);
            if (at >= 0 && lv[i].ext[k].d[4] > lv[i + 1].det_at((int)at)) lv[i + 1].mask[at] = 0;
        }
    }
}

// Do_Subpixel_Refinement: quadratic fit of the determinant around every surviving maximum (d: its 3 x 3
// neighbourhood, row-major)
void refine(const std::vector<HostLevel>& lv, float derivative_factor, std::vector<vo_keypoint>& out) {
    for (size_t i = 0; i < lv.size(); i++) {
        const AkGeom& e = *lv[i].g;
        for (int k = 0; k < lv[i].n; k++) {
            const int p = lv[i].pos(k);
            if (!lv[i].mask[p]) continue;
            const int y = p / e.w, x = p - y * e.w;
            const float* d = lv[i].ext[k].d;
            const float Dx = 0.5f * (d[5] - d[3]), Dy = 0.5f * (d[7] - d[1]);
            const float Dxx = d[5] + d[3] - 2.0f * d[4], Dyy = d[7] + d[1] - 2.0f * d[4];
            const float Dxy = 0.25f * (d[8] + d[0] - d[2] - d[6]);
            const double det = (double)Dxx * Dyy - (double)Dxy * Dxy;  // 2 x 2 solve, as cv::solve does it
            if (det == 0) continue;
            const double inv = 1. / det;
            const float dx = (float)(((double)-Dx * Dyy - (double)-Dy * Dxy) * inv);
            const float dy = (float)(((double)-Dy * Dxx - (double)-Dx * Dxy) * inv);
            if (std::fabs(dx) > 1.0f || std::fabs(dy) > 1.0f) continue;
            vo_keypoint kp;
            kp.x = (float)x * e.ratio + dx * e.ratio + 0.5f * (e.ratio - 1.0f);
            kp.y = (float)y * e.ratio + dy * e.ratio + 0.5f * (e.ratio - 1.0f);
            kp.size = 2.0f * e.esigma * derivative_factor;
            kp.angle = 0.f;
            kp.response = d[4];
            kp.octave = e.octave;
            kp.class_id = (int)i;
            out.push_back(kp);
        }
    }
}

}  // namespace

extern "C" {

void vo_akaze_params_default(vo_akaze_params* p) {
    if (!p) return;
    p->threshold = 0.001f;
    p->n_octaves = 4;
    p->n_sublevels = 4;
    p->channels = 3;
    p->pattern_size = 10;
    p->soffset = 1.6f;
    p->derivative_factor = 1.5f;
    p->kcontrast_percentile = 0.7f;
    p->kcontrast_nbins = 300;
    p->reserved = 0;
}

int vo_akaze_detect_compute(vo_ctx* c, const uint8_t* gray, int w, int h, int stride, const vo_akaze_params* prm,
                            vo_keypoint* kp, uint8_t* desc, int cap, int* n) {
    if (!c) return VO_ERR_INVALID;
    if (!gray || w < 16 || h < 16 || stride < w || !n || cap < 0 || (cap > 0 && (!kp || !desc)))
        return c->fail(VO_ERR_INVALID, "vo_akaze_detect_compute: bad argument");
    if (!ak_params_ok(prm)) return c->fail(VO_ERR_INVALID, "vo_akaze_detect_compute: unsupported parameters");
    if ((size_t)w * h > ((size_t)1 << 26)) return c->fail(VO_ERR_CAPACITY, "vo_akaze_detect_compute: image too large");
    *n = 0;
    VO_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    static const bool timing = getenv("VO_AKAZE_TIMING") != nullptr;  // phase times to stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_start = now();
    size_t total = 0;
    const std::vector<AkGeom> ev = ak_layout(w, h, *prm, &total);
    const int nl = (int)ev.size();
    const size_t P = (size_t)w * h;

    // ---- workspace: 9 full-size work planes, 5 plane sections over all levels, the contrast record -----------
    enum { W_IMG, W_TMP, W_FLOW, W_LTB, W_SX, W_SY, W_LXX, W_LXY, W_LYY, W_COUNT };
    enum { S_LT, S_SM, S_LX, S_LY, S_DET, S_COUNT };
    const size_t floats = W_COUNT * P + S_COUNT * total;
    VO_TRY(vo_ensure(c, c->d_ak, floats * sizeof(float) + sizeof(AkContrast) + 256));
    // every maximum with its 3 x 3 neighbourhood: strict 3 x 3 maxima cannot touch, so a quarter of the pixels
    // (plus the partial blocks at the edges) bounds their number
    const size_t ext_cap = total / 4 + 4096;
    VO_TRY(vo_ensure(c, c->d_ak_mask, 256 + ext_cap * sizeof(AkExt)));
    float* A = c->d_ak.as<float>();
    auto work = [&](int i) { return A + (size_t)i * P; };
    auto plane = [&](int s, int l) { return A + W_COUNT * P + (size_t)s * total + ev[(size_t)l].at; };
    AkContrast* kc = reinterpret_cast<AkContrast*>(A + floats + (64 - floats % 64) % 64);
    unsigned* d_next = c->d_ak_mask.as<unsigned>();
    AkExt* d_ext = reinterpret_cast<AkExt*>(c->d_ak_mask.as<char>() + 256);

    VO_TRY(vo_xfer_begin(c));
    VO_TRY(vo_upload_image(c, c->d_img, gray, w, h, stride));

    // area-resize tables of the octave changes, one upload
    std::vector<int> itab;
    std::vector<float> ftab;
    struct TabAt {
        size_t xf, xs, yf, ys, xa, ya;
    };
    std::vector<TabAt> tabs((size_t)nl);
    for (int l = 1; l < nl; l++) {
        if (ev[(size_t)l].octave == ev[(size_t)l - 1].octave) continue;
        const AreaAxis xt = area_axis(ev[(size_t)l - 1].w, ev[(size_t)l].w), yt = area_axis(ev[(size_t)l - 1].h, ev[(size_t)l].h);
        TabAt& t = tabs[(size_t)l];
        t.xf = itab.size();
        itab.insert(itab.end(), xt.first.begin(), xt.first.end());
        t.xs = itab.size();
        itab.insert(itab.end(), xt.si.begin(), xt.si.end());
        t.yf = itab.size();
        itab.insert(itab.end(), yt.first.begin(), yt.first.end());
        t.ys = itab.size();
        itab.insert(itab.end(), yt.si.begin(), yt.si.end());
        t.xa = ftab.size();
        ftab.insert(ftab.end(), xt.alpha.begin(), xt.alpha.end());
        t.ya = ftab.size();
        ftab.insert(ftab.end(), yt.alpha.begin(), yt.alpha.end());
    }
    // descriptor comparison table: bit dpos compares cell values a and b (vals[cell * 3 + channel])
    std::vector<AkPair> pairs;
    {
        const int base[3] = {0, 4, 13};
        for (int z = 0; z < 3; z++) {
            const int count = (z + 2) * (z + 2);
            for (int pos = 0; pos < prm->channels; pos++)
                for (int i = 0; i < count; i++)
                    for (int j = i + 1; j < count; j++)
                        pairs.push_back({(unsigned short)((base[z] + i) * 3 + pos), (unsigned short)((base[z] + j) * 3 + pos)});
        }
    }
    const size_t ibytes = itab.size() * sizeof(int), fbytes = ftab.size() * sizeof(float),
                 pbytes = pairs.size() * sizeof(AkPair);
    const size_t foff = (ibytes + 255) & ~(size_t)255, poff = foff + ((fbytes + 255) & ~(size_t)255);
    VO_TRY(vo_ensure(c, c->d_ak_tab, poff + pbytes + 256));
    if (ibytes) VO_TRY(vo_xfer_in(c, c->d_ak_tab.p, itab.data(), ibytes));
    if (fbytes) VO_TRY(vo_xfer_in(c, c->d_ak_tab.as<char>() + foff, ftab.data(), fbytes));
    VO_TRY(vo_xfer_in(c, c->d_ak_tab.as<char>() + poff, pairs.data(), pbytes));
    const int* d_itab = c->d_ak_tab.as<int>();
    const float* d_ftab = reinterpret_cast<const float*>(c->d_ak_tab.as<char>() + foff);
    const AkPair* d_pairs = reinterpret_cast<const AkPair*>(c->d_ak_tab.as<char>() + poff);

    // ---- phase 1: Create_Nonlinear_Scale_Space, Compute_Determinant_Hessian_Response, per-level maxima ------------
    // ~300 short launches whose arguments depend only on the image size, the parameters and the workspace
    // addresses: captured once into a hipGraph and replayed per image (VO_AKAZE_GRAPH=0: plain launches).
    auto enqueue = [&]() -> int {
        VO_HIP(c, hipMemsetAsync(d_next, 0, sizeof(unsigned), st));
        ak_launch_cvt(st, c->d_img.as<uint8_t>(), w, h, work(W_IMG));
        {
            int ksize = ceil_int(2.0f * (1.0f + (prm->soffset - 0.8f) / 0.3f));
            ksize |= 1;
            ak_launch_gauss(st, work(W_IMG), work(W_TMP), plane(S_SM, 0), w, h, gauss_taps(ksize, prm->soffset));
            VO_HIP(c, hipMemcpyAsync(plane(S_LT, 0), plane(S_SM, 0), P * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        const AkTaps g5 = gauss_taps(5, 1.0f);
        {
            // contrast factor from the gradient histogram of the lightly smoothed input (un-normalised Scharr)
            ak_launch_gauss(st, work(W_IMG), work(W_TMP), work(W_FLOW), w, h, g5);
            ak_launch_grad(st, work(W_FLOW), work(W_SX), work(W_SY), w, h, 1, 3.f, 10.f);
            ak_launch_kcontrast(st, work(W_SX), work(W_SY), w, h, prm->kcontrast_percentile, prm->kcontrast_nbins,
                                ev.back().octave + 1, kc);
        }
        for (int l = 1; l < nl; l++) {
            const AkGeom &e = ev[(size_t)l], &ep = ev[(size_t)l - 1];
            const size_t n_px = (size_t)e.w * e.h;
            if (e.octave > ep.octave) {
                const TabAt& t = tabs[(size_t)l];
                ak_launch_area(st, plane(S_LT, l - 1), ep.w, plane(S_LT, l), e.w, e.h, d_itab + t.xf, d_itab + t.xs,
                               d_ftab + t.xa, d_itab + t.yf, d_itab + t.ys, d_ftab + t.ya);
            } else {
                VO_HIP(c, hipMemcpyAsync(plane(S_LT, l), plane(S_LT, l - 1), n_px * sizeof(float),
                                         hipMemcpyDeviceToDevice, st));
            }
            ak_launch_gauss(st, plane(S_LT, l), work(W_TMP), plane(S_SM, l), e.w, e.h, g5);
            ak_launch_cond(st, plane(S_SM, l), e.w, e.h, kc, e.octave, work(W_FLOW));
            // FED cycle on Lt
            const std::vector<float> tau = fed_steps(e.etime - ep.etime);
            {
                // one launch per step, ping-pong with a work plane; the result must end in the level's own plane
                float* cur = plane(S_LT, l);
                float* other = work(W_LTB);
                for (size_t j = 0; j < tau.size(); j++) {
                    ak_launch_nld(st, cur, work(W_FLOW), other, e.w, e.h, tau[j] * 0.5f);
                    std::swap(cur, other);
                }
                if (cur != plane(S_LT, l))
                    VO_HIP(c, hipMemcpyAsync(plane(S_LT, l), cur, n_px * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
        }
        for (int l = 0; l < nl; l++) {
            const AkGeom& e = ev[(size_t)l];
            const float w3 = 10.0f / 3.0f, norm = 1.0f / (2.0f * (float)e.sigma_size * (w3 + 2.0f)), wnorm = w3 * norm;
            ak_launch_grad(st, plane(S_SM, l), plane(S_LX, l), plane(S_LY, l), e.w, e.h, e.sigma_size, norm, wnorm);
            const float s4 = (float)(e.sigma_size * e.sigma_size * e.sigma_size * e.sigma_size);
            ak_launch_hessdet(st, plane(S_LX, l), plane(S_LY, l), e.w, e.h, e.sigma_size, norm, wnorm, s4, plane(S_DET, l));
            ak_launch_extrema(st, plane(S_DET, l), e.w, e.h, e.border, prm->threshold, (int)e.at, d_ext, d_next,
                              (unsigned)ext_cap);
        }
        return VO_OK;
    };
    const bool use_graph = c->opt_akaze_graph != 0;  // VO_OPT_AKAZE_GRAPH
    if (use_graph) {
        vo_ctx::AkGraphKey key;
        memset(&key, 0, sizeof(key));
        key.w = w;
        key.h = h;
        key.prm = *prm;
        key.arena = c->d_ak.p;
        key.mask = c->d_ak_mask.p;
        key.ext_cap = ext_cap;
        key.tab = c->d_ak_tab.p;
        key.img = c->d_img.p;
        if (!c->ak_graph || memcmp(&key, &c->ak_key, sizeof(key)) != 0) {
            if (c->ak_graph) {
                (void)hipGraphExecDestroy(c->ak_graph);
                c->ak_graph = nullptr;
            }
            hipGraph_t g = nullptr;
            VO_HIP(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue();
            const hipError_t e = hipStreamEndCapture(st, &g);
            if (rc != VO_OK) {
                if (g) (void)hipGraphDestroy(g);
                return rc;
            }
            VO_HIP(c, e);
            const hipError_t ei = hipGraphInstantiate(&c->ak_graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            VO_HIP(c, ei);
            c->ak_key = key;
        }
        VO_HIP(c, hipGraphLaunch(c->ak_graph, st));
    } else {
        VO_TRY(enqueue());
    }
    VO_TRY(vo_check_launch(c, "AKAZE scale space"));
    const auto t_queued = now();

    // ---- phase 2 (host): duplicates across levels, sub-pixel positions ------------------------------------------
    // the maxima come back as a list (a few thousand entries of 40 bytes), not as planes
    VO_TRY(vo_ensure_pinned(c, c->h_ak, 256 + ext_cap * sizeof(AkExt)));
    unsigned* h_next = c->h_ak.as<unsigned>();
    AkExt* h_ext = reinterpret_cast<AkExt*>(c->h_ak.as<char>() + 256);
    VO_HIP(c, hipMemcpyAsync(h_next, d_next, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    VO_TRY(vo_xfer_finish(c));
    const size_t n_ext = *h_next;
    if (n_ext > ext_cap) return c->fail(VO_ERR_DEVICE, "vo_akaze_detect_compute: %zu maxima exceed their bound", n_ext);
    if (n_ext) {
        VO_HIP(c, hipMemcpyAsync(h_ext, d_ext, n_ext * sizeof(AkExt), hipMemcpyDeviceToHost, st));
        VO_HIP(c, hipStreamSynchronize(st));
    }
    const auto t_phase1 = now();
    // the device appends in any order; cv's order is level after level, row-major: the order of `at`
    // (sorted through 8-byte keys: moving the 40-byte records themselves costs three times as much)
    {
        std::vector<uint64_t> keys(n_ext);
        for (size_t k = 0; k < n_ext; k++) keys[k] = ((uint64_t)(uint32_t)h_ext[k].at << 32) | k;
        std::sort(keys.begin(), keys.end());
        c->ak_ext_sorted.resize(n_ext);
        for (size_t k = 0; k < n_ext; k++) c->ak_ext_sorted[k] = h_ext[keys[k] & 0xffffffffu];
        h_ext = c->ak_ext_sorted.data();
    }
    if (c->ak_hmask.size() < total) c->ak_hmask.assign(total, 0);  // kept all zero between calls
    uint8_t* h_mask = c->ak_hmask.data();
    for (size_t k = 0; k < n_ext; k++) h_mask[h_ext[k].at] = 1;
    std::vector<HostLevel> hl((size_t)nl);
    {
        size_t k = 0;
        for (int l = 0; l < nl; l++) {
            const size_t end = ev[(size_t)l].at + (size_t)ev[(size_t)l].w * ev[(size_t)l].h;
            hl[(size_t)l].g = &ev[(size_t)l];
            hl[(size_t)l].mask = h_mask + ev[(size_t)l].at;
            hl[(size_t)l].ext = h_ext + k;
            size_t k1 = k;
            while (k1 < n_ext && (size_t)h_ext[k1].at < end) k1++;
            hl[(size_t)l].n = (int)(k1 - k);
            k = k1;
        }
    }
    suppress_duplicates(hl);
    std::vector<vo_keypoint> kps;
    refine(hl, prm->derivative_factor, kps);
    c->ak_survivors.clear();
    for (size_t k = 0; k < n_ext; k++) {
        if (h_mask[h_ext[k].at]) c->ak_survivors.push_back(h_ext[k].at);
        h_mask[h_ext[k].at] = 0;
    }
    const auto t_host = now();
    c->ak_levels = nl;
    c->ak_total = total;
    c->ak_plane0 = W_COUNT * P;
    c->ak_lw.assign((size_t)nl, 0);
    c->ak_lh.assign((size_t)nl, 0);
    c->ak_at.assign((size_t)nl, 0);
    for (int l = 0; l < nl; l++) {
        c->ak_lw[(size_t)l] = ev[(size_t)l].w;
        c->ak_lh[(size_t)l] = ev[(size_t)l].h;
        c->ak_at[(size_t)l] = ev[(size_t)l].at;
    }
    *n = (int)kps.size();
    if (kps.empty()) return VO_OK;
    if ((int)kps.size() > cap)
        return c->fail(VO_ERR_CAPACITY, "vo_akaze_detect_compute: %zu keypoints, capacity %d", kps.size(), cap);

    // ---- phase 3: orientation + MLDB -----------------------------------------------------------------------------
    const size_t nk = kps.size();
    VO_TRY(vo_ensure(c, c->d_ak_kp, nk * sizeof(vo_keypoint)));
    VO_TRY(vo_ensure(c, c->d_ak_desc, nk * AK_DESC_BYTES));
    VO_TRY(vo_xfer_begin(c));
    VO_HIP(c, hipMemcpyAsync(c->d_ak_kp.p, kps.data(), nk * sizeof(vo_keypoint), hipMemcpyHostToDevice, st));
    AkLevels lv;
    memset(&lv, 0, sizeof(lv));
    for (int l = 0; l < nl; l++) {
        AkLevel& d = lv.l[l];
        d.w = ev[(size_t)l].w;
        d.h = ev[(size_t)l].h;
        d.ratio = ev[(size_t)l].ratio;
        d.lt = (unsigned)(plane(S_LT, l) - A);
        d.lx = (unsigned)(plane(S_LX, l) - A);
        d.ly = (unsigned)(plane(S_LY, l) - A);
    }
    {
        int k = 0;
        for (int i = -6; i <= 6; ++i)
            for (int j = -6; j <= 6; ++j)
                if (i * i + j * j < 36) {
                    lv.oi[k] = (signed char)i;
                    lv.oj[k] = (signed char)j;
                    k++;
                }
        // SURF's gauss25 table: a sigma 2.5 Gaussian, rounded to 8 decimals as the table in the sources is printed
        for (int i = 0; i < 7; i++)
            for (int j = 0; j < 7; j++) {
                const double g = std::exp(-(double)(i * i + j * j) / 12.5) / (2.0 * M_PI * 6.25);
                lv.gauss25[i][j] = (float)(std::floor(g * 1e8 + 0.5) / 1e8);
            }
        lv.nbits = (int)pairs.size();
    }
    ak_launch_describe(st, c->d_ak_kp.as<vo_keypoint>(), (int)nk, lv, A, d_pairs, prm->pattern_size, prm->channels,
                       c->d_ak_desc.as<uint8_t>());
    VO_TRY(vo_check_launch(c, "AKAZE descriptors"));
    // (the keypoints and descriptors of a full frame exceed the small-transfer arena: direct copies, then one wait)
    VO_HIP(c, hipMemcpyAsync(kp, c->d_ak_kp.p, nk * sizeof(vo_keypoint), hipMemcpyDeviceToHost, st));
    VO_HIP(c, hipMemcpyAsync(desc, c->d_ak_desc.p, nk * AK_DESC_BYTES, hipMemcpyDeviceToHost, st));
    VO_HIP(c, hipStreamSynchronize(st));
    if (timing)
        fprintf(stderr, "[vo_akaze] queue %.2f ms, scale space + copy back %.2f ms, host suppression/refinement %.2f ms, "
                        "orientation + descriptors + copies %.2f ms (%zu keypoints)\n",
                ms(t_start, t_queued), ms(t_queued, t_phase1), ms(t_phase1, t_host), ms(t_host, now()), nk);
    return VO_OK;
}

/* test hook: one plane of one evolution level of the last vo_akaze_detect_compute on this context.
 * which: 0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet, 5 keypoint mask after the cross-level filtering (as floats) */
int vo_debug_akaze_level(vo_ctx* c, int level, int which, float* out, int* lw, int* lh) {
    if (!c) return VO_ERR_INVALID;
    if (level < 0 || level >= c->ak_levels || which < 0 || which > 5)
        return c->fail(VO_ERR_INVALID, "vo_debug_akaze_level: no such plane");
    const int w = c->ak_lw[(size_t)level], h = c->ak_lh[(size_t)level];
    if (lw) *lw = w;
    if (lh) *lh = h;
    if (!out) return VO_OK;
    VO_HIP(c, hipSetDevice(c->device));
    const size_t npx = (size_t)w * h;
    if (which == 5) {
        const size_t a0 = c->ak_at[(size_t)level];
        for (size_t i = 0; i < npx; i++) out[i] = 0.f;
        for (const int at : c->ak_survivors)
            if ((size_t)at >= a0 && (size_t)at < a0 + npx) out[(size_t)at - a0] = 1.f;
        return VO_OK;
    }
    const float* src = c->d_ak.as<float>() + c->ak_plane0 + (size_t)which * c->ak_total + c->ak_at[(size_t)level];
    VO_HIP(c, hipMemcpy(out, src, npx * sizeof(float), hipMemcpyDeviceToHost));
    return VO_OK;
}

}  // extern "C"
