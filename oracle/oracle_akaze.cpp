// oracle_akaze.cpp — TEST INFRASTRUCTURE ONLY (see vo_oracle.h).
//
// CPU restatement of cv::AKAZE::create()->detectAndCompute (reference: src/features/feature_detector.cpp:7,43-45,
// called from src/sfm/bundle_adjustment.cpp:41) — OpenCV 3.4.2 features2d/src/kaze/AKAZEFeatures.cpp,
// nldiffusion_functions.cpp and fed.cpp, restated FROM MEMORY (OpenCV is absent from this image), following the
// published algorithm (Alcantarilla, Nuevo, Bartoli: "Fast Explicit Diffusion for Accelerated Features in
// Nonlinear Scale Spaces", BMVC 2013).  PARITY UNPINNED, and more loosely than the rest of the oracle: this is a
// float pipeline whose last bits depend on cv's SIMD filter kernels, and the details listed below are the
// author's best recollection of the 3.4 sources, not a verified transcription:
//   * cross-level duplicate suppression (Find_Scale_Space_Extrema) and the sub-pixel offset formula,
//   * the border rows / columns of the diffusion step,
//   * scalar evaluation order inside GaussianBlur / Scharr / sepFilter2D,
//   * `modgs *= (nbins - 1) / hmax` taken as a float multiply.
// What is certain: the structure (16 evolution levels at 1241x376, FED cycles, Perona-Malik g2, scaled Scharr
// Hessian determinant, 3x3 extrema, 109-sample orientation, MLDB 2x2/3x3/4x4 grids x 3 channels = 486 bits).
#include <cfloat>

#include "oracle_common.h"

using namespace vo_oracle;

namespace {

struct Img {
    int w = 0, h = 0;
    std::vector<float> d;
    Img() {}
    Img(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_, 0.f) {}
    float& at(int y, int x) { return d[(size_t)y * w + x]; }
    float at(int y, int x) const { return d[(size_t)y * w + x]; }
};

struct Level {
    int w, h, octave, sublevel, sigma_size, border;
    float esigma, etime, ratio;
    Img Lt, Lsmooth, Lx, Ly, Ldet;
};

int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// cv::getGaussianKernel(n, sigma, CV_32F) for sigma > 0
std::vector<float> gauss_kernel(int n, float sigma) {
    std::vector<float> k(n);
    const double scale2x = -0.5 / ((double)sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        const double t = std::exp(scale2x * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
    return k;
}

// cv::GaussianBlur(src, dst, Size(n, n), sigma, sigma, BORDER_REPLICATE) on CV_32F: separable, symmetric taps
// summed as centre + sum_j k[j] (left + right)
Img gaussian_blur(const Img& src, int n, float sigma) {
    const std::vector<float> k = gauss_kernel(n, sigma);
    const int r = n / 2;
    Img tmp(src.w, src.h), dst(src.w, src.h);
    for (int y = 0; y < src.h; y++)
        for (int x = 0; x < src.w; x++) {
            float s = k[r] * src.at(y, x);
            for (int j = 1; j <= r; j++)
                s += k[r + j] * (src.at(y, clampi(x - j, 0, src.w - 1)) + src.at(y, clampi(x + j, 0, src.w - 1)));
            tmp.at(y, x) = s;
        }
    for (int y = 0; y < src.h; y++)
        for (int x = 0; x < src.w; x++) {
            float s = k[r] * tmp.at(y, x);
            for (int j = 1; j <= r; j++)
                s += k[r + j] * (tmp.at(clampi(y - j, 0, src.h - 1), x) + tmp.at(clampi(y + j, 0, src.h - 1), x));
            dst.at(y, x) = s;
        }
    return dst;
}

// Separable derivative filter with three non-zero taps at distance `scale` (compute_derivative_kernels of
// nldiffusion_functions.cpp; scale 1 with norm 3/32, w 10/3 is the normalised Scharr kernel), BORDER_REFLECT_101.
// dx: derivative along x, smoothing along y; otherwise the transpose.  `norm` / `wnorm`: outer / centre
// smoothing taps.  cv::Scharr itself (scale = 1, un-normalised) is norm = 3, wnorm = 10.
Img deriv_filter(const Img& src, bool dx, int scale, float norm, float wnorm) {
    Img tmp(src.w, src.h), dst(src.w, src.h);
    // row pass
    for (int y = 0; y < src.h; y++)
        for (int x = 0; x < src.w; x++) {
            const float a = src.at(y, reflect101(x - scale, src.w)), c = src.at(y, x),
                        b = src.at(y, reflect101(x + scale, src.w));
            tmp.at(y, x) = dx ? b - a : wnorm * c + norm * (a + b);
        }
    // column pass
    for (int y = 0; y < src.h; y++)
        for (int x = 0; x < src.w; x++) {
            const float a = tmp.at(reflect101(y - scale, src.h), x), c = tmp.at(y, x),
                        b = tmp.at(reflect101(y + scale, src.h), x);
            dst.at(y, x) = dx ? wnorm * c + norm * (a + b) : b - a;
        }
    return dst;
}

// compute_kcontrast (AKAZEFeatures.cpp): the `perc` percentile of the gradient magnitude histogram
float compute_kcontrast(const Img& Lx, const Img& Ly, float perc, int nbins) {
    const int rows = Lx.h - 2, cols = Lx.w - 2, total = rows * cols;
    std::vector<float> modg((size_t)total);
    float hmax = 0.f;
    size_t at = 0;
    for (int i = 1; i < Lx.h - 1; i++)
        for (int j = 1; j < Lx.w - 1; j++) {
            const float lx = Lx.at(i, j), ly = Ly.at(i, j);
            const float dist = std::sqrt(lx * lx + ly * ly);
            modg[at++] = dist;
            hmax = std::max(hmax, dist);
        }
    if (hmax == 0.0f) return 0.03f;
    const float sc = (float)(nbins - 1) / hmax;
    std::vector<int> hist((size_t)nbins, 0);
    for (int i = 0; i < total; i++) hist[(size_t)(int)(modg[i] * sc)]++;
    const int nthreshold = (int)((float)(total - hist[0]) * perc);
    int nelements = 0;
    for (int k = 1; k < nbins; k++) {
        if (nelements >= nthreshold) return hmax * (float)k / (float)nbins;
        nelements += hist[k];
    }
    return 0.03f;
}

// pm_g2: 1 / (1 + |grad L|^2 / k^2)
Img pm_g2(const Img& Lx, const Img& Ly, float k) {
    Img dst(Lx.w, Lx.h);
    const float inv_k = 1.0f / (k * k);
    for (size_t i = 0; i < dst.d.size(); i++)
        dst.d[i] = 1.0f / (1.0f + inv_k * (Lx.d[i] * Lx.d[i] + Ly.d[i] * Ly.d[i]));
    return dst;
}

// nld_step_scalar: Lt += step * div(c grad Lt), five-point stencil with mirrored (zero-flux) borders
void nld_step(Img& Lt, const Img& c, float step_size) {
    const int w = Lt.w, h = Lt.h;
    Img out(w, h);
    for (int y = 0; y < h; y++) {
        const int yp = std::max(y - 1, 0), yn = std::min(y + 1, h - 1);
        for (int x = 0; x < w; x++) {
            const int xp = std::max(x - 1, 0), xn = std::min(x + 1, w - 1);
            const float l = Lt.at(y, x), cc = c.at(y, x);
            const float xpos = (cc + c.at(y, xn)) * (Lt.at(y, xn) - l);
            const float xneg = (c.at(y, xp) + cc) * (l - Lt.at(y, xp));
            const float ypos = (cc + c.at(yn, x)) * (Lt.at(yn, x) - l);
            const float yneg = (c.at(yp, x) + cc) * (l - Lt.at(yp, x));
            out.at(y, x) = l + step_size * (xpos - xneg + ypos - yneg);
        }
    }
    Lt.d.swap(out.d);
}

// cv::resize(src, dst, dsize, 0, 0, INTER_AREA) on CV_32F, general (non-integer scale) path of imgproc/resize.cpp:
// computeResizeAreaTab per axis, horizontal weighted sums per source row, then the vertical accumulation in
// table order.  (cv takes a faster path when BOTH scales are integers; its sums are then associated
// differently, which can move the last bit: not restated, this path is used for every size.)
struct AreaTab {
    std::vector<int> si, first;  // source index per entry; first[d] .. first[d + 1]: entries of destination d
    std::vector<float> alpha;
};
AreaTab area_tab(int ssize, int dsize) {
    AreaTab t;
    const double scale = (double)ssize / dsize;
    for (int dx = 0; dx < dsize; dx++) {
        t.first.push_back((int)t.si.size());
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = cvCeil(fsx1), sx2 = cvFloor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) {
            t.si.push_back(sx1 - 1);
            t.alpha.push_back((float)((sx1 - fsx1) / cell));
        }
        for (int sx = sx1; sx < sx2; sx++) {
            t.si.push_back(sx);
            t.alpha.push_back((float)(1.0 / cell));
        }
        if (fsx2 - sx2 > 1e-3) {
            t.si.push_back(sx2);
            t.alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell));
        }
    }
    t.first.push_back((int)t.si.size());
    return t;
}
Img halfsample(const Img& src, int dw, int dh) {
    const AreaTab xt = area_tab(src.w, dw), yt = area_tab(src.h, dh);
    Img dst(dw, dh);
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            float sum = 0.f;
            for (int b = yt.first[y]; b < yt.first[y + 1]; b++) {
                float buf = 0.f;
                for (int a = xt.first[x]; a < xt.first[x + 1]; a++) buf += src.at(yt.si[b], xt.si[a]) * xt.alpha[a];
                if (b == yt.first[y]) sum = yt.alpha[b] * buf;
                else sum += yt.alpha[b] * buf;
            }
            dst.at(y, x) = sum;
        }
    return dst;
}

// fed.cpp
bool fed_is_prime(int n) {
    if (n <= 1) return false;
    if (n == 2 || n == 3 || n == 5 || n == 7) return true;
    if (n % 2 == 0 || n % 3 == 0 || n % 5 == 0 || n % 7 == 0) return false;
    const int upper = (int)std::sqrt(1.0f + n);
    for (int d = 11; d <= upper; d += 2)
        if (n % d == 0) return false;
    return true;
}
std::vector<float> fed_tau_by_process_time(float T, int M, float tau_max, bool reordering) {
    const float t = T / (float)M;
    const int n = cvCeil(std::sqrt(3.0f * t / tau_max + 0.25f) - 0.5f - 1.0e-8f);
    std::vector<float> tau;
    if (n <= 0) return tau;
    const float scale = 3.0f * t / (tau_max * (float)(n * (n + 1)));
    tau.resize(n);
    std::vector<float> tauh(n);
    const float c = 1.0f / (4.0f * (float)n + 2.0f), d = scale * tau_max / 2.0f;
    for (int k = 0; k < n; ++k) {
        const float hh = std::cos((float)M_PI * (2.0f * (float)k + 1.0f) * c);
        tauh[k] = d / (hh * hh);
    }
    if (!reordering) return tauh;
    const int kappa = n / 2;
    int prime = n + 1;
    while (!fed_is_prime(prime)) prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index = 0;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return tau;
}

void layout(int w, int h, const vo_akaze_params& p, std::vector<Level>& ev) {
    const float smax = 10.0f * std::sqrt(2.0f);
    int omax = p.n_octaves;
    for (int i = 0, power = 1; i <= omax - 1; i++, power *= 2) {
        const float rfactor = 1.0f / (float)power;
        const int lh = (int)((float)h * rfactor), lw = (int)((float)w * rfactor);
        if ((lw < 80 || lh < 40) && i != 0) break;
        for (int j = 0; j < p.n_sublevels; j++) {
            Level s;
            s.w = lw;
            s.h = lh;
            s.esigma = p.soffset * std::pow(2.f, (float)j / (float)p.n_sublevels + (float)i);
            s.sigma_size = cvRound(s.esigma * p.derivative_factor / (float)power);
            s.etime = 0.5f * (s.esigma * s.esigma);
            s.octave = i;
            s.sublevel = j;
            s.ratio = (float)power;
            s.border = cvRound(smax * (float)s.sigma_size) + 1;
            ev.push_back(std::move(s));
        }
    }
}

void build_scale_space(const uint8_t* gray, int w, int h, int stride, const vo_akaze_params& p,
                       std::vector<Level>& ev) {
    layout(w, h, p, ev);
    Img img(w, h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img.at(y, x) = (float)gray[(size_t)y * stride + x] * (1.0f / 255.0f);
    int ksize = cvCeil(2.0f * (1.0f + (p.soffset - 0.8f) / 0.3f));
    ksize |= 1;
    ev[0].Lsmooth = gaussian_blur(img, ksize, p.soffset);
    ev[0].Lt = ev[0].Lsmooth;
    float kcontrast;
    {
        const Img sm = gaussian_blur(img, 5, 1.0f);
        const Img Lx = deriv_filter(sm, true, 1, 3.f, 10.f), Ly = deriv_filter(sm, false, 1, 3.f, 10.f);
        kcontrast = compute_kcontrast(Lx, Ly, p.kcontrast_percentile, p.kcontrast_nbins);
    }
    for (size_t i = 1; i < ev.size(); i++) {
        Level& e = ev[i];
        if (e.octave > ev[i - 1].octave) {
            e.Lt = halfsample(ev[i - 1].Lt, e.w, e.h);
            kcontrast *= 0.75f;
        } else {
            e.Lt = ev[i - 1].Lt;
        }
        e.Lsmooth = gaussian_blur(e.Lt, 5, 1.0f);
        const Img Lx = deriv_filter(e.Lsmooth, true, 1, 3.f, 10.f), Ly = deriv_filter(e.Lsmooth, false, 1, 3.f, 10.f);
        const Img flow = pm_g2(Lx, Ly, kcontrast);
        const std::vector<float> tau = fed_tau_by_process_time(e.etime - ev[i - 1].etime, 1, 0.25f, true);
        for (size_t j = 0; j < tau.size(); j++) nld_step(e.Lt, flow, tau[j] * 0.5f);
    }
    // multiscale derivatives and the Hessian determinant (Compute_Determinant_Hessian_Response)
    for (Level& e : ev) {
        const float w3 = 10.0f / 3.0f, norm = 1.0f / (2.0f * (float)e.sigma_size * (w3 + 2.0f)), wnorm = w3 * norm;
        e.Lx = deriv_filter(e.Lsmooth, true, e.sigma_size, norm, wnorm);
        e.Ly = deriv_filter(e.Lsmooth, false, e.sigma_size, norm, wnorm);
        const Img Lxx = deriv_filter(e.Lx, true, e.sigma_size, norm, wnorm);
        const Img Lxy = deriv_filter(e.Lx, false, e.sigma_size, norm, wnorm);
        const Img Lyy = deriv_filter(e.Ly, false, e.sigma_size, norm, wnorm);
        const float s4 = (float)(e.sigma_size * e.sigma_size * e.sigma_size * e.sigma_size);
        e.Ldet = Img(e.w, e.h);
        for (size_t k = 0; k < e.Ldet.d.size(); k++)
            e.Ldet.d[k] = (Lxx.d[k] * Lyy.d[k] - Lxy.d[k] * Lxy.d[k]) * s4;
    }
}

// first keypoint (row-major scan of the square window) of `mask` within `radius` of (x, y)
bool find_neighbor(const std::vector<uint8_t>& mask, int w, int h, int x, int y, int radius, int& idx) {
    for (int i = std::max(y - radius, 0); i < std::min(y + radius, h); i++)
        for (int j = std::max(x - radius, 0); j < std::min(x + radius, w); j++) {
            if (!mask[(size_t)i * w + j]) continue;
            const int dx = j - x, dy = i - y;
            if (dx * dx + dy * dy <= radius * radius) {
                idx = i * w + j;
                return true;
            }
        }
    return false;
}

void find_extrema(const std::vector<Level>& ev, float dthreshold, std::vector<std::vector<uint8_t>>& masks) {
    masks.resize(ev.size());
    for (size_t i = 0; i < ev.size(); i++) {
        const Level& e = ev[i];
        masks[i].assign((size_t)e.w * e.h, 0);
        if (e.border + 1 >= e.h) continue;
        for (int y = e.border; y < e.h - e.border; y++)
            for (int x = e.border; x < e.w - e.border; x++) {
                const float v = e.Ldet.at(y, x);
                if (v <= dthreshold) continue;
                if (v <= e.Ldet.at(y, x - 1) || v <= e.Ldet.at(y, x + 1)) continue;
                if (v <= e.Ldet.at(y - 1, x - 1) || v <= e.Ldet.at(y - 1, x) || v <= e.Ldet.at(y - 1, x + 1)) continue;
                if (v <= e.Ldet.at(y + 1, x - 1) || v <= e.Ldet.at(y + 1, x) || v <= e.Ldet.at(y + 1, x + 1)) continue;
                masks[i][(size_t)y * e.w + x] = 1;
            }
    }
    // a keypoint removes a weaker neighbour one level down; then, top level first, a weaker one one level up
    for (size_t i = 1; i < ev.size(); i++) {
        const Level &e = ev[i], &ep = ev[i - 1];
        const int diff_ratio = (int)e.ratio / (int)ep.ratio;
        const int radius = e.sigma_size * diff_ratio;
        for (int y = 0; y < e.h; y++)
            for (int x = 0; x < e.w; x++) {
                if (!masks[i][(size_t)y * e.w + x]) continue;
                int idx;
                if (find_neighbor(masks[i - 1], ep.w, ep.h, x * diff_ratio, y * diff_ratio, radius, idx)) {
                    if (e.Ldet.at(y, x) > ep.Ldet.d[(size_t)idx]) masks[i - 1][(size_t)idx] = 0;
                }
            }
    }
    for (int i = (int)ev.size() - 2; i >= 0; i--) {
        const Level &e = ev[(size_t)i], &en = ev[(size_t)i + 1];
        const int diff_ratio = (int)en.ratio / (int)e.ratio;
        const int radius = en.sigma_size;  // the upper level's keypoint size in its own pixels
        for (int y = 0; y < e.h; y++)
            for (int x = 0; x < e.w; x++) {
                if (!masks[(size_t)i][(size_t)y * e.w + x]) continue;
                int idx;
                if (find_neighbor(masks[(size_t)i + 1], en.w, en.h, x / diff_ratio, y / diff_ratio, radius, idx)) {
                    if (e.Ldet.at(y, x) > en.Ldet.d[(size_t)idx]) masks[(size_t)i + 1][(size_t)idx] = 0;
                }
            }
    }
}

void subpixel(const std::vector<Level>& ev, const std::vector<std::vector<uint8_t>>& masks, float derivative_factor,
              std::vector<vo_keypoint>& kps) {
    for (size_t i = 0; i < ev.size(); i++) {
        const Level& e = ev[i];
        for (int y = 0; y < e.h; y++)
            for (int x = 0; x < e.w; x++) {
                if (!masks[i][(size_t)y * e.w + x]) continue;
                const Img& L = e.Ldet;
                const float Dx = 0.5f * (L.at(y, x + 1) - L.at(y, x - 1));
                const float Dy = 0.5f * (L.at(y + 1, x) - L.at(y - 1, x));
                const float Dxx = L.at(y, x + 1) + L.at(y, x - 1) - 2.0f * L.at(y, x);
                const float Dyy = L.at(y + 1, x) + L.at(y - 1, x) - 2.0f * L.at(y, x);
                const float Dxy = 0.25f * (L.at(y + 1, x + 1) + L.at(y - 1, x - 1) - L.at(y - 1, x + 1) - L.at(y + 1, x - 1));
                // cv::solve(Matx22f, Vec2f, DECOMP_LU): the 2 x 2 closed form in double
                const double det = (double)Dxx * Dyy - (double)Dxy * Dxy;
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
                kp.response = L.at(y, x);
                kp.octave = e.octave;
                kp.class_id = (int)i;
                kps.push_back(kp);
            }
    }
}

// cv::fastAtan2 (degrees), as in oracle_features.cpp
float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI),
                p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = std::abs(x), ay = std::abs(y);
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

// gauss25[i][j] = exp(-(i^2 + j^2) / (2 * 2.5^2)) / (2 pi 2.5^2), the table of SURF / KAZE rounded to 8 decimals
float gauss25(int i, int j) {
    const double g = std::exp(-(double)(i * i + j * j) / 12.5) / (2.0 * M_PI * 6.25);
    return (float)(std::floor(g * 1e8 + 0.5) / 1e8);
}

void main_orientation(vo_keypoint& kp, const std::vector<Level>& ev) {
    const int ang_size = 109;
    float resX[ang_size], resY[ang_size], Ang[ang_size];
    const Level& e = ev[(size_t)kp.class_id];
    const int s = cvRound(0.5f * kp.size / e.ratio);
    const float xf = kp.x / e.ratio, yf = kp.y / e.ratio;
    int idx = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
            if (i * i + j * j < 36) {
                const int iy = clampi(cvRound(yf + (float)(j * s)), 0, e.h - 1);
                const int ix = clampi(cvRound(xf + (float)(i * s)), 0, e.w - 1);
                const float gw = gauss25(std::abs(i), std::abs(j));
                resX[idx] = gw * e.Lx.at(iy, ix);
                resY[idx] = gw * e.Ly.at(iy, ix);
                ++idx;
            }
    const float deg2rad = (float)(M_PI / 180.0);
    for (int k = 0; k < ang_size; k++) Ang[k] = fast_atan2_deg(resY[k], resX[k]) * deg2rad;
    float maxv = 0.f, best = 0.f;
    for (float ang1 = 0; ang1 < (float)(2.0 * M_PI); ang1 += 0.15f) {
        const float ang2 = (ang1 + (float)(M_PI / 3.0) > (float)(2.0 * M_PI) ? ang1 - (float)(5.0 * M_PI / 3.0)
                                                                           : ang1 + (float)(M_PI / 3.0));
        float sumX = 0.f, sumY = 0.f;
        for (int k = 0; k < ang_size; ++k) {
            const float ang = Ang[k];
            if (ang1 < ang2 && ang1 < ang && ang < ang2) {
                sumX += resX[k];
                sumY += resY[k];
            } else if (ang2 < ang1 && ((ang > 0 && ang < ang2) || (ang > ang1 && ang < (float)(2.0 * M_PI)))) {
                sumX += resX[k];
                sumY += resY[k];
            }
        }
        if (sumX * sumX + sumY * sumY > maxv) {
            maxv = sumX * sumX + sumY * sumY;
            best = fast_atan2_deg(sumY, sumX);
        }
    }
    kp.angle = best;  // degrees, cv's KeyPoint convention
}

// CV_TOGGLE_FLT: floats compared as integers (cv's MLDB_Binary_Comparisons); -0 sorts below +0
int32_t toggle(float f) {
    int32_t i;
    memcpy(&i, &f, 4);
    return i < 0 ? (i ^ 0x7fffffff) : i;
}

void mldb_descriptor(const vo_keypoint& kp, const std::vector<Level>& ev, const vo_akaze_params& p, uint8_t* desc) {
    const Level& e = ev[(size_t)kp.class_id];
    const int pattern = p.pattern_size, chan = p.channels;
    const float scale = (float)cvRound(0.5f * kp.size / e.ratio);
    const float xf = kp.x / e.ratio, yf = kp.y / e.ratio;
    const float angle = kp.angle * (float)M_PI / 180.f;
    const float co = (float)std::cos((double)angle), si = (float)std::sin((double)angle);
    const int steps[3] = {pattern, (pattern * 2 + 2) / 3, (pattern + 1) / 2};
    memset(desc, 0, VO_AKAZE_DESC_BYTES);
    int dpos = 0;
    for (int z = 0; z < 3; z++) {
        float values[16 * 3];
        const int step = steps[z];
        int valpos = 0;
        for (int i = -pattern; i < pattern; i += step)
            for (int j = -pattern; j < pattern; j += step) {
                float di = 0.f, dx = 0.f, dy = 0.f;
                int ns = 0;
                for (int k = i; k < i + step; k++)
                    for (int l = j; l < j + step; l++) {
                        const float sy = yf + ((float)l * co * scale + (float)k * si * scale);
                        const float sx = xf + (-(float)l * si * scale + (float)k * co * scale);
                        const int y1 = cvRound(sy), x1 = cvRound(sx);
                        if (y1 < 0 || y1 >= e.h || x1 < 0 || x1 >= e.w) continue;
                        di += e.Lt.at(y1, x1);
                        if (chan > 1) {
                            const float rx = e.Lx.at(y1, x1), ry = e.Ly.at(y1, x1);
                            if (chan == 2) {
                                dx += std::sqrt(rx * rx + ry * ry);
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
                values[valpos * chan] = di;
                if (chan > 1) values[valpos * chan + 1] = dx;
                if (chan > 2) values[valpos * chan + 2] = dy;
                valpos++;
            }
        const int count = (z + 2) * (z + 2);
        for (int pos = 0; pos < chan; pos++)
            for (int i = 0; i < count; i++)
                for (int j = i + 1; j < count; j++) {
                    if (toggle(values[chan * i + pos]) > toggle(values[chan * j + pos]))
                        desc[dpos >> 3] |= (uint8_t)(1 << (dpos & 7));
                    dpos++;
                }
    }
}

bool params_ok(const vo_akaze_params* p) {
    return p && p->n_octaves >= 1 && p->n_octaves <= 8 && p->n_sublevels >= 1 && p->n_sublevels <= 8 &&
           p->channels >= 1 && p->channels <= 3 && p->pattern_size >= 2 && p->pattern_size <= 16 &&
           (p->pattern_size & 1) == 0 && p->kcontrast_nbins > 2 && p->soffset > 0.8f;
}

}  // namespace

extern "C" {

void vo_oracle_akaze_params_default(vo_akaze_params* p) {
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

int vo_oracle_akaze_layout(int w, int h, const vo_akaze_params* p, int* lw, int* lh, int* sigma_size, int* border,
                           float* esigma, int* nsteps) {
    if (!params_ok(p) || w < 3 || h < 3) return -1;
    std::vector<Level> ev;
    layout(w, h, *p, ev);
    for (size_t i = 0; i < ev.size(); i++) {
        if (lw) lw[i] = ev[i].w;
        if (lh) lh[i] = ev[i].h;
        if (sigma_size) sigma_size[i] = ev[i].sigma_size;
        if (border) border[i] = ev[i].border;
        if (esigma) esigma[i] = ev[i].esigma;
        if (nsteps) nsteps[i] = i == 0 ? 0 : (int)fed_tau_by_process_time(ev[i].etime - ev[i - 1].etime, 1, 0.25f, true).size();
    }
    return (int)ev.size();
}

int vo_oracle_akaze_tau(float T, float* tau, int cap) {
    const std::vector<float> t = fed_tau_by_process_time(T, 1, 0.25f, true);
    for (size_t i = 0; i < t.size() && (int)i < cap; i++) tau[i] = t[i];
    return (int)t.size();
}

/* which: 0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet, 5 extrema mask after the cross-level filtering (0 / 1 as float) */
int vo_oracle_akaze_level(const uint8_t* gray, int w, int h, int stride, const vo_akaze_params* p, int level,
                          int which, float* out) {
    if (!gray || !params_ok(p) || w < 3 || h < 3 || !out) return VO_ERR_INVALID;
    std::vector<Level> ev;
    build_scale_space(gray, w, h, stride, *p, ev);
    if (level < 0 || level >= (int)ev.size()) return VO_ERR_INVALID;
    const Level& e = ev[(size_t)level];
    if (which == 5) {
        std::vector<std::vector<uint8_t>> masks;
        find_extrema(ev, p->threshold, masks);
        for (size_t i = 0; i < masks[(size_t)level].size(); i++) out[i] = masks[(size_t)level][i];
        return VO_OK;
    }
    const Img* src = which == 0 ? &e.Lt : which == 1 ? &e.Lsmooth : which == 2 ? &e.Lx : which == 3 ? &e.Ly : &e.Ldet;
    memcpy(out, src->d.data(), src->d.size() * sizeof(float));
    return VO_OK;
}

int vo_oracle_akaze_detect_compute(const uint8_t* gray, int w, int h, int stride, const vo_akaze_params* p,
                                   vo_keypoint* kp, uint8_t* desc, int cap, int* n) {
    if (!gray || !params_ok(p) || w < 3 || h < 3 || !n) return VO_ERR_INVALID;
    std::vector<Level> ev;
    build_scale_space(gray, w, h, stride, *p, ev);
    std::vector<std::vector<uint8_t>> masks;
    find_extrema(ev, p->threshold, masks);
    std::vector<vo_keypoint> kps;
    subpixel(ev, masks, p->derivative_factor, kps);
    *n = (int)kps.size();
    if ((int)kps.size() > cap) return VO_ERR_CAPACITY;
    for (size_t i = 0; i < kps.size(); i++) {
        main_orientation(kps[i], ev);
        if (kp) kp[i] = kps[i];
        if (desc) mldb_descriptor(kps[i], ev, *p, desc + i * VO_AKAZE_DESC_BYTES);
    }
    return VO_OK;
}

}  // extern "C"
