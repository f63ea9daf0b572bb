// host_feature_tracker.cpp — FeatureTracker::trackPoints and removePoints of the host mirror (reference:
// src/features/feature_tracker.cpp:10-17, src/features/utils.h:9-58) over vo_lk_track.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <numeric>

#include "feature_tracker.h"
#include "host_batch.h"
#include "host_internal.h"
#include "utils.h"

uint64_t vo::Image::next_id() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1);
}

FeatureTracker::FeatureTracker(int channels) {
    const char* e = std::getenv("VO_HIP_DEVICE");
    const int rc = vo_ctx_create(e ? std::atoi(e) : 0, nullptr, &ctx_);
    if (rc != VO_OK || !ctx_)
        throw vo::Error(rc, "vo_ctx_create failed: no usable HIP device (there is no CPU fallback)");
    vo_lk_params_default(&lk_);
    lk_.channels = channels;
}

FeatureTracker::~FeatureTracker() {
    if (ctx_) vo_ctx_destroy(ctx_);
}

float FeatureTracker::trackPoints(VOFrame* vo0, VOFrame* vo1) {
    const vo::Image& a = vo0->image;
    const vo::Image& b = vo1->image;
    if (a.width != b.width || a.height != b.height || a.empty())
        throw vo::Error(VO_ERR_INVALID, "trackPoints: images differ in size or are empty");
    const int n = (int)vo0->points.size();
    std::vector<unsigned char> status(n);
    vo1->points.resize(n);
    const bool cached = cached_id_ != 0 && cached_id_ == a.id;
    cached_id_ = 0;
    static_assert(sizeof(vo::Point2f) == 8, "packed points");
    {
        StageScope ts(3);
        if (vo::BatchGroup* grp = vo::current_group()) {
            // lock-step mode (vo_host_vo_run_concurrent): this call joins those of the group's other drivers in one
            // vo_lk_track_batch on the group's context, which holds the pyramids; same results
            const uint8_t* next_px = b.px.data();
            if (uint8_t* slot = grp->image_slot(vo::current_slot(), b.width, b.height)) {  // this thread's share of the gather
                memcpy(slot, next_px, b.px.size());
                next_px = slot;
            }
            const int rc = grp->lk_track(vo::current_slot(), a.px.data(), cached, next_px, a.width, a.height,
                                         n ? &vo0->points[0].x : nullptr, n, &lk_, n ? &vo1->points[0].x : nullptr,
                                         n ? status.data() : nullptr);
            if (rc != VO_OK) throw vo::Error(rc, grp->last_error());
        } else {
            check(ctx_, vo_lk_track(ctx_, cached ? nullptr : a.px.data(), b.px.data(), a.width, a.height, a.width,
                                    n ? &vo0->points[0].x : nullptr, n, &lk_, n ? &vo1->points[0].x : nullptr,
                                    n ? status.data() : nullptr));
        }
    }
    cached_id_ = b.id;
    // Remove bad points
    StageScope ts(4);
    return (float)removePoints(vo0, vo1, &status);
}

// The reference erases from the back of every vector inside the loop; an erase at index i only
// moves elements behind i, which the loop has already visited, so every decision sees original
// data and the result is this stable compaction.  `dists` is sorted first and then erased by the
// ORIGINAL point index: kept as is (utils.h:16,42,57).
namespace {
// Ascending sort of floats that are norms (>= 0, or NaN, which sorts last as a bit pattern): their bit
// patterns order like unsigned integers, so three 11-bit counting passes do it in a fraction of
// std::sort's time for the few thousand points of a frame.  Equal values are indistinguishable, so the
// result is the array std::sort produces.
void sort_nonnegative(std::vector<float>* v) {
    const size_t n = v->size();
    bool plain = n >= 256;
    for (size_t i = 0; i < n && plain; i++) plain = !std::signbit((*v)[i]) && !std::isnan((*v)[i]);
    if (!plain) {
        std::sort(v->begin(), v->end());
        return;
    }
    std::vector<uint32_t> a(n), b(n);
    memcpy(a.data(), v->data(), n * sizeof(float));
    for (int pass = 0; pass < 3; pass++) {
        const int shift = 11 * pass;
        uint32_t count[2049] = {0};
        for (size_t i = 0; i < n; i++) count[((a[i] >> shift) & 2047u) + 1]++;
        for (int k = 0; k < 2048; k++) count[k + 1] += count[k];
        for (size_t i = 0; i < n; i++) b[count[(a[i] >> shift) & 2047u]++] = a[i];
        a.swap(b);
    }
    memcpy(v->data(), a.data(), n * sizeof(float));
}
}  // namespace

int removePoints(VOFrame* vo0, VOFrame* vo1, std::vector<unsigned char>* status) {
    const size_t n = status->size();
    std::vector<float> orig(n);
    for (size_t i = 0; i < n; i++) {
        const float dx = vo0->points[i].x - vo1->points[i].x, dy = vo0->points[i].y - vo1->points[i].y;
        orig[i] = (float)std::sqrt((double)dx * dx + (double)dy * dy);  // cv::norm(Point2f) -> double -> float
    }
    std::vector<float> dists = orig;
    sort_nonnegative(&dists);  // std::sort(dists) in the reference: the sorted ORDER is what the sums below see
    const double sum = std::accumulate(dists.begin(), dists.end(), 0.0);
    const double mean = sum / dists.size();
    const double sq_sum = std::inner_product(dists.begin(), dists.end(), dists.begin(), 0.0);
    const double stdev = std::sqrt(sq_sum / dists.size() - mean * mean);

    std::vector<uint8_t>& mask = vo0->mask.d;
    const size_t n3 = vo0->points_3d.size(), nm = mask.size();
    size_t k = 0, k3 = 0, km = 0;
    std::vector<float> kept;
    kept.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const vo::Point2f pt = vo1->points[i];
        const float dist = orig[i];
        if ((*status)[i] == 0 || pt.x < 0 || pt.y < 0 || std::abs(dist - mean) > stdev * 2) {
            if (pt.x < 0 || pt.y < 0) (*status)[i] = 0;
            continue;
        }
        vo0->points[k] = vo0->points[i];
        vo1->points[k] = pt;
        k++;
        kept.push_back(dists[i]);
        if (i < n3) vo0->points_3d[k3++] = vo0->points_3d[i];
        if (i < nm) mask[km++] = mask[i];
    }
    for (size_t i = n; i < n3; i++) vo0->points_3d[k3++] = vo0->points_3d[i];
    for (size_t i = n; i < nm; i++) mask[km++] = mask[i];
    vo0->points.resize(k);
    vo1->points.resize(k);
    vo0->points_3d.resize(k3);
    mask.resize(km);
    return kept.empty() ? 0 : (int)kept[kept.size() / 2];  // empty: undefined behaviour in the reference
}

