// host_common.cpp — small matrix helpers, the per-thread default context and the stage timers of libvo_host.so.
// Everything that computes runs in the device library (libvo_hip.so); the host side only converts carriers
// and chains poses (the same glue the reference keeps on the host).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "host_internal.h"

namespace vo {

Matd operator*(const Matd& a, const Matd& b) {
    if (a.cols != b.rows) throw Error(VO_ERR_INVALID, "Matd product: inner dimensions differ");
    Matd m(a.rows, b.cols);
    for (int r = 0; r < a.rows; r++)
        for (int c = 0; c < b.cols; c++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += a(r, k) * b(k, c);
            m(r, c) = s;
        }
    return m;
}
Matd operator+(const Matd& a, const Matd& b) {
    if (a.rows != b.rows || a.cols != b.cols) throw Error(VO_ERR_INVALID, "Matd sum: shapes differ");
    Matd m = a;
    for (size_t i = 0; i < m.d.size(); i++) m.d[i] += b.d[i];
    return m;
}
Matd operator*(double s, const Matd& a) {
    Matd m = a;
    for (auto& v : m.d) v *= s;
    return m;
}
Matd operator-(const Matd& a) { return -1.0 * a; }
Matd hconcat(const Matd& a, const Matd& b) {
    if (a.rows != b.rows) throw Error(VO_ERR_INVALID, "hconcat: row counts differ");
    Matd m(a.rows, a.cols + b.cols);
    for (int r = 0; r < a.rows; r++) {
        for (int c = 0; c < a.cols; c++) m(r, c) = a(r, c);
        for (int c = 0; c < b.cols; c++) m(r, a.cols + c) = b(r, c);
    }
    return m;
}

namespace {
struct CtxHolder {
    vo_ctx* ctx = nullptr;
    ~CtxHolder() {
        if (ctx) vo_ctx_destroy(ctx);
    }
};
int env_device() {
    const char* e = std::getenv("VO_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}
vo_ctx* make_ctx() {
    vo_ctx* c = nullptr;
    const int rc = vo_ctx_create(env_device(), nullptr, &c);
    if (rc != VO_OK || !c) throw Error(rc, "vo_ctx_create failed: no usable HIP device (there is no CPU fallback)");
    return c;
}
}  // namespace

vo_ctx* default_ctx() {
    thread_local CtxHolder h;
    if (!h.ctx) h.ctx = make_ctx();
    return h.ctx;
}

void check(vo_ctx* ctx, int rc) {
    if (rc != VO_OK) throw Error(rc, vo_last_error(ctx));
}

}  // namespace vo

StageTimers::~StageTimers() {
    if (!on) return;
    for (int i = 0; i < N; i++)
        if (calls[i])
            std::fprintf(stderr, "[vo_host timing] %-18s %6ld calls  %8.1f us each\n", names[i], calls[i],
                         sec[i] / calls[i] * 1e6);
}
StageTimers g_timers;
