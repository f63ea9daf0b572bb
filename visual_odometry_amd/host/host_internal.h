// host_internal.h — shared by the translation units of libvo_host.so (not installed).
#ifndef VO_HOST_INTERNAL_H
#define VO_HOST_INTERNAL_H

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "vo_host_types.h"

using vo::check;

// VO_HOST_TIMING=1: wall time per stage of the per-frame driver, printed when the process ends
// (one set of counters for the process: meant for single-threaded runs of the driver).
struct StageTimers {
    static constexpr int N = 10;
    const char* names[N] = {"image copy in", "frame buffer", "detectFAST", "vo_lk_track", "removePoints",
                            "vo_essential_pose", "points4DToVec", "getScale", "pose compose", "other"};
    double sec[N] = {0};
    long calls[N] = {0};
    bool on = std::getenv("VO_HOST_TIMING") != nullptr;
    ~StageTimers();
};
extern StageTimers g_timers;
struct StageScope {
    int i;
    std::chrono::steady_clock::time_point t0;
    explicit StageScope(int idx) : i(idx) {
        if (g_timers.on) t0 = std::chrono::steady_clock::now();
    }
    ~StageScope() {
        if (!g_timers.on) return;
        g_timers.sec[i] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        g_timers.calls[i]++;
    }
};

#endif
