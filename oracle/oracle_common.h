// oracle_common.h — TEST INFRASTRUCTURE ONLY (see vo_oracle.h). Small helpers shared by the
// CPU restatement: OpenCV's rounding primitives and cv::RNG.
#ifndef VO_ORACLE_COMMON_H
#define VO_ORACLE_COMMON_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#include "vo_oracle.h"

namespace vo_oracle {

// cvRound: round-half-to-even (SSE cvtsd2si / cvtss2si under the default rounding mode).
static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvRound(float v) { return (int)std::lrintf(v); }
static inline int cvRound(int v) { return v; }
static inline int cvFloor(double v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cvCeil(double v) {
    int i = (int)v;
    return i + (i < v);
}

// cv::RNG — multiply-with-carry generator (core/operations.hpp, RNG::next / RNG::uniform).
struct RNG {
    uint64_t state;
    explicit RNG(uint64_t s) : state(s ? s : 0xffffffffu) {}
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
};

}  // namespace vo_oracle
#endif
