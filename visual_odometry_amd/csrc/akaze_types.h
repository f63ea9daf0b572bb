// akaze_types.h — shared between k_akaze.hip and vo_akaze.cpp (not part of the ABI).
#ifndef VO_AKAZE_TYPES_H
#define VO_AKAZE_TYPES_H

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/vo_types.h"

#define AK_MAX_LEVELS 32
#define AK_MAX_OCTAVES 8
#define AK_MAX_BINS 1024
#define AK_DESC_BYTES VO_AKAZE_DESC_BYTES

struct AkTaps {  // half of a symmetric kernel: k[0] centre, k[j] at distance j
    int r;
    float k[8];
};

struct AkContrast {  // device-resident result of compute_kcontrast
    unsigned int hmax_bits;
    unsigned int hist[AK_MAX_BINS];
    float k[AK_MAX_OCTAVES];  // contrast factor per octave
};

struct AkLevel {
    int w, h;
    float ratio;           // 2^octave
    unsigned lt, lx, ly;   // float offsets of the planes in the arena
};

struct AkLevels {  // by value into the descriptor kernels
    AkLevel l[AK_MAX_LEVELS];
    signed char oi[112], oj[112];  // the 109 orientation samples in scan order
    float gauss25[7][7];
    int nbits;
};

struct AkExt {  // one 3 x 3 maximum of the determinant: pixel offset over all levels, its neighbourhood row-major
    int at;
    float d[9];
};

struct AkPair {  // one descriptor bit: vals[a] > vals[b]
    unsigned short a, b;
};

void ak_launch_cvt(hipStream_t st, const uint8_t* gray, int w, int h, float* out);
void ak_launch_gauss(hipStream_t st, const float* src, float* tmp, float* dst, int w, int h, const AkTaps& t);
void ak_launch_grad(hipStream_t st, const float* S, float* Lx, float* Ly, int w, int h, int scale, float norm,
                    float wnorm);
void ak_launch_cond(hipStream_t st, const float* S, int w, int h, const AkContrast* kc, int octave, float* flow);
void ak_launch_hessdet(hipStream_t st, const float* Lx, const float* Ly, int w, int h, int scale, float norm,
                       float wnorm, float s4, float* Ldet);
void ak_launch_kcontrast(hipStream_t st, const float* Lx, const float* Ly, int w, int h, float perc, int nbins,
                         int n_octaves, AkContrast* kc);
void ak_launch_nld(hipStream_t st, const float* Lt, const float* flow, float* out, int w, int h, float step);
void ak_launch_area(hipStream_t st, const float* src, int sw, float* dst, int dw, int dh, const int* xfirst,
                    const int* xsi, const float* xalpha, const int* yfirst, const int* ysi, const float* yalpha);
void ak_launch_extrema(hipStream_t st, const float* L, int w, int h, int border, float thr, int level_at, AkExt* list,
                       unsigned* count, unsigned cap);
void ak_launch_describe(hipStream_t st, vo_keypoint* kps, int n, const AkLevels& lv, const float* arena,
                        const AkPair* pairs, int pattern, int chan, uint8_t* desc);

#endif
