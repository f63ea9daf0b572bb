// Micro-benchmark: where one five-point solve spends its time on a gfx950 lane (phase timestamps
// with s_memtime), launched like k_solve5<L>: L lanes per block.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I visual_odometry_amd/csrc \
//        -o tools/micro/solve_time tools/micro/solve_time.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

__device__ long long* g_ticks;
#define FP_TICK(i) (g_ticks[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 8 + (i)] = clock64())
__device__ unsigned long long g_cnt[4];
#ifdef COUNT_STEPS  // -DCOUNT_STEPS: iteration counts (the atomics distort the timings)
#define FP_COUNT(i) atomicAdd(&g_cnt[i], 1ull)
#else
#define FP_COUNT(i)
#endif
__device__ int g_stop = 99;
#define FP_STOP(i, v)            \
    if (g_stop == (i)) {         \
        E_out[0] = (v);          \
        return 1;                \
    }
#include "five_point.h"

template <int LANES>
struct LaneScratch {
    double* base;
    __device__ __forceinline__ double& operator()(int i) const { return base[i * LANES]; }
};

template <int LANES>
__global__ __launch_bounds__(LANES) void k(const double* pts, double* models, int* nm) {
    __shared__ double lds[FP_SCRATCH * LANES];
    const int h = blockIdx.x * LANES + threadIdx.x;
    double a[10], b[10];
    for (int i = 0; i < 10; i++) {
        a[i] = pts[h * 20 + i];
        b[i] = pts[h * 20 + 10 + i];
    }
    FP_TICK(0);
    LaneScratch<LANES> s{lds + threadIdx.x};
    nm[h] = five_point_solve(a, b, s, models + (size_t)h * 90);
    FP_TICK(6);
}

template <int LANES>
void run(int nh) {
    std::mt19937 rng(1);
    std::uniform_real_distribution<double> U(-0.5, 0.5), Z(4, 40);
    std::vector<double> pts((size_t)nh * 20);
    for (int h = 0; h < nh; h++)
        for (int i = 0; i < 5; i++) {
            // a real two-view geometry: small rotation about y, translation mostly forward
            const double X = U(rng) * 20, Y = U(rng) * 4, Zz = Z(rng);
            const double th = 0.02, tx = 0.1, tz = 1.0;
            const double X2 = cos(th) * X + sin(th) * Zz + tx, Z2 = -sin(th) * X + cos(th) * Zz - tz;
            pts[h * 20 + 2 * i] = X / Zz + U(rng) * 1e-3;
            pts[h * 20 + 2 * i + 1] = Y / Zz + U(rng) * 1e-3;
            pts[h * 20 + 10 + 2 * i] = X2 / Z2 + U(rng) * 1e-3;
            pts[h * 20 + 10 + 2 * i + 1] = Y / Z2 + U(rng) * 1e-3;
        }
    double *d_pts, *d_models;
    int* d_nm;
    long long* d_t;
    hipMalloc(&d_pts, pts.size() * 8);
    hipMalloc(&d_models, (size_t)nh * 90 * 8);
    hipMalloc(&d_nm, nh * 4);
    hipMalloc(&d_t, (size_t)nh * 8 * 8);
    hipMemcpy(d_pts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), &d_t, sizeof(d_t));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<LANES><<<nh / LANES, LANES>>>(d_pts, d_models, d_nm);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> t((size_t)nh * 8);
    std::vector<int> nm(nh);
    hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(nm.data(), d_nm, nh * 4, hipMemcpyDeviceToHost);
    double ph[6] = {0}, mx[6] = {0};
    double nmavg = 0;
    for (int h = 0; h < nh; h++) {
        nmavg += nm[h];
        for (int i = 0; i < 6; i++) {
            const double d = (double)(t[h * 8 + i + 1] - t[h * 8 + i]);
            ph[i] += d / nh;
            if (d > mx[i]) mx[i] = d;
        }
    }
    unsigned long long cnt[4] = {0, 0, 0, 0}, zero[4] = {0, 0, 0, 0};
    hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_cnt), sizeof(cnt));
    hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), zero, sizeof(zero));
    printf("per solve: %.1f newton/bisection steps, %.1f outward steps, %.1f roots bracketed\n", cnt[0] / 3.0 / nh,
           cnt[1] / 3.0 / nh, cnt[2] / 3.0 / nh);
    const char* names[6] = {"nullspace", "constraints", "gauss-jordan", "B(z)+det", "roots", "backsub"};
    printf("LANES %d, %d hypotheses, kernel %.1f us, models/hyp %.2f\n", LANES, nh, ms * 1e3, nmavg / nh);
    for (int i = 0; i < 6; i++) printf("  %-14s mean %9.0f  max %9.0f  clock64 ticks\n", names[i], ph[i], mx[i]);
    hipFree(d_pts);
    hipFree(d_models);
    hipFree(d_nm);
    hipFree(d_t);
}

int main(int argc, char** argv) {
    if (argc > 1) {  // cumulative time up to the end of phase argv[1] (1 null space .. 5 roots, 99 all)
        for (int stop : {1, 2, 3, 4, 5, 99}) {
            hipMemcpyToSymbol(HIP_SYMBOL(g_stop), &stop, sizeof(int));
            printf("== stop after phase %d\n", stop);
            run<4>(2048);
        }
        return 0;
    }
    run<4>(2048);
    run<64>(2048);
    run<64>(65536);
    return 0;
}
