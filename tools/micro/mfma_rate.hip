// Issue rate of the int8 matrix instructions next to the bf16 one: N back-to-back MFMAs per wave on one
// or two accumulators, W waves per SIMD, every CU busy.  Prints cycles (at the measured wall time and the
// s_memrealtime 100 MHz clock) per instruction and SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_rate tools/micro/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE> __global__ __launch_bounds__(256) void k(int* out, int iters) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    i32x16 c0 = {0}, c1 = {0};
    i32x4 d0 = {0}, d1 = {0};
    f32x16 f0 = {0}, f1 = {0};
    bf16x8 ha, hb;
    for (int i = 0; i < 8; i++) { ha[i] = (__bf16)(float)(threadIdx.x + i); hb[i] = (__bf16)(float)i; }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                f0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, f0, 0, 0, 0);
                f1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, f1, 0, 0, 0);
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + (int)f0[i] + (int)f1[i];
    for (int i = 0; i < 4; i++) s += d0[i] + d1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int* d, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks / 256.0 * iters * 16;  // one wave of each block per SIMD
    printf("%-22s blocks/CU %d: %.3f ms, %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz, %.1f at 2.0)\n", name,
           blocks / 256, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, ms * 1e6 / per_simd * 2.0);
}
int main() {
    int* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(int));
    for (int w = 1; w <= 2; w++) {
        run<0>("i32_32x32x32_i8", d, 256 * w);
        run<1>("i32_16x16x64_i8", d, 256 * w);
        run<2>("f32_32x32x16_bf16", d, 256 * w);
    }
    return 0;
}
