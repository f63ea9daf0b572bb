// What does a vector instruction cost beside v_mfma_i32_32x32x32_i8?  Each trip: 16 x (one MFMA, N fillers of
// one kind on independent registers), W waves per SIMD on every CU.  Prints shader cycles per MFMA gap
// (s_memtime of one wave) for N = 0, 4, 8, 12.
// W waves per SIMD are forced by the block size (256 W threads, one block per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_valu tools/micro/mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define FILL(OP)                                                                                               \
    template <int N> struct OP##_f {                                                                           \
        static __device__ __forceinline__ void run(uint32_t (&x)[12], uint32_t y) {                            \
            _Pragma("unroll") for (int i = 0; i < N; i++) OP(x[i], y);                                          \
        }                                                                                                      \
    };
static __device__ __forceinline__ void op_and(uint32_t& x, uint32_t y) { asm volatile("v_and_b32 %0, 0x10101010, %0" : "+v"(x)); }
static __device__ __forceinline__ void op_and_r(uint32_t& x, uint32_t y) { asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(y)); }
static __device__ __forceinline__ void op_min_i16(uint32_t& x, uint32_t y) { asm volatile("v_min_i16 %0, %1, %0" : "+v"(x) : "v"(y)); }
static __device__ __forceinline__ void op_min_u16(uint32_t& x, uint32_t y) { asm volatile("v_min_u16 %0, %1, %0" : "+v"(x) : "v"(y)); }
static __device__ __forceinline__ void op_max_i16(uint32_t& x, uint32_t y) { asm volatile("v_max_i16 %0, %1, %0" : "+v"(x) : "v"(y)); }
static __device__ __forceinline__ void op_min3_i16(uint32_t& x, uint32_t y) { asm volatile("v_min3_i16 %0, %1, %0, %1" : "+v"(x) : "v"(y)); }
static __device__ __forceinline__ void op_min_u32(uint32_t& x, uint32_t y) { asm volatile("v_min_u32 %0, %1, %0" : "+v"(x) : "v"(y)); }
static __device__ __forceinline__ void op_add_u32(uint32_t& x, uint32_t y) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(y)); }
static __device__ __forceinline__ void op_pk_min_i16(uint32_t& x, uint32_t y) { asm volatile("v_pk_min_i16 %0, %1, %0" : "+v"(x) : "v"(y)); }
FILL(op_and) FILL(op_and_r) FILL(op_min_i16) FILL(op_min_u16) FILL(op_max_i16) FILL(op_min3_i16) FILL(op_min_u32) FILL(op_add_u32) FILL(op_pk_min_i16)

template <template <int> class F, int N, bool MFMA> __global__ __launch_bounds__(1024) void k(int* out, int iters, unsigned long long* clk) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    i32x16 c0 = {0}, c1 = {0};
    uint32_t x[12];
    for (int i = 0; i < 12; i++) x[i] = threadIdx.x * (i + 3);
    const uint32_t y = threadIdx.x * 77u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MFMA) c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            F<N>::run(x, y);
            if (MFMA) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            F<N>::run(x, y);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i];
    for (int i = 0; i < 12; i++) s += x[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicMax(clk, t1 - t0);  // the slowest wave of the block
}
template <template <int> class F, int N, bool MFMA> double run(int* d, unsigned long long* dc, int w) {
    const int iters = 500;
    if (hipMemset(dc, 0, 8) != hipSuccess) return -1;
    hipLaunchKernelGGL((k<F, N, MFMA>), dim3(256), dim3(256 * w), 0, 0, d, iters, dc);
    unsigned long long c = 0;
    if (hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (double)c / (iters * 16.0) / w;  // SIMD cycles per (MFMA + N fillers) of any wave
}
template <template <int> class F> void row(const char* name, int* d, unsigned long long* dc) {
    for (int w = 1; w <= 4; w *= 2)
        printf("%-12s %d wave(s)/SIMD: gap with 0 / 4 / 8 / 12 fillers: %5.1f %5.1f %5.1f %5.1f   fillers alone (8): %5.1f\n", name, w,
               run<F, 0, true>(d, dc, w), run<F, 4, true>(d, dc, w), run<F, 8, true>(d, dc, w), run<F, 12, true>(d, dc, w),
               run<F, 8, false>(d, dc, w));
}
int main() {
    int* d; unsigned long long* dc;
    if (hipMalloc(&d, 256 * 1024 * sizeof(int)) != hipSuccess || hipMalloc(&dc, 8) != hipSuccess) return 1;
    row<op_and_f>("and literal", d, dc);
    row<op_and_r_f>("and reg", d, dc);
    row<op_add_u32_f>("add_u32", d, dc);
    row<op_min_u32_f>("min_u32", d, dc);
    row<op_min_u16_f>("min_u16", d, dc);
    row<op_min_i16_f>("min_i16", d, dc);
    row<op_max_i16_f>("max_i16", d, dc);
    row<op_min3_i16_f>("min3_i16", d, dc);
    row<op_pk_min_i16_f>("pk_min_i16", d, dc);
    return 0;
}
