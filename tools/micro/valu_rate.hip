// Micro-benchmark: issue cost in cycles of single VALU instructions and of the matcher's per-row
// instruction mix on gfx950, written as inline asm so that the stream is exactly what is listed.
// Every block is 256 threads (one wave per SIMD); `blocks per CU` resident blocks give that many
// waves per SIMD.  Cycles per wave-instruction per SIMD = elapsed * clock / (instructions per wave *
// waves per SIMD); the clock is measured in the kernel (s_memtime ticks per s_memrealtime 100 MHz tick).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rate tools/micro/valu_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP7(x) x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, unsigned long long* clk) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17,
             a7 = a0 * 19;
    uint32_t b0 = 1, b1 = 2, b2 = 3, b3 = 4, b4 = 5, b5 = 6, b6 = 7, b7 = 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {  // 64 x v_xor_b32 (8 independent chains)
            REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                              "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(b0));)
        } else if (MODE == 1) {  // 64 x v_bcnt_u32_b32 v, v, 0
            REP8(asm volatile("v_bcnt_u32_b32 %0, %8, 0\n v_bcnt_u32_b32 %1, %9, 0\n v_bcnt_u32_b32 %2, %10, 0\n v_bcnt_u32_b32 %3, %11, 0\n"
                              "v_bcnt_u32_b32 %4, %12, 0\n v_bcnt_u32_b32 %5, %13, 0\n v_bcnt_u32_b32 %6, %14, 0\n v_bcnt_u32_b32 %7, %15, 0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));)
        } else if (MODE == 2) {  // 64 x v_bcnt_u32_b32 v, v, v (accumulating, 8 independent chains)
            REP8(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3\n"
                              "v_bcnt_u32_b32 %4, %12, %4\n v_bcnt_u32_b32 %5, %13, %5\n v_bcnt_u32_b32 %6, %14, %6\n v_bcnt_u32_b32 %7, %15, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));)
        } else if (MODE == 3) {  // 64 x v_med3_u32
            REP8(asm volatile("v_med3_u32 %0, %0, %8, %9\n v_med3_u32 %1, %1, %8, %9\n v_med3_u32 %2, %2, %8, %9\n v_med3_u32 %3, %3, %8, %9\n"
                              "v_med3_u32 %4, %4, %8, %9\n v_med3_u32 %5, %5, %8, %9\n v_med3_u32 %6, %6, %8, %9\n v_med3_u32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(b0), "v"(b1));)
        } else if (MODE == 4) {  // 64 x v_lshl_or_b32 v, v, 22, v
            REP8(asm volatile("v_lshl_or_b32 %0, %0, 22, %8\n v_lshl_or_b32 %1, %1, 22, %8\n v_lshl_or_b32 %2, %2, 22, %8\n v_lshl_or_b32 %3, %3, 22, %8\n"
                              "v_lshl_or_b32 %4, %4, 22, %8\n v_lshl_or_b32 %5, %5, 22, %8\n v_lshl_or_b32 %6, %6, 22, %8\n v_lshl_or_b32 %7, %7, 22, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(b0));)
        } else if (MODE == 5) {  // 64 x v_min_u32
            REP8(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n"
                              "v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(b0));)
        } else if (MODE == 6) {  // the matcher row: 8 xor + 8 bcnt (chain) + lshl_or + med3 + min, 4 rows interleaved = 76 instr
            asm volatile(
                "v_xor_b32 %[x0], %[q0], %[t0]\n v_xor_b32 %[x1], %[q1], %[t0]\n v_xor_b32 %[x2], %[q2], %[t0]\n v_xor_b32 %[x3], %[q3], %[t0]\n"
                "v_bcnt_u32_b32 %[d0], %[x0], 0\n v_bcnt_u32_b32 %[d1], %[x1], 0\n v_bcnt_u32_b32 %[d2], %[x2], 0\n v_bcnt_u32_b32 %[d3], %[x3], 0\n"
                REP7("v_xor_b32 %[x0], %[q0], %[t1]\n v_xor_b32 %[x1], %[q1], %[t1]\n v_xor_b32 %[x2], %[q2], %[t1]\n v_xor_b32 %[x3], %[q3], %[t1]\n"
                     "v_bcnt_u32_b32 %[d0], %[x0], %[d0]\n v_bcnt_u32_b32 %[d1], %[x1], %[d1]\n v_bcnt_u32_b32 %[d2], %[x2], %[d2]\n v_bcnt_u32_b32 %[d3], %[x3], %[d3]\n")
                "v_lshl_or_b32 %[d0], %[d0], 22, %[t0]\n v_lshl_or_b32 %[d1], %[d1], 22, %[t0]\n v_lshl_or_b32 %[d2], %[d2], 22, %[t0]\n v_lshl_or_b32 %[d3], %[d3], 22, %[t0]\n"
                "v_med3_u32 %[k2], %[d0], %[k1], %[k2]\n v_min_u32 %[k1], %[d0], %[k1]\n v_med3_u32 %[k2], %[d1], %[k1], %[k2]\n v_min_u32 %[k1], %[d1], %[k1]\n"
                "v_med3_u32 %[k2], %[d2], %[k1], %[k2]\n v_min_u32 %[k1], %[d2], %[k1]\n v_med3_u32 %[k2], %[d3], %[k1], %[k2]\n v_min_u32 %[k1], %[d3], %[k1]\n"
                : [d0] "+v"(a0), [d1] "+v"(a1), [d2] "+v"(a2), [d3] "+v"(a3), [x0] "+v"(a4), [x1] "+v"(a5), [x2] "+v"(a6),
                  [x3] "+v"(a7), [k1] "+v"(b6), [k2] "+v"(b7)
                : [q0] "v"(b0), [q1] "v"(b1), [q2] "v"(b2), [q3] "v"(b3), [t0] "v"(b4), [t1] "v"(b5));
        } else if (MODE == 7) {  // 32 x (v_xor + v_bcnt acc) alternating: the core of the row
            REP8(asm volatile("v_xor_b32 %4, %8, %9\n v_bcnt_u32_b32 %0, %4, %0\n v_xor_b32 %5, %8, %10\n v_bcnt_u32_b32 %1, %5, %1\n"
                              "v_xor_b32 %6, %8, %11\n v_bcnt_u32_b32 %2, %6, %2\n v_xor_b32 %7, %8, %12\n v_bcnt_u32_b32 %3, %7, %3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b6 ^ b7;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

template <int MODE>
void run(const char* name, int instr_per_iter, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu;
    uint32_t* d;
    unsigned long long* c;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipMalloc(&c, 16);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters, c);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, c);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)instr_per_iter * iters * blocks_per_cu);
    printf("%-44s %d waves/SIMD  %.3f ms  clock %.2f GHz  %.2f cycles per wave-instruction\n", name, blocks_per_cu, ms,
           ghz, cyc);
    hipFree(d);
    hipFree(c);
}

int main() {
    for (int w : {1, 2, 4, 6, 8}) {
        run<0>("v_xor_b32", 64, w);
        run<1>("v_bcnt_u32_b32 v, v, 0", 64, w);
        run<2>("v_bcnt_u32_b32 v, v, v (accumulate)", 64, w);
        run<3>("v_med3_u32", 64, w);
        run<4>("v_lshl_or_b32", 64, w);
        run<5>("v_min_u32", 64, w);
        run<7>("xor + bcnt-accumulate alternating", 64, w);
        run<6>("matcher rows x4 (8 xor 8 bcnt pack med3 min)", 76, w);
    }
    return 0;
}
