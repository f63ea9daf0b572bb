// Micro-benchmark: cost of fp64 VALU instructions for ONE wave per SIMD on gfx950 (the regime the
// five-point solver runs in): dependent chains vs independent chains, fma vs mul+add, LDS b64.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/dp_latency tools/micro/dp_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
    __shared__ double lds[64 * 32];
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double x = 1.0000001, y = 1e-9;
    for (int i = 0; i < 32; i++) lds[i * 64 + threadIdx.x] = seed + i;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {  // one dependent chain of 32 fma
#pragma unroll
            for (int i = 0; i < 32; i++) a[0] = __builtin_fma(a[0], x, y);
        } else if (MODE == 1) {  // 8 independent chains, 32 fma in total
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] = __builtin_fma(a[j], x, y);
        } else if (MODE == 2) {  // dependent mul, add pairs (no contraction): 32 instructions
#pragma unroll
            for (int i = 0; i < 16; i++) {
                a[0] = a[0] * x;
                asm volatile("" : "+v"(a[0]));
                a[0] = a[0] + y;
                asm volatile("" : "+v"(a[0]));
            }
        } else if (MODE == 3) {  // f32 dependent chain of 32 fma
            float f = (float)a[0];
#pragma unroll
            for (int i = 0; i < 32; i++) f = __builtin_fmaf(f, 1.0000001f, 1e-9f);
            a[0] = f;
        } else if (MODE == 4) {  // 32 LDS reads b64 (independent) + 32 adds
#pragma unroll
            for (int i = 0; i < 32; i++) a[i & 7] += lds[i * 64 + threadIdx.x];
        } else if (MODE == 5) {  // dependent rcp chain x8 (counted as 32 for the report: 4 per)
#pragma unroll
            for (int i = 0; i < 32; i++) a[0] = __builtin_amdgcn_rcp(a[0]) + 1.5;
        }
    }
    double r = 0;
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, int blocks) {
    double* d;
    (void)hipMalloc(&d, (size_t)blocks * 64 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int r = 0; r < 2; r++) {
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 64>>>(d, iters, 1.0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-44s blocks %5d  %.2f ns per instruction per wave\n", name, blocks, ms * 1e6 / ((double)iters * 32));
    (void)hipFree(d);
}

int main() {
    for (int blocks : {256, 1024, 4096}) {
        run<0>("fma_f64 dependent chain", blocks);
        run<1>("fma_f64 8 independent chains", blocks);
        run<2>("mul_f64 / add_f64 dependent", blocks);
        run<3>("fma_f32 dependent chain", blocks);
        run<4>("ds_read_b64 + add_f64", blocks);
        run<5>("rcp_f64 + add_f64 dependent (2 instr each)", blocks);
    }
    return 0;
}
