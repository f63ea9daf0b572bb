// Micro-benchmark: issue rate of the matcher's instruction mix on gfx950 (lane-ops/s, whole chip).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_peak tools/micro/valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8], acc[4] = {0, 0, 0, 0};
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = seed * (threadIdx.x + 1) + i * 0x9e3779b9u;
        f[i] = (float)a[i];
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE == 0) {  // xor + bcnt (accumulating), the matcher's body
#pragma unroll
                for (int i = 0; i < 8; i++) acc[u] = __builtin_popcount(a[i] ^ (acc[(u + 1) & 3] + i)) + acc[u];
            } else if (MODE == 1) {  // xor only
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] ^= acc[u] + i;
                acc[u] ^= a[u];
            } else if (MODE == 2) {  // fma f32
#pragma unroll
                for (int i = 0; i < 8; i++) f[i] = f[i] * 1.0001f + 0.5f;
            } else if (MODE == 3) {  // bcnt only
#pragma unroll
                for (int i = 0; i < 8; i++) acc[u] = __builtin_popcount(a[i]) + acc[u];
                a[u] += acc[u];
            } else if (MODE == 4) {  // min/max u32
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = min(a[i], max(a[(i + 1) & 7], acc[u]));
            }
        }
    }
    uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ (uint32_t)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, double ops_per_iter, int blocks) {
    uint32_t* d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * iters * ops_per_iter;
    printf("%-28s blocks %5d  %.3f ms  %.1f T lane-ops/s (counting source-level ops)\n", name, blocks, ms,
           lane_ops / ms / 1e9);
    hipFree(d);
}

int main() {
    for (int blocks : {1024, 2048, 4096}) {
        run<0>("xor+add+bcnt (3 ops) x32", 32 * 3.0, blocks);
        run<1>("add+xor (2 ops) x32", 32 * 2.0 + 4, blocks);
        run<2>("fma_f32 x32", 32 * 1.0, blocks);
        run<3>("bcnt x32", 32 * 1.0 + 4, blocks);
        run<4>("max+min x32", 32 * 2.0, blocks);
    }
    return 0;
}
