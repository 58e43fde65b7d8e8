#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 17 + i;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)a[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = a[i] + a[(i + 1) & 7];                  // v_add_u32
            if (KIND == 1) a[i] = a[i] ^ (a[(i + 1) & 7] >> 3);            // v_lshrrev + v_xor
            if (KIND == 2) a[i] = max(a[i], a[(i + 3) & 7] + 1u);          // v_add + v_max
            if (KIND == 3) a[i] = a[i] * 2654435761u + 1u;                  // v_mul_lo (+add)
            if (KIND == 4) f[i] = f[i] * 1.0001f + 0.5f;                    // v_fma_f32
            if (KIND == 5) a[i] = (a[i] > a[(i + 1) & 7]) ? a[(i + 2) & 7] : a[i] + 1;  // cmp+cndmask+add
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> void run(const char *name, int ops_per_elem, uint32_t *d) {
    const int blocks = 256 * 8, iters = 4096;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 8 * ops_per_elem;      // wave-instructions
    double per_simd_per_s = winstr / (ms * 1e-3) / 1024.0;
    printf("%-28s %.3f ms  %.2f G wave-instr/s per SIMD -> %.2f cycles/wave-instr @2.4GHz (%.2f @2.0GHz)\n", name, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, 2.0e9 / per_simd_per_s);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", 1, d);
    run<1>("v_lshrrev+v_xor", 2, d);
    run<2>("v_add+v_max", 2, d);
    run<3>("v_mul_lo(+add)", 2, d);
    run<4>("v_fma_f32", 1, d);
    run<5>("cmp+cndmask+add", 3, d);
    return 0;
}
