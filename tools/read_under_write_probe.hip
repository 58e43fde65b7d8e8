// tools/read_under_write_probe.hip -- why does ONE look-ahead load per bin that misses the caches cost the step kernel 6 us of 28?
// A stand-in for the step kernel's memory behaviour: 4096 workgroups x 256 threads, every workgroup streams 32 KB of
// float4 stores (the observation / mask writes) and its first wave reads ONE dword per group of four lanes from a pool:
//   mode 0  no pool read;                         mode 1  random rows anywhere in a 512 MB pool;
//   mode 2  the 16 reads of a workgroup inside ONE 128 KB window that belongs to the workgroup (bin-major ring layout);
//   mode 3  random rows inside 64 MB;             mode 4  as 1, but the value is only consumed at the end of the kernel.
// Prints microseconds per launch for each mode (outputs rotated over 1 GB so that stores go to HBM).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(float4 *out, const uint32_t *pool, size_t pool_words, uint32_t salt, uint32_t *sink) {
    const int b = blockIdx.x, t = threadIdx.x;
    uint32_t v = 0;
    if (MODE != 0 && t < 64) {
        const uint32_t h = mix(salt ^ (uint32_t)(b * 16 + (t >> 2)));
        size_t idx;
        if (MODE == 2) idx = (size_t)b * 32768 + (h & 32767u);             // 128 KB window of this workgroup (32768 words)
        else if (MODE == 3) idx = h & ((16u << 20) - 1u);                  // inside 64 MB
        else idx = (size_t)h % pool_words;                                 // anywhere
        v = pool[idx];
        if (MODE != 4) {   // consumed right away, like a value the decision needs
            __shared__ uint32_t s[64];
            s[t] = v;
        }
    }
    __syncthreads();
    float4 *o = out + (size_t)b * 2048 + t;      // 32 KB per workgroup
    const float f = (float)b;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k * 256] = make_float4(f, f + 1.f, f + 2.f, (float)k);
    if (MODE == 4 && v == 0xffffffffu) sink[0] = v;
    if (MODE != 0 && MODE != 4 && t == 0 && v == 0xffffffffu) sink[1] = v;
}

int main() {
    const size_t pool_bytes = (size_t)512 << 20, pool_words = pool_bytes / 4;
    const int blocks = 4096, sets = 8;
    uint32_t *pool, *sink;
    float4 *out;
    CHECK(hipMalloc(&pool, pool_bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&out, (size_t)sets * blocks * 32768));
    CHECK(hipMemset(pool, 1, pool_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode <= 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            for (int it = 0; it < 40; ++it) {
                float4 *o = out + (size_t)(it % sets) * blocks * 2048;
                const uint32_t salt = (uint32_t)(rep * 1000 + it) * 2654435761u;
                switch (mode) {
                    case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, o, pool, pool_words, salt, sink); break;
                    case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, o, pool, pool_words, salt, sink); break;
                    case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, o, pool, pool_words, salt, sink); break;
                    case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, o, pool, pool_words, salt, sink); break;
                    default: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, o, pool, pool_words, salt, sink); break;
                }
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("{\"mode\": %d, \"us_per_launch\": %.2f}\n", mode, best / 40 * 1e3);
    }
    return 0;
}
