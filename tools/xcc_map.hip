// Which XCD does workgroup b of a launch run on?  For a few grid sizes, alone and with a second launch running beside it
// on another stream: how many workgroups per XCC, and the offset c for which XCC_ID == (b + c) % 8 holds for ALL
// workgroups of the launch (-1: no single offset) -- the round-robin dealing the step kernels' XCD-aware block remap
// relies on (for speed only).   hipcc --offload-arch=gfx950 -O2 -o tools/xcc_map tools/xcc_map.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void record(int *xcc, int spin) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = (int)__builtin_amdgcn_s_getreg(6164) & 15;   // hwreg(HW_REG_XCC_ID, 0, 4)
    for (volatile int k = 0; k < spin; ++k) {}
}

int main() {
    hipStream_t a, b;
    hipStreamCreate(&a);
    hipStreamCreate(&b);
    for (int n : {256, 4096, 4099, 65536}) {
        int *d0, *d1;
        hipMalloc(&d0, n * sizeof(int));
        hipMalloc(&d1, n * sizeof(int));
        for (int both = 0; both < 2; ++both) {
            hipLaunchKernelGGL(record, dim3(n), dim3(256), 0, a, d0, 200);
            if (both) hipLaunchKernelGGL(record, dim3(n), dim3(256), 0, b, d1, 200);
            hipDeviceSynchronize();
            std::vector<int> h(n);
            hipMemcpy(h.data(), d0, n * sizeof(int), hipMemcpyDeviceToHost);
            int hist[16] = {0}, off = (h[0] + 8) % 8;
            for (int k = 0; k < n; ++k) {
                if (h[k] != (k + off) % 8) off = -1 - 100 * (off < 0);
                if (off < -1) off = -1;
                hist[h[k] & 15]++;
            }
            printf("{\"workgroups\": %d, \"second_stream_busy\": %d, \"offset\": %d, \"per_xcc\": [%d,%d,%d,%d,%d,%d,%d,%d]}\n", n, both,
                   off, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
        }
        hipFree(d0);
        hipFree(d1);
    }
    return 0;
}
