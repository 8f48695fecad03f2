// dev micro-benchmark: cost of ds_add_f32 under different address patterns (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(1024) void k(float *out, int mode, int iters, unsigned seed) {
    extern __shared__ float tile[];
    for (int e = threadIdx.x; e < 24000; e += 1024) tile[e] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = lane & 15, row = lane >> 4;
    unsigned rng = seed + blockIdx.x * 7919u + (threadIdx.x >> 6) * 104729u;
    float v = 1.0f + lane;
    for (int i = 0; i < iters; ++i) {
        rng = rng * 1664525u + 1013904223u;
        int pix;
        const unsigned r = rng >> 8;
        if (mode == 0) pix = (r % 370) * 4 + row;                 // 64 consecutive floats (4 adjacent pixels)
        else if (mode == 1) pix = r % 1500;                        // all 4 rows SAME pixel
        else if (mode == 2) pix = (r * (row * 2 + 1) + row * 977) % 1500;   // rows at unrelated pixels
        else if (mode == 3) pix = (r % 1500) / 2 * 2 + (row & 1);  // pairs of rows share a pixel
        else pix = (r % 370) * 4 + (row ^ (i & 3));
        if (mode == 5) { if (row == 0) atomicAdd(&tile[(r % 1500) * 16 + sub], v * 4); }   // one row active only
        else atomicAdd(&tile[pix * 16 + sub], v);
    }
    __syncthreads();
    float s = 0;
    for (int e = threadIdx.x; e < 24000; e += 1024) s += tile[e];
    if (s == 12345.f) out[0] = s;
}
int main() {
    float *out; hipMalloc(&out, 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96000);
    const int iters = 4096, blocks = 256;
    for (int mode = 0; mode <= 5; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 96000, 0, out, mode, 16, 1u);
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 96000, 0, out, mode, iters, 1u);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // per CU: 16 waves x iters instructions
        const double inst_per_cu = 16.0 * iters;
        printf("mode %d: %.3f ms  -> %.1f clk per ds_add_f32 wave-instr per CU (2.4 GHz)\n", mode, ms, ms * 1e-3 * 2.4e9 / inst_per_cu);
    }
    return 0;
}
