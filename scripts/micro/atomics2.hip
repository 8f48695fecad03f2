// dev micro-benchmark: LDS int/float atomics and global float atomics, lane-ops per second (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(1024) void k(float *g, int npix, int iters, int same_rows) {
    extern __shared__ float tile[];
    for (int e = threadIdx.x; e < 24000; e += 1024) tile[e] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = lane & 15, row = lane >> 4;
    unsigned rng = 1u + blockIdx.x * 7919u + (threadIdx.x >> 6) * 104729u;
    float v = 1.0f + lane;
    for (int i = 0; i < iters; ++i) {
        rng = rng * 1664525u + 1013904223u;
        const unsigned r = (rng >> 8) * (same_rows ? 1 : (2 * row + 1)) + (same_rows ? 0 : row * 977);
        if (KIND == 0) atomicAdd(&tile[(r % 1500) * 16 + sub], v);
        if (KIND == 1) atomicAdd((unsigned *)&tile[(r % 1500) * 16 + sub], (unsigned)lane);
        if (KIND == 2) atomicAdd((unsigned long long *)&tile[(r % 750) * 32 + sub * 2], (unsigned long long)lane);
        if (KIND == 3) unsafeAtomicAdd(g + (size_t)(r % npix) * 16 + sub, v);
        if (KIND == 4) { float o = atomicAdd(&tile[(r % 1500) * 16 + sub], v); v += o * 1e-30f; }
        if (KIND == 6) unsafeAtomicAdd((double *)&tile[(r % 750) * 32 + sub * 2], (double)v);
        if (KIND == 7) unsafeAtomicAdd((double *)g + (size_t)(r % npix) * 16 + sub, (double)v);
        if (KIND == 5) { tile[(r % 1500) * 16 + sub] += v; }   // racy plain RMW, cost floor
    }
    __syncthreads();
    float s = 0;
    for (int e = threadIdx.x; e < 24000; e += 1024) s += tile[e];
    if (s == 12345.f) g[0] = s + v;
}
template <int KIND>
void run(const char *name, float *g, int npix, int same) {
    (void)hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 96000);
    const int iters = 2048, blocks = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 96000, 0, g, npix, 16, same);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 96000, 0, g, npix, iters, same);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double ops = 256.0 * 1024 * iters;
    printf("%-34s npix=%8d same_rows=%d : %7.3f ms  %7.1f G lane-ops/s  (%.1f clk / wave-instr / CU)\n", name, npix, same, ms,
           ops / ms / 1e6, ms * 1e-3 * 2.4e9 / (16.0 * iters));
}
int main() {
    float *g; (void)hipMalloc(&g, (size_t)16 << 20 << 2); (void)hipMemset(g, 0, (size_t)16 << 20 << 2);
    for (int same = 0; same < 2; ++same) {
        run<0>("lds f32 add", g, 0, same);
        run<4>("lds f32 add rtn", g, 0, same);
        run<1>("lds u32 add", g, 0, same);
        run<2>("lds u64 add", g, 0, same);
        run<5>("lds plain rmw (racy)", g, 0, same);
        run<6>("lds f64 add", g, 0, same);
        run<7>("global f64 add, 19200 px", g, 19200, same);
        run<3>("global f32 add, 1M px (64 MB)", g, 1 << 20, same);
        run<3>("global f32 add, 19200 px", g, 19200, same);
        run<3>("global f32 add, 1200 px", g, 1200, same);
        run<3>("global f32 add, 300 px", g, 300, same);
    }
    return 0;
}
