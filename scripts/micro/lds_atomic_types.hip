// dev micro-benchmark: issue cost of LDS atomics by operand type on gfx950 (16-lane rows at random pixels of a tile,
// the address pattern of msda_bwd_band_list_kernel / rb_brick_kernel).  hipcc --offload-arch=gfx950 -O3 -o lds_atomic_types
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __device__ T mk(float v);
template <> __device__ float mk<float>(float v) { return v; }
template <> __device__ double mk<double>(float v) { return (double)v; }
template <> __device__ unsigned mk<unsigned>(float v) { return (unsigned)(int)(v * 1024.f); }
template <> __device__ unsigned long long mk<unsigned long long>(float v) { return (unsigned long long)(long long)(v * 1048576.f); }

template <typename T, int ROWW, int ACTIVE_ROWS = 64>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned seed, int n_pix) {
    extern __shared__ unsigned char raw[];
    T *tile = (T *)raw;
    for (int e = threadIdx.x; e < n_pix * ROWW; e += 512) tile[e] = T(0);
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = lane % ROWW, row = lane / ROWW;
    unsigned rng = seed + blockIdx.x * 7919u + (threadIdx.x >> 6) * 104729u;
    const float v = 1.0f + lane;
    for (int i = 0; i < iters; ++i) {
        rng = rng * 1664525u + 1013904223u;
        const unsigned r = (rng >> 8) * (2 * row + 1) + row * 977u;
        const int pix = r % n_pix;
        if (row < ACTIVE_ROWS) atomicAdd(&tile[pix * ROWW + sub], mk<T>(v));
    }
    __syncthreads();
    double s = 0;
    for (int e = threadIdx.x; e < n_pix * ROWW; e += 512) s += (double)tile[e];
    if (s == 12345.0) out[0] = (float)s;
}
template <typename T, int ROWW, int ACTIVE_ROWS = 64>
void run(const char *name, float *out) {
    const int n_pix = 400, iters = 8192, blocks = 512;
    const size_t shm = (size_t)n_pix * ROWW * sizeof(T);
    hipFuncSetAttribute((const void *)k<T, ROWW, ACTIVE_ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, ROWW, ACTIVE_ROWS>), dim3(blocks), dim3(512), shm, 0, out, 16, 1u, n_pix);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<T, ROWW, ACTIVE_ROWS>), dim3(blocks), dim3(512), shm, 0, out, iters, 1u, n_pix);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // 512 blocks x 8 waves on 256 CUs: 16 waves per CU (2 blocks resident if LDS allows), iters instructions each
    const double inst_per_cu = 16.0 * iters;
    printf("%-28s rows of %2d lanes: %.3f ms -> %.1f clk per 64-lane LDS atomic per CU (2.4 GHz)\n", name, ROWW, ms, ms * 1e-3 * 2.4e9 / inst_per_cu);
}
int main() {
    float *out; hipMalloc(&out, 4);
    run<float, 16>("ds_add_f32", out);
    run<double, 16>("ds_add_f64", out);
    run<unsigned, 16>("ds_add_u32", out);
    run<unsigned long long, 16>("ds_add_u64", out);
    run<float, 32>("ds_add_f32", out);
    run<double, 32>("ds_add_f64", out);
    run<unsigned, 32>("ds_add_u32", out);
    run<unsigned long long, 32>("ds_add_u64", out);
    run<double, 16, 1>("ds_add_f64 1 of 4 rows active", out);
    run<double, 16, 2>("ds_add_f64 2 of 4 rows active", out);
    run<double, 16, 3>("ds_add_f64 3 of 4 rows active", out);
    return 0;
}
