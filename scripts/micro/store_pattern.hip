// Dev micro-benchmark (round 6): how fast can 256 CUs WRITE a tall (T x N) float matrix in the tile order of linear_fwd_b3_kernel?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern scripts/micro/store_pattern.hip && /tmp/store_pattern
// Patterns (all write every element exactly once, float4 per lane, no loads, no arithmetic):
//   0: the kernel's: persistent blocks, block = (column block of 96, row group), wave tile 32 rows x 96 columns, 12 float4 stores
//   1: as 0, but a wave tile's 384-byte row segments are written by ONE instruction per 2-3 rows (lane = 16-byte chunk of a row)
//   2: linear fill (each wave writes 1 KiB contiguous per instruction, grid-stride): the ceiling
//   3: as 0 with the column block as the SLOW index (every block walks all rows of one column block band by band)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ unsigned xcd_block() {
    const unsigned nb = gridDim.x, b = blockIdx.x, x = b & 7u, k = b >> 3, q = nb >> 3, r = nb & 7u;
    return x * q + (x < r ? x : r) + k;
}

template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(float *y, long long T, int N, int ncb, int groups) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
    if (PAT == 2) {
        const long long n4 = T * N / 4;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) ((float4 *)y)[i] = v;
        return;
    }
    const unsigned logical = xcd_block();
    const int cb = logical % ncb;
    const long long rc = logical / ncb;
    const long long nwt = T / 32, step = 4LL * groups;
    const int n0 = cb * 96;
    for (long long wt = rc * 4 + wave; wt < nwt; wt += step) {
        const long long row0 = wt * 32;
        if (PAT == 0 || PAT == 3) {
            const int r = lane & 15, kq = lane >> 4;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 6; ++t) *(float4 *)(y + (row0 + 16 * h + r) * N + n0 + 16 * t + 4 * kq) = v;
        } else {
            // 24 lanes cover one row's 384 bytes; 64 lanes = 2.67 rows per instruction: 12 instructions for 32 rows
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int idx = i * 64 + lane, r = idx / 24, c = idx - r * 24;
                *(float4 *)(y + (row0 + r) * N + n0 + 4 * c) = v;
            }
        }
    }
}

int main() {
    const long long T = 78880;     // 2465 full tiles
    float *y;
    CK(hipMalloc(&y, (size_t)T * 672 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int N : {576, 288, 96}) {
        const int ncb = N / 96;
        for (int pat = 0; pat < 3; ++pat) {
            for (int percu : {2, 4, 8}) {
                int groups = std::max(1, 256 * percu / ncb);
                const int nblk = pat == 2 ? 256 * percu : groups * ncb;
                float best = 1e9f;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipEventRecord(e0));
                    for (int k = 0; k < 10; ++k) {
                        if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(nblk), dim3(256), 0, 0, y, T, N, ncb, groups);
                        else if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(nblk), dim3(256), 0, 0, y, T, N, ncb, groups);
                        else hipLaunchKernelGGL(store_kernel<2>, dim3(nblk), dim3(256), 0, 0, y, T, N, ncb, groups);
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms / 10);
                }
                printf("N=%4d pattern %d blocks/CU %d: %7.1f us  %6.0f GB/s\n", N, pat, percu, best * 1e3, (double)T * N * 4 / best / 1e6);
            }
        }
    }
    return 0;
}
