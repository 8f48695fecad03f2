// eikonal.hip — the eikonal regulariser of the SDF head as one launch per direction (gfx950).
//
// Reference: loss/eikonal_loss.py:19-22 (EikonalLoss.eikonal_loss): mean over all ray samples of (||grad sdf||_2 - 1)^2 on the (R * S, 3)
// per-sample metre gradients of the render (7.4 M rows per nuscenes_occ iteration).  torch runs it as a norm REDUCTION over a
// dimension of three (67 us), sub, pow, mean forward and pow / norm backward (a copy, div, mul, masked_fill): ~0.26 ms per
// iteration for 88 MB of input.  Here: forward = one streaming pass writing one partial sum per block (the caller adds the
// partials: deterministic), backward = one streaming pass
//     d/dg = scale * 2 (||g|| - 1) g / ||g||      (0 where ||g|| = 0, as torch's norm backward),   scale read from device memory.
#include "so_device.h"
#include <algorithm>

namespace {

constexpr int kEikRowsPerThread = 4;

__global__ __launch_bounds__(256) void eikonal_fwd_kernel(const float *__restrict__ g, float *__restrict__ partial, long long n) {
    __shared__ float red[4];
    float s = 0.0f;
    const long long stride = (long long)gridDim.x * 256;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
        const float x = g[3 * r], y = g[3 * r + 1], z = g[3 * r + 2];
        const float d = sqrtf((x * x + y * y) + z * z) - 1.0f;
        s += d * d;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void eikonal_bwd_kernel(const float *__restrict__ g, const float *__restrict__ scale,
                                                          float *__restrict__ gg, long long n) {
    const float sc = 2.0f * scale[0];
    const long long stride = (long long)gridDim.x * 256;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
        const float x = g[3 * r], y = g[3 * r + 1], z = g[3 * r + 2];
        const float nn = sqrtf((x * x + y * y) + z * z);
        const float f = nn > 0.0f ? sc * (nn - 1.0f) / nn : 0.0f;
        gg[3 * r] = f * x; gg[3 * r + 1] = f * y; gg[3 * r + 2] = f * z;
    }
}

int so_eik_blocks(long long n) {
    return (int)std::max<long long>(1, std::min<long long>(2048, (n + 256LL * kEikRowsPerThread - 1) / (256LL * kEikRowsPerThread)));
}

}  // namespace

extern "C" int selfocc_eikonal_partials(int64_t n) { return n <= 0 ? 0 : so_eik_blocks(n); }

extern "C" int selfocc_eikonal_fwd(const float *grad, float *partial, int64_t n, void *stream) {
    SO_REQUIRE(n >= 0, "eikonal_fwd: negative size");
    if (n == 0) return 0;
    SO_REQUIRE(grad && partial, "eikonal_fwd: NULL pointer");
    hipLaunchKernelGGL(eikonal_fwd_kernel, dim3((unsigned)so_eik_blocks(n)), dim3(256), 0, (hipStream_t)stream, grad, partial,
                       (long long)n);
    return so_launch_status();
}

extern "C" int selfocc_eikonal_bwd(const float *grad, const float *scale, float *g_grad, int64_t n, void *stream) {
    SO_REQUIRE(n >= 0, "eikonal_bwd: negative size");
    if (n == 0) return 0;
    SO_REQUIRE(grad && scale && g_grad, "eikonal_bwd: NULL pointer");
    hipLaunchKernelGGL(eikonal_bwd_kernel, dim3((unsigned)so_eik_blocks(n)), dim3(256), 0, (hipStream_t)stream, grad, scale, g_grad,
                       (long long)n);
    return so_launch_status();
}
