// layernorm.hip — LayerNorm over the channel dimension of the (rows, C) plane tensors of the TPV / BEV
// encoder (the `norm` entries of TPVFormerLayer's operation_order: mmcv build_norm_layer(dict(type='LN')),
// model/encoder/tpvformer/tpvformer_encoder_layer.py; 12 calls per frame on 78 899 x 96 floats).
//
// Pure streaming work: 30 MB in, 30 MB out per call.  torch's kernel reaches ~0.75 TB/s on this
// shape (80 us per call in profiles/r2_a_eval_kernel_trace.txt).  Here a row lives in one 32-lane half
// wave, C / 4 lanes x float4 (C <= 128, a multiple of 4): one 16-byte load and one 16-byte store per
// lane, mean and variance by two 5-step shuffle reductions on the register copy (two-pass: no
// cancellation).  The backward recomputes x-hat from the saved (mean, rstd), reduces the two row sums
// the same way and keeps the weight / bias gradient partials in registers across a block's rows; the
// per-block partials are summed by a second tiny kernel in a fixed order (deterministic).
#include "so_device.h"

namespace {

SO_DEVFN float so_half_sum(float v) {   // sum over the 32 lanes of a half wave
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ y,
                                                            float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                            long long rows, int C, float eps) {
    const int C4 = C >> 2;
    const int l = threadIdx.x & 31;
    const bool lane_on = l < C4;
    const float4 g4 = lane_on ? ((const float4 *)gamma)[l] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b4 = lane_on ? ((const float4 *)beta)[l] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_c = 1.0f / (float)C;
    const long long half0 = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;
    const long long nhalf = ((long long)gridDim.x * 256) >> 5;
    for (long long row = half0; row < rows; row += nhalf) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_on) v = ((const float4 *)(x + row * C))[l];
        const float mean = so_half_sum((v.x + v.y) + (v.z + v.w)) * inv_c;
        float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
        if (!lane_on) d = make_float4(0.f, 0.f, 0.f, 0.f);
        const float var = so_half_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * inv_c;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (lane_on) {
            float4 o;
            o.x = fmaf(d.x * rstd, g4.x, b4.x); o.y = fmaf(d.y * rstd, g4.y, b4.y);
            o.z = fmaf(d.z * rstd, g4.z, b4.z); o.w = fmaf(d.w * rstd, g4.w, b4.w);
            ((float4 *)(y + row * C))[l] = o;
        }
        if (l == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                            const float *__restrict__ dy, float *__restrict__ dx,
                                                            float *__restrict__ partial /* [gridDim.x][2][C] */,
                                                            long long rows, int C) {
    __shared__ float s_red[8][2][128];
    const int C4 = C >> 2;
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const bool lane_on = l < C4;
    const float4 g4 = lane_on ? ((const float4 *)gamma)[l] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_c = 1.0f / (float)C;
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long half0 = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;
    const long long nhalf = ((long long)gridDim.x * 256) >> 5;
    for (long long row = half0; row < rows; row += nhalf) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane_on) { v = ((const float4 *)(x + row * C))[l]; g = ((const float4 *)(dy + row * C))[l]; }
        const float mean = mean_in[row], rstd = rstd_in[row];
        float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
        if (!lane_on) xh = make_float4(0.f, 0.f, 0.f, 0.f);
        dg.x = fmaf(g.x, xh.x, dg.x); dg.y = fmaf(g.y, xh.y, dg.y); dg.z = fmaf(g.z, xh.z, dg.z); dg.w = fmaf(g.w, xh.w, dg.w);
        db.x += g.x; db.y += g.y; db.z += g.z; db.w += g.w;
        const float4 gg = make_float4(g.x * g4.x, g.y * g4.y, g.z * g4.z, g.w * g4.w);
        const float a = so_half_sum((gg.x + gg.y) + (gg.z + gg.w)) * inv_c;
        const float b = so_half_sum((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * inv_c;
        if (lane_on) {
            float4 o;
            o.x = rstd * ((gg.x - a) - xh.x * b); o.y = rstd * ((gg.y - a) - xh.y * b);
            o.z = rstd * ((gg.z - a) - xh.z * b); o.w = rstd * ((gg.w - a) - xh.w * b);
            ((float4 *)(dx + row * C))[l] = o;
        }
    }
    if (lane_on) {
        ((float4 *)s_red[hw][0])[l] = dg;
        ((float4 *)s_red[hw][1])[l] = db;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        const int which = e / C, c = e - which * C;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += s_red[k][which][c];
        partial[((size_t)blockIdx.x * 2 + which) * C + c] = s;
    }
}

__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float *__restrict__ partial, int nblocks, int C,
                                                                   float *__restrict__ dgamma, float *__restrict__ dbeta) {
    // one wave per output element (2 * C of them): fixed-order tree over the blocks' partials
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= 2 * C) return;
    const int which = e / C, c = e - which * C;
    float s = 0.0f;
    for (int b = lane; b < nblocks; b += 64) s += partial[((size_t)b * 2 + which) * C + c];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) (which == 0 ? dgamma : dbeta)[c] = s;
}

int ln_blocks(long long rows) {
    const long long need = (rows + 7) / 8;
    return (int)(need < 2048 ? (need < 1 ? 1 : need) : 2048);
}

}  // namespace

extern "C" int selfocc_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                                     float *rstd, int64_t rows, int32_t C, float eps, void *stream) {
    SO_REQUIRE(rows >= 0, "layernorm: rows must be >= 0");
    SO_REQUIRE(C >= 4 && C <= 128 && C % 4 == 0, "layernorm: C must be a multiple of 4 in [4, 128] (got %d)", C);
    if (rows == 0) return 0;
    SO_REQUIRE(x && gamma && beta && y, "layernorm: NULL pointer");
    SO_REQUIRE((mean == nullptr) == (rstd == nullptr), "layernorm: mean and rstd go together");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ln_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                       mean, rstd, (long long)rows, C, eps);
    return so_launch_status();
}

extern "C" size_t selfocc_layernorm_bwd_workspace(int64_t rows, int32_t C) {
    return (size_t)ln_blocks(rows) * 2 * (size_t)C * sizeof(float);
}

extern "C" int selfocc_layernorm_bwd(const float *x, const float *gamma, const float *mean, const float *rstd,
                                     const float *dy, float *dx, float *dgamma, float *dbeta, int64_t rows, int32_t C,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    SO_REQUIRE(rows >= 0, "layernorm: rows must be >= 0");
    SO_REQUIRE(C >= 4 && C <= 128 && C % 4 == 0, "layernorm: C must be a multiple of 4 in [4, 128] (got %d)", C);
    SO_REQUIRE(dgamma && dbeta, "layernorm_bwd: NULL gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {   // empty input: the parameter gradients are zero, nothing else to write
        (void)hipMemsetAsync(dgamma, 0, C * sizeof(float), st);
        return (int)hipMemsetAsync(dbeta, 0, C * sizeof(float), st);
    }
    SO_REQUIRE(x && gamma && mean && rstd && dy && dx, "layernorm_bwd: NULL pointer");
    const int nb = ln_blocks(rows);
    SO_REQUIRE(workspace && workspace_bytes >= (size_t)nb * 2 * C * sizeof(float), "layernorm_bwd: workspace too small");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), 0, st, x, gamma, mean, rstd, dy, dx, (float *)workspace,
                       (long long)rows, C);
    hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, st, (const float *)workspace, nb, C,
                       dgamma, dbeta);
    return so_launch_status();
}
