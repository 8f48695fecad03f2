// dropout.hip — `identity + dropout(x)` of the encoder's attention / FFN outputs as ONE streaming pass per direction (gfx950).
//
// Reference: mmcv's attention / FFN modules end in `self.dropout(output) + identity`
// (model/encoder/bevformer/attention/image_cross_attention.py:137-139, tpvformer/attention/cross_view_hybrid_attention.py:119-124,
// mmcv FFN `identity + self.dropout_layer(out)`), which torch runs as native_dropout (reads x, writes the scaled copy and a
// bool mask) + add (reads two, writes one) forward and a masked scale (reads g and the mask) backward: 24 + 12 launches and
// ~1.3 GB of traffic per nuscenes_occ iteration.  Here the keep / drop decision is a counter-based hash of (seed, element
// index) — a dropped element is decided by the index alone, so no mask tensor exists: forward reads x and identity and writes
// y, backward reads g and writes g_x.  Same distribution as torch's dropout (keep with probability 1 - p, scale 1 / (1 - p));
// not torch's Philox stream — dropout masks are not reproducible across implementations in the reference either.
#include "so_device.h"
#include <algorithm>

namespace {

SO_DEVFN bool so_keep(unsigned long long seed, unsigned long long idx, float p) {
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;          // splitmix64 of (seed, index)
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f) >= p;      // 24 uniform bits in [0, 1)
}

__global__ __launch_bounds__(256) void dropout_add_fwd_kernel(const float *__restrict__ x, const float *__restrict__ identity,
                                                              float *__restrict__ y, long long n, float p, float scale,
                                                              unsigned long long seed) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 a = *(const float4 *)(x + i), b = *(const float4 *)(identity + i);
            float4 o;
            o.x = b.x + (so_keep(seed, i, p) ? a.x * scale : 0.0f);
            o.y = b.y + (so_keep(seed, i + 1, p) ? a.y * scale : 0.0f);
            o.z = b.z + (so_keep(seed, i + 2, p) ? a.z * scale : 0.0f);
            o.w = b.w + (so_keep(seed, i + 3, p) ? a.w * scale : 0.0f);
            *(float4 *)(y + i) = o;
        } else {
            for (long long k = i; k < n; ++k) y[k] = identity[k] + (so_keep(seed, k, p) ? x[k] * scale : 0.0f);
        }
    }
}

__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float *__restrict__ g, float *__restrict__ gx, long long n, float p,
                                                          float scale, unsigned long long seed) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 a = *(const float4 *)(g + i);
            float4 o;
            o.x = so_keep(seed, i, p) ? a.x * scale : 0.0f;
            o.y = so_keep(seed, i + 1, p) ? a.y * scale : 0.0f;
            o.z = so_keep(seed, i + 2, p) ? a.z * scale : 0.0f;
            o.w = so_keep(seed, i + 3, p) ? a.w * scale : 0.0f;
            *(float4 *)(gx + i) = o;
        } else {
            for (long long k = i; k < n; ++k) gx[k] = so_keep(seed, k, p) ? g[k] * scale : 0.0f;
        }
    }
}

int so_drop_blocks(long long n) { return (int)std::max<long long>(1, std::min<long long>(8192, (n + 1023) / 1024)); }

}  // namespace

extern "C" int selfocc_dropout_add_fwd(const float *x, const float *identity, float *y, int64_t n, float p, uint64_t seed,
                                       void *stream) {
    SO_REQUIRE(n >= 0 && p >= 0.0f && p < 1.0f, "dropout_add: need n >= 0 and 0 <= p < 1 (got n = %lld, p = %g)", (long long)n, p);
    if (n == 0) return 0;
    SO_REQUIRE(x && identity && y, "dropout_add: NULL pointer");
    SO_REQUIRE((((uintptr_t)x | (uintptr_t)identity | (uintptr_t)y) & 15) == 0, "dropout_add: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(dropout_add_fwd_kernel, dim3(so_drop_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, identity, y,
                       (long long)n, p, 1.0f / (1.0f - p), (unsigned long long)seed);
    return so_launch_status();
}

extern "C" int selfocc_dropout_bwd(const float *g, float *g_x, int64_t n, float p, uint64_t seed, void *stream) {
    SO_REQUIRE(n >= 0 && p >= 0.0f && p < 1.0f, "dropout_bwd: need n >= 0 and 0 <= p < 1 (got n = %lld, p = %g)", (long long)n, p);
    if (n == 0) return 0;
    SO_REQUIRE(g && g_x, "dropout_bwd: NULL pointer");
    SO_REQUIRE((((uintptr_t)g | (uintptr_t)g_x) & 15) == 0, "dropout_bwd: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(so_drop_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, g_x, (long long)n, p,
                       1.0f / (1.0f - p), (unsigned long long)seed);
    return so_launch_status();
}
