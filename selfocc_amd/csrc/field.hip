// field.hip — tri-plane -> dense SDF / colour / semantic volume in ONE kernel (gfx950, MFMA f32).
//
// Replaces the head's `pre_compute_density_color(representation)` (sdfstudio-fork SDFCustomField, called
// from model/head/neus_head/neus_head.py:295-306; in-repo analogue BEVNeRF,
// model/head/nerfacc_head/bev_nerf.py:74-95):
//     feat[h,w,d,:] = hw[h,w,:] + zh[d,h,:] + wz[w,d,:]                      (H*W*D x C, 634 MB at occ sizes)
//     out = Linear_out(Softplus(Linear_hidden(Softplus(feat))))              ([Softplus, Linear] x density_layers)
//     sdf = out[..., 0];  colour / semantic channels = out[..., 1:]
// as separate torch kernels: broadcast add, softplus, GEMM, softplus, GEMM, slice copies (2.7 ms per
// nuscenes_depth frame, the 634 MB intermediate written and re-read three times).
//
// Here a wavefront owns 32 consecutive voxels (rows).  Lane (i = lane & 31, half = lane >> 5) builds
// HALF of row i's Softplus(feat) directly in registers from the three planes (12 float4 loads per
// plane, planes stay L2-resident) — that register file IS the A operand of v_mfma_f32_32x32x2_f32:
// MFMA step ks consumes k = half * C/2 + ks (the k labelling is free as long as A and B agree), so a
// lane's 48 A values are one contiguous run of its row.  B = the hidden weight, staged once per block
// in LDS in [ks][half][n] order (conflict-free ds_read_b32).  The 32 x C result (C/32 accumulators of
// 16 VGPRs) gets bias + Softplus in registers, goes through a wave-private LDS tile to become the A
// operand of the output layer (C layout -> A layout), and the second MFMA chain's result is stored
// straight into the layouts the render kernels read (sdf (H,W,D) and feat (H,W,D,F)).
// f32 MFMA on gfx950 is an exact k-ordered fmaf chain (no TF32), so this is float32 arithmetic.
#include "so_device.h"
#include <hip/hip_bf16.h>
#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// torch.nn.Softplus(beta=1, threshold=20): x > 20 ? x : log1p(exp(x))
SO_DEVFN float so_softplus(float x) {
    const float e = __expf(x);
    // log1p(e): alternating series below 0.05 (remainder e^5/5 < 7e-8 relative 1.3e-6), log(1 + e) above
    const float series = e * (1.0f - e * (0.5f - e * (0.33333334f - 0.25f * e)));
    const float lg = __logf(1.0f + e);
    const float r = e < 0.05f ? series : lg;
    return x > 20.0f ? x : r;
}

// waves per block: the weights + one 32 x (C + 4) transpose tile per wave must fit 160 KB of LDS
constexpr int field_waves(int C) { return C > 96 ? 4 : 8; }

struct FieldArgs {
    const float *hw, *zh, *wz;
    int H, W, D;
    const float *w_hidden, *b_hidden;   // (C, C), (C)   [n_hidden == 1]
    const float *w_out, *b_out;         // (out_dim, C), (out_dim)
    int n_hidden, out_dim;
    float *sdf;
    void *feat;
    int feat_stride;                    // F (floats / bf16 per voxel), 0 = no feature volume
    long long M;                        // H * W * D
    int n_tiles;
};

template <int C, bool BF16>
__global__ __launch_bounds__(field_waves(C) * 64) void field_volume_kernel(FieldArgs a) {
    constexpr int kFieldWaves = field_waves(C);
    constexpr int KS = C / 2;          // MFMA k-steps (2 k per step)
    constexpr int NT = C / 32;         // 32-column tiles of the hidden layer
    constexpr int YS = C + 4;          // row stride of the transpose tile (16-byte aligned rows)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *w1s = smem;                               // [KS][2][C]    hidden weight, B-operand order
    float *w2s = w1s + (size_t)C * C;                // [KS][2][32]   output weight, zero-padded to 32 columns
    float *ytiles = w2s + (size_t)C * 32;            // [waves][32][YS]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;

    // ---- stage the weights (once per block; blocks are persistent) ------------------------------
    if (a.n_hidden == 1) {
        for (int e = threadIdx.x; e < C * C; e += kFieldWaves * 64) {
            const int n = e % C, kk = e / C;           // kk = ks * 2 + half
            const int k = (kk & 1) * KS + (kk >> 1);
            w1s[e] = a.w_hidden[(size_t)n * C + k];
        }
    }
    for (int e = threadIdx.x; e < C * 32; e += kFieldWaves * 64) {
        const int n = e & 31, kk = e >> 5;
        const int k = (kk & 1) * KS + (kk >> 1);
        w2s[e] = n < a.out_dim ? a.w_out[(size_t)n * C + k] : 0.0f;
    }
    __syncthreads();
    float *ytile = ytiles + (size_t)wave * 32 * YS;

    for (int tile = blockIdx.x * kFieldWaves + wave; tile < a.n_tiles; tile += gridDim.x * kFieldWaves) {
        // ---- A operand: Softplus(hw + zh + wz) of row m, columns half * KS .. + KS ---------------
        const long long m = (long long)tile * 32 + i;
        const long long mc = m < a.M ? m : a.M - 1;
        const int d = (int)(mc % a.D);
        const int hwi = (int)(mc / a.D);               // h * W + w
        const int w = hwi % a.W, h = hwi / a.W;
        const float4 *p0 = (const float4 *)(a.hw + (size_t)hwi * C + half * KS);
        const float4 *p1 = (const float4 *)(a.zh + ((size_t)d * a.H + h) * C + half * KS);
        const float4 *p2 = (const float4 *)(a.wz + ((size_t)w * a.D + d) * C + half * KS);
        float av[KS];
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const float4 x0 = p0[q], x1 = p1[q], x2 = p2[q];
            av[4 * q + 0] = so_softplus((x0.x + x1.x) + x2.x);
            av[4 * q + 1] = so_softplus((x0.y + x1.y) + x2.y);
            av[4 * q + 2] = so_softplus((x0.z + x1.z) + x2.z);
            av[4 * q + 3] = so_softplus((x0.w + x1.w) + x2.w);
        }
        if (a.n_hidden == 1) {   // uniform
            f32x16 acc[NT];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[ct][v] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float *brow = w1s + (ks * 2 + half) * C + i;
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], brow[ct * 32], acc[ct], 0, 0, 0);
            }
            // C layout (col = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)) -> bias, Softplus -> LDS [row][col]
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const float bias = a.b_hidden[ct * 32 + i];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int r = (v & 3) + 8 * (v >> 2) + 4 * half;
                    ytile[r * YS + ct * 32 + i] = so_softplus(acc[ct][v] + bias);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float4 *yr = (const float4 *)(ytile + i * YS + half * KS);
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const float4 t = yr[q];
                av[4 * q + 0] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- output layer --------------------------------------------------------------------------
        f32x16 o;
#pragma unroll
        for (int v = 0; v < 16; ++v) o[v] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], w2s[(ks * 2 + half) * 32 + i], o, 0, 0, 0);
        const int n = i;                                   // output channel of this lane
        const float bias = n < a.out_dim ? a.b_out[n] : 0.0f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = (v & 3) + 8 * (v >> 2) + 4 * half;
            const long long mm = (long long)tile * 32 + r;
            if (mm >= a.M) continue;
            const float val = o[v] + bias;
            if (n == 0) {
                a.sdf[mm] = val;
            } else if (n - 1 < a.feat_stride) {
                const float f = n < a.out_dim ? val : 0.0f;   // padding channels of the feature volume
                if (BF16) ((__hip_bfloat16 *)a.feat)[(size_t)mm * a.feat_stride + (n - 1)] = __float2bfloat16(f);
                else ((float *)a.feat)[(size_t)mm * a.feat_stride + (n - 1)] = f;
            }
        }
    }
}

}  // namespace

extern "C" int selfocc_field_volume_fwd(const float *hw, const float *zh, const float *wz, int32_t H, int32_t W,
                                        int32_t D, int32_t C, const float *w_hidden, const float *b_hidden,
                                        int32_t n_hidden, const float *w_out, const float *b_out, int32_t out_dim,
                                        float *sdf, void *feat, int32_t feat_dtype, int32_t feat_stride,
                                        void *stream) {
    SO_REQUIRE(H >= 1 && W >= 1 && D >= 1, "field_volume: bad volume size (%d, %d, %d)", H, W, D);
    SO_REQUIRE(C == 64 || C == 96 || C == 128, "field_volume: embed_dims must be 64, 96 or 128 (got %d)", C);
    SO_REQUIRE(n_hidden == 0 || n_hidden == 1, "field_volume: density_layers must be 1 or 2 (n_hidden = %d)", n_hidden);
    SO_REQUIRE(out_dim >= 1 && out_dim <= 32, "field_volume: 1 + color_dims must be <= 32 (got %d)", out_dim);
    SO_REQUIRE(hw && zh && wz && w_out && b_out && sdf, "field_volume: NULL pointer");
    SO_REQUIRE(n_hidden == 0 || (w_hidden && b_hidden), "field_volume: NULL hidden weight");
    SO_REQUIRE(feat_stride >= 0 && feat_stride <= 31 && (feat_stride == 0 || feat != nullptr),
               "field_volume: feat_stride must be 0..31 with a feature buffer");
    SO_REQUIRE(feat_stride == 0 || feat_stride >= out_dim - 1, "field_volume: feat_stride %d < %d colour channels",
               feat_stride, out_dim - 1);
    SO_REQUIRE(feat_dtype == SO_DTYPE_F32 || feat_dtype == SO_DTYPE_BF16, "field_volume: bad feat_dtype");
    const long long M = (long long)H * W * D;
    SO_REQUIRE(M < (1LL << 31) * 32, "field_volume: volume too large");
    FieldArgs a{hw, zh, wz, H, W, D, w_hidden, b_hidden, w_out, b_out, n_hidden, out_dim, sdf, feat, feat_stride, M,
                (int)((M + 31) / 32)};
    const int nw = field_waves(C);
    const size_t shm = ((size_t)C * C + (size_t)C * 32 + (size_t)nw * 32 * (C + 4)) * sizeof(float);
    const int blocks = std::min((a.n_tiles + nw - 1) / nw, 256);
    hipStream_t st = (hipStream_t)stream;
#define SO_LAUNCH(CC, BF)                                                                                    \
    {                                                                                                        \
        static bool attr_set = false;                                                                        \
        if (!attr_set) {                                                                                     \
            (void)hipFuncSetAttribute((const void *)field_volume_kernel<CC, BF>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);         \
            attr_set = true;                                                                                 \
        }                                                                                                    \
        hipLaunchKernelGGL((field_volume_kernel<CC, BF>), dim3(blocks), dim3(nw * 64), shm, st, a);             \
    }
    const bool bf = feat_dtype == SO_DTYPE_BF16;
    if (C == 64) { if (bf) SO_LAUNCH(64, true) else SO_LAUNCH(64, false) }
    else if (C == 96) { if (bf) SO_LAUNCH(96, true) else SO_LAUNCH(96, false) }
    else { if (bf) SO_LAUNCH(128, true) else SO_LAUNCH(128, false) }
#undef SO_LAUNCH
    return so_launch_status();
}
