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
#include <atomic>
#include <cstdlib>

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

// Forward kernel, "transposed chain": the hidden layer is computed as Y^T = W1 X^T (W1 = the MFMA's A operand, read
// from LDS with ds_read_b128 = four k-steps per read; X^T = the B operand, the lane's own 48 Softplus(x) registers), so
// a lane ends up holding 48 hidden units OF ITS OWN VOXEL (units 32 ct + 8 a + 4 half + b).  Because the k labelling of
// an MFMA is free as long as A and B agree, exactly those registers are the next layer's A operand (Z[voxel][unit],
// with W2 read from LDS at the same permuted units): the 32 x 96 hidden tile never goes through LDS, there is no
// C-layout -> A-layout transpose, no wave barrier, and the block's LDS drops from 151 KB to 52 KB (three 4-wave
// blocks per CU instead of one 8-wave block).
constexpr int kFieldFwdWaves = 4;

template <int C, bool BF16>
__global__ __launch_bounds__(kFieldFwdWaves * 64) void field_volume_kernel(FieldArgs a) {
    constexpr int KS = C / 2;          // k-steps of the hidden layer (2 k per MFMA)
    constexpr int NT = C / 32;         // 32-unit tiles of the hidden layer
    constexpr int KP = C + 4;          // LDS row stride: conflict-free ds_read_b128 (cf. linear_fwd.hip)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *w1s = smem;                               // [C units][KP]   hidden weight, row = unit, column = input k
    float *w2s = w1s + (size_t)C * KP;               // [32 outputs][KP] output weight (zero rows beyond out_dim)
    float *b1s = w2s + (size_t)32 * KP;              // [C]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;

    // ---- stage the weights (once per block; blocks are persistent) ------------------------------
    if (a.n_hidden == 1) {
        for (int e = threadIdx.x; e < C * (C / 4); e += kFieldFwdWaves * 64) {
            const int u = e / (C / 4), k4 = e - u * (C / 4);
            *(float4 *)(w1s + u * KP + 4 * k4) = ((const float4 *)(a.w_hidden + (size_t)u * C))[k4];
        }
        for (int e = threadIdx.x; e < C; e += kFieldFwdWaves * 64) b1s[e] = a.b_hidden[e];
    }
    for (int e = threadIdx.x; e < 32 * (C / 4); e += kFieldFwdWaves * 64) {
        const int n = e / (C / 4), k4 = e - n * (C / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < a.out_dim) v = ((const float4 *)(a.w_out + (size_t)n * C))[k4];
        *(float4 *)(w2s + n * KP + 4 * k4) = v;
    }
    __syncthreads();

    for (int tile = blockIdx.x * kFieldFwdWaves + wave; tile < a.n_tiles; tile += gridDim.x * kFieldFwdWaves) {
        // ---- Softplus(hw + zh + wz) of voxel i, inputs half * KS .. + KS -----------------------------
        // (32-bit index arithmetic: the host guarantees M < 2^31; 64-bit division costs ~100 VALU instructions each)
        const unsigned m = (unsigned)tile * 32u + (unsigned)i;
        const unsigned mc = m < (unsigned)a.M ? m : (unsigned)a.M - 1u;
        const unsigned hwi = mc / (unsigned)a.D;       // h * W + w
        const unsigned d = mc - hwi * (unsigned)a.D;
        const unsigned h = hwi / (unsigned)a.W, w = hwi - h * (unsigned)a.W;
        const float4 *p0 = (const float4 *)(a.hw + (size_t)hwi * C + half * KS);
        const float4 *p1 = (const float4 *)(a.zh + ((size_t)d * a.H + h) * C + half * KS);
        const float4 *p2 = (const float4 *)(a.wz + ((size_t)w * a.D + d) * C + half * KS);
        float xv[KS];
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const float4 x0 = p0[q], x1 = p1[q], x2 = p2[q];
            xv[4 * q + 0] = so_softplus((x0.x + x1.x) + x2.x);
            xv[4 * q + 1] = so_softplus((x0.y + x1.y) + x2.y);
            xv[4 * q + 2] = so_softplus((x0.z + x1.z) + x2.z);
            xv[4 * q + 3] = so_softplus((x0.w + x1.w) + x2.w);
        }
        f32x16 o;
#pragma unroll
        for (int v = 0; v < 16; ++v) o[v] = 0.0f;
        const float *w2row = w2s + i * KP;               // this lane's output channel as the B operand
        if (a.n_hidden == 1) {   // uniform
            // hidden layer, transposed: acc[ct][v] = y[voxel i][unit 32 ct + 8 (v >> 2) + 4 half + (v & 3)]
            f32x16 acc[NT];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[ct][v] = 0.0f;
            constexpr int QN = KS / 4, NS = NT * QN;
            const float *abase = w1s + i * KP + half * KS;
            float4 wa = *(const float4 *)abase;
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) {
                const int ct = sidx / QN, q = sidx - ct * QN;
                float4 wn = wa;
                if (sidx + 1 < NS) {
                    const int ct2 = (sidx + 1) / QN, q2 = (sidx + 1) - ct2 * QN;
                    wn = *(const float4 *)(abase + 32 * ct2 * KP + 4 * q2);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the next group's read ahead of this group's MFMAs
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.x, xv[4 * q], acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.y, xv[4 * q + 1], acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.z, xv[4 * q + 2], acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.w, xv[4 * q + 3], acc[ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                wa = wn;
            }
            // bias + Softplus in place, then the output layer straight from these registers:
            // o[voxel][channel] += z[voxel][unit] W2[channel][unit], unit = 32 ct + 8 a + 4 half + b
            if (a.out_dim == 1) {
                // SDF only (the depth configs): a 32-column MFMA chain for ONE useful column is a quarter of the tile's
                // matrix-pipe time; the lane already holds 48 of its voxel's 96 hidden units -> 48 fmas + one exchange
                float dot = 0.0f;
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                    for (int aa = 0; aa < 4; ++aa) {
                        const int u0 = 32 * ct + 8 * aa + 4 * half;
                        const float4 bb = *(const float4 *)(b1s + u0);
                        const float4 w2 = *(const float4 *)(w2s + u0);       // row 0 of W2: broadcast read
                        dot = fmaf(so_softplus(acc[ct][4 * aa + 0] + bb.x), w2.x, dot);
                        dot = fmaf(so_softplus(acc[ct][4 * aa + 1] + bb.y), w2.y, dot);
                        dot = fmaf(so_softplus(acc[ct][4 * aa + 2] + bb.z), w2.z, dot);
                        dot = fmaf(so_softplus(acc[ct][4 * aa + 3] + bb.w), w2.w, dot);
                    }
                }
                dot += __shfl_xor(dot, 32, 64);
                if (half == 0 && m < (unsigned)a.M) a.sdf[m] = dot + a.b_out[0];
                continue;
            }
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) {
                    const int u0 = 32 * ct + 8 * aa + 4 * half;
                    const float4 bb = *(const float4 *)(b1s + u0);        // broadcast read (same address per half wave)
                    const float4 w2 = *(const float4 *)(w2row + u0);
                    const float z0 = so_softplus(acc[ct][4 * aa + 0] + bb.x), z1 = so_softplus(acc[ct][4 * aa + 1] + bb.y);
                    const float z2 = so_softplus(acc[ct][4 * aa + 2] + bb.z), z3 = so_softplus(acc[ct][4 * aa + 3] + bb.w);
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(z0, w2.x, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(z1, w2.y, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(z2, w2.z, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(z3, w2.w, o, 0, 0, 0);
                }
            }
        } else {
            // single linear layer: o[voxel][channel] = sum_k x[voxel][k] W2[channel][k], k = half * KS + ks
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const float4 w2 = *(const float4 *)(w2row + half * KS + 4 * q);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[4 * q], w2.x, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[4 * q + 1], w2.y, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[4 * q + 2], w2.z, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[4 * q + 3], w2.w, o, 0, 0, 0);
            }
        }
        // ---- store: C layout of o: column (channel) = lane & 31, row (voxel) = (v & 3) + 8 (v >> 2) + 4 half ----
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


// ---------------------------------------------------------------------------------------
// The same forward with both GEMMs on the bf16 matrix pipe through the exact three-way split of linear_fwd.hip (round 3;
// C = 96, one hidden layer).  f32 MFMA on gfx950 executes at the vector rate and competes with the 96 + 96 Softplus
// evaluations per voxel for the same lanes; v_mfma_f32_32x32x16_bf16 does 16 x the rate on the matrix pipe.  Every operand
// is the exact sum of three bfloat16 and the six products with i + j <= 4 are accumulated in float32: float32-level
// accuracy (tests/test_field_gpu.py unchanged), 2.67 x fewer matrix-pipe cycles.  The transposed chain survives as is:
//   hidden layer  Y^T = W1 X^T: A = W1 (three bf16 planes in LDS, lane (unit i, kb) reads W1[32 ct + i][48 kb + 8 s ..+8]),
//                 B = the lane's own Softplus(x) run x[voxel i][48 half + 8 s ..+8], s < 6 (k labelling free, as before);
//   result        acc[ct][v] = y[voxel i][unit 32 ct + 8 (v >> 2) + 4 half + (v & 3)] — so the 8 hidden units a lane owns
//                 for output step (ct, p) are acc[ct][8 p .. 8 p + 7]: consecutive registers = the A operand of the output
//                 layer, W2 (three planes) read from LDS at units 32 ct + 16 p + 4 half + {0..3, 8..11}.
constexpr int kFieldB3Waves = 12, kFieldB3KPB = 104;
typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4f __attribute__((ext_vector_type(4)));

SO_DEVFN void so_fsplit3(const float (&x)[8], bf16x8f &a1, bf16x8f &a2, bf16x8f &a3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 b1 = (__bf16)x[j];
        const float r1 = x[j] - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const float r2 = r1 - (float)b2;
        a1[j] = b1; a2[j] = b2; a3[j] = (__bf16)r2;
    }
}

template <bool BF16>
__global__ __launch_bounds__(kFieldB3Waves * 64) void field_volume_b3_kernel(FieldArgs a) {
    constexpr int C = 96, KS = 48, NT = 3, KPB = kFieldB3KPB, THREADS = kFieldB3Waves * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16 *w1b = (__bf16 *)smem;                          // [3 planes][96 units][KPB]
    __bf16 *w2b = w1b + (size_t)3 * C * KPB;               // [3 planes][32 outputs][KPB] (zero rows beyond out_dim)
    float *b1s = (float *)(w2b + (size_t)3 * 32 * KPB);    // [96]
    float *w2r = b1s + C;                                  // [96] row 0 of W2 in float32 (the SDF-only dot)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;

    for (int e = threadIdx.x; e < (C + 32) * (C / 8); e += THREADS) {
        const int r = e / (C / 8), k8 = e - r * (C / 8);
        const float *src = r < C ? a.w_hidden + (size_t)r * C : (r - C < a.out_dim ? a.w_out + (size_t)(r - C) * C : nullptr);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (src) {
            const float4 lo = ((const float4 *)src)[2 * k8], hi = ((const float4 *)src)[2 * k8 + 1];
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        }
        bf16x8f p1, p2, p3;
        so_fsplit3(v, p1, p2, p3);
        __bf16 *dst = r < C ? w1b + (size_t)r * KPB : w2b + (size_t)(r - C) * KPB;
        const size_t ps = r < C ? (size_t)C * KPB : (size_t)32 * KPB;
        *(bf16x8f *)(dst + 8 * k8) = p1;
        *(bf16x8f *)(dst + ps + 8 * k8) = p2;
        *(bf16x8f *)(dst + 2 * ps + 8 * k8) = p3;
    }
    for (int e = threadIdx.x; e < C; e += THREADS) { b1s[e] = a.b_hidden[e]; w2r[e] = a.w_out[e]; }
    __syncthreads();

    for (int tile = blockIdx.x * kFieldB3Waves + wave; tile < a.n_tiles; tile += gridDim.x * kFieldB3Waves) {
        const unsigned m = (unsigned)tile * 32u + (unsigned)i;
        const unsigned mc = m < (unsigned)a.M ? m : (unsigned)a.M - 1u;
        const unsigned hwi = mc / (unsigned)a.D;
        const unsigned d = mc - hwi * (unsigned)a.D;
        const unsigned h = hwi / (unsigned)a.W, w = hwi - h * (unsigned)a.W;
        const float4 *p0 = (const float4 *)(a.hw + (size_t)hwi * C + half * KS);
        const float4 *p1 = (const float4 *)(a.zh + ((size_t)d * a.H + h) * C + half * KS);
        const float4 *p2 = (const float4 *)(a.wz + ((size_t)w * a.D + d) * C + half * KS);
        // hidden layer, transposed: acc[ct][v] = y[voxel i][unit 32 ct + 8 (v >> 2) + 4 half + (v & 3)].  Per 8-k step: the
        // lane's Softplus(hw + zh + wz) run (voxel i, inputs 48 half + 8 s ..+8) is split once and used by the three unit tiles
        f32x16 acc[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ct][v] = 0.0f;
        const __bf16 *abase = w1b + (size_t)i * KPB + half * KS;
        float4 xn[6];       // the next step's three plane rows (2 float4 each): one step of prefetch, no more (registers)
#pragma unroll
        for (int q = 0; q < 2; ++q) { xn[q] = p0[q]; xn[2 + q] = p1[q]; xn[4 + q] = p2[q]; }
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float xv[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 x0 = xn[q], x1 = xn[2 + q], x2 = xn[4 + q];
                xv[4 * q + 0] = so_softplus((x0.x + x1.x) + x2.x);
                xv[4 * q + 1] = so_softplus((x0.y + x1.y) + x2.y);
                xv[4 * q + 2] = so_softplus((x0.z + x1.z) + x2.z);
                xv[4 * q + 3] = so_softplus((x0.w + x1.w) + x2.w);
            }
            if (s < 5) {
#pragma unroll
                for (int q = 0; q < 2; ++q) { xn[q] = p0[2 * s + 2 + q]; xn[2 + q] = p1[2 * s + 2 + q]; xn[4 + q] = p2[2 * s + 2 + q]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8f xb1, xb2, xb3;
            so_fsplit3(xv, xb1, xb2, xb3);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const __bf16 *ap = abase + (size_t)(32 * ct) * KPB + 8 * s;
                const bf16x8f a1 = *(const bf16x8f *)ap, a2 = *(const bf16x8f *)(ap + (size_t)C * KPB),
                              a3 = *(const bf16x8f *)(ap + (size_t)2 * C * KPB);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, xb1, acc[ct], 0, 0, 0);      // small terms first
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xb3, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb2, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb1, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xb2, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xb1, acc[ct], 0, 0, 0);
            }
        }
        if (a.out_dim == 1) {      // SDF only (the depth configs): 48 fmas + one exchange, as in the f32 kernel
            float dot = 0.0f;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) {
                    const int u0 = 32 * ct + 8 * aa + 4 * half;
                    const float4 bb = *(const float4 *)(b1s + u0);
                    const float4 w2 = *(const float4 *)(w2r + u0);
                    dot = fmaf(so_softplus(acc[ct][4 * aa + 0] + bb.x), w2.x, dot);
                    dot = fmaf(so_softplus(acc[ct][4 * aa + 1] + bb.y), w2.y, dot);
                    dot = fmaf(so_softplus(acc[ct][4 * aa + 2] + bb.z), w2.z, dot);
                    dot = fmaf(so_softplus(acc[ct][4 * aa + 3] + bb.w), w2.w, dot);
                }
            }
            dot += __shfl_xor(dot, 32, 64);
            if (half == 0 && m < (unsigned)a.M) a.sdf[m] = dot + a.b_out[0];
            continue;
        }
        // output layer: o[voxel][channel] += z[voxel][unit] W2[channel][unit]; step (ct, p): the lane's units acc[ct][8 p ..+8]
        f32x16 o;
#pragma unroll
        for (int v = 0; v < 16; ++v) o[v] = 0.0f;
        const __bf16 *w2row = w2b + (size_t)i * KPB + 4 * half;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float z[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int v = 8 * p + j, u = 32 * ct + 8 * (v >> 2) + 4 * half + (v & 3);
                    z[j] = so_softplus(acc[ct][v] + b1s[u]);
                }
                bf16x8f z1, z2, z3;
                so_fsplit3(z, z1, z2, z3);
                bf16x8f wq[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const __bf16 *wp = w2row + (size_t)pl * 32 * KPB + 32 * ct + 16 * p;
                    const bf16x4f lo = *(const bf16x4f *)wp, hi = *(const bf16x4f *)(wp + 8);
                    wq[pl][0] = lo[0]; wq[pl][1] = lo[1]; wq[pl][2] = lo[2]; wq[pl][3] = lo[3];
                    wq[pl][4] = hi[0]; wq[pl][5] = hi[1]; wq[pl][6] = hi[2]; wq[pl][7] = hi[3];
                }
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z3, wq[0], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z1, wq[2], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z2, wq[1], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z2, wq[0], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z1, wq[1], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z1, wq[0], o, 0, 0, 0);
            }
        }
        const int n = i;
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
                const float f = n < a.out_dim ? val : 0.0f;
                if (BF16) ((__hip_bfloat16 *)a.feat)[(size_t)mm * a.feat_stride + (n - 1)] = __float2bfloat16(f);
                else ((float *)a.feat)[(size_t)mm * a.feat_stride + (n - 1)] = f;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// backward of the fused volume MLP (training), C = 96, one hidden layer.  Nothing of the forward is kept:
// per 32-voxel tile the wave recomputes a = Softplus(x) and y = W1 a + b1, then runs the four gradient GEMMs on
// the same MFMA operand scheme (528 MFMAs per tile, all operands from registers / LDS):
//     dZ = dOut W2                (K = 32 outputs)          dY = dZ * sigmoid(y)
//     dW2 += z^T dOut             (K = the tile's 32 rows)  dW1 += a^T dY
//     dA = dY W1                                            dX = dA * sigmoid(x)
// and scatters dX to the three plane gradients: g_zh / g_wz one 128-byte row segment per voxel, g_hw summed
// over the tile's rows that share (h, w) first (registers + one shuffle).  The weight / bias gradients stay
// in accumulators for all tiles of the wave and are added to global memory once at the end.
// One wave per SIMD (the 192 accumulator registers of dW1 / dW2 need the full register file); LDS = the
// hidden weight in B-operand order (row stride 97: conflict-free as B[k][n] for the forward AND as B[n][k]
// for dA), W2 in [o][n] order, and two 32 x 100 transpose tiles per wave.
// ---------------------------------------------------------------------------------------
struct FieldBwdArgs {
    const float *hw, *zh, *wz;
    int H, W, D;
    const float *w1, *b1, *w2;      // (96, 96), (96), (out_dim, 96)
    int out_dim;
    const float *g_sdf;             // (M) or NULL
    const float *g_feat;            // (M, feat_stride) or NULL
    int feat_stride;
    float *g_hw, *g_zh, *g_wz;      // zero-initialised, accumulated
    float *g_w1, *g_b1, *g_w2, *g_b2;
    long long M;
    int n_tiles;            // PH * PW * PD patches of 4 x 4 x 2 voxels
    int PW, PD;
};

constexpr int kFB_C = 96, kFB_KS = 48, kFB_LD = 97, kFB_TS = 100, kFB_WAVES = 4;

SO_DEVFN int so_crow(int v, int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }   // C-layout row of register v

__global__ __launch_bounds__(kFB_WAVES * 64) void field_volume_bwd_kernel(FieldBwdArgs a) {
    constexpr int C = kFB_C, KS = kFB_KS, LD = kFB_LD, TS = kFB_TS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *w1s = smem;                          // [kk = ks * 2 + half][n], k = half * 48 + ks     96 x 97
    float *w2k = w1s + C * LD;                  // [oo = os * 2 + ohalf][n], o = ohalf * 16 + os    32 x 97
    float *tiles = w2k + 32 * LD;               // per wave: T1 (a), T2 (z, then dY)               2 x 32 x 100
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;
    for (int e = threadIdx.x; e < C * C; e += kFB_WAVES * 64) {
        const int k = e % C, n = e / C;                       // coalesced read of w1 (n, k)
        w1s[((k % KS) * 2 + k / KS) * LD + n] = a.w1[e];
    }
    for (int e = threadIdx.x; e < 32 * C; e += kFB_WAVES * 64) {
        const int n = e % C, o = e / C;
        w2k[((o % 16) * 2 + o / 16) * LD + n] = o < a.out_dim ? a.w2[(size_t)o * C + n] : 0.0f;
    }
    __syncthreads();
    float *T1 = tiles + (size_t)wave * 2 * 32 * TS, *T2 = T1 + 32 * TS;

    f32x16 dW1[3][3], dW2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int v = 0; v < 16; ++v) dW2[r][v] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int v = 0; v < 16; ++v) dW1[r][c][v] = 0.0f;
    }
    float db1[3] = {0.0f, 0.0f, 0.0f};
    float db2[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) db2[k] = 0.0f;
    float bias1[3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) bias1[ct] = a.b1[ct * 32 + i];

    // d-walk (round 4): a wave takes a CONTIGUOUS range of tiles (d-patches fastest: a column of (h, w) patches after the other)
    // and carries the 16 hw rows of the current column in registers across its d-patches: 16 + 16 / 13 instead of 32 plane-row
    // atomics per tile (WRITE_SIZE 0.70 -> 0.4 GB per launch; the time does not move — 2.22 ms either way, the kernel is
    // instruction-bound — but g_hw no longer depends on the order in which the d-patches of a column arrive)
    float hw_acc[3][8];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int q = 0; q < 8; ++q) hw_acc[ct][q] = 0.0f;
    int col_prev = -1;
    auto flush_hw = [&](int col) {
        const int pw_ = col % a.PW, ph_ = col / a.PW;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int hh = 4 * ph_ + (q >> 1), ww = 4 * pw_ + 2 * half + (q & 1);
                if (hh < a.H && ww < a.W && hw_acc[ct][q] != 0.0f)
                    unsafeAtomicAdd(a.g_hw + ((size_t)hh * a.W + ww) * C + ct * 32 + i, hw_acc[ct][q]);
                hw_acc[ct][q] = 0.0f;
            }
        }
    };
    const int n_waves = gridDim.x * kFB_WAVES, wid = blockIdx.x * kFB_WAVES + wave;
    const int t_lo = (int)((long long)a.n_tiles * wid / n_waves), t_hi = (int)((long long)a.n_tiles * (wid + 1) / n_waves);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        if (tile / a.PD != col_prev) {
            if (col_prev >= 0) flush_hw(col_prev);
            col_prev = tile / a.PD;
        }
        // a tile = a 4 (h) x 4 (w) x 2 (d) patch of voxels, row r of the tile = voxel (r >> 3, (r >> 1) & 3, r & 1) of the patch:
        // every plane row (h, w) / (d, h) / (w, d) is then shared by 2 / 4 / 4 rows of the tile whose C-layout registers sit
        // in one lane (or its partner half), and the scatter below adds them up before it issues an atomic — 32 plane-row
        // atomics per tile instead of 66 with the 32 consecutive voxels of round 2 (1.33 GB of fabric writes per launch,
        // profiles/r3_k_train_bwd_pmc.txt)
        const int pd = tile % a.PD, tq = tile / a.PD;
        const int pw = tq % a.PW, ph = tq / a.PW;
        const int h_b = 4 * ph, w_b = 4 * pw, d_b = 2 * pd;
        auto row_voxel = [&](int r, bool &live) {                // linear index (h W + w) D + d of row r (clamped when dead)
            const int hh = h_b + (r >> 3), ww = w_b + ((r >> 1) & 3), dd = d_b + (r & 1);
            live = (hh < a.H) & (ww < a.W) & (dd < a.D);
            return ((long long)min(hh, a.H - 1) * a.W + min(ww, a.W - 1)) * a.D + min(dd, a.D - 1);
        };
        // ---- S1: a = Softplus(x), A layout (row i, columns half * 48 ..) -> registers and T1 ---------------
        bool mlive;
        const long long m = row_voxel(i, mlive);
        const int h = min(h_b + (i >> 3), a.H - 1), w = min(w_b + ((i >> 1) & 3), a.W - 1), d = min(d_b + (i & 1), a.D - 1);
        const int hwi = h * a.W + w;
        float av[KS];
        {
            const float4 *p0 = (const float4 *)(a.hw + (size_t)hwi * C + half * KS);
            const float4 *p1 = (const float4 *)(a.zh + ((size_t)d * a.H + h) * C + half * KS);
            const float4 *p2 = (const float4 *)(a.wz + ((size_t)w * a.D + d) * C + half * KS);
            float4 *t1 = (float4 *)(T1 + i * TS + half * KS);
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const float4 x0 = p0[q], x1 = p1[q], x2 = p2[q];
                float4 s4;
                s4.x = so_softplus((x0.x + x1.x) + x2.x);
                s4.y = so_softplus((x0.y + x1.y) + x2.y);
                s4.z = so_softplus((x0.z + x1.z) + x2.z);
                s4.w = so_softplus((x0.w + x1.w) + x2.w);
                av[4 * q + 0] = s4.x; av[4 * q + 1] = s4.y; av[4 * q + 2] = s4.z; av[4 * q + 3] = s4.w;
                t1[q] = s4;
                // the 192 weight-gradient accumulators leave ~300 registers: keep at most 4 x 3 loads in flight
                if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- S2: y = W1 a + b1 (C layout); z = Softplus(y) -> T2 --------------------------------------------
        f32x16 acc[3];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ct][v] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float *brow = w1s + (ks * 2 + half) * LD + i;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], brow[ct * 32], acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) T2[so_crow(v, half) * TS + ct * 32 + i] = so_softplus(acc[ct][v] + bias1[ct]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- S3: dOut, A layout: lane (row i, half) holds outputs half * 16 .. + 16 ------------------------
        float dav[16];
#pragma unroll
        for (int os = 0; os < 16; ++os) {
            const int o = half * 16 + os;
            float g = 0.0f;
            if (mlive && o < a.out_dim) {
                if (o == 0) g = a.g_sdf ? a.g_sdf[m] : 0.0f;
                else g = a.g_feat ? a.g_feat[(size_t)m * a.feat_stride + (o - 1)] : 0.0f;
            }
            dav[os] = g;
            db2[os] += g;
        }
        // ---- S4 / S5: dZ = dOut W2 ; dY = dZ sigmoid(y) ----------------------------------------------------
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ct][v] = 0.0f;
#pragma unroll
        for (int os = 0; os < 16; ++os) {
            const float *brow = w2k + (os * 2 + half) * LD + i;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(dav[os], brow[ct * 32], acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            float sum = 0.0f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float z = T2[so_crow(v, half) * TS + ct * 32 + i];   // this lane's own element (written in S2)
                acc[ct][v] = acc[ct][v] * (1.0f - __expf(-z));           // sigmoid(y) = 1 - exp(-softplus(y))
                sum += acc[ct][v];
            }
            db1[ct] += sum;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- S6: dW2 += dOut^T z : rows o, columns n (3 tiles), K = the 32 rows of the tile ----------------
#pragma unroll
        for (int ms = 0; ms < 16; ++ms) {
            bool rlive;
            const long long mr = row_voxel(2 * ms + half, rlive);  // A' operand: dOut[mr][o = i]
            float g = 0.0f;
            if (rlive && i < a.out_dim) {
                if (i == 0) g = a.g_sdf ? a.g_sdf[mr] : 0.0f;
                else g = a.g_feat ? a.g_feat[(size_t)mr * a.feat_stride + (i - 1)] : 0.0f;
            }
            const float *zrow = T2 + (2 * ms + half) * TS + i;     // B' operand: z[mr][n = ct * 32 + i]
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)                          // rows o, columns n: row-contiguous in g_w2 (o, n)
                dW2[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(g, zrow[ct * 32], dW2[ct], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- S7: dY -> T2 ----------------------------------------------------------------------------------
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) T2[so_crow(v, half) * TS + ct * 32 + i] = acc[ct][v];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- S8: dW1 += dY^T a : rows n (3 tiles), columns k (3 tiles) -------------------------------------
#pragma unroll
        for (int ms = 0; ms < 16; ++ms) {
            const float *arow = T1 + (2 * ms + half) * TS + i;
            const float *yrow = T2 + (2 * ms + half) * TS + i;
            float bv[3];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) bv[ct] = yrow[ct * 32];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {                        // rows n (from dY), columns k (from a):
                const float avv = arow[ct * 32];                    // row-contiguous in g_w1 (n, k)
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
                    dW1[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[rt], avv, dW1[rt][ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- S9: dA = dY W1 : A = dY in A layout (from T2), B[n][k] = W1[n][k] (w1s read the other way) ----
        {
            const float4 *yr = (const float4 *)(T2 + i * TS + half * KS);
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const float4 t = yr[q];
                av[4 * q + 0] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
            }
        }
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ct][v] = 0.0f;
        {
            int krow[3];                                           // w1s row of column k = ct * 32 + i
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const int k = ct * 32 + i;
                krow[ct] = ((k % KS) * 2 + k / KS) * LD;
            }
#pragma unroll
            for (int ns = 0; ns < KS; ++ns) {
                const int n = half * KS + ns;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ns], w1s[krow[ct] + n], acc[ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- S10 / S11: dX = dA sigmoid(x) (C layout) -> plane gradients -----------------------------------
        // C-layout register v of lane (i, half) = tile row r = (v & 3) + 8 (v >> 2) + 4 half = patch voxel
        //   dd = v & 1,   ww = 2 half + ((v >> 1) & 1),   hh = v >> 2
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int k = ct * 32 + i;
            float dx[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = so_crow(v, half);
                const bool live = (h_b + (v >> 2) < a.H) & (w_b + 2 * half + ((v >> 1) & 1) < a.W) & (d_b + (v & 1) < a.D);
                const float ar = T1[r * TS + k];
                dx[v] = live ? acc[ct][v] * (1.0f - __expf(-ar)) : 0.0f;
            }
            // g_hw[h][w] += sum over d: registers v, v ^ 1 (8 rows per lane, the halves hold different w)
#pragma unroll
            for (int q = 0; q < 8; ++q) hw_acc[ct][q] += dx[2 * q] + dx[2 * q + 1];
            // g_wz[w][d] += sum over h: registers v, v + 4, v + 8, v + 12 (4 rows per lane)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ww = w_b + 2 * half + (q >> 1), dd = d_b + (q & 1);
                if (ww < a.W && dd < a.D)
                    unsafeAtomicAdd(a.g_wz + ((size_t)ww * a.D + dd) * C + k, (dx[q] + dx[q + 4]) + (dx[q + 8] + dx[q + 12]));
            }
            // g_zh[d][h] += sum over w: registers v, v ^ 2 and the partner half; half 0 issues d = d_b, half 1 d = d_b + 1
#pragma unroll
            for (int hq = 0; hq < 4; ++hq) {
                const float s0 = dx[4 * hq] + dx[4 * hq + 2], s1 = dx[4 * hq + 1] + dx[4 * hq + 3];     // dd = 0 / 1, this half's two w
                const float t0 = s0 + __shfl_xor(s0, 32, 64), t1 = s1 + __shfl_xor(s1, 32, 64);
                const int hh = h_b + hq, dd = d_b + half;
                if (hh < a.H && dd < a.D) unsafeAtomicAdd(a.g_zh + ((size_t)dd * a.H + hh) * C + k, half ? t1 : t0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (col_prev >= 0) flush_hw(col_prev);

    // ---- the wave's weight / bias gradients -> global (once) -------------------------------------------------
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int cr = so_crow(v, half);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)            // dW1[n = rt * 32 + cr][k = ct * 32 + i] -> g_w1 (n, k): 128-byte rows
                unsafeAtomicAdd(a.g_w1 + (size_t)(rt * 32 + cr) * C + ct * 32 + i, dW1[rt][ct][v]);
            if (cr < a.out_dim) unsafeAtomicAdd(a.g_w2 + (size_t)cr * C + ct * 32 + i, dW2[ct][v]);   // dW2[o = cr][n]
        }
    }
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const float tot = db1[ct] + __shfl_xor(db1[ct], 32, 64);
        if (half == 0) unsafeAtomicAdd(a.g_b1 + ct * 32 + i, tot);
    }
#pragma unroll
    for (int os = 0; os < 16; ++os) {
        float t = db2[os];
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1) t += __shfl_xor(t, msk, 64);
        const int o = half * 16 + os;
        if (i == 0 && o < a.out_dim) unsafeAtomicAdd(a.g_b2 + o, t);
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
    SO_REQUIRE(M < (1LL << 31) - 64, "field_volume: volume too large");
    FieldArgs a{hw, zh, wz, H, W, D, w_hidden, b_hidden, w_out, b_out, n_hidden, out_dim, sdf, feat, feat_stride, M,
                (int)((M + 31) / 32)};
    hipStream_t st = (hipStream_t)stream;
    // round 3: both GEMMs on the bf16 matrix pipe (exact three-way split); SELFOCC_FIELD_B3=0 keeps the f32-MFMA kernel
    static const bool use_b3 = !(getenv("SELFOCC_FIELD_B3") && atoi(getenv("SELFOCC_FIELD_B3")) == 0);
    if (use_b3 && C == 96 && n_hidden == 1) {
        const size_t shm3 = (size_t)3 * (96 + 32) * kFieldB3KPB * 2 + 2 * 96 * sizeof(float);
        const int blocks3 = std::min((a.n_tiles + kFieldB3Waves - 1) / kFieldB3Waves, 256);      // one 12-wave block per CU
        static std::atomic<unsigned long long> done3{0};
        int dev3 = 0;
        (void)hipGetDevice(&dev3);
        if (!(done3.load(std::memory_order_relaxed) & (1ull << (dev3 & 63)))) {
            (void)hipFuncSetAttribute((const void *)field_volume_b3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            (void)hipFuncSetAttribute((const void *)field_volume_b3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            done3.fetch_or(1ull << (dev3 & 63), std::memory_order_relaxed);
        }
        if (feat_dtype == SO_DTYPE_BF16) hipLaunchKernelGGL(field_volume_b3_kernel<true>, dim3(blocks3), dim3(kFieldB3Waves * 64), shm3, st, a);
        else hipLaunchKernelGGL(field_volume_b3_kernel<false>, dim3(blocks3), dim3(kFieldB3Waves * 64), shm3, st, a);
        return so_launch_status();
    }
    const int nw = kFieldFwdWaves;
    const size_t shm = ((size_t)(C + 32) * (C + 4) + C) * sizeof(float);
    // persistent blocks: as many as the LDS of a CU holds (52 KB each at C = 96 -> 3 per CU), at most one tile each
    const int per_cu = std::max(1, std::min(4, (int)((160 * 1024) / (shm + 1024))));
    const int blocks = std::min((a.n_tiles + nw - 1) / nw, 256 * per_cu);
#define SO_LAUNCH(CC, BF)                                                                                    \
    {                                                                                                        \
        /* per launch: the attribute is per device (a process may drive several GPUs) */                     \
        (void)hipFuncSetAttribute((const void *)field_volume_kernel<CC, BF>,                                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);             \
        hipLaunchKernelGGL((field_volume_kernel<CC, BF>), dim3(blocks), dim3(nw * 64), shm, st, a);             \
    }
    const bool bf = feat_dtype == SO_DTYPE_BF16;
    if (C == 64) { if (bf) SO_LAUNCH(64, true) else SO_LAUNCH(64, false) }
    else if (C == 96) { if (bf) SO_LAUNCH(96, true) else SO_LAUNCH(96, false) }
    else { if (bf) SO_LAUNCH(128, true) else SO_LAUNCH(128, false) }
#undef SO_LAUNCH
    return so_launch_status();
}


int so_field_volume_bwd_b3(const float *hw, const float *zh, const float *wz, int H, int W, int D, const float *w1,
                           const float *b1, const float *w2, int out_dim, const float *g_sdf, const float *g_feat,
                           int feat_stride, float *g_hw, float *g_zh, float *g_wz, float *g_w1, float *g_b1, float *g_w2,
                           float *g_b2, hipStream_t st);      // field_bwd_b3.hip

extern "C" int selfocc_field_volume_bwd(const float *hw, const float *zh, const float *wz, int32_t H, int32_t W,
                                        int32_t D, int32_t C, const float *w_hidden, const float *b_hidden,
                                        const float *w_out, int32_t out_dim, const float *g_sdf, const float *g_feat,
                                        int32_t feat_stride, float *g_hw, float *g_zh, float *g_wz, float *g_w_hidden,
                                        float *g_b_hidden, float *g_w_out, float *g_b_out, void *stream) {
    SO_REQUIRE(H >= 1 && W >= 1 && D >= 1, "field_volume_bwd: bad volume size (%d, %d, %d)", H, W, D);
    SO_REQUIRE(C == 96, "field_volume_bwd: embed_dims must be 96 (got %d); use the autograd path", C);
    SO_REQUIRE(out_dim >= 1 && out_dim <= 32, "field_volume_bwd: 1 + color_dims must be <= 32 (got %d)", out_dim);
    SO_REQUIRE(hw && zh && wz && w_hidden && b_hidden && w_out, "field_volume_bwd: NULL input pointer");
    SO_REQUIRE(g_hw && g_zh && g_wz && g_w_hidden && g_b_hidden && g_w_out && g_b_out,
               "field_volume_bwd: NULL gradient pointer");
    SO_REQUIRE(g_feat == nullptr || feat_stride >= out_dim - 1, "field_volume_bwd: feat_stride %d < %d colour channels",
               feat_stride, out_dim - 1);
    const long long M = (long long)H * W * D;
    SO_REQUIRE(M < (1LL << 31) * 32, "field_volume_bwd: volume too large");
    const long long PH = (H + 3) / 4, PW = (W + 3) / 4, PD = (D + 1) / 2;
    SO_REQUIRE(PH * PW * PD < (1LL << 31), "field_volume_bwd: volume too large");
    // round 5: the five GEMMs on the bf16 matrix pipe (field_bwd_b3.hip); SELFOCC_FIELD_BWD_B3=0 keeps the f32-MFMA kernel.
    // (its 16-byte loads of the upstream feature gradient need rows of a multiple of four floats)
    static const bool bwd_b3 = !(getenv("SELFOCC_FIELD_BWD_B3") && atoi(getenv("SELFOCC_FIELD_BWD_B3")) == 0);
    // (and it indexes voxels / feature rows in 32 bits: larger volumes stay on the 64-bit-indexed f32 kernel)
    if (bwd_b3 && (g_feat == nullptr || feat_stride % 4 == 0) && M * std::max(feat_stride, 1) < (1LL << 31))
        return so_field_volume_bwd_b3(hw, zh, wz, H, W, D, w_hidden, b_hidden, w_out, out_dim, g_sdf, g_feat, feat_stride, g_hw,
                                      g_zh, g_wz, g_w_hidden, g_b_hidden, g_w_out, g_b_out, (hipStream_t)stream);
    FieldBwdArgs a{hw, zh, wz, H, W, D, w_hidden, b_hidden, w_out, out_dim, g_sdf, g_feat, feat_stride,
                   g_hw, g_zh, g_wz, g_w_hidden, g_b_hidden, g_w_out, g_b_out, M, (int)(PH * PW * PD), (int)PW, (int)PD};
    const size_t shm = ((size_t)kFB_C * kFB_LD + 32 * kFB_LD + (size_t)kFB_WAVES * 2 * 32 * kFB_TS) * sizeof(float);
    // per launch: the attribute is per device (a process may drive several GPUs)
    (void)hipFuncSetAttribute((const void *)field_volume_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024 - 256);
    const int blocks = std::min((a.n_tiles + kFB_WAVES - 1) / kFB_WAVES, 256);
    hipLaunchKernelGGL(field_volume_bwd_kernel, dim3(blocks), dim3(kFB_WAVES * 64), shm, (hipStream_t)stream, a);
    return so_launch_status();
}
