// field_bwd_b3.hip — backward of the fused tri-plane -> volume MLP (training, C = 96, one hidden layer) with all FIVE
// GEMMs on the bf16 matrix pipe through the exact three-way split (round 5; field.hip's field_volume_bwd_kernel keeps the
// float32-MFMA form for A/B: SELFOCC_FIELD_BWD_B3=0).
//
// Reference semantics: autograd through the field the head drives (sdfstudio-fork SDFCustomField; in-repo analogue
// model/head/nerfacc_head/bev_nerf.py:74-95):  x = hw + zh + wz,  a = Softplus(x),  y = W1 a + b1,  z = Softplus(y),
// out = W2 z + b2.  Nothing of the forward is kept; per 32-voxel tile (a 4 x 4 x 2 patch) a wave recomputes a, y, z and runs
//     y   = a W1^T            (K = 96 inputs)        dZ  = dOut W2          (K = 32 outputs)      dY = dZ * sigmoid(y)
//     dW2 += dOut^T z         (K = 32 voxels)        dW1 += dY^T a          (K = 32 voxels)
//     dA  = dY W1             (K = 96 units)         dX  = dA * sigmoid(x)  -> the three plane gradients
// f32 MFMA on gfx950 runs at the vector rate (528 v_mfma_f32_32x32x2 = 33.8 k matrix cycles per tile); a float32 is the exact
// sum of three bfloat16 and bf16 products are exact in the f32 accumulator, so six v_mfma_f32_32x32x16_bf16 per 16-k step
// (x_i w_j, i + j <= 4) give float32-level accuracy at 396 MFMAs = 12.7 k matrix cycles per tile.
//
// Orientation ("lane = unit"): every chain GEMM is computed as rows = voxels, columns = units / channels, so that its
// C-layout result — lane (column c = lane & 31, half = lane >> 5) holds rows r(v, half) = 8 (v >> 2) + 4 half + (v & 3) —
//   * is, eight consecutive registers at a time, DIRECTLY the bf16 A / B operand of the weight-gradient MFMAs
//     (k = the 16 voxels {16 s + 8 q + 4 half + b}; dY -> A of dW1, z -> B of dW2: no LDS round trip),
//   * carries the bias / db1 as one register per lane, and
//   * leaves dX with a lane per channel, i.e. the plane-row atomics stay 128-byte row segments.
// Two operands need the other layout and go through ONE wave-private 13.8 KB LDS tile: a (built lane-per-voxel for the
// y GEMM, read back lane-per-channel for dW1 / dX) and dY (lane-per-unit -> lane-per-voxel for the dA GEMM).
// The k labelling of an MFMA is free as long as A and B agree: a lane (row, kb) owns the k values
//     K(ct, p, kb)[j] = 32 ct + 16 p + 8 (j >> 2) + 4 kb + (j & 3)        (ct < 3, p < 2, j < 8)
// of k-step (ct, p) — the C-layout rows of registers 8 p .. 8 p + 7 — in every GEMM that contracts 96 features.
// W1 sits in LDS ONCE ([unit][input], three bf16 planes): the y GEMM contracts inputs (8-byte reads along a row), the dA
// GEMM contracts units — k-strided in that image — and reads it with ds_read_b64_tr_b16 (gfx950's transpose read: within
// 16 lanes, lane s supplies the address of M[k = s >> 2][columns 4 (s & 3) .. + 3] and lane l receives M[k = 0..3][column l];
// mapping measured with scripts/micro/tr16_map.hip).  A second [input][unit] copy (60 KB) would not fit next to the tiles.
#include "so_device.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// torch.nn.Softplus(beta=1, threshold=20) — the same evaluation as field.hip's forward
// (1 + e >= 1 is never denormal: the raw v_log_f32 / v_exp_f32 instructions, without the denormal-range scaling that
// __logf emits under -fno-fast-math — 26 -> 14 vector instructions per evaluation, 192 evaluations per voxel row pair)
SO_DEVFN float fb_softplus(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
    const float series = e * (1.0f - e * (0.5f - e * (0.33333334f - 0.25f * e)));
    const float lg = __builtin_amdgcn_logf(1.0f + e) * 0.69314718055994531f;
    const float r = e < 0.05f ? series : lg;
    return x > 20.0f ? x : r;
}
SO_DEVFN float fb_exp_neg(float z) { return __builtin_amdgcn_exp2f(z * -1.44269504088896341f); }     // exp(-z)

SO_DEVFN void fb_split3(const float (&x)[8], bf16x8 &a1, bf16x8 &a2, bf16x8 &a3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 b1 = (__bf16)x[j];
        const float r1 = x[j] - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const float r2 = r1 - (float)b2;
        a1[j] = b1; a2[j] = b2; a3[j] = (__bf16)r2;
    }
}

// acc += A B with A = a1 + a2 + a3, B = b1 + b2 + b3 (all exact), the six products with i + j <= 4, small terms first
SO_DEVFN void fb_mfma6(f32x16 &acc, const bf16x8 &a1, const bf16x8 &a2, const bf16x8 &a3, const bf16x8 &b1, const bf16x8 &b2,
                       const bf16x8 &b3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
}


typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// three-way split of TWO floats into packed bf16 pairs (one dword per plane)
SO_DEVFN void fb_split_pair(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3) {
    const bf16x2 b1 = {(__bf16)x0, (__bf16)x1};
    const float r0 = x0 - (float)b1[0], r1 = x1 - (float)b1[1];
    const bf16x2 b2 = {(__bf16)r0, (__bf16)r1};
    const float s0 = r0 - (float)b2[0], s1 = r1 - (float)b2[1];
    const bf16x2 b3 = {(__bf16)s0, (__bf16)s1};
    p1 = __builtin_bit_cast(unsigned, b1); p2 = __builtin_bit_cast(unsigned, b2); p3 = __builtin_bit_cast(unsigned, b3);
}
SO_DEVFN bf16x8 fb_as8(const u32x4 &v) { return __builtin_bit_cast(bf16x8, v); }

// fb_mfma6 with a work item after each of the six MFMAs, the order pinned by scheduling fences: with ONE wave per SIMD the
// vector work of the next operand only hides in the shadow of the dependent MFMA chain (32 cycles each) if it sits between
// the MFMAs in program order, and the compiler's scheduler, left alone, issues the six MFMAs first
#define FB_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef FB_IL_S2
#define FB_IL_S2 1
#endif
#ifndef FB_IL_DW1
#define FB_IL_DW1 0   // measured: 1.37 ms with, 1.31 ms without (the up-front a splits are not hidden, and the fences cost more than they save here)
#endif
#ifndef FB_IL_S6
#define FB_IL_S6 1
#endif
template <bool IL, class F>
SO_DEVFN void fb_mfma6_il(f32x16 &acc, const bf16x8 &a1, const bf16x8 &a2, const bf16x8 &a3, const bf16x8 &b1, const bf16x8 &b2,
                          const bf16x8 &b3, F &&gap) {
    if constexpr (!IL) {          // A/B: the same work items, left to the compiler's scheduler
#pragma unroll
        for (int k = 0; k < 6; ++k) gap(k);
        fb_mfma6(acc, a1, a2, a3, b1, b2, b3);
        return;
    }
    FB_SB();
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0); FB_SB(); gap(0); FB_SB();
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0); FB_SB(); gap(1); FB_SB();
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0); FB_SB(); gap(2); FB_SB();
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0); FB_SB(); gap(3); FB_SB();
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0); FB_SB(); gap(4); FB_SB();
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0); FB_SB(); gap(5); FB_SB();
}

SO_DEVFN bf16x8 fb_cat(const bf16x4 &lo, const bf16x4 &hi) {
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

SO_DEVFN bf16x4 fb_tr4(const __bf16 *p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p);
    return __builtin_bit_cast(bf16x4, v);
}

SO_DEVFN int fb_crow(int v, int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }   // C-layout row of register v

struct FieldBwdB3Args {
    const float *hw, *zh, *wz;
    int H, W, D;
    const float *w1, *b1, *w2;      // (96, 96), (96), (out_dim, 96)
    int out_dim;
    const float *g_sdf;             // (M) or NULL
    const float *g_feat;            // (M, feat_stride) or NULL
    int feat_stride;
    float *g_hw, *g_zh, *g_wz;      // zero-initialised, accumulated
    float *g_w1, *g_b1, *g_w2, *g_b2;
    int n_tiles;                    // PH * PW * PD patches of 4 x 4 x 2 voxels
    int PW, PD;
    int dbg;                        // dev switch SELFOCC_FIELD_BWD_DBG: 1 = no zh / wz plane-gradient atomics (timing only)
};

constexpr int kB3_C = 96, kB3_WAVES = 4;
constexpr int kB3_KPB = 100;   // W1 row stride (bf16): 50 dwords — conflict-free for 8-byte reads of 32 rows AND for the transpose reads
constexpr int kB3_OPB = 40;    // W2^T row stride (bf16): 20 dwords — conflict-free 16-byte reads
constexpr int kB3_TA = 36;     // TA [96 channels][32 voxels + 4]  (float)
constexpr int kB3_TY = 100;    // TY [32 voxels][96 units + 4]     (float), same buffer
constexpr size_t kB3_W1_BYTES = (size_t)3 * kB3_C * kB3_KPB * 2;      // 57 600
constexpr size_t kB3_W2_BYTES = (size_t)3 * kB3_C * kB3_OPB * 2;      // 23 040
constexpr size_t kB3_T_FLOATS = (size_t)kB3_C * kB3_TA;               // 3 456 floats = 13 824 B per wave (TY needs 3 200)
constexpr int kB3_HWA = 24;    // per lane: the 3 x 8 carried hw-row sums of the d-walk (lane-private LDS slots, not registers)
constexpr size_t kB3_LDS = kB3_W1_BYTES + kB3_W2_BYTES + kB3_WAVES * kB3_T_FLOATS * 4 + (size_t)kB3_WAVES * 64 * kB3_HWA * 4;   // 160 512 B

// output columns are re-ordered so that the feature channels start at o' = 0 (16-byte aligned rows of g_feat):
//   o' < 31: feature channel o' (row o' + 1 of W2) when o' < out_dim - 1, else nothing;   o' = 31: the SDF (row 0 of W2)
SO_DEVFN int fb_w2_row(int op, int out_dim) { return op == 31 ? 0 : (op < out_dim - 1 ? op + 1 : -1); }

__global__ __launch_bounds__(kB3_WAVES * 64) void field_volume_bwd_b3_kernel(FieldBwdB3Args a) {
    constexpr int C = kB3_C, KPB = kB3_KPB, OPB = kB3_OPB, THREADS = kB3_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16 *w1b = (__bf16 *)smem_raw;                               // [3 planes][96 units][KPB]   W1[unit][input]
    __bf16 *w2t = (__bf16 *)(smem_raw + kB3_W1_BYTES);              // [3 planes][96 units][OPB]   W2[o'][unit] transposed
    float *tiles = (float *)(smem_raw + kB3_W1_BYTES + kB3_W2_BYTES);
    const int lane_ = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lane = lane_, i = lane & 31, half = lane >> 5;

    // ---- stage the weights: split once per block --------------------------------------------------------------
    for (int e = threadIdx.x; e < C * (C / 4); e += THREADS) {
        const int n = e / (C / 4), k4 = e - n * (C / 4);
        const float4 v = ((const float4 *)(a.w1 + (size_t)n * C))[k4];
        const float x[4] = {v.x, v.y, v.z, v.w};
        bf16x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 b1 = (__bf16)x[j];
            const float r1 = x[j] - (float)b1;
            const __bf16 b2 = (__bf16)r1;
            p1[j] = b1; p2[j] = b2; p3[j] = (__bf16)(r1 - (float)b2);
        }
        __bf16 *dst = w1b + (size_t)n * KPB + 4 * k4;
        *(bf16x4 *)dst = p1;
        *(bf16x4 *)(dst + (size_t)C * KPB) = p2;
        *(bf16x4 *)(dst + (size_t)2 * C * KPB) = p3;
    }
    for (int e = threadIdx.x; e < C * 32; e += THREADS) {
        const int n = e >> 5, op = e & 31;
        const int row = fb_w2_row(op, a.out_dim);
        const float x = row >= 0 ? a.w2[(size_t)row * C + n] : 0.0f;
        const __bf16 b1 = (__bf16)x;
        const float r1 = x - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        __bf16 *dst = w2t + (size_t)n * OPB + op;
        dst[0] = b1; dst[(size_t)C * OPB] = b2; dst[(size_t)2 * C * OPB] = (__bf16)(r1 - (float)b2);
    }
    __syncthreads();

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
    float db2 = 0.0f;              // lane (o' = i, half): sum over the voxels this lane has seen

    // d-walk (field.hip, round 4): a wave takes a contiguous range of tiles, d-patches fastest, and carries the 16 hw rows of
    // its current (h, w) column across the column's d-patches — here in 24 lane-private LDS floats (the register file is full:
    // 192 weight-gradient accumulators + the chain's 3 x 48)
    float *hwa = tiles + (size_t)kB3_WAVES * kB3_T_FLOATS + ((size_t)wave * 64 + lane) * kB3_HWA;
#pragma unroll
    for (int q = 0; q < kB3_HWA / 4; ++q) ((float4 *)hwa)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    int col_prev = -1;
    auto flush_hw = [&](int col) {
        const int pw_ = col % a.PW, ph_ = col / a.PW;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const float4 lo = ((const float4 *)hwa)[2 * ct], hi = ((const float4 *)hwa)[2 * ct + 1];
            const float acc8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int hh = 4 * ph_ + (q >> 1), ww = 4 * pw_ + 2 * half + (q & 1);
                if (hh < a.H && ww < a.W && acc8[q] != 0.0f)
                    unsafeAtomicAdd(a.g_hw + ((size_t)hh * a.W + ww) * C + ct * 32 + i, acc8[q]);
            }
            ((float4 *)hwa)[2 * ct] = make_float4(0.f, 0.f, 0.f, 0.f);
            ((float4 *)hwa)[2 * ct + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int n_waves = gridDim.x * kB3_WAVES, wid = blockIdx.x * kB3_WAVES + wave;
    const int t_lo = (int)((long long)a.n_tiles * wid / n_waves), t_hi = (int)((long long)a.n_tiles * (wid + 1) / n_waves);
    const int nfeat = a.g_feat ? a.out_dim - 1 : 0;          // feature channels that carry a gradient
    // lane constants of the patch geometry (tile row r = voxel (r >> 3, (r >> 1) & 3, r & 1) of the 4 x 4 x 2 patch)
    // One tile.  FULL: the 4 x 4 x 2 patch lies inside the volume (89 % of the tiles at 257 x 257 x 25) — no clamps, no liveness
    // masks, no divergent branches around loads and atomics, 32-bit offsets from uniform tile bases.
    auto tile_body = [&](auto full_tag, int h_b, int w_b, int d_b) {
        constexpr bool FULL = decltype(full_tag)::value;
        // The register file is full (192 weight-gradient accumulators + the chain): lane constants that the compiler would hoist
        // out of the tile loop (a dozen pre-multiplied LDS / global offsets) end up in scratch.  Laundering the lane id once per
        // tile makes every offset derived from it a per-tile value: a few integer instructions instead of long-lived registers.
        int lane = lane_, i, half;
        asm volatile("" : "+v"(lane));
        i = lane & 31; half = lane >> 5;
        float *T = tiles + (size_t)wave * kB3_T_FLOATS;
        float *hwa = tiles + (size_t)kB3_WAVES * kB3_T_FLOATS + ((size_t)wave * 64 + lane) * kB3_HWA;
        const int hhi = i >> 3, wwi = (i >> 1) & 3, ddi = i & 1;
        auto row_voxel = [&](int hh_, int ww_, int dd_, bool &live) {      // linear voxel index of patch voxel (hh_, ww_, dd_)
            const int hh = h_b + hh_, ww = w_b + ww_, dd = d_b + dd_;
            if constexpr (FULL) {
                live = true;
                return (hh * a.W + ww) * a.D + dd;
            } else {
                live = (hh < a.H) & (ww < a.W) & (dd < a.D);
                return (min(hh, a.H - 1) * a.W + min(ww, a.W - 1)) * a.D + min(dd, a.D - 1);
            }
        };
        bool mlive;
        const int m = row_voxel(hhi, wwi, ddi, mlive);            // the voxel this lane owns in the lane-per-voxel steps
        // dOut, lane (voxel i, kb = half): float4 q of k-step s holds o' = 16 s + 8 half + 4 q .. + 3; o' >= nfeat carries nothing
        // (the load is clamped into the row instead of branched around: feat_stride is a multiple of 4)
        int oa_off[4];
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) oa_off[sq] = min(16 * (sq >> 1) + 8 * half + 4 * (sq & 1), max(a.feat_stride - 4, 0));
        const float *oc_src = i < nfeat ? a.g_feat + i : ((i == 31 && a.g_sdf) ? a.g_sdf : nullptr);
        const int oc_stride = i < nfeat ? a.feat_stride : 1;

        // ---- dOut, lane (voxel i, kb = half) ------------------------------------------------------------------------
        float dOA[2][8];
        {
            const float *gf = a.g_feat ? a.g_feat + (size_t)m * a.feat_stride : nullptr;
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
                const int s = sq >> 1, q = sq & 1, o0 = 16 * s + 8 * half + 4 * q;
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nfeat > 0) g = *(const float4 *)(gf + oa_off[sq]);       // uniform branch
                const bool ok = FULL ? true : mlive;
                dOA[s][4 * q + 0] = (ok && o0 + 0 < nfeat) ? g.x : 0.0f;
                dOA[s][4 * q + 1] = (ok && o0 + 1 < nfeat) ? g.y : 0.0f;
                dOA[s][4 * q + 2] = (ok && o0 + 2 < nfeat) ? g.z : 0.0f;
                dOA[s][4 * q + 3] = (ok && o0 + 3 < nfeat) ? g.w : 0.0f;
            }
            if (a.g_sdf) {                                                   // o' = 31: element 7 of (s = 1, half = 1)
                const float gs = a.g_sdf[m];
                if (half == 1 && (FULL || mlive)) dOA[1][7] = gs;
            }
        }
        // dOut, lane (o' = i, half): the 16 voxels r(v, half)  (A operand of dW2; its sum over voxels is db2)
        float dOC[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            bool rl;
            const int mr = row_voxel(v >> 2, 2 * half + ((v >> 1) & 1), v & 1, rl);
            dOC[v] = (rl && oc_src) ? oc_src[(size_t)mr * oc_stride] : 0.0f;
            db2 += dOC[v];
        }

        // ---- S1 + S2: a = Softplus(hw + zh + wz) -> TA, and y = a W1^T k-step by k-step --------------------------------
        // lane (voxel i, kb = half): k-step (kc, p) = inputs K(kc, p, half); the raw plane rows of the NEXT step are loaded
        // before this step's MFMAs
        f32x16 y[3];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) y[ct][v] = 0.0f;
        {
            int h, w, d;
            if constexpr (FULL) { h = h_b + hhi; w = w_b + wwi; d = d_b + ddi; }
            else { h = min(h_b + hhi, a.H - 1); w = min(w_b + wwi, a.W - 1); d = min(d_b + ddi, a.D - 1); }
            const float *p0 = a.hw + ((size_t)h * a.W + w) * C + 4 * half;
            const float *p1 = a.zh + ((size_t)d * a.H + h) * C + 4 * half;
            const float *p2 = a.wz + ((size_t)w * a.D + d) * C + 4 * half;
            // software pipeline, spelled out gap by gap: while step st's 18 MFMAs run, the NEXT step's operand is built (8 x
            // Softplus + TA write, 4 pair splits), the next accumulator's W1 rows are read from LDS and the plane rows of step
            // st + 2 are requested.  Work items sit in the gaps of the dependent MFMA chain (fb_mfma6_il).
            float4 xn[6];
            auto load_step = [&](int st) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    xn[q] = *(const float4 *)(p0 + 16 * st + 8 * q); xn[2 + q] = *(const float4 *)(p1 + 16 * st + 8 * q);
                    xn[4 + q] = *(const float4 *)(p2 + 16 * st + 8 * q);
                }
            };
            float xs[8];
            auto soft_elem = [&](int st, int e) {            // element e of step st's operand: Softplus + TA write
                const int q = e >> 2, c = e & 3;
                const float x0 = c == 0 ? xn[q].x : c == 1 ? xn[q].y : c == 2 ? xn[q].z : xn[q].w;
                const float x1 = c == 0 ? xn[2 + q].x : c == 1 ? xn[2 + q].y : c == 2 ? xn[2 + q].z : xn[2 + q].w;
                const float x2 = c == 0 ? xn[4 + q].x : c == 1 ? xn[4 + q].y : c == 2 ? xn[4 + q].z : xn[4 + q].w;
                xs[e] = fb_softplus((x0 + x1) + x2);
                asm volatile("" : "+v"(xs[e]));               // keeps the evaluation in ITS gap (the SLP vectoriser would pair it with
                T[(16 * st + 8 * q + 4 * half + c) * kB3_TA + i] = xs[e];      // a later gap's element and sink both)
            };
            u32x4 xq[3];                                      // the operand under construction, three planes of packed pairs
            auto split_pr = [&](int pj) {
                unsigned q1, q2, q3;
                fb_split_pair(xs[2 * pj], xs[2 * pj + 1], q1, q2, q3);
                xq[0][pj] = q1; xq[1][pj] = q2; xq[2][pj] = q3;
            };
            bf16x8 wn[3];
            auto read_w = [&](int st, int ct) {
                const __bf16 *bp = w1b + (size_t)(32 * ct + i) * KPB + 16 * st + 4 * half;
                wn[0] = fb_cat(*(const bf16x4 *)bp, *(const bf16x4 *)(bp + 8));
                wn[1] = fb_cat(*(const bf16x4 *)(bp + (size_t)C * KPB), *(const bf16x4 *)(bp + (size_t)C * KPB + 8));
                wn[2] = fb_cat(*(const bf16x4 *)(bp + (size_t)2 * C * KPB), *(const bf16x4 *)(bp + (size_t)2 * C * KPB + 8));
            };
            load_step(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) soft_elem(0, e);
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) split_pr(pj);
            load_step(1);
            read_w(0, 0);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                const bf16x8 xa1 = fb_as8(xq[0]), xa2 = fb_as8(xq[1]), xa3 = fb_as8(xq[2]);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    const bf16x8 wc0 = wn[0], wc1 = wn[1], wc2 = wn[2];
                    const bool more = st < 5;
                    fb_mfma6_il<FB_IL_S2 != 0>(y[ct], xa1, xa2, xa3, wc0, wc1, wc2, [&](int k) {
                        if (k == 0) { if (ct < 2) read_w(st, ct + 1); else if (more) read_w(st + 1, 0); }
                        if (!more) return;
                        if (ct == 0) { if (k == 1) soft_elem(st + 1, 0); if (k == 2) soft_elem(st + 1, 1); if (k == 3) split_pr(0);
                                       if (k == 4) soft_elem(st + 1, 2); if (k == 5) soft_elem(st + 1, 3); }
                        if (ct == 1) { if (k == 1) split_pr(1); if (k == 2) soft_elem(st + 1, 4); if (k == 3) soft_elem(st + 1, 5);
                                       if (k == 4) split_pr(2); if (k == 5) soft_elem(st + 1, 6); }
                        if (ct == 2) { if (k == 1) soft_elem(st + 1, 7); if (k == 2) split_pr(3); if (k == 3 && st < 4) load_step(st + 2); }
                    });
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // z = Softplus(y + b1): lane (unit 32 ct + i, half), rows = voxels r(v, half); the bias is re-read per tile (L1)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const float bias = a.b1[ct * 32 + i];
#pragma unroll
            for (int v = 0; v < 16; ++v) y[ct][v] = fb_softplus(y[ct][v] + bias);
        }

        // ---- dW2'[o'][n] += dOut^T z : A = dOut (lane o'), B = z registers, K = the tile's 32 voxels ----------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float gs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) gs[j] = dOC[8 * s + j];
            bf16x8 g1, g2, g3;
            fb_split3(gs, g1, g2, g3);
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                float zs[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) zs[j] = y[ct][8 * s + j];
                bf16x8 z1, z2, z3;
                fb_split3(zs, z1, z2, z3);
                fb_mfma6(dW2[ct], g1, g2, g3, z1, z2, z3);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // sigmoid(y) = 1 - exp(-Softplus(y)), in place
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) y[ct][v] = 1.0f - fb_exp_neg(y[ct][v]);

        // ---- S3: dZ = dOut W2 ; dY = dZ sigmoid(y) ------------------------------------------------------------------
        f32x16 dY[3];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) dY[ct][v] = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 g1, g2, g3;
            fb_split3(dOA[s], g1, g2, g3);
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                const __bf16 *bp = w2t + (size_t)(32 * ct + i) * OPB + 16 * s + 8 * half;
                const bf16x8 w1_ = *(const bf16x8 *)bp, w2_ = *(const bf16x8 *)(bp + (size_t)C * OPB),
                             w3_ = *(const bf16x8 *)(bp + (size_t)2 * C * OPB);
                fb_mfma6(dY[ct], g1, g2, g3, w1_, w2_, w3_);
            }
        }
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            float sum = 0.0f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                dY[ct][v] *= y[ct][v];
                sum += dY[ct][v];
            }
            db1[ct] += sum;
        }

        __builtin_amdgcn_sched_barrier(0);
        // ---- a, lane (channel 32 ct + i, half): rows r(v, half) from TA (B operand of dW1, and sigmoid(x) of dX) ------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float aC[3][16];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 t = *(const float4 *)(T + (size_t)(32 * ct + i) * kB3_TA + 8 * g4 + 4 * half);
                aC[ct][4 * g4 + 0] = t.x; aC[ct][4 * g4 + 1] = t.y; aC[ct][4 * g4 + 2] = t.z; aC[ct][4 * g4 + 3] = t.w;
            }
        }
        // ---- dW1[n][k] += dY^T a : A = dY registers (lane = unit n), B = a registers (lane = input k) ----------------
        // groups (s, rt, ct) of six MFMAs.  Per s the three a operands are split up front (kept: 36 registers); the dY operand of
        // the NEXT (s, rt) is split in the gaps of the current rt's first group.
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            u32x4 aq[3][3], yq[2][3];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int pj = 0; pj < 4; ++pj) {
                    unsigned q1, q2, q3;
                    fb_split_pair(aC[ct][8 * s_ + 2 * pj], aC[ct][8 * s_ + 2 * pj + 1], q1, q2, q3);
                    aq[ct][0][pj] = q1; aq[ct][1][pj] = q2; aq[ct][2][pj] = q3;
                }
            auto split_y = [&](int buf, int rt, int pj) {
                unsigned q1, q2, q3;
                fb_split_pair(dY[rt][8 * s_ + 2 * pj], dY[rt][8 * s_ + 2 * pj + 1], q1, q2, q3);
                yq[buf][0][pj] = q1; yq[buf][1][pj] = q2; yq[buf][2][pj] = q3;
            };
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) split_y(0, 0, pj);
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                const int cur = rt & 1;
                const bf16x8 y1 = fb_as8(yq[cur][0]), y2 = fb_as8(yq[cur][1]), y3 = fb_as8(yq[cur][2]);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
                    fb_mfma6_il<FB_IL_DW1 != 0>(dW1[rt][ct], y1, y2, y3, fb_as8(aq[ct][0]), fb_as8(aq[ct][1]), fb_as8(aq[ct][2]), [&](int k) {
                        if (ct == 0 && rt < 2 && k < 4) split_y(cur ^ 1, rt + 1, k);
                    });
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- dY -> TY [voxel][unit] (the tile buffer is free: every lane has read its a) ------------------------------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) T[(size_t)fb_crow(v, half) * kB3_TY + 32 * ct + i] = dY[ct][v];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- S6: dA = dY W1 : A = dY (lane-per-voxel, from TY), B = W1 columns via the transpose read ------------------
        f32x16 dA[3];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int v = 0; v < 16; ++v) dA[ct][v] = 0.0f;
        // lane s of a 16-lane group supplies &W1[u0 + (s >> 2)][c0 + 4 (s & 3)] and receives W1[u0 .. u0 + 3][c0 + s]
        {
            const __bf16 *trb = w1b + (size_t)(4 * half + ((lane & 15) >> 2)) * KPB + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const float *yrow = T + (size_t)i * kB3_TY + 4 * half;
            bf16x8 wn[3];
            auto read_w = [&](int st, int ct) {
                const __bf16 *bp = trb + (size_t)(16 * st) * KPB + 32 * ct;
                wn[0] = fb_cat(fb_tr4(bp), fb_tr4(bp + (size_t)8 * KPB));
                wn[1] = fb_cat(fb_tr4(bp + (size_t)C * KPB), fb_tr4(bp + (size_t)C * KPB + (size_t)8 * KPB));
                wn[2] = fb_cat(fb_tr4(bp + (size_t)2 * C * KPB), fb_tr4(bp + (size_t)2 * C * KPB + (size_t)8 * KPB));
            };
            float4 ylo, yhi;
            auto read_y = [&](int st) { ylo = *(const float4 *)(yrow + 16 * st); yhi = *(const float4 *)(yrow + 16 * st + 8); };
            u32x4 yq[3];
            auto split_pr = [&](int pj) {
                const float v0 = pj == 0 ? ylo.x : pj == 1 ? ylo.z : pj == 2 ? yhi.x : yhi.z;
                const float v1 = pj == 0 ? ylo.y : pj == 1 ? ylo.w : pj == 2 ? yhi.y : yhi.w;
                unsigned q1, q2, q3;
                fb_split_pair(v0, v1, q1, q2, q3);
                yq[0][pj] = q1; yq[1][pj] = q2; yq[2][pj] = q3;
            };
            read_y(0);
            read_w(0, 0);
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) split_pr(pj);
            read_y(1);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                const bf16x8 y1 = fb_as8(yq[0]), y2 = fb_as8(yq[1]), y3 = fb_as8(yq[2]);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    const bf16x8 wc0 = wn[0], wc1 = wn[1], wc2 = wn[2];
                    fb_mfma6_il<FB_IL_S6 != 0>(dA[ct], y1, y2, y3, wc0, wc1, wc2, [&](int k) {
                        if (k == 0) { if (ct < 2) read_w(st, ct + 1); else if (st < 5) read_w(st + 1, 0); }
                        if (st < 5 && ct == 1 && k >= 1 && k <= 4) split_pr(k - 1);       // the next step's operand
                        if (st < 4 && ct == 2 && k == 1) read_y(st + 2);
                    });
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- dX = dA sigmoid(x) (lane = channel) -> plane gradients ----------------------------------------------------
        // register v of lane (i, half) = patch voxel  dd = v & 1,  ww = 2 half + ((v >> 1) & 1),  hh = v >> 2
        float *gwz = a.g_wz + ((size_t)(w_b + 2 * half) * a.D + d_b) * C + i;
        float *gzh = a.g_zh + ((size_t)(d_b + half) * a.H + h_b) * C + i;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            float dx[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const bool live = FULL ? true : ((h_b + (v >> 2) < a.H) & (w_b + 2 * half + ((v >> 1) & 1) < a.W) & (d_b + (v & 1) < a.D));
                dx[v] = live ? dA[ct][v] * (1.0f - fb_exp_neg(aC[ct][v])) : 0.0f;
            }
            // g_hw[h][w] += sum over d: registers v, v ^ 1 (carried along the d-walk in the lane's LDS slots)
            {
                float4 lo = ((const float4 *)hwa)[2 * ct], hi = ((const float4 *)hwa)[2 * ct + 1];
                lo.x += dx[0] + dx[1]; lo.y += dx[2] + dx[3]; lo.z += dx[4] + dx[5]; lo.w += dx[6] + dx[7];
                hi.x += dx[8] + dx[9]; hi.y += dx[10] + dx[11]; hi.z += dx[12] + dx[13]; hi.w += dx[14] + dx[15];
                ((float4 *)hwa)[2 * ct] = lo;
                ((float4 *)hwa)[2 * ct + 1] = hi;
            }
            // g_wz[w][d] += sum over h: registers v, v + 4, v + 8, v + 12
            if (!(a.dbg & 1))
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = FULL ? true : ((w_b + 2 * half + (q >> 1) < a.W) & (d_b + (q & 1) < a.D));
                if (ok) unsafeAtomicAdd(gwz + ((q >> 1) * a.D + (q & 1)) * C + 32 * ct, (dx[q] + dx[q + 4]) + (dx[q + 8] + dx[q + 12]));
            }
            // g_zh[d][h] += sum over w: registers v, v ^ 2 and the partner half; half 0 issues d = d_b, half 1 d = d_b + 1
            if (!(a.dbg & 1))
#pragma unroll
            for (int hq = 0; hq < 4; ++hq) {
                const float s0 = dx[4 * hq] + dx[4 * hq + 2], s1 = dx[4 * hq + 1] + dx[4 * hq + 3];
                const float t0 = s0 + __shfl_xor(s0, 32, 64), t1 = s1 + __shfl_xor(s1, 32, 64);
                const bool ok = FULL ? true : ((h_b + hq < a.H) & (d_b + half < a.D));
                if (ok) unsafeAtomicAdd(gzh + hq * C + 32 * ct, half ? t1 : t0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();      // the next tile's S1 overwrites the buffer
    };

    for (int tile = t_lo; tile < t_hi; ++tile) {
        if (tile / a.PD != col_prev) {
            if (col_prev >= 0) flush_hw(col_prev);
            col_prev = tile / a.PD;
        }
        const int pd = tile % a.PD, tq = tile / a.PD;
        const int pw = tq % a.PW, ph = tq / a.PW;
        const int h_b = 4 * ph, w_b = 4 * pw, d_b = 2 * pd;
        if (h_b + 4 <= a.H && w_b + 4 <= a.W && d_b + 2 <= a.D) tile_body(std::true_type{}, h_b, w_b, d_b);     // wave-uniform
        else tile_body(std::false_type{}, h_b, w_b, d_b);
    }
    if (col_prev >= 0) flush_hw(col_prev);

    // ---- the wave's weight / bias gradients -> global (once) -------------------------------------------------------
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int cr = fb_crow(v, half);
        const int w2row = fb_w2_row(cr, a.out_dim);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)            // dW1[n = rt * 32 + cr][k = ct * 32 + i] -> g_w1 (n, k): 128-byte rows
                unsafeAtomicAdd(a.g_w1 + (size_t)(rt * 32 + cr) * C + ct * 32 + i, dW1[rt][ct][v]);
            if (w2row >= 0 && (w2row > 0 ? a.g_feat != nullptr : a.g_sdf != nullptr))
                unsafeAtomicAdd(a.g_w2 + (size_t)w2row * C + ct * 32 + i, dW2[ct][v]);   // dW2'[o' = cr][n]
        }
    }
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const float tot = db1[ct] + __shfl_xor(db1[ct], 32, 64);
        if (half == 0) unsafeAtomicAdd(a.g_b1 + ct * 32 + i, tot);
    }
    {
        const float t = db2 + __shfl_xor(db2, 32, 64);
        const int row = fb_w2_row(i, a.out_dim);
        if (half == 0 && row >= 0) unsafeAtomicAdd(a.g_b2 + row, t);
    }
}

}  // namespace

// launched by selfocc_field_volume_bwd (field.hip)
int so_field_volume_bwd_b3(const float *hw, const float *zh, const float *wz, int H, int W, int D, const float *w1,
                           const float *b1, const float *w2, int out_dim, const float *g_sdf, const float *g_feat,
                           int feat_stride, float *g_hw, float *g_zh, float *g_wz, float *g_w1, float *g_b1, float *g_w2,
                           float *g_b2, hipStream_t st) {
    const long long PH = (H + 3) / 4, PW = (W + 3) / 4, PD = (D + 1) / 2;
    FieldBwdB3Args a{hw, zh, wz, H, W, D, w1, b1, w2, out_dim, g_sdf, g_feat, feat_stride,
                     g_hw, g_zh, g_wz, g_w1, g_b1, g_w2, g_b2, (int)(PH * PW * PD), (int)PW, (int)PD,
                     getenv("SELFOCC_FIELD_BWD_DBG") ? atoi(getenv("SELFOCC_FIELD_BWD_DBG")) : 0};
    // the attribute is per device: set once per device, and a failure (LDS carve-out refused) is an error, not a silent launch failure
    static std::atomic<unsigned long long> done_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void *)field_volume_bwd_b3_kernel,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        SO_REQUIRE(e == hipSuccess, "field_volume_bwd: cannot reserve %d bytes of LDS (%s)", 160 * 1024 - 256, hipGetErrorString(e));
        done_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    const int blocks = std::min((a.n_tiles + kB3_WAVES - 1) / kB3_WAVES, 256);
    hipLaunchKernelGGL(field_volume_bwd_b3_kernel, dim3(blocks), dim3(kB3_WAVES * 64), kB3_LDS, st, a);
    return so_launch_status();
}
