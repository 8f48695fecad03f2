// linear_fwd.hip — forward of the encoder's tall-skinny projections, with the elementwise work that follows them
// in TPVFormerLayer / BEVFormerLayer folded into the epilogue:
//     Y = X W^T + b   [ReLU]   [+ residual]   [-> LayerNorm over the N outputs]
// (mmcv Linear / FFN / build_norm_layer under torch in the reference: sampling_offsets, attention_weights,
// value_proj, output_proj of model/encoder/bevformer/attention/image_cross_attention.py:149-351 and
// tpvformer/attention/cross_view_hybrid_attention.py:35-60, the FFN and `norm` steps of
// tpvformer/tpvformer_encoder_layer.py:150-219).
//
// Shapes: 6 k - 180 k rows, K = 96 (192) inputs, N = 96 .. 2304 outputs, float32.  The vendor GEMMs picked for
// these shapes run at 35 - 50 TFLOP/s and ~1 TB/s of output (profiles/r2_c_eval_kernel_trace.txt: 3.7 ms of the
// 8.9 ms eval encoder), then a residual add and a LayerNorm stream the same rows twice more.  The work is bound by
// the OUTPUT write (N >= K) and by the f32 MFMA rate at the same time (66 049 x 384 x 96: 101 MB out = 25 us at
// 4 TB/s, 4.9 GFLOP = 31 us at 155 TFLOP/s), so the kernel keeps both busy and does nothing else:
//   * a block owns a 96-column slice of W in LDS (row stride K + 4 floats: conflict-free ds_read_b128) and walks
//     over 128-row tiles; a wave owns 32 rows x 96 columns = three v_mfma_f32_32x32x2_f32 accumulators;
//   * the A operand (32 rows x K) is loaded ONCE per tile straight into MFMA layout: the reduction index is
//     permuted so that lane (row i, half h) holds x[row][h K/2 .. (h+1) K/2) — K/8 float4 loads per lane — and the
//     B operand read from LDS uses the same permutation (W[n][h K/2 + j]);
//   * the epilogue works on the accumulator layout (a lane holds one column of 16 rows; the 32 lanes of a half wave
//     hold 32 consecutive columns of a row): bias / ReLU / residual per element, LayerNorm by two 5-step shuffle
//     reductions per row, every store a 128-byte row segment.
// f32 MFMA on gfx950 is an exact fmaf chain: float32 arithmetic, only the summation order differs from a BLAS.
#include "so_device.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LinearFwdArgs {
    const float *x, *w, *bias, *residual, *gamma, *beta;
    float *y, *y_pre, *mean, *rstd;
    long long T;
    int N, K, ldy, ldr;
    int relu, ncb, groups, col0;
    float eps;
};

SO_DEVFN unsigned so_lin_xcd_block() {   // workgroup b runs on XCD b % 8: give each XCD a contiguous range
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned x = b & 7u, k = b >> 3;
    const unsigned q = nb >> 3, r = nb & 7u;
    return x * q + (x < r ? x : r) + k;
}

// Epilogue of one wave's 32-row x (32 NT)-column tile.  FULL: all 32 rows and every column of the NT tiles exist
// — no per-element predicates (the predicated form compiles to a branch per load / store and keeps every value alive:
// 230 registers); the ragged edge blocks take the predicated instantiation.
template <bool LN, bool FULL, int NT>
SO_DEVFN void so_linear_epilogue(f32x16 (&acc)[NT], const float (&bv)[NT], const float (&gv)[NT], const float (&bt)[NT],
                                 const bool (&cok)[NT], float relu_lo, const float *rb, int ldr, float *yb, int ldy,
                                 float *pb, float *mb, float *sb, int N, float eps, int rem, int i, int half) {
    const float inv_n = 1.0f / (float)N;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int rl = (v & 3) + 8 * (v >> 2) + 4 * half;
        const bool rok = FULL || rl < rem;
        float o[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) o[t] = fmaxf(acc[t][v] + bv[t], relu_lo);
        if (rb) {
            const unsigned ro = (unsigned)(rl * ldr + i);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (FULL) o[t] += rb[ro + 32 * t];
                else o[t] += (rok && cok[t]) ? rb[ro + 32 * t] : 0.0f;
            }
        }
        const unsigned yo = (unsigned)(rl * ldy + i);
        if (LN) {
            if (pb) {
                const unsigned po = (unsigned)(rl * N + i);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (FULL || (rok && cok[t])) pb[po + 32 * t] = o[t];
            }
            float s = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) s += cok[t] ? o[t] : 0.0f;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) s += __shfl_xor(s, m, 64);
            const float mean = s * inv_n;
            float d[NT], q2 = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                d[t] = cok[t] ? o[t] - mean : 0.0f;
                q2 += d[t] * d[t];
            }
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) q2 += __shfl_xor(q2, m, 64);
            const float rstd = 1.0f / sqrtf(q2 * inv_n + eps);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (FULL || (rok && cok[t])) yb[yo + 32 * t] = fmaf(d[t] * rstd, gv[t], bt[t]);
            if (mb && i == 0 && rok) { mb[rl] = mean; sb[rl] = rstd; }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (FULL || (rok && cok[t])) yb[yo + 32 * t] = o[t];
        }
    }
}

template <int KH, bool LN, int NT>
__global__ __launch_bounds__(256, ((KH <= 64 && !LN) ? 3 : 2)) void linear_fwd_kernel(LinearFwdArgs a) {
    constexpr int K = 2 * KH, KP = K + 4;
    extern __shared__ __attribute__((aligned(16))) float wl[];    // [96][KP]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, half = lane >> 5;
    const unsigned logical = so_lin_xcd_block();
    // (the division runs on the vector ALU; readfirstlane brings the results back to scalar registers so that every
    // base pointer below is scalar and the loads / stores take the saddr + 32-bit lane offset form)
    const int cb = __builtin_amdgcn_readfirstlane((int)(logical % (unsigned)a.ncb));
    const long long rc = __builtin_amdgcn_readfirstlane((int)(logical / (unsigned)a.ncb));
    const int n0 = a.col0 + cb * 96;

    {   // stage the block's slice of W: all of a thread's loads in flight, then the LDS writes (a dependent
        // load -> write chain per float4 cost ~5 us per block)
        constexpr int NV = NT * 32 * (K / 4), PER = (NV + 255) / 256;
        float4 wv[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = threadIdx.x + 256 * j;
            const int n = idx / (K / 4), k4 = idx - n * (K / 4);
            wv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NV && n0 + n < a.N) wv[j] = ((const float4 *)(a.w + (size_t)(n0 + n) * K))[k4];
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = threadIdx.x + 256 * j;
            const int n = idx / (K / 4), k4 = idx - n * (K / 4);
            if (idx < NV) *(float4 *)(wl + n * KP + 4 * k4) = wv[j];
        }
    }
    __syncthreads();

    const bool full_cols = a.N - n0 >= 32 * NT;          // every column of the block's NT tiles exists
    const float relu_lo = a.relu ? 0.0f : -__builtin_huge_valf();
    float bv[NT], gv[NT], bt[NT];
    bool cok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = n0 + 32 * t + i;
        cok[t] = col < a.N;
        bv[t] = (a.bias && cok[t]) ? a.bias[col] : 0.0f;
        gv[t] = (LN && cok[t]) ? a.gamma[col] : 0.0f;
        bt[t] = (LN && cok[t]) ? a.beta[col] : 0.0f;
    }

    // persistent: block (column block cb, row group rc) walks over the 32-row wave tiles rc * 4 + wave + 4 G j.
    // Everything is addressed as (wave-uniform 64-bit base) + (32-bit lane offset): saddr-form loads / stores, no
    // 64-bit address per element in registers.
    const long long nwt = (a.T + 31) / 32, wt_step = 4LL * a.groups;
    long long wt = rc * 4 + wave;
    float av[KH];
    auto load_a = [&](long long wtile) {
        const long long r0 = wtile * 32;
        const int rm = (int)min(32LL, a.T - r0);
        const float *xb = a.x + r0 * K;
        const unsigned xoff = (unsigned)(min(i, rm - 1) * K + half * KH);
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            const float4 v = *(const float4 *)(xb + xoff + 4 * q);
            av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
        }
    };
    if (wt < nwt) load_a(wt);
    for (; wt < nwt; wt += wt_step) {
        const long long row0 = wt * 32;
        const int rem = (int)min(32LL, a.T - row0);                   // live rows of this wave's tile
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float *bp = wl + (32 * t + i) * KP + half * KH;
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) {
                const float4 b = *(const float4 *)(bp + 4 * q);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * q], b.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * q + 1], b.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * q + 2], b.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * q + 3], b.w, acc[t], 0, 0, 0);
            }
        }
        // the A registers are dead now: the next tile's loads fly while this tile's epilogue runs
        if (wt + wt_step < nwt) load_a(wt + wt_step);
        // epilogue on the accumulator layout: element v of tile t = row (v & 3) + 8 (v >> 2) + 4 half, column 32 t + i
        const float *rb = a.residual ? a.residual + row0 * a.ldr + n0 : nullptr;
        float *yb = a.y + row0 * a.ldy + n0;
        float *pb = (LN && a.y_pre) ? a.y_pre + row0 * a.N : nullptr;
        float *mb = (LN && a.mean) ? a.mean + row0 : nullptr, *sb = (LN && a.mean) ? a.rstd + row0 : nullptr;
        // the per-row offsets of the epilogue are loop-invariant; hoisted out of the tile loop they would occupy
        // 30 - 60 registers across the MFMA phase — an opaque copy of the lane index keeps them inside
        int io = i;
        asm volatile("" : "+v"(io));
        if (rem == 32 && full_cols)
            so_linear_epilogue<LN, true, NT>(acc, bv, gv, bt, cok, relu_lo, rb, a.ldr, yb, a.ldy, pb, mb, sb, a.N, a.eps, rem,
                                             io, half);
        else
            so_linear_epilogue<LN, false, NT>(acc, bv, gv, bt, cok, relu_lo, rb, a.ldr, yb, a.ldy, pb, mb, sb, a.N, a.eps, rem,
                                              io, half);
    }
}

bool so_linear_fwd_ok(long long T, int N, int K) {
    if (!(K == 32 || K == 64 || K == 96 || K == 128 || K == 192)) return false;
    return T >= 1 && N >= 1 && (long long)N * K < (1LL << 30) && T < (1LL << 40);
}

}  // namespace

extern "C" int selfocc_linear_fwd_supported(int64_t T, int32_t N, int32_t K) { return so_linear_fwd_ok(T, N, K) ? 1 : 0; }

extern "C" int selfocc_linear_fwd(const float *x, const float *w, const float *bias, const float *residual, int32_t ldr,
                                  const float *ln_gamma, const float *ln_beta, float ln_eps, float *y, int32_t ldy,
                                  float *y_pre, float *mean, float *rstd, int64_t T, int32_t N, int32_t K,
                                  uint32_t flags, void *stream) {
    SO_REQUIRE(so_linear_fwd_ok(T, N, K), "linear_fwd: unsupported shape (T = %lld rows, N = %d, K = %d; K must be 32, 64, 96, "
               "128 or 192)", (long long)T, N, K);
    SO_REQUIRE(x && w && y, "linear_fwd: NULL pointer");
    SO_REQUIRE(ldy >= N && (!residual || ldr >= N), "linear_fwd: ldy / ldr smaller than N");
    const bool ln = ln_gamma != nullptr;
    SO_REQUIRE(!ln || (ln_beta && N <= 96), "linear_fwd: the LayerNorm epilogue needs beta and N <= 96 (one column block)");
    SO_REQUIRE(ln || (!y_pre && !mean && !rstd), "linear_fwd: y_pre / mean / rstd are outputs of the LayerNorm epilogue");
    SO_REQUIRE((mean == nullptr) == (rstd == nullptr), "linear_fwd: mean and rstd come together");
    LinearFwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.residual = residual; a.gamma = ln_gamma; a.beta = ln_beta;
    a.y = y; a.y_pre = y_pre; a.mean = mean; a.rstd = rstd;
    a.T = T; a.N = N; a.K = K; a.ldy = ldy; a.ldr = ldr;
    a.relu = (flags & SO_LINEAR_RELU) ? 1 : 0;
    a.eps = ln_eps;
    hipStream_t st = (hipStream_t)stream;
    // main launch: the full 96-column blocks (three accumulator tiles per wave); tail launch: the last 1 - 64 columns with
    // one or two tiles (compile-time, so that no accumulator tile sits behind a branch)
    const int ncb_full = N / 96, tail_cols = N - 96 * ncb_full;
    const int tail_nt = (tail_cols + 31) / 32;
    for (int pass = 0; pass < 2; ++pass) {
        const int nt = pass == 0 ? 3 : tail_nt;
        a.ncb = pass == 0 ? ncb_full : 1;
        a.col0 = pass == 0 ? 0 : 96 * ncb_full;
        if ((pass == 0 && ncb_full == 0) || (pass == 1 && tail_nt == 0)) continue;
        // persistent grid: as many blocks as the chip holds at once (LDS: 160 KB per CU), split evenly over the column
        // blocks; each block keeps its slice of W and walks over its share of the 32-row wave tiles
        const size_t lds_blk = (size_t)nt * 32 * (K + 4) * sizeof(float);
        // blocks per CU: registers allow 3 waves per SIMD (2 with the LayerNorm epilogue or K = 192), LDS 160 KB
        const long long by_regs = (K <= 128 && !ln) ? 3 : 2;
        const long long per_cu = std::max<long long>(1, std::min<long long>(by_regs, (160 * 1024) / (long long)(lds_blk + 512)));
        static const long long slots_env = getenv("SELFOCC_LINEAR_SLOTS") ? atoll(getenv("SELFOCC_LINEAR_SLOTS")) : 0;   // dev A/B
        const long long slots = slots_env > 0 ? slots_env : 256 * per_cu;
        const long long nwt = (T + 31) / 32;
        long long groups = std::max(1LL, slots / a.ncb);
        groups = std::min(groups, (nwt + 3) / 4);
        a.groups = (int)groups;
        const long long nblk = groups * a.ncb;
        const size_t lds = (size_t)nt * 32 * (K + 4) * sizeof(float);
#define SO_LAUNCH1(KH_, LN_, NT_)                                                                                    \
    do {                                                                                                             \
        static bool attr_set = false;                                                                                \
        if (!attr_set && lds > 48 * 1024) {                                                                          \
            (void)hipFuncSetAttribute((const void *)linear_fwd_kernel<KH_, LN_, NT_>,                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((linear_fwd_kernel<KH_, LN_, NT_>), dim3((unsigned)nblk), dim3(256), lds, st, a);         \
    } while (0)
#define SO_LAUNCH2(KH_, LN_)                                                                                         \
    do {                                                                                                             \
        if (nt == 3) SO_LAUNCH1(KH_, LN_, 3);                                                                        \
        else if (nt == 2) SO_LAUNCH1(KH_, LN_, 2);                                                                   \
        else SO_LAUNCH1(KH_, LN_, 1);                                                                                \
    } while (0)
#define SO_LAUNCH(KH_)                                                                                               \
    do {                                                                                                             \
        if (ln) SO_LAUNCH2(KH_, true);                                                                               \
        else SO_LAUNCH2(KH_, false);                                                                                 \
    } while (0)
        switch (K) {
            case 32: SO_LAUNCH(16); break;
            case 64: SO_LAUNCH(32); break;
            case 96: SO_LAUNCH(48); break;
            case 128: SO_LAUNCH(64); break;
            default: SO_LAUNCH(96); break;
        }
#undef SO_LAUNCH
#undef SO_LAUNCH2
#undef SO_LAUNCH1
    }
    return so_launch_status();
}
