// linear_fwd.hip — forward of the encoder's tall-skinny projections, with the elementwise work that follows them
// in TPVFormerLayer / BEVFormerLayer folded into the epilogue:
//     Y = X W^T + b   [ReLU]   [+ residual]   [-> LayerNorm over the N outputs]
// (mmcv Linear / FFN / build_norm_layer under torch in the reference: sampling_offsets, attention_weights,
// value_proj, output_proj of model/encoder/bevformer/attention/image_cross_attention.py:149-351 and
// tpvformer/attention/cross_view_hybrid_attention.py:35-60, the FFN and `norm` steps of
// tpvformer/tpvformer_encoder_layer.py:150-219).
//
// Shapes: 6 k - 180 k rows, K = 96 (192) inputs, N = 96 .. 2304 outputs, float32.  The vendor GEMMs picked for
// these shapes run at 35 - 65 TFLOP/s (profiles/r2_c_eval_kernel_trace.txt: 3.7 ms of the 8.9 ms eval encoder), then a
// residual add and a LayerNorm stream the same rows twice more.  The work is bound by the OUTPUT write (N >= K) and by
// the f32 MFMA rate at the same time (66 049 x 384 x 96: 101 MB out = 25 us at 4 TB/s, 4.9 GFLOP = 31 us at the
// 155 TFLOP/s of f32 MFMA — 38 us at the ~1.95 GHz the chip sustains under dense MFMA), so the kernel does nothing else:
//   * a persistent block owns a 96-column slice of W in LDS (row stride K + 4 floats: conflict-free ds_read_b128,
//     staged with all loads in flight) and its waves walk over 16-row tiles of x;
//   * the A operand (16 rows x K) is loaded ONCE per tile straight into MFMA layout: the reduction index is
//     permuted so that lane (row m, quarter kq) holds x[row][16 q + 4 kq + j] (q < K/16, j < 4) — K/16 float4 loads per
//     lane, the four kq lanes of a row reading 64 contiguous bytes — and the B operand read from LDS uses the same
//     permutation (W[n][16 q + 4 kq + j]);
//   * the epilogue works on the accumulator layout (a lane holds one column of 4 rows; 16 lanes hold 16 consecutive
//     columns of a row): bias / ReLU / residual per element, LayerNorm by two 4-step shuffle reductions per row.
// f32 MFMA on gfx950 is an exact fmaf chain: float32 arithmetic, only the summation order differs from a BLAS.
// Measured (scripts/bench_linear.py, one layer's twelve projections): hipBLASLt 742 us -> 543 us; output_proj +
// residual + LayerNorm of a 78 899 x 96 plane tensor 124 us (three torch kernels) -> 34 us.  What was tried on the way
// (round-2 runs of scripts/bench_linear.py): 32-row tiles on v_mfma_f32_32x32x2_f32 with three 16-register accumulators
// (118 - 164 VGPRs, 2 - 3 waves per SIMD, half as many work units: 576 - 672 us); more resident waves (4 per SIMD:
// 612 us, 8-wave blocks 664 us) — fewer, longer-lived waves win because every extra resident block re-stages its
// slice of W and scatters the output stream over more DRAM pages at once.  Then: the next tile's x operand loaded
// before the current tile's MFMAs and the W operand double-buffered in registers across MFMA groups (543 -> 483 us =
// 85 TFLOP/s over the twelve shapes; PMC: the MFMA pipes are busy 64 % of the kernel on the 178 500 x 288 shape).  Tried
// without gain: W as the MFMA's A operand so that a lane owns four consecutive columns of a row and the epilogue is
// float4 loads / stores (512 us; the SGPR budget overflows), a contiguous range of row tiles per block (504 us), the
// epilogue of tile i cut into slices between the MFMA groups of tile i + 1 (507 us: anything between two MFMAs costs
// more than its issue slot); the first x tile loaded before the W staging, and the k permutation that makes the x
// loads touch 16 half lines per instruction instead of 48 lines (both kept, neither measurable: 491 / 492 us).
// The HBM side is far from its limit (torch fill_ writes the same 100 MB in 16 us = 6 TB/s).
#include "so_device.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>

#ifdef SO_LIN_TRACE
// dev build only (scripts/build_variant.sh lintrace linear_fwd.hip -DSO_LIN_TRACE): per-wave s_memtime stamps of
// linear_fwd_b3_kernel — [wave][0] = start, [1] = W staged, then per tile: top, x arrived, MFMAs done, stores issued, stores acknowledged
__device__ unsigned long long so_lin_trace[4096 * 64];
extern "C" int selfocc_diag_lin_trace(unsigned long long *dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(so_lin_trace), n * sizeof(unsigned long long));
}
#define SO_TR(slot) do { if (lane == 0 && (slot) < 64) so_lin_trace[(size_t)(blockIdx.x * WAVES + wave) * 64 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SO_TR(slot) do { } while (0)
#endif

namespace {

struct LinearFwdArgs {
    const float *x, *w, *bias, *residual, *gamma, *beta;
    float *y, *y_pre, *mean, *rstd;
    long long T;
    int N, K, ldy, ldr;
    int relu, ncb, groups, col0;
    int vec;                 // b3 kernel: x / y / residual / y_pre rows start 16-byte aligned (float4 epilogue)
    int hm_nv;               // > 0: head-major output (selfocc_linear_fwd_heads): rows per batch item
    unsigned hm_sg;          // floats per 96-column group of the head-major output: (T / nv) * 6 * nv * 16
    float eps;
};

SO_DEVFN unsigned so_lin_xcd_block() {   // workgroup b runs on XCD b % 8: give each XCD a contiguous range
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned x = b & 7u, k = b >> 3;
    const unsigned q = nb >> 3, r = nb & 7u;
    return x * q + (x < r ? x : r) + k;
}

// 16-row wave tiles on v_mfma_f32_16x16x4_f32 (same f32 rate as 32x32x2, half the registers per wave: A operand K / 4,
// six 4-register accumulators for 96 columns; twice as many work units to balance over the 1 024 SIMDs).
//   A[m][k]: lane (m = l % 16, kq = l / 16) holds x[row m][16 q + 4 kq + j], q < K / 16, j < 4   (K / 16 float4 loads per lane)
//   B[k][n]: lane (n = l % 16, kq) reads W[n0 + 16 t + n][16 q + 4 kq + j] from LDS (row stride K + 4: conflict-free b128)
//   D[m][n]: lane holds column n = l % 16 of rows 4 kq + j, j = 0..3
// Two column tiles are interleaved so that no MFMA waits for its own accumulator (40 cycles dependent latency vs 32 issue).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LN, bool FULL, int NT16>
SO_DEVFN void so_linear_epilogue16(f32x4 (&acc)[NT16], const float (&bv)[NT16], const float (&gv)[NT16],
                                   const float (&bt)[NT16], const bool (&cok)[NT16], float relu_lo, const float *rb, int ldr,
                                   float *yb, int ldy, float *pb, float *mb, float *sb, int N, float eps, int rem, int n,
                                   int kq) {
    const float inv_n = 1.0f / (float)N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = 4 * kq + j;
        const bool rok = FULL || rl < rem;
        float o[NT16];
#pragma unroll
        for (int t = 0; t < NT16; ++t) o[t] = fmaxf(acc[t][j] + bv[t], relu_lo);
        if (rb) {
            const unsigned ro = (unsigned)(rl * ldr + n);
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                if (FULL) o[t] += rb[ro + 16 * t];
                else o[t] += (rok && cok[t]) ? rb[ro + 16 * t] : 0.0f;
            }
        }
        const unsigned yo = (unsigned)(rl * ldy + n);
        if (LN) {
            if (pb) {
                const unsigned po = (unsigned)(rl * N + n);
#pragma unroll
                for (int t = 0; t < NT16; ++t)
                    if (FULL || (rok && cok[t])) pb[po + 16 * t] = o[t];
            }
            float s = 0.0f;
#pragma unroll
            for (int t = 0; t < NT16; ++t) s += cok[t] ? o[t] : 0.0f;
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
            const float mean = s * inv_n;
            float d[NT16], q2 = 0.0f;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                d[t] = cok[t] ? o[t] - mean : 0.0f;
                q2 += d[t] * d[t];
            }
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) q2 += __shfl_xor(q2, m, 64);
            const float rstd = 1.0f / sqrtf(q2 * inv_n + eps);
#pragma unroll
            for (int t = 0; t < NT16; ++t)
                if (FULL || (rok && cok[t])) yb[yo + 16 * t] = fmaf(d[t] * rstd, gv[t], bt[t]);
            if (mb && n == 0 && rok) { mb[rl] = mean; sb[rl] = rstd; }
        } else {
#pragma unroll
            for (int t = 0; t < NT16; ++t)
                if (FULL || (rok && cok[t])) yb[yo + 16 * t] = o[t];
        }
    }
}

// Head-major epilogue (selfocc_linear_fwd_heads): the 96 columns of a block are 6 heads x 16 channels of one deformable
// attention's `value`, and a 16-column accumulator tile IS one head — the tile's 64-byte row segments go to
// y[group][b][head][pix][0..16) instead of y[row][16 head ..]: the (bs, heads, nv, d) layout the MSDA kernels gather
// fastest from (a cache line then holds x-neighbours of one head), written by the projection itself — no transposing
// copy.  Rows of a tile are consecutive, so (b, pix) advance without a division per lane (nv >= 16).
template <int NT16>
SO_DEVFN void so_linear_epilogue16_hm(f32x4 (&acc)[NT16], const float (&bv)[NT16], float relu_lo, float *yg, long long row0,
                                      int rem, int nv, int n, int kq) {
    const long long b0 = row0 / nv;
    const int pix0 = (int)(row0 - b0 * nv);
    const unsigned hs = (unsigned)nv * 16u;                       // floats per (b, head)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rl = 4 * kq + j;
        int pix = pix0 + rl;
        unsigned bo = (unsigned)b0 * 6u * hs;
        if (pix >= nv) { pix -= nv; bo += 6u * hs; }
        const unsigned o0 = bo + (unsigned)pix * 16u + (unsigned)n;
        if (rl < rem) {
#pragma unroll
            for (int t = 0; t < NT16; ++t) yg[o0 + (unsigned)t * hs] = fmaxf(acc[t][j] + bv[t], relu_lo);
        }
    }
}

template <int KQ /* K / 4 */, bool LN, int NT /* 32-column tiles per block */, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void linear_fwd16_kernel(LinearFwdArgs a) {
    constexpr int K = 4 * KQ, KP = K + 4, NT16 = 2 * NT, THREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float wl[];    // [32 NT][KP]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = lane & 15, kq = lane >> 4;
    const unsigned logical = so_lin_xcd_block();
    const int cb = __builtin_amdgcn_readfirstlane((int)(logical % (unsigned)a.ncb));
    const long long rc = __builtin_amdgcn_readfirstlane((int)(logical / (unsigned)a.ncb));
    const int n0 = a.col0 + cb * 96;
    const long long nwt = (a.T + 15) / 16, wt_step = (long long)WAVES * a.groups;
    // the A operand of the NEXT tile is loaded before the MFMAs of the current one (24 - 48 more registers; the kernel
    // runs 2 waves per SIMD by choice, so they are free): a wave never waits for HBM between its tiles
    float av[KQ], an[KQ];
    auto load_a = [&](long long wtile, float (&dst)[KQ]) {
        const long long r0 = wtile * 16;
        const int rm = (int)min(16LL, a.T - r0);
        const float *xb = a.x + r0 * K;
        // k permutation: load q of lane (row n, kq) covers k = 16 q + 4 kq .. + 3, so the four kq lanes of a row read 64
        // CONTIGUOUS bytes per instruction (a load instruction touches 16 half lines; with k = kq K/4 + 4 q + j it
        // touched 64 sixteen-byte pieces in 48 lines — three times the line accesses of the whole tile per pass)
        const unsigned xoff = (unsigned)(min(n, rm - 1) * K + 4 * kq);
#pragma unroll
        for (int q = 0; q < KQ / 4; ++q) {
            const float4 v = *(const float4 *)(xb + xoff + 16 * q);
            dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
        }
    };
    long long wt = rc * WAVES + wave;
    if (wt < nwt) load_a(wt, av);        // the first tile's x is in flight while the block stages its slice of W
    {
        constexpr int NV = NT * 32 * (K / 4), PER = (NV + THREADS - 1) / THREADS;
        float4 wv[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = threadIdx.x + THREADS * j;
            const int r = idx / (K / 4), k4 = idx - r * (K / 4);
            wv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NV && n0 + r < a.N) wv[j] = ((const float4 *)(a.w + (size_t)(n0 + r) * K))[k4];
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int idx = threadIdx.x + THREADS * j;
            const int r = idx / (K / 4), k4 = idx - r * (K / 4);
            if (idx < NV) *(float4 *)(wl + r * KP + 4 * k4) = wv[j];
        }
    }
    __syncthreads();

    const bool full_cols = a.N - n0 >= 32 * NT;
    const float relu_lo = a.relu ? 0.0f : -__builtin_huge_valf();
    float bv[NT16], gv[NT16], bt[NT16];
    bool cok[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        const int col = n0 + 16 * t + n;
        cok[t] = col < a.N;
        bv[t] = (a.bias && cok[t]) ? a.bias[col] : 0.0f;
        gv[t] = (LN && cok[t]) ? a.gamma[col] : 0.0f;
        bt[t] = (LN && cok[t]) ? a.beta[col] : 0.0f;
    }

    for (; wt < nwt; wt += wt_step) {
        const long long row0 = wt * 16;
        const int rem = (int)min(16LL, a.T - row0);
        const bool more = wt + wt_step < nwt;
        if (more) load_a(wt + wt_step, an);
        f32x4 acc[NT16];
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = 0.0f;
        // B operand double-buffered in registers: the ds_read_b128 pair of group s + 1 is issued before the eight MFMAs
        // of group s (with one or two waves per SIMD nothing else hides the LDS latency)
        constexpr int QN = KQ / 4, NS = NT * QN;
        const float *bbase = wl + n * KP + 4 * kq;      // the same k permutation as the x operand: k = 16 q + 4 kq + j
        float4 b0 = *(const float4 *)(bbase), b1 = *(const float4 *)(bbase + 16 * KP);
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int tp = sidx / QN, q = sidx - tp * QN;
            float4 nb0 = b0, nb1 = b1;
            if (sidx + 1 < NS) {
                const int tp2 = (sidx + 1) / QN, q2 = (sidx + 1) - tp2 * QN;
                nb0 = *(const float4 *)(bbase + 32 * tp2 * KP + 16 * q2);
                nb1 = *(const float4 *)(bbase + (32 * tp2 + 16) * KP + 16 * q2);
            }
            __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the reads next to their first use)
            acc[2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q], b0.x, acc[2 * tp], 0, 0, 0);
            acc[2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q], b1.x, acc[2 * tp + 1], 0, 0, 0);
            acc[2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 1], b0.y, acc[2 * tp], 0, 0, 0);
            acc[2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 1], b1.y, acc[2 * tp + 1], 0, 0, 0);
            acc[2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 2], b0.z, acc[2 * tp], 0, 0, 0);
            acc[2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 2], b1.z, acc[2 * tp + 1], 0, 0, 0);
            acc[2 * tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 3], b0.w, acc[2 * tp], 0, 0, 0);
            acc[2 * tp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 3], b1.w, acc[2 * tp + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            b0 = nb0; b1 = nb1;
        }
        const float *rb = a.residual ? a.residual + row0 * a.ldr + n0 : nullptr;
        float *yb = a.y + row0 * a.ldy + n0;
        float *pb = (LN && a.y_pre) ? a.y_pre + row0 * a.N : nullptr;
        float *mb = (LN && a.mean) ? a.mean + row0 : nullptr, *sb = (LN && a.mean) ? a.rstd + row0 : nullptr;
        int no = n;
        asm volatile("" : "+v"(no));      // keeps the loop-invariant row offsets of the epilogue inside the loop
        if (!LN && NT == 3 && a.hm_nv > 0) {
            if constexpr (!LN && NT == 3)
                so_linear_epilogue16_hm<NT16>(acc, bv, relu_lo, a.y + (size_t)cb * a.hm_sg, row0, rem, a.hm_nv, no, kq);
        } else if (rem == 16 && full_cols)
            so_linear_epilogue16<LN, true, NT16>(acc, bv, gv, bt, cok, relu_lo, rb, a.ldr, yb, a.ldy, pb, mb, sb, a.N, a.eps,
                                                 rem, no, kq);
        else
            so_linear_epilogue16<LN, false, NT16>(acc, bv, gv, bt, cok, relu_lo, rb, a.ldr, yb, a.ldy, pb, mb, sb, a.N, a.eps,
                                                  rem, no, kq);
        if (more) {
#pragma unroll
            for (int j = 0; j < KQ; ++j) av[j] = an[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same projection on the bf16 matrix pipe with an EXACT three-way split (round 3).  f32 MFMA on gfx950 runs at the
// vector rate (64 FLOP / clk / SIMD); v_mfma_f32_16x16x32_bf16 does 16 x that.  Every float32 is the exact sum of three
// bfloat16 (8 + 8 + 8 mantissa bits: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2), the residuals are exact in
// float32), so x w = sum_ij xi wj, and the six products with i + j <= 4 carry everything above 2^-24 |x w|: float32-level
// accuracy (the tests' 2e-6 against the float64 product) for 6 bf16 MFMAs of 16 cycles per 32 k instead of 8 f32 MFMAs
// of 32 cycles — 2.67 x fewer matrix-pipe cycles.  Products of bfloat16 pairs are exact in the float32 accumulator; the
// three small terms are accumulated first.
//   * W: split once per persistent block into three bf16 planes in LDS ([column][K + 8]: 16-byte reads, conflict-free);
//   * x: a wave owns 32 rows where column blocks share the rows (two 16-row MFMA tiles use every W fragment read from LDS
//     twice), 16 rows on the short N <= 192 launches; lane (m, kb) holds x[row m][32 ks + 8 kb ..+8]; the whole tile is
//     split into its planes first (the raw registers then take the NEXT tile's loads), the MFMAs follow as one phase with the
//     W fragments of the next (k step, column tile) read from LDS ahead of the current one's MFMAs;
//   * epilogues: the W fragment is the MFMA's A operand, so a lane ends up with four consecutive COLUMNS of a row
//     (so_linear_epilogue_t below): float4 stores / residual / bias reads, two shuffle steps per LayerNorm row.
// Where the time goes (round 6; scripts/diag/linear_fwd_trace.py on the -DSO_LIN_TRACE build, s_memtime per wave, 66 049 x 576):
// a wave needs ~14 k cycles per 32-row tile (x wait 2.7 k before the prefetch / 0.9 k with it, split + 216 MFMAs 5.6 - 6.7 k,
// epilogue 2.2 k, the rest contention with the SIMD's other wave); the 216 MFMAs are 3 456 cycles of the matrix pipe, so two
// waves per SIMD keep it 49 % busy while they live and 31 - 34 % over the launch (SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA
// exactly).  The output stream by itself — the same tile order, float4 stores, no loads, no arithmetic
// (scripts/micro/store_pattern.hip) — takes 30 us at this shape (6.0 TB/s), the kernel 55: the remaining factor is matrix +
// VALU issue per wave, not DRAM page locality and not the x re-reads (FETCH_SIZE = 1.2 x the size of x: the column blocks'
// re-reads hit the XCD's L2).  Tried on the way, same-box A/B, none moved a wide launch by more than +-3 %: the next tile's x
// in a second register set (one or two tiles ahead), straight-line iterations per epilogue kind so that the wait for the
// prefetch is s_waitcnt vmcnt(<stores>) instead of vmcnt(0), the tail column block folded into the main launch, four
// accumulators taking turns, 6 or 8 waves per block (3 - 4 per SIMD: 10 - 20 % SLOWER).  What did pay: the transposed tile
// (LayerNorm launches 33 -> 22 us at 78 899 x 96) and 16-row tiles on the N <= 192 launches (21 -> 19.5, 17.4 -> 15.9 us).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

SO_DEVFN void so_split3(const float (&x)[8], bf16x8 &a1, bf16x8 &a2, bf16x8 &a3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 b1 = (__bf16)x[j];
        const float r1 = x[j] - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const float r2 = r1 - (float)b2;
        a1[j] = b1; a2[j] = b2; a3[j] = (__bf16)r2;
    }
}

// Epilogues of the b3 kernel.  Its MFMAs take the W fragment as the A operand and the x fragment as B, so the accumulator tile
// is the TRANSPOSE of the one above: lane (r = l % 16, kq = l / 16) holds row r of the 16-row half and the four CONSECUTIVE
// columns 16 t + 4 kq + j — a float4 per (lane, tile).  Twelve 16-byte stores per 32-row tile instead of forty-eight 4-byte
// ones (round 6, s_memtime stamps per wave: the scalar-store epilogue was 2 900 of a tile's 13 400 cycles, issue-bound), the
// bias / gamma / beta / residual reads are float4 too and a LayerNorm row reduces over 24 values in the lane plus two
// shuffle steps (lanes r, r + 16, r + 32, r + 48) instead of six values and four steps.
//   VEC: every row start is 16-byte aligned (pointers and leading dimensions, checked by the launcher); otherwise scalar stores.
//   nval[t]: how many of the lane's four columns of tile t exist (4 everywhere in a full column block).
template <bool LN, bool FULL, bool VEC, int NT16>
SO_DEVFN void so_linear_epilogue_t(f32x4 (&acc)[NT16], const f32x4 (&bv)[NT16], const float *lnp /* LDS: [bias | gamma | beta][96] */,
                                   const int (&nval)[NT16], float relu_lo, const float *rb, int ldr, float *yb, int ldy, float *pb,
                                   float *mb, float *sb, int N, float eps, int rem, int r, int kq) {
    const bool rok = FULL || r < rem;
    const unsigned c0 = 4u * (unsigned)kq;
    f32x4 o[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        f32x4 bq;
        if (LN) bq = *(const f32x4 *)(lnp + 16 * t + c0);       // LayerNorm instances keep bias / gamma / beta in LDS (registers: the
        else bq = bv[t];                                         // row's 24 values twice over, the accumulators and the next tile's x)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[t][j] = fmaxf(acc[t][j] + bq[j], relu_lo);
    }
    if (rb) {
        const float *rr = rb + (size_t)r * ldr + c0;
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            if (FULL && VEC) {
                const float4 v = *(const float4 *)(rr + 16 * t);
                o[t][0] += v.x; o[t][1] += v.y; o[t][2] += v.z; o[t][3] += v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[t][j] += (rok && j < nval[t]) ? rr[16 * t + j] : 0.0f;
            }
        }
    }
    auto put = [&](float *dst, int t, const f32x4 &v) __attribute__((always_inline)) {
        if (FULL && VEC) *(float4 *)dst = make_float4(v[0], v[1], v[2], v[3]);
        else if (rok) {
            if (VEC && nval[t] == 4) *(float4 *)dst = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nval[t]) dst[j] = v[j];
            }
        }
    };
    float *yr = yb + (size_t)r * ldy + c0;
    if (LN) {
        if (pb) {
            float *pr = pb + (size_t)r * N + c0;
#pragma unroll
            for (int t = 0; t < NT16; ++t) put(pr + 16 * t, t, o[t]);
        }
        const float inv_n = 1.0f / (float)N;
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (FULL || j < nval[t]) ? o[t][j] : 0.0f;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = s * inv_n;
        f32x4 dlt[NT16];
        float q2 = 0.0f;
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dlt[t][j] = (FULL || j < nval[t]) ? o[t][j] - mean : 0.0f;
                q2 += dlt[t][j] * dlt[t][j];
            }
        q2 += __shfl_xor(q2, 16, 64);
        q2 += __shfl_xor(q2, 32, 64);
        const float rstd = 1.0f / sqrtf(q2 * inv_n + eps);
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
            const f32x4 gq = *(const f32x4 *)(lnp + 96 + 16 * t + c0), tq = *(const f32x4 *)(lnp + 192 + 16 * t + c0);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaf(dlt[t][j] * rstd, gq[j], tq[j]);
            put(yr + 16 * t, t, v);
        }
        if (mb && kq == 0 && rok) { mb[r] = mean; sb[r] = rstd; }
    } else {
#pragma unroll
        for (int t = 0; t < NT16; ++t) put(yr + 16 * t, t, o[t]);
    }
}

// head-major output (selfocc_linear_fwd_heads) from the transposed tile: column tile t IS head t, the lane's float4 is
// channels 4 kq .. 4 kq + 3 of pixel row r -> y[group][b][t][pix][4 kq ..]
template <int NT16>
SO_DEVFN void so_linear_epilogue_t_hm(f32x4 (&acc)[NT16], const f32x4 (&bv)[NT16], float relu_lo, float *yg, long long row0, int rem,
                                      int nv, int r, int kq) {
    const long long b0 = row0 / nv;
    int pix = (int)(row0 - b0 * nv) + r;
    const unsigned hs = (unsigned)nv * 16u;                       // floats per (b, head)
    unsigned bo = (unsigned)b0 * 6u * hs;
    if (pix >= nv) { pix -= nv; bo += 6u * hs; }
    float *dst = yg + bo + (unsigned)pix * 16u + 4u * (unsigned)kq;
    if (r < rem) {
#pragma unroll
        for (int t = 0; t < NT16; ++t)
            *(float4 *)(dst + (unsigned)t * hs) = make_float4(fmaxf(acc[t][0] + bv[t][0], relu_lo), fmaxf(acc[t][1] + bv[t][1], relu_lo),
                                                              fmaxf(acc[t][2] + bv[t][2], relu_lo), fmaxf(acc[t][3] + bv[t][3], relu_lo));
    }
}

template <int KS /* K / 32 */, bool LN, int NT /* 32-column tiles per block */, int WAVES, int H = 2 /* 16-row halves per wave tile */>
__global__ __launch_bounds__(WAVES * 64, 2) void linear_fwd_b3_kernel(LinearFwdArgs a) {
    constexpr int TR = 16 * H;
    constexpr int K = 32 * KS, KPB = K + 8, NT16 = 2 * NT, THREADS = WAVES * 64, NCOL = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) __bf16 wb[];    // [3][NCOL][KPB]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = lane & 15, kb = lane >> 4;
    const unsigned logical = so_lin_xcd_block();
    const int cb = __builtin_amdgcn_readfirstlane((int)(logical % (unsigned)a.ncb));
    const long long rc = __builtin_amdgcn_readfirstlane((int)(logical / (unsigned)a.ncb));
    const int n0 = a.col0 + cb * 96;
    const long long nwt = (a.T + TR - 1) / TR, wt_step = (long long)WAVES * a.groups;
    SO_TR(0);
    // ---- stage W: 8 consecutive k of one column per thread and step, split into the three planes ----
    {
        constexpr int NV = NCOL * (K / 8);
        for (int idx = threadIdx.x; idx < NV; idx += THREADS) {
            const int r = idx / (K / 8), k8 = idx - r * (K / 8);
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (n0 + r < a.N) {
                const float4 lo = ((const float4 *)(a.w + (size_t)(n0 + r) * K))[2 * k8], hi = ((const float4 *)(a.w + (size_t)(n0 + r) * K))[2 * k8 + 1];
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            }
            bf16x8 w1, w2, w3;
            so_split3(v, w1, w2, w3);
            *(bf16x8 *)(wb + (size_t)r * KPB + 8 * k8) = w1;
            *(bf16x8 *)(wb + (size_t)(NCOL + r) * KPB + 8 * k8) = w2;
            *(bf16x8 *)(wb + (size_t)(2 * NCOL + r) * KPB + 8 * k8) = w3;
        }
    }
    __syncthreads();

    const bool full_cols = a.N - n0 >= 32 * NT;
    const float relu_lo = a.relu ? 0.0f : -__builtin_huge_valf();
    f32x4 bv[NT16];                         // the lane's four consecutive columns 16 t + 4 kb + j of every tile
    int nval[NT16];
    __shared__ __attribute__((aligned(16))) float lnp[LN ? 3 * 96 : 4];
    if (LN) {
        for (int c = threadIdx.x; c < 96; c += THREADS) {
            const bool ok = n0 + c < a.N;
            lnp[c] = (a.bias && ok) ? a.bias[n0 + c] : 0.0f;
            lnp[96 + c] = ok ? a.gamma[n0 + c] : 0.0f;
            lnp[192 + c] = ok ? a.beta[n0 + c] : 0.0f;
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT16; ++t) {
        const int col = n0 + 16 * t + 4 * kb;
        nval[t] = max(0, min(4, a.N - col));
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[t][j] = (!LN && a.bias && j < nval[t]) ? a.bias[col + j] : 0.0f;
    }
    const bool vec = a.vec != 0;

    SO_TR(1);
#ifdef SO_LIN_TRACE
    int tr_i = 0;
#endif
    // raw x of both 16-row halves of a tile, all k steps in flight at once.  The registers are free again as soon as the tile
    // is split into its bf16 planes, so the NEXT tile's x is requested right there — before the matrix phase — into the same
    // registers: its latency runs under this tile's MFMAs and stores at no register cost.
    float xr[H][KS][8];
    auto load_tile = [&](long long wt_) __attribute__((always_inline)) {
        const long long row0_ = wt_ * TR;
        const int rem_ = (int)min((long long)TR, a.T - row0_);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int rl = min(16 * h + n, rem_ - 1);
            const float *xb = a.x + (row0_ + rl) * K + 8 * kb;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float4 lo = *(const float4 *)(xb + 32 * ks), hi = *(const float4 *)(xb + 32 * ks + 4);
                xr[h][ks][0] = lo.x; xr[h][ks][1] = lo.y; xr[h][ks][2] = lo.z; xr[h][ks][3] = lo.w;
                xr[h][ks][4] = hi.x; xr[h][ks][5] = hi.y; xr[h][ks][6] = hi.z; xr[h][ks][7] = hi.w;
            }
        }
    };
    if (rc * WAVES + wave < nwt) load_tile(rc * WAVES + wave);
    for (long long wt = rc * WAVES + wave; wt < nwt; wt += wt_step) {
        const long long row0 = wt * TR;
        const int rem = (int)min((long long)TR, a.T - row0);
#ifdef SO_LIN_TRACE
        SO_TR(2 + 5 * tr_i);
#endif
#ifdef SO_LIN_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SO_TR(3 + 5 * tr_i);
#endif
        f32x4 acc[H][NT16];
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int t = 0; t < NT16; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][t][j] = 0.0f;
        // a pure matrix phase: every split of the tile first (VALU, beside the SIMD's other wave's MFMAs), then 36 K / 32 MFMAs per
        // 16-column tile with the W fragments of group g + 1 read from LDS BEFORE the MFMAs of group g — round 6, s_memtime
        // stamps: with the three ds_read_b128 of a group issued right in front of its MFMAs and the next k-step's ~100 split
        // instructions between the groups, the phase took >= 5 600 cycles for 216 MFMAs x 16 cycles
        const __bf16 *bbase = wb + (size_t)n * KPB + 8 * kb;
        bf16x8 a1[KS][H], a2[KS][H], a3[KS][H];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int h = 0; h < H; ++h) so_split3(xr[h][ks], a1[ks][h], a2[ks][h], a3[ks][h]);
        }
        if (wt + wt_step < nwt) load_tile(wt + wt_step);
        bf16x8 wq[2][3];
        auto wfrag = [&](int g, bf16x8 (&dst)[3]) __attribute__((always_inline)) {
            const int ks = g / NT16, t = g - ks * NT16;
            const __bf16 *bp = bbase + (size_t)(16 * t) * KPB + 32 * ks;
            dst[0] = *(const bf16x8 *)bp; dst[1] = *(const bf16x8 *)(bp + (size_t)NCOL * KPB);
            dst[2] = *(const bf16x8 *)(bp + (size_t)2 * NCOL * KPB);
        };
        wfrag(0, wq[0]);
#pragma unroll
        for (int g = 0; g < KS * NT16; ++g) {
            const int ks = g / NT16, t = g - ks * NT16;
            if (g + 1 < KS * NT16) wfrag(g + 1, wq[(g + 1) & 1]);
            const bf16x8 b1 = wq[g & 1][0], b2 = wq[g & 1][1], b3 = wq[g & 1][2];
#pragma unroll
            for (int h = 0; h < H; ++h) {      // small terms first
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a3[ks][h], acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b3, a1[ks][h], acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, a2[ks][h], acc[h][t], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a2[ks][h], acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, a1[ks][h], acc[h][t], 0, 0, 0);
                acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a1[ks][h], acc[h][t], 0, 0, 0);
            }
        }
#ifdef SO_LIN_TRACE
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int t = 0; t < NT16; ++t) asm volatile("" : "+v"(acc[h][t]));
        SO_TR(4 + 5 * tr_i);
#endif
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const long long r0 = row0 + 16 * h;
            const int rm = rem - 16 * h;
            if (rm <= 0) break;
            const int rmc = min(rm, 16);
            const float *rb = a.residual ? a.residual + r0 * a.ldr + n0 : nullptr;
            float *yb = a.y + r0 * a.ldy + n0;
            float *pb = (LN && a.y_pre) ? a.y_pre + r0 * a.N : nullptr;
            float *mb = (LN && a.mean) ? a.mean + r0 : nullptr, *sb = (LN && a.mean) ? a.rstd + r0 : nullptr;
            int no = n;
            asm volatile("" : "+v"(no));
            if (!LN && NT == 3 && a.hm_nv > 0) {
                if constexpr (!LN && NT == 3)
                    so_linear_epilogue_t_hm<NT16>(acc[h], bv, relu_lo, a.y + (size_t)cb * a.hm_sg, r0, rmc, a.hm_nv, no, kb);
            } else if (rmc == 16 && full_cols && vec)
                so_linear_epilogue_t<LN, true, true, NT16>(acc[h], bv, lnp, nval, relu_lo, rb, a.ldr, yb, a.ldy, pb, mb, sb, a.N,
                                                           a.eps, rmc, no, kb);
            else if (vec)
                so_linear_epilogue_t<LN, false, true, NT16>(acc[h], bv, lnp, nval, relu_lo, rb, a.ldr, yb, a.ldy, pb, mb, sb, a.N,
                                                            a.eps, rmc, no, kb);
            else
                so_linear_epilogue_t<LN, false, false, NT16>(acc[h], bv, lnp, nval, relu_lo, rb, a.ldr, yb, a.ldy, pb, mb, sb, a.N,
                                                             a.eps, rmc, no, kb);
        }
#ifdef SO_LIN_TRACE
        SO_TR(5 + 5 * tr_i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SO_TR(6 + 5 * tr_i);
        ++tr_i;
#endif
    }
    SO_TR(63);
}

// ---------------------------------------------------------------------------------------------------------------------
// Input gradient of a tall Linear, dx = dy W (round 3: the training encoder's dgrad GEMMs, 2.7 ms per iteration on the
// vendor library), on the b3 scheme without LDS:
//   1. linear_dgrad_split_kernel: W (N, K) as stored -> three bf16 planes [p][column c < K][k < Npad] (transposed: the B
//      operand wants 8 consecutive reduction indices per lane; zero beyond N), <= 442 KB, one tiny launch;
//   2. linear_dgrad_b3_kernel: a wave owns 32 rows x 96 columns and walks the reduction in 32-wide steps; dy streams from
//      HBM straight into the A operand (split in registers), the B operand's three planes come from global memory — every
//      wave of the chip reads the same <= 442 KB, so they are vector-L1 / L2 hits (18 KB per step and wave: ~1/2 of the L1 rate
//      the MFMAs of the step leave room for).  No staging, no barriers: the first version of this kernel staged 96-wide chunks
//      of W through LDS and was a chain of load -> barrier -> stage -> barrier -> MFMA latencies (63 us at 66 049 x 384 x 96
//      where the dy stream takes 16).
__global__ __launch_bounds__(256) void linear_dgrad_split_kernel(const float *w, __bf16 *planes, int N, int K, int Npad) {
    const int nk8 = Npad / 8;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < K * nk8; idx += gridDim.x * 256) {
        const int k8 = idx / K, c = idx - k8 * K;                 // consecutive threads: consecutive columns (coalesced reads)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * k8 + j;
            v[j] = kk < N ? w[(size_t)kk * K + c] : 0.0f;
        }
        bf16x8 w1, w2, w3;
        so_split3(v, w1, w2, w3);
        const size_t o = (size_t)c * Npad + 8 * k8, ps = (size_t)K * Npad;
        *(bf16x8 *)(planes + o) = w1;
        *(bf16x8 *)(planes + ps + o) = w2;
        *(bf16x8 *)(planes + 2 * ps + o) = w3;
    }
}

struct LinearDgradArgs {
    const float *dy;         // (T, N)
    const __bf16 *planes;    // [3][K][Npad], Npad = 96 nchunks; NULL (one chunk only): the kernel splits W itself
    const float *w;          // (N, K) as stored: read when planes == NULL
    float *dx;               // (T, K)
    long long T;
    int N, K, Npad, groups;
};

// Work items = (32-row tile of the wave, 96-wide chunk of the reduction), walked in order by the block's four waves together.
// Before the MFMAs of item i every lane issues the dy loads of item i + 1, so that the HBM stream and the matrix pipe overlap
// inside one wave; the chunk's planes are copied from L2 between two barriers, which the CU's other block (60 KB of LDS each)
// covers with its MFMAs.
template <int DUMMY>
__global__ __launch_bounds__(256) void linear_dgrad_b3_kernel(LinearDgradArgs a) {
    constexpr int KC = 96, KS = 3, KPB = KC + 8, NT16 = 6, NCOL = 96, WAVES = 4, THREADS = 256;
    constexpr int NST = (3 * NCOL * (KC / 8) + THREADS - 1) / THREADS;     // 16-byte staging copies per thread and chunk: 14
    extern __shared__ __attribute__((aligned(16))) __bf16 wb[];            // [3][NCOL][KPB]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = lane & 15, kb = lane >> 4;
    const unsigned logical = so_lin_xcd_block();
    const int ncb = a.K / 96;
    const int cb = __builtin_amdgcn_readfirstlane((int)(logical % (unsigned)ncb));
    const long long rc = __builtin_amdgcn_readfirstlane((int)(logical / (unsigned)ncb));
    const int n0 = cb * 96;
    const long long nwt = (a.T + 31) / 32, wt_step = (long long)WAVES * a.groups;
    const int nchunks = a.Npad / KC;
    const long long npass = (nwt - rc * WAVES + wt_step - 1) / wt_step;    // passes of this block (uniform), >= 1 by the grid size
    const long long nitems = npass * nchunks;
    const size_t ps = (size_t)a.K * a.Npad;

    float xc[2][KS][8], xn[2][KS][8];
    auto tile_row0 = [&](long long pass) { return min(rc * WAVES + pass * wt_step + wave, nwt - 1) * 32; };
    auto load_item = [&](long long it) {            // dy run of item it -> xn
        const long long pass = it / nchunks;
        const int c = (int)(it - pass * nchunks), k0 = c * KC;
        const long long row0 = tile_row0(pass);
        const int rem = (int)min(32LL, a.T - row0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float *xb = a.dy + (row0 + min(16 * h + n, rem - 1)) * a.N + k0 + 8 * kb;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                if (k0 + 32 * ks + 8 * kb < a.N) { lo = *(const float4 *)(xb + 32 * ks); hi = *(const float4 *)(xb + 32 * ks + 4); }
                xn[h][ks][0] = lo.x; xn[h][ks][1] = lo.y; xn[h][ks][2] = lo.z; xn[h][ks][3] = lo.w;
                xn[h][ks][4] = hi.x; xn[h][ks][5] = hi.y; xn[h][ks][6] = hi.z; xn[h][ks][7] = hi.w;
            }
        }
    };
    load_item(0);
    f32x4 acc[2][NT16];
    const float zero6[NT16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool cok[NT16] = {true, true, true, true, true, true};
    for (long long it = 0; it < nitems; ++it) {
        const long long pass = it / nchunks;
        const int c = (int)(it - pass * nchunks);
        if (nchunks > 1 || it == 0) {
            __syncthreads();                         // the previous chunk's B reads are done
            if (a.planes == nullptr) {
                // one chunk (N <= 96): no pre-split planes — transpose and split W here, once per (persistent) block:
                // thread (column r, run k8) reads W[8 k8 + j][n0 + r] (coalesced along r)
                for (int idx = threadIdx.x; idx < NCOL * (KC / 8); idx += THREADS) {
                    const int k8 = idx / NCOL, r = idx - k8 * NCOL;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kk = 8 * k8 + j;
                        v[j] = kk < a.N ? a.w[(size_t)kk * a.K + n0 + r] : 0.0f;
                    }
                    bf16x8 w1, w2, w3;
                    so_split3(v, w1, w2, w3);
                    *(bf16x8 *)(wb + (size_t)r * KPB + 8 * k8) = w1;
                    *(bf16x8 *)(wb + (size_t)(NCOL + r) * KPB + 8 * k8) = w2;
                    *(bf16x8 *)(wb + (size_t)(2 * NCOL + r) * KPB + 8 * k8) = w3;
                }
            } else
            // the chunk's planes: 3 x 96 x 12 16-byte runs, straight copies (L2 hits: every block of the chip reads the same)
#pragma unroll 2
            for (int j0 = 0; j0 < NST; j0 += 7) {
                bf16x8 sv[7];
                int so[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int idx = min((int)threadIdx.x + (j0 + j) * THREADS, 3 * NCOL * (KC / 8) - 1);
                    const int pl = idx / (NCOL * (KC / 8)), rem = idx - pl * (NCOL * (KC / 8));
                    const int r = rem / (KC / 8), k8 = rem - r * (KC / 8);
                    sv[j] = *(const bf16x8 *)(a.planes + pl * ps + (size_t)(n0 + r) * a.Npad + c * KC + 8 * k8);
                    so[j] = (pl * NCOL + r) * KPB + 8 * k8;
                }
#pragma unroll
                for (int j = 0; j < 7; ++j) *(bf16x8 *)(wb + so[j]) = sv[j];      // (the clamped tail rewrites the last run)
            }
            __syncthreads();
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) xc[h][ks][j] = xn[h][ks][j];
        if (it + 1 < nitems) load_item(it + 1);
        if (c == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT16; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[h][t][j] = 0.0f;
        }
        const __bf16 *bbase = wb + (size_t)n * KPB + 8 * kb;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 a1[2], a2[2], a3[2];
            so_split3(xc[0][ks], a1[0], a2[0], a3[0]);
            so_split3(xc[1][ks], a1[1], a2[1], a3[1]);
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                const __bf16 *bp = bbase + (size_t)(16 * t) * KPB + 32 * ks;
                const bf16x8 b1 = *(const bf16x8 *)bp, b2 = *(const bf16x8 *)(bp + (size_t)NCOL * KPB),
                             b3 = *(const bf16x8 *)(bp + (size_t)2 * NCOL * KPB);
#pragma unroll
                for (int h = 0; h < 2; ++h) {      // small terms first
                    acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[h], b1, acc[h][t], 0, 0, 0);
                    acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[h], b3, acc[h][t], 0, 0, 0);
                    acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[h], b2, acc[h][t], 0, 0, 0);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[h], b1, acc[h][t], 0, 0, 0);
                    acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[h], b2, acc[h][t], 0, 0, 0);
                    acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[h], b1, acc[h][t], 0, 0, 0);
                }
            }
        }
        if (c != nchunks - 1) continue;
        const long long wt = rc * WAVES + pass * wt_step + wave;
        if (wt >= nwt) continue;
        const long long row0 = wt * 32;
        const int rem = (int)min(32LL, a.T - row0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = row0 + 16 * h;
            const int rm = rem - 16 * h;
            if (rm <= 0) break;
            const int rmc = min(rm, 16);
            float *yb = a.dx + r0 * a.K + n0;
            int no = n;
            asm volatile("" : "+v"(no));
            if (rmc == 16)
                so_linear_epilogue16<false, true, NT16>(acc[h], zero6, zero6, zero6, cok, -__builtin_huge_valf(), nullptr, 0, yb, a.K,
                                                        nullptr, nullptr, nullptr, a.K, 0.0f, rmc, no, kb);
            else
                so_linear_epilogue16<false, false, NT16>(acc[h], zero6, zero6, zero6, cok, -__builtin_huge_valf(), nullptr, 0, yb, a.K,
                                                         nullptr, nullptr, nullptr, a.K, 0.0f, rmc, no, kb);
        }
    }
}

bool so_linear_dgrad_ok(long long T, int N, int K) {     // N = the layer's output features (reduced), K = its input features
    return T >= 1 && T < (1LL << 40) && N >= 8 && N % 8 == 0 && N <= 4096 && K >= 96 && K % 96 == 0 && K <= 384;
}

bool so_linear_fwd_ok(long long T, int N, int K) {
    if (!(K == 32 || K == 64 || K == 96 || K == 128 || K == 192)) return false;
    return T >= 1 && N >= 1 && (long long)N * K < (1LL << 30) && T < (1LL << 40);
}

}  // namespace

extern "C" int selfocc_linear_fwd_supported(int64_t T, int32_t N, int32_t K) { return so_linear_fwd_ok(T, N, K) ? 1 : 0; }

static int so_linear_fwd_launch(const float *x, const float *w, const float *bias, const float *residual, int32_t ldr,
                                const float *ln_gamma, const float *ln_beta, float ln_eps, float *y, int32_t ldy,
                                float *y_pre, float *mean, float *rstd, int64_t T, int32_t N, int32_t K,
                                uint32_t flags, int32_t hm_nv, void *stream) {
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
    a.hm_nv = hm_nv;
    a.vec = (((uintptr_t)y | (uintptr_t)residual | (uintptr_t)y_pre) & 15) == 0 && ldy % 4 == 0 && (!residual || ldr % 4 == 0) && N % 4 == 0;
    a.hm_sg = hm_nv > 0 ? (unsigned)(T * 96) : 0u;
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
        // round 3: the bf16 three-way-split kernel (same results to float32 rounding, 2.67 x fewer matrix-pipe cycles);
        // SELFOCC_LINEAR_B3=0 keeps the f32-MFMA kernel (A/B)
        static const bool b3_env = !(getenv("SELFOCC_LINEAR_B3") && atoi(getenv("SELFOCC_LINEAR_B3")) == 0);
        // where it pays (measured inside the eval encoder, profiles/r3_h_*): enough 32-row tiles per block to amortise the
        // three-plane split of W at block start (the 6 - 8 k-row zh / wz planes at N = 96 ran 17 vs 9 us); K = 192 has its own
        // launch above (round 6: 78 899 x 192 -> 96 with residual + LayerNorm 50.2 -> 37.3 us, plain 39.4 -> 31.0 us against
        // the f32-MFMA kernel, same box)
        // K = 192 (the FFN's second Linear + residual + LayerNorm): the planes take 115 KB, so ONE block of eight waves per CU shares
        // them (two waves per SIMD, like two four-wave blocks at K <= 128), 16-row tiles
        if (b3_env && K == 192 && T >= 16384) {
            const size_t lds_b3 = (size_t)3 * nt * 32 * (K + 8) * 2;
            const long long nwt3 = (T + 15) / 16;
            long long groups3 = std::max(1LL, 256LL / a.ncb);
            groups3 = std::min(groups3, (nwt3 + 7) / 8);
            a.groups = (int)groups3;
            const long long nblk3 = groups3 * a.ncb;
#define SO_B3_K192(LN_, NT_)                                                                                          \
    do {                                                                                                             \
        static std::atomic<unsigned long long> done_mask{0};                                                         \
        int dev_ = 0;                                                                                                \
        (void)hipGetDevice(&dev_);                                                                                   \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                         \
        if (!(done_mask.load(std::memory_order_relaxed) & bit_)) {                                                   \
            (void)hipFuncSetAttribute((const void *)linear_fwd_b3_kernel<6, LN_, NT_, 8, 1>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);                \
            done_mask.fetch_or(bit_, std::memory_order_relaxed);                                                     \
        }                                                                                                            \
        hipLaunchKernelGGL((linear_fwd_b3_kernel<6, LN_, NT_, 8, 1>), dim3((unsigned)nblk3), dim3(512), lds_b3, st, a); \
    } while (0)
            if (ln) { if (nt == 3) SO_B3_K192(true, 3); else if (nt == 2) SO_B3_K192(true, 2); else SO_B3_K192(true, 1); }
            else { if (nt == 3) SO_B3_K192(false, 3); else if (nt == 2) SO_B3_K192(false, 2); else SO_B3_K192(false, 1); }
#undef SO_B3_K192
            continue;
        }
        const bool use_b3 = b3_env && K <= 128 && (T >= 16384 || N >= 384);
        if (use_b3) {
            const size_t lds_b3 = (size_t)3 * nt * 32 * (K + 8) * 2;
            long long per_cu3 = std::max<long long>(1, std::min<long long>(2, (160 * 1024) / (long long)(lds_b3 + 512)));
            // wave tile: 32 rows x 96 columns where the column blocks share the x rows (N >= 288: the two 16-row halves use
            // every W fragment read from LDS twice); 16 rows where one or two column blocks make short launches (N <= 192: the
            // launch is a ramp, a W staging and 1 - 2 tiles per wave — half-size tiles spread it better; round 6, same box:
            // 78 899 x 96 24.5 -> 19.5 us, 66 049 x 96 19.3 -> 16.7 us, 78 899 x 192 30.0 -> 28.1 us, the wide shapes unchanged)
            const bool half_tiles = N <= 192;
            const long long nwt3 = half_tiles ? (T + 15) / 16 : (T + 31) / 32;
            long long groups3 = std::max(1LL, 256 * per_cu3 / a.ncb);
            groups3 = std::min(groups3, (nwt3 + 3) / 4);
            a.groups = (int)groups3;
            const long long nblk3 = groups3 * a.ncb;
#define SO_B3_0(KS_, LN_, NT_, H_)                                                                                    \
    do {                                                                                                             \
        /* the attribute is per device: set once per (device, instantiation) — the call is a driver round trip of tens of */ \
        /* microseconds, which at ~45 launches per frame would make the host the bottleneck                                */ \
        static std::atomic<unsigned long long> done_mask{0};                                                         \
        int dev_ = 0;                                                                                                \
        (void)hipGetDevice(&dev_);                                                                                   \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                         \
        if (lds_b3 > 48 * 1024 && !(done_mask.load(std::memory_order_relaxed) & bit_)) {                             \
            (void)hipFuncSetAttribute((const void *)linear_fwd_b3_kernel<KS_, LN_, NT_, 4, H_>,                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);                \
            done_mask.fetch_or(bit_, std::memory_order_relaxed);                                                     \
        }                                                                                                            \
        hipLaunchKernelGGL((linear_fwd_b3_kernel<KS_, LN_, NT_, 4, H_>), dim3((unsigned)nblk3), dim3(256), lds_b3, st, a); \
    } while (0)
#define SO_B3_1(KS_, LN_, NT_)                                                                                        \
    do {                                                                                                             \
        if (half_tiles) SO_B3_0(KS_, LN_, NT_, 1);                                                                   \
        else SO_B3_0(KS_, LN_, NT_, 2);                                                                              \
    } while (0)
#define SO_B3_2(KS_, LN_)                                                                                             \
    do {                                                                                                             \
        if (nt == 3) SO_B3_1(KS_, LN_, 3);                                                                           \
        else if (nt == 2) SO_B3_1(KS_, LN_, 2);                                                                      \
        else SO_B3_1(KS_, LN_, 1);                                                                                   \
    } while (0)
#define SO_B3_3(KS_)                                                                                                  \
    do {                                                                                                             \
        if (ln) SO_B3_2(KS_, true);                                                                                  \
        else SO_B3_2(KS_, false);                                                                                    \
    } while (0)
            switch (K) {
                case 32: SO_B3_3(1); break;
                case 64: SO_B3_3(2); break;
                case 96: SO_B3_3(3); break;
                default: SO_B3_3(4); break;      // K = 128
            }
#undef SO_B3_3
#undef SO_B3_2
#undef SO_B3_1
#undef SO_B3_0
            continue;
        }
        const size_t lds_blk = (size_t)nt * 32 * (K + 4) * sizeof(float);
        static const long long percu_env = getenv("SELFOCC_LINEAR_PERCU") ? atoll(getenv("SELFOCC_LINEAR_PERCU")) : 0;  // dev A/B
        // persistent grid: two 4-wave blocks per CU (measured with the prefetches: 1 / 2 / 3 per CU = 495 / 483 / 527 us over
        // the encoder's twelve shapes; 4 per CU 607 us), split evenly over the column blocks; a block's waves take the 16-row tiles
        // (group * 4 + wave) + 4 groups j
        long long per_cu = std::max<long long>(1, std::min<long long>(2, (160 * 1024) / (long long)(lds_blk + 512)));
        if (percu_env > 0) per_cu = percu_env;
        const long long nwt = (T + 15) / 16;
        long long groups = std::max(1LL, 256 * per_cu / a.ncb);
        groups = std::min(groups, (nwt + 3) / 4);
        a.groups = (int)groups;
        const long long nblk = groups * a.ncb;
#define SO_L16_1(KQ_, LN_, NT_)                                                                                       \
    do {                                                                                                             \
        /* per launch, not once per process: the attribute is per device, and a second GPU in the same process */    \
        /* would otherwise launch without it (cheap: no driver round trip after the first call on a device)      */    \
        if (lds_blk > 48 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)linear_fwd16_kernel<KQ_, LN_, NT_, 4>,                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_blk);                     \
        hipLaunchKernelGGL((linear_fwd16_kernel<KQ_, LN_, NT_, 4>), dim3((unsigned)nblk), dim3(256), lds_blk, st, a); \
    } while (0)
#define SO_L16_2(KQ_, LN_)                                                                                            \
    do {                                                                                                             \
        if (nt == 3) SO_L16_1(KQ_, LN_, 3);                                                                          \
        else if (nt == 2) SO_L16_1(KQ_, LN_, 2);                                                                     \
        else SO_L16_1(KQ_, LN_, 1);                                                                                  \
    } while (0)
#define SO_L16_3(KQ_)                                                                                                 \
    do {                                                                                                             \
        if (ln) SO_L16_2(KQ_, true);                                                                                 \
        else SO_L16_2(KQ_, false);                                                                                   \
    } while (0)
        switch (K) {
            case 32: SO_L16_3(8); break;
            case 64: SO_L16_3(16); break;
            case 96: SO_L16_3(24); break;
            case 128: SO_L16_3(32); break;
            default: SO_L16_3(48); break;
        }
#undef SO_L16_3
#undef SO_L16_2
#undef SO_L16_1
    }
    return so_launch_status();
}

extern "C" int selfocc_linear_fwd(const float *x, const float *w, const float *bias, const float *residual, int32_t ldr,
                                  const float *ln_gamma, const float *ln_beta, float ln_eps, float *y, int32_t ldy,
                                  float *y_pre, float *mean, float *rstd, int64_t T, int32_t N, int32_t K,
                                  uint32_t flags, void *stream) {
    return so_linear_fwd_launch(x, w, bias, residual, ldr, ln_gamma, ln_beta, ln_eps, y, ldy, y_pre, mean, rstd, T, N, K, flags, 0,
                                stream);
}

extern "C" int selfocc_linear_fwd_heads(const float *x, const float *w, const float *bias, float *y, int64_t T, int32_t N,
                                        int32_t K, int32_t nv, uint32_t flags, void *stream) {
    SO_REQUIRE(N >= 96 && N % 96 == 0, "linear_fwd_heads: N = %d must be a multiple of 96 (6 heads x 16 channels per group)", N);
    SO_REQUIRE(nv >= 16 && T % nv == 0, "linear_fwd_heads: T = %lld rows must be a whole number of batch items of nv = %d >= 16 rows",
               (long long)T, nv);
    SO_REQUIRE((long long)T * N < (1LL << 31), "linear_fwd_heads: output too large for 32-bit offsets");
    SO_REQUIRE(((uintptr_t)y & 15) == 0, "linear_fwd_heads: y must be 16-byte aligned (the head-major rows are written as float4)");
    return so_linear_fwd_launch(x, w, bias, nullptr, 0, nullptr, nullptr, 0.0f, y, N, nullptr, nullptr, nullptr, T, N, K, flags, nv,
                                stream);
}

// dx = dy W for a tall Linear (the input gradient; replaces the `dy @ weight` GEMM of torch.nn.functional.linear's backward
// on the encoder's projections, /root/reference/model/encoder/tpvformer/tpvformer_encoder.py:257-291 runs them under autograd)
extern "C" int selfocc_linear_dgrad_supported(int64_t T, int32_t N, int32_t K) { return so_linear_dgrad_ok(T, N, K) ? 1 : 0; }

extern "C" size_t selfocc_linear_dgrad_workspace(int32_t N, int32_t K) {
    return (size_t)3 * K * ((N + 95) / 96 * 96) * 2;
}

extern "C" int selfocc_linear_dgrad(const float *dy, const float *w, float *dx, int64_t T, int32_t N, int32_t K, void *workspace,
                                    int64_t workspace_bytes, void *stream) {
    SO_REQUIRE(so_linear_dgrad_ok(T, N, K), "linear_dgrad: unsupported shape (T = %lld rows, N = %d reduced features (multiple of 8, "
               "<= 4096), K = %d input features (96, 192, 288 or 384))", (long long)T, N, K);
    SO_REQUIRE(dy && w && dx && workspace, "linear_dgrad: NULL pointer");
    SO_REQUIRE(workspace_bytes >= (int64_t)selfocc_linear_dgrad_workspace(N, K), "linear_dgrad: workspace of %lld bytes, need %lld",
               (long long)workspace_bytes, (long long)selfocc_linear_dgrad_workspace(N, K));
    SO_REQUIRE(((uintptr_t)workspace & 15) == 0, "linear_dgrad: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int Npad = (N + 95) / 96 * 96;
    LinearDgradArgs a;
    a.dy = dy; a.planes = (const __bf16 *)workspace; a.w = w; a.dx = dx; a.T = T; a.N = N; a.K = K; a.Npad = Npad;
    if (Npad == 96) a.planes = nullptr;        // one chunk: the main kernel splits W while it stages it (one launch less)
    else hipLaunchKernelGGL(linear_dgrad_split_kernel, dim3((unsigned)std::min(256, (K * (Npad / 8) + 255) / 256)), dim3(256), 0,
                            st, w, (__bf16 *)workspace, N, K, Npad);
    const int ncb = K / 96;
    const long long nwt = (T + 31) / 32;
    static const int percu = getenv("SELFOCC_DGRAD_PERCU") ? atoi(getenv("SELFOCC_DGRAD_PERCU")) : 2;      // dev A/B
    long long groups = std::max<long long>(1, 256 * percu / ncb);
    groups = std::min<long long>(groups, (nwt + 3) / 4);
    a.groups = (int)groups;
    const size_t lds = (size_t)3 * 96 * (96 + 8) * 2;
    static std::atomic<unsigned long long> done_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute((const void *)linear_dgrad_b3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        done_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(linear_dgrad_b3_kernel<0>, dim3((unsigned)(groups * ncb)), dim3(256), lds, st, a);
    return so_launch_status();
}
