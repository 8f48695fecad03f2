// linear.hip — weight and bias gradient of the encoder's tall-skinny projections in ONE pass over dY and X:
//     dW[n][k] = sum_r dY[r][n] X[r][k]        db[n] = sum_r dY[r][n]
// (the Linear layers of TPVFormerLayer / BEVFormerLayer and their attention modules: sampling_offsets,
// attention_weights, value_proj, output_proj, FFN — mmcv Linear under torch autograd in the reference,
// model/encoder/tpvformer/attention/image_cross_attention.py:130-160, cross_view_hybrid_attention.py:35-60).
//
// Shapes: r runs over 66 k - 180 k rows, K = 96 (192) input features, N = 96 .. 2304 outputs.  The vendor GEMM for
// dY^T X with that reduction length runs on a handful of workgroups; round 1 split it into <= 256 batched GEMMs
// + a sum, and took the bias gradient with torch's column reduction: dY is read twice, 56 + 20 + ~10 us per layer
// call, 36 calls per training iteration.  Here the reduction over rows is the K dimension of
// v_mfma_f32_32x32x2_f32 and BOTH operands load from global memory straight in MFMA layout:
//     A[m = n][kk = row pair]: lane (i, half) <- dY[r + half][n0 + i]     (32 consecutive floats of one row)
//     B[kk = row pair][col = k]: lane (i, half) <- X[r + half][k0 + i]
// no LDS, no transposes; the bias gradient is the running sum of the A operand.  A wave keeps NTW x KT 32 x 32
// accumulator tiles (<= 12: 192 registers) for its quarter of the block's rows; the four waves of a block are
// summed through one LDS tile at the end, and the per-block partials by a second tiny kernel in a fixed order
// (deterministic).  f32 MFMA on gfx950 is an exact fmaf chain (no TF32): float32 arithmetic.
//
// Measured (66 049 x 384 x 96, scripts/micro/wgrad_bench.py): 65 us = 70 TFLOP/s, 1.9 TB/s.  Built without the MFMAs
// it takes 35 us (3.6 TB/s), with the loads hoisted out of the loop 45 us (the 1.2 M MFMAs at 64 clk on 1 024 SIMDs
// are 31 us): at 2 waves per SIMD (144 accumulator + 48 prefetch registers) the two overlap only partly.  Tried without
// gain: branch-free clamped loads + ping-pong register buffers + sched_barrier so that the compiler emits counted
// vmcnt waits (74 us: 64-bit address math per load, spills), an XCD-aware block order that lets the n-groups of a row
// chunk share x in one L2 (70 -> 70 us; 96 -> 138 us for N = 432); and (end of round 2) the LDS-staged form — a block
// stages 32 rows of dY / X with coalesced float4 loads (next stage in registers, one barrier per stage) and its four
// waves split the 96 x K output 2 x 2 on v_mfma_f32_16x16x4_f32 (36 accumulator registers, ds_read_b32 operands): correct,
// but 83 / 113 / 202 us where this kernel takes 70 / 95 / 153 us (a barrier per 72 MFMAs per wave and one LDS read per
// operand cost more than the overlap buys) — dropped.
#include "so_device.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRowPairsAhead = 4;      // row pairs loaded per step (all loads of a step are in flight together)

template <int KT, int NTW>
__global__ __launch_bounds__(256, 2) void linear_wgrad_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                              float *__restrict__ part_w /* [chunks][N][K] */,
                                                              float *__restrict__ part_b /* [chunks][N] */,
                                                              long long T, int N, int K, long long rows_per_block) {
    constexpr int U = (NTW * KT > 9) ? kRowPairsAhead / 2 : kRowPairsAhead;   // 12 tiles: 192 accumulator registers
    __shared__ float red[NTW * KT * 1024];
    __shared__ float red_b[4][NTW * 32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, half = lane >> 5;
    const int chunk = blockIdx.x;
    const int n0 = blockIdx.y * (NTW * 32);
    const long long r_blk = (long long)chunk * rows_per_block;
    const long long rpw = rows_per_block / 4;                         // rows per wave (host: a multiple of 2)
    const long long r0 = r_blk + wave * rpw, r1 = min(T, r0 + rpw);

    f32x16 acc[NTW][KT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][u][v] = 0.0f;
    float bsum[NTW];
    bool ncol[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) { bsum[t] = 0.0f; ncol[t] = n0 + t * 32 + i < N; }

    auto load = [&](long long r, float (&a)[NTW], float (&b)[KT]) {
        const long long rr = r + half;
        const bool ok = rr < r1;
        const float *dyr = dy + rr * N + n0 + i;
        const float *xr = x + rr * K + i;
#pragma unroll
        for (int t = 0; t < NTW; ++t) a[t] = (ok && ncol[t]) ? dyr[t * 32] : 0.0f;
#pragma unroll
        for (int u = 0; u < KT; ++u) b[u] = ok ? xr[u * 32] : 0.0f;
    };
    float a_cur[U][NTW], b_cur[U][KT], a_nxt[U][NTW], b_nxt[U][KT];
#pragma unroll
    for (int s = 0; s < U; ++s) load(r0 + 2 * s, a_cur[s], b_cur[s]);
    for (long long r = r0; r < r1; r += 2 * U) {
#pragma unroll
        for (int s = 0; s < U; ++s) load(r + 2 * (U + s), a_nxt[s], b_nxt[s]);
#pragma unroll
        for (int s = 0; s < U; ++s) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                bsum[t] += a_cur[s][t];
#pragma unroll
                for (int u = 0; u < KT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s][t], b_cur[s][u], acc[t][u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < U; ++s) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) a_cur[s][t] = a_nxt[s][t];
#pragma unroll
            for (int u = 0; u < KT; ++u) b_cur[s][u] = b_nxt[s][u];
        }
    }

    // block reduction: waves 1..3 hand their tiles to wave 0 through one LDS tile set, one after the other
#pragma unroll
    for (int t = 0; t < NTW; ++t) bsum[t] += __shfl_xor(bsum[t], 32, 64);
    if (half == 0) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) red_b[wave][t * 32 + i] = bsum[t];
    }
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int u = 0; u < KT; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) red[((t * KT + u) * 16 + v) * 64 + lane] = acc[t][u][v];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int u = 0; u < KT; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[t][u][v] += red[((t * KT + u) * 16 + v) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // C layout: col = lane & 31 (k), row = (v & 3) + 8 (v >> 2) + 4 half (n)
    float *pw = part_w + (size_t)chunk * N * K;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int n = n0 + t * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
            if (n >= N) continue;
#pragma unroll
            for (int u = 0; u < KT; ++u) pw[(size_t)n * K + u * 32 + i] = acc[t][u][v];
        }
        if (half == 0 && ncol[t])
            part_b[(size_t)chunk * N + n0 + t * 32 + i] =
                (red_b[0][t * 32 + i] + red_b[1][t * 32 + i]) + (red_b[2][t * 32 + i] + red_b[3][t * 32 + i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same reduction on the bf16 matrix pipe with the exact three-way split of linear_fwd.hip (round 3): every loaded
// float32 of dY and X becomes three bfloat16 (x = x1 + x2 + x3 exactly), six v_mfma_f32_32x32x16_bf16 (the products with
// i + j <= 4, small terms first) replace eight v_mfma_f32_32x32x2_f32 per 16 rows — 2.67 x fewer matrix-pipe cycles at
// float32-level accuracy.  Operand layout of the 16-row step (reduction index = row):
//     A[m = n][k = row]: lane (i, kb) <- dY[r + 8 kb + j][n0 + 32 t + i], j < 8     (per j: 32 consecutive floats of a row)
//     B[k = row][col]:   lane (i, kb) <- X[r + 8 kb + j][32 u + i]
// X's parts are split once per step and shared by the NTW tile rows; dY's parts per tile row.  The bias gradient is the
// float32 running sum of the raw A operand, as before.  Accumulator layout and everything after the loop are the f32
// kernel's.
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));

SO_DEVFN void so_wsplit3(const float (&x)[8], bf16x8w &a1, bf16x8w &a2, bf16x8w &a3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 b1 = (__bf16)x[j];
        const float r1 = x[j] - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const float r2 = r1 - (float)b2;
        a1[j] = b1; a2[j] = b2; a3[j] = (__bf16)r2;
    }
}

template <int KT, int NTW>
__global__ __launch_bounds__(256, 2) void linear_wgrad_b3_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                 float *__restrict__ part_w, float *__restrict__ part_b,
                                                                 long long T, int N, int K, long long rows_per_block) {
    __shared__ float red[NTW * KT * 1024];
    __shared__ float red_b[4][NTW * 32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, half = lane >> 5;                       // half = k block (rows 8 half .. 8 half + 7 of a step)
    const int chunk = blockIdx.x;
    const int n0 = blockIdx.y * (NTW * 32);
    const long long r_blk = (long long)chunk * rows_per_block;
    const long long rpw = rows_per_block / 4;
    const long long r0 = r_blk + wave * rpw, r1 = min(T, r0 + rpw);

    f32x16 acc[NTW][KT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][u][v] = 0.0f;
    float bsum[NTW];
    bool ncol[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) { bsum[t] = 0.0f; ncol[t] = n0 + t * 32 + i < N; }

    // the operands of step r + 16 are loaded before the MFMAs of step r (NTW = 1: the registers are there): a wave has
    // 2 - 3 neighbours on its SIMD, not enough to cover the HBM latency of 16 rows by themselves
    constexpr bool PF = NTW == 1 && KT <= 3;      // (NTW = 2 with the second operand set: 255 registers, measured 58 -> 74 us)
    float a[NTW][8], b[KT][8], an[PF ? NTW : 1][8], bn[PF ? KT : 1][8];
    auto load_step = [&](long long r, float (&da)[NTW][8], float (&db)[KT][8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long rr = r + 8 * half + j;
            const bool ok = rr < r1;
            const float *dyr = dy + rr * N + n0 + i;
            const float *xr = x + rr * K + i;
#pragma unroll
            for (int t = 0; t < NTW; ++t) da[t][j] = (ok && ncol[t]) ? dyr[t * 32] : 0.0f;
#pragma unroll
            for (int u = 0; u < KT; ++u) db[u][j] = ok ? xr[u * 32] : 0.0f;
        }
    };
    if constexpr (PF) load_step(r0, a, b);
    for (long long r = r0; r < r1; r += 16) {
        if constexpr (PF) {
            if (r + 16 < r1) load_step(r + 16, an, bn);
        } else {
            load_step(r, a, b);
        }
        bf16x8w b1[KT], b2[KT], b3[KT];
#pragma unroll
        for (int u = 0; u < KT; ++u) so_wsplit3(b[u], b1[u], b2[u], b3[u]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[t] += a[t][j];
            bf16x8w a1, a2, a3;
            so_wsplit3(a[t], a1, a2, a3);
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1[u], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3[u], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2[u], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1[u], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[u], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[u], acc[t][u], 0, 0, 0);
            }
        }
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) a[t][j] = an[t][j];
#pragma unroll
                for (int u = 0; u < KT; ++u) b[u][j] = bn[u][j];
            }
        }
    }

    // block reduction and stores: as in linear_wgrad_kernel
#pragma unroll
    for (int t = 0; t < NTW; ++t) bsum[t] += __shfl_xor(bsum[t], 32, 64);
    if (half == 0) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) red_b[wave][t * 32 + i] = bsum[t];
    }
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int u = 0; u < KT; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) red[((t * KT + u) * 16 + v) * 64 + lane] = acc[t][u][v];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int u = 0; u < KT; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[t][u][v] += red[((t * KT + u) * 16 + v) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    float *pw = part_w + (size_t)chunk * N * K;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int n = n0 + t * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
            if (n >= N) continue;
#pragma unroll
            for (int u = 0; u < KT; ++u) pw[(size_t)n * K + u * 32 + i] = acc[t][u][v];
        }
        if (half == 0 && ncol[t])
            part_b[(size_t)chunk * N + n0 + t * 32 + i] =
                (red_b[0][t * 32 + i] + red_b[1][t * 32 + i]) + (red_b[2][t * 32 + i] + red_b[3][t * 32 + i]);
    }
}

// dW / db = sum over the chunks in a fixed order: wave w of a block adds the chunks c = w (mod 4) of 64 consecutive
// elements (8 loads in flight), the four waves are combined through LDS
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ part_w,
                                                                  const float *__restrict__ part_b,
                                                                  float *__restrict__ dw, float *__restrict__ db, int chunks,
                                                                  int NK, int N) {
    __shared__ float red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 64 + lane;
    const bool is_w = e < NK, is_b = !is_w && e < NK + N && db != nullptr;
    const float *src = is_w ? part_w + e : part_b + (e - NK);
    const size_t stride = is_w ? (size_t)NK : (size_t)N;
    float s = 0.0f;
    if (is_w || is_b) {
        int c = wave;
        for (; c + 28 < chunks; c += 32) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = src[(size_t)(c + 4 * j) * stride];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j];
        }
        for (; c < chunks; c += 4) s += src[(size_t)c * stride];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        const float tot = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (is_w) dw[e] = tot;
        else if (is_b) db[e - NK] = tot;
    }
}

struct WgradPlan {
    int kt, ntw, ngroups, chunks;
    long long rows_per_block;
};

// round 3: the bf16 three-way-split kernels (same results to float32 rounding); SELFOCC_LINEAR_B3=0 keeps the f32-MFMA ones
bool so_wgrad_b3() {
    static const bool on = !(getenv("SELFOCC_LINEAR_B3") && atoi(getenv("SELFOCC_LINEAR_B3")) == 0);
    return on;
}

bool so_wgrad_plan(long long T, int N, int K, WgradPlan &p) {
    static const int env_chunks = getenv("SELFOCC_WGRAD_CHUNKS") ? atoi(getenv("SELFOCC_WGRAD_CHUNKS")) : 0;     // dev A/B
    const bool b3 = so_wgrad_b3();
    // tile rows of dY per wave (NTW).  bf16x3 kernels, K = 96: one or two (round 3, scripts/micro/wgrad_bench.py: N <= 96 had
    // 128 blocks = one wave on half of the SIMDs with three; 40 -> 28 us at 66 049 x 96 x 96, 108 -> 84 us at 153 000 x 288 x 96
    // with two); K = 192: one (222 registers; 66 -> 44 us at 78 899 x 96 x 192); K = 128 stays on the f32-MFMA kernel
    if (K == 96) { p.kt = 3; p.ntw = b3 ? ((N <= 96 || T < 16384) ? 1 : 2) : 3; }
    else if (K == 192) { p.kt = 6; p.ntw = b3 ? 1 : 2; }
    else if (K == 32) { p.kt = 1; p.ntw = 4; }
    else if (K == 64) { p.kt = 2; p.ntw = b3 ? 2 : 4; }
    else if (K == 128) { p.kt = 4; p.ntw = 3; }
    else return false;
    if (T < 1 || N < 1 || (long long)N * K >= (1LL << 30)) return false;
    const int nt = (N + 31) / 32;
    p.ngroups = (nt + p.ntw - 1) / p.ntw;
    // ~2 blocks per CU, at most 128 partial sums per element; at least 64 rows per block (16 per wave)
    const int target = env_chunks > 0 ? 4 * env_chunks : 512;
    long long chunks = std::max(1LL, std::min((long long)(target / std::max(1, std::min(p.ngroups, target))), (T + 63) / 64));
    chunks = std::min(chunks, env_chunks > 0 ? (long long)env_chunks : 128LL);
    long long rpb = (T + chunks - 1) / chunks;
    rpb = (rpb + 7) / 8 * 8;                          // rows per wave a multiple of 2
    chunks = (T + rpb - 1) / rpb;
    p.chunks = (int)chunks;
    p.rows_per_block = rpb;
    return true;
}

}  // namespace

extern "C" int selfocc_linear_wgrad_supported(int64_t T, int32_t N, int32_t K) {
    WgradPlan p;
    return so_wgrad_plan(T, N, K, p) ? 1 : 0;
}

extern "C" size_t selfocc_linear_wgrad_workspace(int64_t T, int32_t N, int32_t K) {
    WgradPlan p;
    if (!so_wgrad_plan(T, N, K, p)) return 0;
    return (size_t)p.chunks * ((size_t)N * K + N) * sizeof(float);
}

extern "C" int selfocc_linear_wgrad(const float *dy, const float *x, float *dw, float *db, int64_t T, int32_t N,
                                    int32_t K, void *workspace, size_t workspace_bytes, void *stream) {
    WgradPlan p;
    SO_REQUIRE(so_wgrad_plan(T, N, K, p), "linear_wgrad: unsupported shape (T = %lld rows, N = %d, K = %d; K must be 32, 64, "
               "96, 128 or 192)", (long long)T, N, K);
    SO_REQUIRE(dy && x && dw, "linear_wgrad: NULL pointer");
    const size_t need = (size_t)p.chunks * ((size_t)N * K + N) * sizeof(float);
    SO_REQUIRE(workspace != nullptr && workspace_bytes >= need, "linear_wgrad: workspace too small (%zu bytes, need %zu)",
               workspace_bytes, need);
    float *part_w = (float *)workspace, *part_b = part_w + (size_t)p.chunks * N * K;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)p.chunks, (unsigned)p.ngroups);
    const bool use_b3 = so_wgrad_b3();
#define SO_LAUNCH_F32(KT_, NTW_)                                                                                   \
    hipLaunchKernelGGL((linear_wgrad_kernel<KT_, NTW_>), grid, dim3(256), 0, st, dy, x, part_w, part_b, T, N, K, p.rows_per_block)
#define SO_LAUNCH_B3(KT_, NTW_)                                                                                    \
    hipLaunchKernelGGL((linear_wgrad_b3_kernel<KT_, NTW_>), grid, dim3(256), 0, st, dy, x, part_w, part_b, T, N, K, p.rows_per_block)
    // (KT, NTW) pairs of the plan; only the pairs a plan can produce are instantiated (the b3 kernel at 3 x 3 / 4 x 3 / 6 x 2
    // accumulator tiles needs more than 256 registers)
    switch (p.kt) {
        case 1: if (use_b3) SO_LAUNCH_B3(1, 4); else SO_LAUNCH_F32(1, 4); break;
        case 2: if (use_b3) SO_LAUNCH_B3(2, 2); else SO_LAUNCH_F32(2, 4); break;
        case 3:
            if (!use_b3) SO_LAUNCH_F32(3, 3);
            else if (p.ntw == 1) SO_LAUNCH_B3(3, 1);
            else SO_LAUNCH_B3(3, 2);
            break;
        case 4: SO_LAUNCH_F32(4, 3); break;
        default: if (use_b3) SO_LAUNCH_B3(6, 1); else SO_LAUNCH_F32(6, 2); break;
    }
#undef SO_LAUNCH_F32
#undef SO_LAUNCH_B3
    const int NK = N * K;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((NK + N + 63) / 64)), dim3(256), 0, st, part_w, part_b,
                       dw, db, p.chunks, NK, N);
    return so_launch_status();
}
