// msda_pro.hip — deformable attention with its two query linears in the kernel prologue (inference).
//
// The reference computes, per deformable attention (bevformer/attention/image_cross_attention.py:296-345,
// tpvformer/attention/cross_view_hybrid_attention.py:78-116):
//     off    = sampling_offsets(query)    (nq, heads * L * P * 2)      — a Linear
//     logits = attention_weights(query)   (nq, heads * L * P)          — a Linear
//     A = softmax(logits), loc = ref + off / (W_l, H_l), out = MSDA(value, loc, A)
// i.e. it writes 3 * heads * L * P floats per query to memory (152 MB on the hw plane of nuscenes_occ, 204 MB for the
// cross-view self-attention, per layer) only to read them back in the sampling kernel.  Here the two linears are an
// MFMA tile in the sampling kernel's prologue (on the bf16 matrix pipe through an exact three-way split: float32 accuracy): a (query, head) needs 3 * L * P of those outputs, a block works on
// ONE head (head-outer order, as in msda.hip), so only that head's 3 * L * P rows of the weights are needed — 96 rows
// x 96 inputs = 36 KB on the hw plane — and they stay in LDS for the lifetime of a persistent 16-wave block.
//
//   per wave, per tile of 16 consecutive queries:
//     1. x tile (16 x 96) straight into the MFMA A layout (lane (row, k block): 8 consecutive k per 32-k step), split into
//        three bfloat16 parts in registers; W^T (three bf16 planes) from LDS: six v_mfma_f32_16x16x32_bf16 per 32 k and
//        16-column tile, NT16 = ceil(3 L P / 16) tiles;
//     2. bias added, the 16 x 3LP result parked in a wave-private LDS tile (accumulator layout -> row layout);
//     3. the sampling stage of msda_fused_fwd_kernel / msda_cross_fwd_kernel unchanged — softmax over the group's logits,
//        offsets / (W_l, H_l), bilinear set-up, channel-team gathers, camera loop, group reduce — reading its logits
//        and offsets from that tile instead of global memory.
// The split is exact (csrc/linear_fwd.hip: linear_fwd_b3_kernel), so off / logits differ from the separate Linear by float32
// rounding only.  (The first version used f32 MFMA: same results, but f32 MFMA runs at the vector rate and did not hide
// under the gathers — 381 / 387 us per call instead of 352 / 322)
// Applies where a head's weight slice (3 planes x 3LP x 208 B) fits LDS next to 16 (12) result tiles: 3 L P <= 112 (hw
// plane: 96, self-attention: 108); the zh / wz planes (3 L P = 576: 221 KB per head, 6 425 queries) keep the separate linears.
#include "so_device.h"
#include <algorithm>
#include <atomic>

namespace {

#include "msda_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MsdaProArgs {
    const void *value;
    const int32_t *shapes, *starts;
    const float *ref;          // cross: (cams, nq, P, 2); fused: per ref_kind
    const uint8_t *vis;        // cross: (cams, nq)
    const float *x;            // (bs * nq, 96): the rows the two linears are applied to
    const float *w_off, *b_off, *w_aw, *b_aw;      // (heads*L*P*2, 96), (heads*L*P*2), (heads*L*P, 96), (heads*L*P)
    float *out;
    int ref_kind, cams, nbh, n_tiles;
    MsdaDims dm;
};

constexpr int kProK = 96, kProKPB = 104;       // K; bf16 elements per LDS weight row (208 B: 16-byte reads on distinct banks)

// float32 = exact sum of three bfloat16 (csrc/linear_fwd.hip: linear_fwd_b3_kernel): the prologue's product runs on the
// bf16 matrix pipe as six MFMAs per 32 k — float32-level accuracy, 2.67 x fewer matrix-pipe cycles than f32 MFMA, and,
// unlike f32 MFMA (which executes at the vector rate and showed no overlap), off the lanes the other waves' gathers need
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
SO_DEVFN void so_pro_split3(const float (&x)[8], bf16x8 &a1, bf16x8 &a2, bf16x8 &a3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 b1 = (__bf16)x[j];
        const float r1 = x[j] - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const float r2 = r1 - (float)b2;
        a1[j] = b1; a2[j] = b2; a3[j] = (__bf16)r2;
    }
}

template <int D, int LOGG, int NT16, bool CROSS, typename VT, int kProWaves>
__global__ __launch_bounds__(kProWaves * 64) void msda_pro_fwd_kernel(MsdaProArgs a) {
    constexpr int K = kProK, KPB = kProKPB, NROW = NT16 * 16;
    constexpr int G = 1 << LOGG;
    constexpr int QL = D / 4, LOGQ = so_ilog2(QL);
    constexpr int NJ = LOGG > LOGQ ? LOGG - LOGQ : 0;
    constexpr int MAXR = so_maxr(LOGG);
    constexpr int RS = NT16 * 16 + 1;                    // row stride of the result tile (odd: rows on different banks)
    constexpr int GPW = 64 / G;                          // (query, head) groups a wave samples at once
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16 *wb = (__bf16 *)lds;                          // [3 planes][NT16 * 16][KPB]: this head's rows of W_off then W_aw
    __shared__ int s_next;
    const MsdaDims dm = a.dm;
    const int LP = dm.L * dm.P, NC = 3 * LP;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float *outl = lds + (3 * NROW * KPB * 2) / 4 + wave * (16 * RS);
    const unsigned lb = so_xcd_block();
    const int h = (int)(lb / (unsigned)a.nbh), bi = (int)(lb - (unsigned)h * a.nbh);
    // this block's contiguous range of 16-query tiles
    const int t_lo = (int)((long long)a.n_tiles * bi / a.nbh), t_hi = (int)((long long)a.n_tiles * (bi + 1) / a.nbh);
    if (threadIdx.x == 0) s_next = t_lo + kProWaves;     // tiles t_lo .. t_lo + 15 are the waves' first ones
    const long long T = (long long)dm.bs * dm.nq;        // rows of x

    // ---- stage the head's weight rows (column c < 2 LP -> W_off row h * 2LP + c, else W_aw row h * LP + c - 2LP), split
    // into the three bf16 planes: 8 consecutive k of one row per thread and step ----
    {
        constexpr int NV = NROW * (K / 8);
        for (int idx = threadIdx.x; idx < NV; idx += kProWaves * 64) {
            const int r = idx / (K / 8), k8 = idx - r * (K / 8);
            const float *src = nullptr;
            if (r < 2 * LP) src = a.w_off + (size_t)(h * 2 * LP + r) * K;
            else if (r < NC) src = a.w_aw + (size_t)(h * LP + r - 2 * LP) * K;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (src) {
                const float4 lo = ((const float4 *)src)[2 * k8], hi = ((const float4 *)src)[2 * k8 + 1];
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            }
            bf16x8 w1, w2, w3;
            so_pro_split3(v, w1, w2, w3);
            *(bf16x8 *)(wb + (size_t)r * KPB + 8 * k8) = w1;
            *(bf16x8 *)(wb + (size_t)(NROW + r) * KPB + 8 * k8) = w2;
            *(bf16x8 *)(wb + (size_t)(2 * NROW + r) * KPB + 8 * k8) = w3;
        }
    }
    const int n = lane & 15, kq = lane >> 4;     // MFMA lane roles: row / column n, k block (bf16 operands) = row group (result) kq
    // the head's biases, one per result column, behind the waves' result tiles (in registers they were the 7 that spilled)
    float *bias_s = lds + (3 * NROW * KPB * 2) / 4 + kProWaves * (16 * RS);
    for (int c = threadIdx.x; c < NROW; c += kProWaves * 64)
        bias_s[c] = c < 2 * LP ? a.b_off[h * 2 * LP + c] : (c < NC ? a.b_aw[h * LP + c - 2 * LP] : 0.0f);
    __syncthreads();

    const int pix_stride = so_pix_stride(dm, D);
    const int gl = lane & (G - 1), gi = lane >> LOGG;    // lane within its group, group within the wave
    const int s = gl & (QL - 1);
    int cam_stride = 0;
    if constexpr (CROSS) cam_stride = (int)(so_value_base(dm, D, 1, h, 0) - so_value_base(dm, D, 0, h, 0));

    for (int tile = t_lo + wave; tile < t_hi;) {
        const long long r0 = (long long)tile * 16;
        const int rm = (int)min(16LL, T - r0);
        // ---- 1. off / logits of the tile's 16 queries for head h: six bf16 MFMAs per 32 k (exact three-way split) ----
        {
            bf16x8 a1[3], a2[3], a3[3];
            {
                const float *xb = a.x + (r0 + min(n, rm - 1)) * K + 8 * kq;
                float xr[3][8];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const float4 lo = *(const float4 *)(xb + 32 * ks), hi = *(const float4 *)(xb + 32 * ks + 4);
                    xr[ks][0] = lo.x; xr[ks][1] = lo.y; xr[ks][2] = lo.z; xr[ks][3] = lo.w;
                    xr[ks][4] = hi.x; xr[ks][5] = hi.y; xr[ks][6] = hi.z; xr[ks][7] = hi.w;
                }
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) so_pro_split3(xr[ks], a1[ks], a2[ks], a3[ks]);
            }
            const __bf16 *bbase = wb + (size_t)n * KPB + 8 * kq;
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const __bf16 *bp = bbase + (size_t)(16 * t) * KPB + 32 * ks;
                    const bf16x8 b1 = *(const bf16x8 *)bp, b2 = *(const bf16x8 *)(bp + (size_t)NROW * KPB),
                                 b3 = *(const bf16x8 *)(bp + (size_t)2 * NROW * KPB);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[ks], b1, acc, 0, 0, 0);      // small terms first
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ks], b3, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[ks], b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[ks], b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ks], b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ks], b1, acc, 0, 0, 0);
                }
                // ---- 2. accumulator layout (lane: column n of rows 4 kq + j) -> row-major result tile in LDS ----
                const float bvt = bias_s[16 * t + n];
#pragma unroll
                for (int j = 0; j < 4; ++j) outl[(4 * kq + j) * RS + 16 * t + n] = acc[j] + bvt;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- 3. sampling: GPW (query, head) groups per pass ----
        for (int pass = 0; pass < 16 / GPW; ++pass) {
            const int ql = pass * GPW + gi;
            const bool live = ql < rm;
            const long long bq = r0 + (live ? ql : 0);                 // row of x = b * nq + q
            const int b = CROSS ? 0 : (int)(bq / dm.nq);
            const int q = (int)(bq - (long long)b * dm.nq);
            const long long gq = bq * dm.heads + h;
            const float *orow = outl + (live ? ql : 0) * RS;
            const VT *vb = (const VT *)a.value + so_value_base(dm, D, b, h, 0) + 4 * s;

            float lg[MAXR], ox[MAXR], oy[MAXR];
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int pt = gl + r * G;
                lg[r] = (pt < LP) ? orow[2 * LP + pt] : -INFINITY;
                mx = fmaxf(mx, lg[r]);
            }
#pragma unroll
            for (int m = 1; m < G; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
            float den = 0.0f;
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int pt = gl + r * G;
                lg[r] = (pt < LP) ? __expf(lg[r] - mx) : 0.0f;
                den += lg[r];
                ox[r] = oy[r] = 0.0f;
                if (pt < LP) {
                    const int l = so_level_of(pt, dm.P, dm.L);
                    ox[r] = orow[2 * pt] / (float)a.shapes[2 * l + 1];
                    oy[r] = orow[2 * pt + 1] / (float)a.shapes[2 * l];
                }
            }
#pragma unroll
            for (int m = 1; m < G; m <<= 1) den += __shfl_xor(den, m, 64);
            const float iden = 1.0f / den;

            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (CROSS) {
                int count = 0;
                for (int cam = 0; cam < a.cams; ++cam) {
                    const bool seen = live && a.vis[(size_t)cam * dm.nq + q] != 0;
                    count += seen ? 1 : 0;
                    if (!__any(seen)) continue;
                    const int cam_off = cam * cam_stride;
#pragma unroll
                    for (int r = 0; r < MAXR; ++r) {
                        if (r * G >= LP) break;   // uniform
                        const int pt = gl + r * G;
                        MsdaPoint mp = so_point_none();
                        if (seen && pt < LP) {
                            const int l = so_level_of(pt, dm.P, dm.L);
                            const int pp = pt - l * dm.P;
                            const float2 rf = *(const float2 *)(a.ref + 2 * (((size_t)cam * dm.nq + q) * dm.P + pp));
                            mp = so_point_setup(rf.x + ox[r], rf.y + oy[r], lg[r] * iden, a.shapes[2 * l], a.shapes[2 * l + 1],
                                                cam_off + a.starts[l] * pix_stride, pix_stride);
                        }
                        if constexpr (LOGG >= 5) {
                            const int steps = so_compact_points<D, LOGG>(mp);
                            so_team_gather_steps<D>(vb, mp, acc, steps);
                        } else {
                            so_team_gather<D>(vb, mp, acc);
                        }
                    }
                }
                const float cnt = (float)max(count, 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = acc[c] / cnt;
            } else {
#pragma unroll
                for (int r = 0; r < MAXR; ++r) {
                    if (r * G >= LP) break;   // uniform
                    const int pt = gl + r * G;
                    MsdaPoint mp = so_point_none();
                    if (live && pt < LP) {
                        const int l = so_level_of(pt, dm.P, dm.L);
                        const int pp = pt - l * dm.P;
                        size_t ri;
                        if (a.ref_kind == 1) ri = (size_t)bq * dm.P + pp;
                        else if (a.ref_kind == 2) ri = ((size_t)bq * dm.L + l) * dm.P + pp;
                        else ri = (size_t)bq * dm.L + l;
                        const float2 rf = *(const float2 *)(a.ref + 2 * ri);
                        mp = so_point_setup(rf.x + ox[r], rf.y + oy[r], lg[r] * iden, a.shapes[2 * l], a.shapes[2 * l + 1],
                                            a.starts[l] * pix_stride, pix_stride);
                    }
                    so_team_gather<D>(vb, mp, acc);
                }
            }
            so_group_reduce_store<NJ, LOGQ>(acc, gl >> LOGQ, live, a.out + (size_t)gq * D + 4 * s);
        }
        // the next tile's result must not overtake this tile's reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // dynamic tile hand-out inside the block's range: waves that drew short camera loops take more tiles
        int nt = 0;
        if (lane == 0) nt = atomicAdd(&s_next, 1);
        tile = __builtin_amdgcn_readfirstlane(nt);
    }
}

int so_num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus;
}

bool so_pro_shape_ok(int heads, int d, int L, int P, int K, int &logG, int &nt16) {
    if (d != 16 || K != kProK || heads < 1 || L < 1 || P < 1) return false;
    const int LP = L * P;
    nt16 = (3 * LP + 15) / 16;
    if (nt16 != 6 && nt16 != 7) return false;
    // group size as in the separate kernels (msda.hip: so_pick_group_fused)
    int best_l = 6;
    double best_u = -1.0;
    for (int lg = 6; lg >= 0; --lg) {
        const int g = 1 << lg;
        if (g < d / 4) break;
        const int rounds = (LP + g - 1) / g;
        if (rounds > so_maxr(lg)) { if (lg >= 4) continue; else break; }
        const double u = (double)LP / ((double)rounds * g);
        if (u > best_u + 1e-12) { best_u = u; best_l = lg; }
    }
    logG = best_l;
    return (logG == 5 && nt16 == 6) || (logG == 3 && nt16 == 7);      // the two shipped shapes are instantiated
}

}  // namespace

extern "C" int selfocc_msda_pro_supported(int32_t heads, int32_t d, int32_t L, int32_t P, int32_t K) {
    int lg, nt;
    return so_pro_shape_ok(heads, d, L, P, K, lg, nt) ? 1 : 0;
}

extern "C" int selfocc_msda_pro_fwd(const void *value, const int32_t *shapes, const int32_t *starts, const float *ref,
                                    int32_t ref_kind, const uint8_t *vis, const float *x, const float *w_off,
                                    const float *b_off, const float *w_aw, const float *b_aw, float *out, int32_t cams,
                                    int32_t bs, int32_t nv, int32_t nq, int32_t heads, int32_t d, int32_t L, int32_t P,
                                    int32_t K, int32_t value_stride, int32_t value_layout, int32_t value_dtype,
                                    void *stream) {
    const bool cross = vis != nullptr;
    int logG = 0, nt16 = 0;
    SO_REQUIRE(so_pro_shape_ok(heads, d, L, P, K, logG, nt16),
               "msda_pro_fwd: unsupported shape (d = %d, K = %d, L * P = %d): needs d = 16, K = 96 and 3 L P in (80, 112]", d, K,
               L * P);
    SO_REQUIRE(value && shapes && starts && ref && x && w_off && b_off && w_aw && b_aw && out, "msda_pro_fwd: NULL pointer");
    SO_REQUIRE(value_dtype == SO_DTYPE_F32 || value_dtype == SO_DTYPE_BF16, "msda_pro_fwd: bad value_dtype");
    SO_REQUIRE(value_layout == SO_VALUE_PIXEL_MAJOR || (value_layout == SO_VALUE_HEAD_MAJOR && value_stride == 0),
               "msda_pro_fwd: bad value_layout");
    SO_REQUIRE(value_stride == 0 || (value_stride >= heads * d && value_stride % 4 == 0), "msda_pro_fwd: bad value_stride");
    SO_REQUIRE(cross ? (cams >= 1 && bs == 1) : (ref_kind >= 0 && ref_kind <= 2 && bs >= 1),
               "msda_pro_fwd: camera loop needs cams >= 1 and bs == 1; the plain form a ref_kind in 0..2");
    SO_REQUIRE(nq >= 0 && nv >= 0, "msda_pro_fwd: negative size");
    const long long T = (long long)bs * nq;
    if (T == 0) return 0;
    SO_REQUIRE(T * heads < (1LL << 31), "msda_pro_fwd: bs * nq * heads must be < 2^31");
    SO_REQUIRE((long long)(cross ? cams : bs) * nv * (value_stride ? value_stride : heads * d) < (1LL << 31),
               "msda_pro_fwd: value must span < 2^31 floats");
    hipStream_t st = (hipStream_t)stream;
    if (nv == 0) return (int)hipMemsetAsync(out, 0, (size_t)T * heads * d * sizeof(float), st);
    MsdaProArgs a;
    a.value = value; a.shapes = shapes; a.starts = starts; a.ref = ref; a.vis = vis; a.x = x;
    a.w_off = w_off; a.b_off = b_off; a.w_aw = w_aw; a.b_aw = b_aw; a.out = out;
    a.ref_kind = ref_kind; a.cams = cams;
    a.n_tiles = (int)((T + 15) / 16);
    a.nbh = std::max(1, std::min(so_num_cus() / heads, a.n_tiles));
    a.dm = MsdaDims{bs, nv, nq, heads, L, P, 0, value_stride, value_layout};
    const unsigned blocks = (unsigned)(a.nbh * heads);
#define SO_LAUNCH_PRO(LG, NT, CR, VTT, WV)                                                                              \
    do {                                                                                                                \
        const size_t shm = (size_t)3 * NT * 16 * kProKPB * 2 + (size_t)WV * 16 * (NT * 16 + 1) * sizeof(float) +        \
                           (size_t)NT * 16 * sizeof(float);         \
        static std::atomic<unsigned long long> done_mask{0};      /* per-device attribute: one driver call per device */ \
        int dev_ = 0;                                                                                                   \
        (void)hipGetDevice(&dev_);                                                                                      \
        if (!(done_mask.load(std::memory_order_relaxed) & (1ull << (dev_ & 63)))) {                                     \
            (void)hipFuncSetAttribute((const void *)msda_pro_fwd_kernel<16, LG, NT, CR, VTT, WV>,                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);                    \
            done_mask.fetch_or(1ull << (dev_ & 63), std::memory_order_relaxed);                                         \
        }                                                                                                               \
        hipLaunchKernelGGL((msda_pro_fwd_kernel<16, LG, NT, CR, VTT, WV>), dim3(blocks), dim3(WV * 64), shm, st, a);    \
    } while (0)
#define SO_LAUNCH_PRO_V(LG, NT, CR, WV)                                          \
    do {                                                                         \
        if (value_dtype == SO_DTYPE_BF16) SO_LAUNCH_PRO(LG, NT, CR, uint16_t, WV); \
        else SO_LAUNCH_PRO(LG, NT, CR, float, WV);                               \
    } while (0)
    // waves per block: what the 160 KB of LDS leave next to the three weight planes (16 at 3LP <= 96, 12 at <= 112)
    if (logG == 5) { if (cross) SO_LAUNCH_PRO_V(5, 6, true, 16); else SO_LAUNCH_PRO_V(5, 6, false, 16); }
    else { if (cross) SO_LAUNCH_PRO_V(3, 7, true, 12); else SO_LAUNCH_PRO_V(3, 7, false, 12); }
#undef SO_LAUNCH_PRO_V
#undef SO_LAUNCH_PRO
    return so_launch_status();
}
