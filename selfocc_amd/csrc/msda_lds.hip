// msda_lds.hip — camera-loop deformable attention (inference) with the COARSE FPN levels of `value` staged in LDS.
//
// The zh / wz planes of the TPV lifter (bevformer/attention/image_cross_attention.py:46-139 through
// tpvformer/attention/image_cross_attention.py:71-95) sample, per (query, head, camera), 48 pillar points on each of 4
// FPN levels.  Counters of the camera-loop kernel on this workload (profiles/r2_i_l1_pmc.txt): vector-L1 hit rate 77 - 83 %,
// the L1s stalled on pending misses 48 % of the kernel — the gathers wait for lines, they are not short of bytes.  Half
// of the points go to the two coarsest levels (24x50 and 12x25 pixels at the shipped size: 1 500 pixels x 64 B per
// (camera, head) = 96 KB), which fit the CU's 160 KB LDS.  So the work is re-cut per (CAMERA, HEAD) pair:
//   * a 16-wave block stages levels >= `lds_level0` of value[cam][head] in LDS once (head-major value: one contiguous
//     run), then walks the queries that camera sees (dealt round-robin to the pair's blocks: visibility is spatially
//     coherent, contiguous ranges would be unequal); a wave = one (query, head) group, skipped in one test when the
//     camera does not see the query;
//   * the lane <-> point map is one LEVEL per round (lanes 0 .. P-1 own the level's P points), so the source of a
//     round's gathers is wave-uniform: rounds of the fine levels gather from global memory as before (channel teams,
//     inside-point compaction), rounds of the staged levels read their 16-byte corner quarters with ds_read_b128 and
//     never touch the texture addresser / vector L1 — whose capacity is then left to the two fine levels;
//   * the per-camera result goes to partial[cam][q][head*16 ..]; a second, tiny kernel adds the visible cameras in
//     camera order and divides by their count (image_cross_attention.py:129-136) — deterministic, no atomics.
// Same arithmetic per (query, head, camera) as msda_cross_fwd_kernel; the camera sum is associated per camera instead
// of point by point (differences at the 1e-7 level).
//
// MEASURED AND NOT USED BY DEFAULT (round 3; profiles/r3_d_*): zh plane of the eval encoder, 6 425 queries
// x 6 heads x 6 cameras: 312 us per call against 204 us for the query-major msda_cross_fwd_kernel.  Built with every level
// gathered from global memory (-DSO_LDS_FORCE_GLOBAL) the same re-cut takes 329 us: the LDS staging is worth 5 %, the
// (camera, head)-stationary structure it requires — 252 sixteen-wave blocks, one (query, head) per wave, the softmax
// redone per camera, four level rounds at 75 % lane use instead of three full ones — costs 60 %.  The coarse levels were
// never where the gathers wait: 19 + 77 KB per (camera, head) sit in the vector L1 / L2 already; the misses come from the
// 96x200 and 48x100 levels (1.5 MB per pair), which no LDS holds.  Kept as an A/B (SELFOCC_MSDA_LDS=1) with its parity test.
#include "so_device.h"
#include <algorithm>

namespace {

#include "msda_device.h"

struct MsdaLdsArgs {
    const float *value;          // (cams, heads, nv, 16) head-major float32
    const int32_t *shapes, *starts;
    const float *ref;            // (cams, nq, P, 2)
    const uint8_t *vis;          // (cams, nq)
    const float *off_raw;        // (nq, heads, L, P, 2)
    const float *logits;         // (nq, heads, L * P)
    float *partial;              // (cams, nq, heads * 16)
    int cams, nq, heads, L, P, nv;
    int lds_level0, lds_px0, lds_px;   // first staged level, its first pixel, number of staged pixels
    int nbp;                     // blocks per (camera, head) pair
};

constexpr int kLdsWaves = 16;

// one team step from LDS: the 16-byte channel quarter of each corner with ds_read_b128
template <int I>
SO_DEVFN void so_team_step_lds(const float *lb, const MsdaPoint &mp, float (&acc)[4]) {
    constexpr int QL = 4;
    const float aw = so_team_bcastf<QL, I>(mp.aw);
    float val[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int off = so_team_bcast<QL, I>(mp.off[k]);
        const float w = so_team_bcastf<QL, I>(mp.w[k]);
        const float4 t = *(const float4 *)(lb + off);
        val[0] = fmaf(w, t.x, val[0]);
        val[1] = fmaf(w, t.y, val[1]);
        val[2] = fmaf(w, t.z, val[2]);
        val[3] = fmaf(w, t.w, val[3]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = fmaf(aw, val[c], acc[c]);
}

SO_DEVFN void so_team_gather_steps_lds(const float *lb, const MsdaPoint &mp, float (&acc)[4], int steps) {
    if (steps > 0) so_team_step_lds<0>(lb, mp, acc);
    if (steps > 1) so_team_step_lds<1>(lb, mp, acc);
    if (steps > 2) so_team_step_lds<2>(lb, mp, acc);
    if (steps > 3) so_team_step_lds<3>(lb, mp, acc);
}

template <int MAXL>
__global__ __launch_bounds__(kLdsWaves * 64) void msda_cross_lds_fwd_kernel(MsdaLdsArgs a) {
    constexpr int D = 16, LOGG = 6, QL = 4, LOGQ = 2, NJ = 4;
    extern __shared__ __attribute__((aligned(16))) float lds_val[];       // [lds_px][16]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pair = blockIdx.x / a.nbp, bi = blockIdx.x - pair * a.nbp;
    const int cam = pair / a.heads, h = pair - cam * a.heads;
    const float *vch = a.value + ((size_t)(cam * a.heads + h) * a.nv) * D;      // this (camera, head)'s map, all levels
    // ---- stage the coarse levels: one contiguous run of lds_px * 16 floats ----
    {
        const float4 *src = (const float4 *)(vch + (size_t)a.lds_px0 * D);
        float4 *dst = (float4 *)lds_val;
        const int n4 = a.lds_px * (D / 4);
        for (int i = threadIdx.x; i < n4; i += kLdsWaves * 64) dst[i] = src[i];
    }
    __syncthreads();

    const int LP = a.L * a.P;
    const int s = lane & (QL - 1);
    const float *vb = vch + 4 * s;                   // global gathers: element offsets include starts[l] * 16
    const float *lb = lds_val + 4 * s;               // LDS gathers: offsets relative to the first staged pixel
    const bool own = lane < a.P;                     // lanes 0 .. P-1 own the P points of the round's level
    float shw[MAXL], shh[MAXL];
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        shh[l] = l < a.L ? (float)a.shapes[2 * l] : 1.0f;
        shw[l] = l < a.L ? (float)a.shapes[2 * l + 1] : 1.0f;
    }

    for (int q = bi * kLdsWaves + wave; q < a.nq; q += a.nbp * kLdsWaves) {
        if (a.vis[(size_t)cam * a.nq + q] == 0) continue;                       // wave-uniform: one (query, head) per wave
        const int gq = q * a.heads + h;
        // softmax over the group's L * P logits (lane: point `lane` of every level) and the raw offsets / (W_l, H_l)
        float lg[MAXL], ox[MAXL], oy[MAXL];
        float mx = -INFINITY;
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            const bool on = own && l < a.L;
            lg[l] = on ? a.logits[(size_t)gq * LP + l * a.P + lane] : -INFINITY;
            mx = fmaxf(mx, lg[l]);
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        float den = 0.0f;
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            const bool on = own && l < a.L;
            lg[l] = on ? __expf(lg[l] - mx) : 0.0f;
            den += lg[l];
            ox[l] = oy[l] = 0.0f;
            if (on) {
                const float2 o = *(const float2 *)(a.off_raw + 2 * ((size_t)gq * LP + l * a.P + lane));
                ox[l] = o.x / shw[l];
                oy[l] = o.y / shh[l];
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) den += __shfl_xor(den, m, 64);
        const float iden = 1.0f / den;
        float2 rf = make_float2(0.0f, 0.0f);
        if (own) rf = *(const float2 *)(a.ref + 2 * (((size_t)cam * a.nq + q) * a.P + lane));

        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            if (l >= a.L) break;                                                // uniform
#ifdef SO_LDS_FORCE_GLOBAL      /* A/B: the re-cut per (camera, head) alone, every level gathered from global memory */
            const bool staged = false;
#else
            const bool staged = l >= a.lds_level0;                              // uniform
#endif
            MsdaPoint mp = so_point_none();
            if (own) {
                const int level_off = (staged ? a.starts[l] - a.lds_px0 : a.starts[l]) * D;
                mp = so_point_setup(rf.x + ox[l], rf.y + oy[l], lg[l] * iden, (int)shh[l], (int)shw[l], level_off, D);
            }
            const int steps = so_compact_points<D, LOGG>(mp);
            if (staged) so_team_gather_steps_lds(lb, mp, acc, steps);
            else so_team_gather_steps<D>(vb, mp, acc, steps);
        }
        so_group_reduce_store<NJ, LOGQ>(acc, lane >> LOGQ, true, a.partial + ((size_t)cam * a.nq + q) * (a.heads * D) + h * D + 4 * s);
    }
}

// out[q][c] = sum over the cameras that see q (camera order) of partial[cam][q][c], / max(#cameras, 1)
__global__ __launch_bounds__(256) void msda_cross_lds_reduce_kernel(const float *__restrict__ partial, const uint8_t *__restrict__ vis,
                                                                    float *__restrict__ out, int cams, int nq, int width4) {
    const int idx = blockIdx.x * 256 + threadIdx.x;            // one float4 of one query
    if (idx >= nq * width4) return;
    const int q = idx / width4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int count = 0;
    for (int cam = 0; cam < cams; ++cam) {
        if (vis[(size_t)cam * nq + q] == 0) continue;
        const float4 v = ((const float4 *)partial)[(size_t)cam * nq * width4 + idx];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        ++count;
    }
    const float c = (float)max(count, 1);
    ((float4 *)out)[idx] = make_float4(s.x / c, s.y / c, s.z / c, s.w / c);
}

// first level staged in LDS so that the staged run fits `budget` bytes; returns L when nothing fits
int so_lds_first_level(const int32_t *host_shapes, int L, size_t budget, int &px0, int &px) {
    int total = 0;
    for (int l = 0; l < L; ++l) total += host_shapes[2 * l] * host_shapes[2 * l + 1];
    int start = 0;
    for (int l = 0; l < L; ++l) {
        const int rest = total - start;
        if ((size_t)rest * 64 <= budget) { px0 = start; px = rest; return l; }
        start += host_shapes[2 * l] * host_shapes[2 * l + 1];
    }
    px0 = total; px = 0;
    return L;
}

int so_lds_num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus;
}

constexpr size_t kLdsBudget = 128 * 1024;

}  // namespace

extern "C" int selfocc_msda_cross_lds_supported(const int32_t *host_shapes, int32_t heads, int32_t d, int32_t L, int32_t P) {
    if (!host_shapes || d != 16 || L < 1 || L > 4 || P < 33 || P > 64 || heads < 1) return 0;
    int px0, px;
    return so_lds_first_level(host_shapes, L, kLdsBudget, px0, px) < L ? 1 : 0;
}

extern "C" size_t selfocc_msda_cross_lds_workspace(int32_t cams, int32_t nq, int32_t heads, int32_t d) {
    return (size_t)cams * nq * heads * d * sizeof(float);
}

extern "C" int selfocc_msda_cross_lds_fwd(const float *value, const int32_t *shapes, const int32_t *starts,
                                          const int32_t *host_shapes, const float *ref, const uint8_t *vis,
                                          const float *off_raw, const float *logits, float *out, int32_t cams, int32_t nv,
                                          int32_t nq, int32_t heads, int32_t d, int32_t L, int32_t P, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    SO_REQUIRE(selfocc_msda_cross_lds_supported(host_shapes, heads, d, L, P),
               "msda_cross_lds_fwd: unsupported shape (needs d = 16, L <= 4, 33 <= P <= 64 and a coarse level that fits LDS)");
    SO_REQUIRE(value && shapes && starts && ref && vis && off_raw && logits && out, "msda_cross_lds_fwd: NULL pointer");
    SO_REQUIRE(cams >= 1 && nq >= 0 && nv >= 1, "msda_cross_lds_fwd: bad sizes");
    if (nq == 0) return 0;
    SO_REQUIRE((long long)cams * nv * heads * d < (1LL << 31) && (long long)nq * heads * L * P < (1LL << 31),
               "msda_cross_lds_fwd: tensors must span < 2^31 elements");
    SO_REQUIRE(workspace && workspace_bytes >= selfocc_msda_cross_lds_workspace(cams, nq, heads, d),
               "msda_cross_lds_fwd: workspace too small");
    int total = 0;
    for (int l = 0; l < L; ++l) total += host_shapes[2 * l] * host_shapes[2 * l + 1];
    SO_REQUIRE(total == nv, "msda_cross_lds_fwd: host_shapes do not add up to nv");
    MsdaLdsArgs a;
    a.value = value; a.shapes = shapes; a.starts = starts; a.ref = ref; a.vis = vis; a.off_raw = off_raw; a.logits = logits;
    a.partial = (float *)workspace;
    a.cams = cams; a.nq = nq; a.heads = heads; a.L = L; a.P = P; a.nv = nv;
    a.lds_level0 = so_lds_first_level(host_shapes, L, kLdsBudget, a.lds_px0, a.lds_px);
    const int pairs = cams * heads;
    a.nbp = std::max(1, std::min(so_lds_num_cus() / pairs, (nq + kLdsWaves - 1) / kLdsWaves));   // one block per CU (LDS)
    hipStream_t st = (hipStream_t)stream;
    const size_t shm = (size_t)a.lds_px * 64;
    (void)hipFuncSetAttribute((const void *)msda_cross_lds_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(msda_cross_lds_fwd_kernel<4>, dim3((unsigned)(pairs * a.nbp)), dim3(kLdsWaves * 64), shm, st, a);
    const int width4 = heads * d / 4;
    hipLaunchKernelGGL(msda_cross_lds_reduce_kernel, dim3((unsigned)((nq * width4 + 255) / 256)), dim3(256), 0, st,
                       (const float *)workspace, vis, out, cams, nq, width4);
    return so_launch_status();
}
