// msda.hip — multi-scale deformable attention sampling for gfx950 (MI355X).
//
// Drop-in for mmcv==2.0.1 MultiScaleDeformableAttnFunction (ms_deform_attn_forward /
// _backward) at the reference call sites
//   model/encoder/bevformer/attention/image_cross_attention.py:340-342
//   model/encoder/tpvformer/attention/cross_view_hybrid_attention.py:111-113
//
//   out[b,q,h*D+c] = sum_{l,p} A[b,q,h,l,p] * bilinear(V_l[b, :, h, c], loc[b,q,h,l,p])
//   h_im = loc_y * H_l - 0.5, w_im = loc_x * W_l - 0.5, zero padding outside the map.
//
// Hardware mapping (not mmcv's thread-per-output-channel): A LANE OWNS A SAMPLING POINT, A TEAM GATHERS IT.
//   * the L*P points of one (b, q, head) are consecutive in memory, so the loc / attw (or raw offset /
//     logit) streams — ~80 % of the compulsory bytes — are read fully coalesced, once, and the bilinear
//     set-up is done once per point, not once per channel lane;
//   * d/4 adjacent lanes (a channel team) walk through their d/4 points together, each lane fetching its own
//     16-byte quarter of every corner: a wave-level load touches 16 corner segments instead of 64 (the L1 tag
//     rate, not bytes, bounds the gather); the point record travels through the team with DPP quad permutes;
//   * the teams of a group combine with a static reduce-scatter over wavefront shuffles;
//   * fused forms do the reference's prologue (softmax, loc = ref + off / (W, H)) in registers, in inference
//     and training; the camera-loop forms replace BEVCrossAttention's re-batch / scatter-add / mean;
//   * backward: a point kernel (grad_loc / grad_attw or the raw-output gradients; keys + records in band
//     order) and a banded, output-stationary scatter of grad_value through ds_add_f64 in LDS — no global atomics
//     in the scatter (selfocc_msda_bwd keeps the global-atomic form for callers without host shapes / workspace).
#include "so_device.h"
#include <algorithm>
#include <type_traits>

namespace {

#include "msda_device.h"

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int D, int LOGG>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const float *__restrict__ value,
                                                       const int32_t *__restrict__ shapes,
                                                       const int32_t *__restrict__ starts,
                                                       const float *__restrict__ loc,
                                                       const float *__restrict__ attw,
                                                       float *__restrict__ out, MsdaDims dm) {
    constexpr int G = 1 << LOGG;             // lanes per (b, q, head) group; host: G >= QL
    constexpr int QL = D / 4, LOGQ = so_ilog2(QL);
    constexpr int NJ = LOGG > LOGQ ? LOGG - LOGQ : 0;
    const int LP = dm.L * dm.P;
    const int groups_per_block = 256 / G;
    const long long n_groups = (long long)dm.bs * dm.nq * dm.heads;
    const long long gid = (long long)so_xcd_block() * groups_per_block + (threadIdx.x / G);
    const int gl = threadIdx.x & (G - 1);
    const bool live = gid < n_groups;  // whole groups are live or dead; exchanges stay in-group
    int h, b;
    long long bq, gq;
    so_split_group_head_outer(live ? gid : 0, dm, h, b, bq, gq);
    const int pix_stride = so_pix_stride(dm, D);
    const int s = gl & (QL - 1);
    const float *vb = value + so_value_base(dm, D, b, h, 0) + 4 * s;

    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int r0 = 0; r0 < LP; r0 += G) {   // uniform trip count: the team exchanges need every lane
        const int pt = r0 + gl;
        MsdaPoint mp = so_point_none();
        if (live && pt < LP) {
            const int l = so_level_of(pt, dm.P, dm.L);
            const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
            const size_t idx = (size_t)gq * LP + pt;
            const float2 xy = *(const float2 *)(loc + 2 * idx);
            mp = so_point_setup(xy.x, xy.y, attw[idx], Hl, Wl, starts[l] * pix_stride, pix_stride);
        }
        so_team_gather<D>(vb, mp, acc);
    }
    so_group_reduce_store<NJ, LOGQ>(acc, gl >> LOGQ, live, out + (size_t)gq * D + 4 * s);
}

// ---------------------------------------------------------------------------------------
// fused forward (inference): the prologue the reference runs as separate torch kernels around the
// op — softmax over the L*P attention logits of a (query, head) and
// loc = reference + offset / (W_l, H_l) (bevformer/attention/image_cross_attention.py:314-328,
// tpvformer/attention/cross_view_hybrid_attention.py:88-99) — happens in registers, so the
// 100-400 MB `sampling_locations` / `attention_weights` tensors are never written or re-read.
//   ref_kind 0: ref (bs, nq, L, 2)      (mmcv base class)
//            1: ref (bs, nq, P, 2)      (one anchor per point, all levels: BEVDeformableAttention)
//            2: ref (bs, nq, L, P, 2)   (CrossViewHybridAttention)
// ---------------------------------------------------------------------------------------
template <int D, int LOGG, typename VT>
__global__ __launch_bounds__(256, SO_MSDA_FWD_WAVES(D)) void msda_fused_fwd_kernel(const VT *__restrict__ value,
                                                             const int32_t *__restrict__ shapes,
                                                             const int32_t *__restrict__ starts,
                                                             const float *__restrict__ ref, int ref_kind,
                                                             const float *__restrict__ off_raw,
                                                             const float *__restrict__ logits,
                                                             float *__restrict__ out, MsdaDims dm) {
    constexpr int G = 1 << LOGG;
    constexpr int QL = D / 4, LOGQ = so_ilog2(QL);
    constexpr int NJ = LOGG > LOGQ ? LOGG - LOGQ : 0;
    constexpr int MAXR = so_maxr(LOGG);   // points per lane (host guarantees L * P <= MAXR * G)
    const int LP = dm.L * dm.P;
    const int groups_per_block = 256 / G;
    const long long n_groups = (long long)dm.bs * dm.nq * dm.heads;
    const long long gid = (long long)so_xcd_block() * groups_per_block + (threadIdx.x / G);
    const int gl = threadIdx.x & (G - 1);
    const bool live = gid < n_groups;
    int h, b;
    long long bq, gq;                                   // b * nq + q; (b, q, h) index of the group
    so_split_group_head_outer(live ? gid : 0, dm, h, b, bq, gq);
    const int pix_stride = so_pix_stride(dm, D);
    const int s = gl & (QL - 1);
    const VT *vb = value + so_value_base(dm, D, b, h, 0) + 4 * s;

    // softmax over the group's L * P logits
    float lg[MAXR];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int pt = gl + r * G;
        lg[r] = (pt < LP) ? logits[so_lg_index(dm, bq, h, LP, pt)] : -INFINITY;
        mx = fmaxf(mx, lg[r]);
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    float den = 0.0f;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        lg[r] = (gl + r * G < LP) ? __expf(lg[r] - mx) : 0.0f;
        den += lg[r];
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) den += __shfl_xor(den, m, 64);
    const float iden = 1.0f / den;

    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r * G >= LP) break;   // uniform
        const int pt = gl + r * G;
        MsdaPoint mp = so_point_none();
        if (live && pt < LP) {
            const int l = so_level_of(pt, dm.P, dm.L);
            const int pp = pt - l * dm.P;
            const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
            const float2 o = *(const float2 *)(off_raw + so_off_index(dm, bq, h, LP, pt));
            size_t ri;
            if (ref_kind == 1) ri = (size_t)bq * dm.P + pp;
            else if (ref_kind == 2) ri = ((size_t)bq * dm.L + l) * dm.P + pp;
            else ri = (size_t)bq * dm.L + l;
            const float2 rf = *(const float2 *)(ref + 2 * ri);
            const float lx = rf.x + o.x / (float)Wl, ly = rf.y + o.y / (float)Hl;
            mp = so_point_setup(lx, ly, lg[r] * iden, Hl, Wl, starts[l] * pix_stride, pix_stride);
        }
        so_team_gather<D>(vb, mp, acc);
    }
    so_group_reduce_store<NJ, LOGQ>(acc, gl >> LOGQ, live, out + (size_t)gq * D + 4 * s);
}

// ---------------------------------------------------------------------------------------
// camera-loop forward (inference) — the whole of BEVCrossAttention's sampling stage
// (bevformer/attention/image_cross_attention.py:90-136) in one launch.  The reference re-batches the
// queries each camera sees into (cams, max_len) (a host sync for max_len, python double loops), runs the
// offset / weight linears and MSDA on the padded re-batch, scatter-adds the result back per camera and
// divides by the number of cameras that saw the query.  But the offsets and attention logits are
// functions of the QUERY only — every camera's copy of a query produces the same ones; only the
// reference points and the value map differ per camera.  So here a (query, head) group does the
// softmax and the offsets once and loops over the cameras that see the query, accumulating the
// sampled output in registers in camera order (the order of the reference's `slots +=`), and divides
// by the visible-camera count at the end: no index lists, no padding, no atomics, no host sync,
// and the two linears run on nq rows instead of cams * max_len.
//   value (cams, nv, heads, D)   ref (cams, nq, P, 2)   vis (cams, nq) u8   off_raw (nq, heads, L, P, 2)
//   logits (nq, heads, L*P)      out (nq, heads*D) = sum_{cam visible} msda_cam(q) / max(#visible, 1)
// ---------------------------------------------------------------------------------------
template <int D, int LOGG, typename VT>
__global__ __launch_bounds__(256, SO_MSDA_CROSS_WAVES(D, LOGG)) void msda_cross_fwd_kernel(const VT *__restrict__ value,
                                                             const int32_t *__restrict__ shapes,
                                                             const int32_t *__restrict__ starts,
                                                             const float *__restrict__ ref,
                                                             const uint8_t *__restrict__ vis,
                                                             const float *__restrict__ off_raw,
                                                             const float *__restrict__ logits,
                                                             float *__restrict__ out, int cams, MsdaDims dm) {
    constexpr int G = 1 << LOGG;
    constexpr int QL = D / 4, LOGQ = so_ilog2(QL);
    constexpr int NJ = LOGG > LOGQ ? LOGG - LOGQ : 0;
    constexpr int MAXR = so_maxr(LOGG);   // points per lane (host guarantees L * P <= MAXR * G)
    const int LP = dm.L * dm.P;
    const int groups_per_block = 256 / G;
    const int n_groups = dm.nq * dm.heads;                       // < 2^31 (validated)
    const int gid = (int)so_xcd_block() * groups_per_block + (threadIdx.x / G);
    const int gl = threadIdx.x & (G - 1);
    const bool live = gid < n_groups;
    const int gq0 = live ? gid : 0;                            // head-outer order, see so_split_group_head_outer
    const int h = gq0 / dm.nq, q = gq0 - h * dm.nq;
    const int gq = q * dm.heads + h;
    const int pix_stride = so_pix_stride(dm, D);     // `value` may be head-major or a column block of a wider matrix
    const int s = gl & (QL - 1);
    const VT *vb = value + so_value_base(dm, D, 0, h, 0) + 4 * s;
    const int cam_stride = (int)(so_value_base(dm, D, 1, h, 0) - so_value_base(dm, D, 0, h, 0));   // < 2^31 (validated)

    // softmax over the group's L * P logits; raw offsets of the lane's own points, already / (W_l, H_l)
    float lg[MAXR], ox[MAXR], oy[MAXR];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int pt = gl + r * G;
        lg[r] = (pt < LP) ? logits[so_lg_index(dm, q, h, LP, pt)] : -INFINITY;
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
            const float2 o = *(const float2 *)(off_raw + so_off_index(dm, q, h, LP, pt));
            ox[r] = o.x / (float)shapes[2 * l + 1];
            oy[r] = o.y / (float)shapes[2 * l];
        }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) den += __shfl_xor(den, m, 64);
    const float iden = 1.0f / den;

    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int count = 0;
    for (int cam = 0; cam < cams; ++cam) {
        const bool seen = live && vis[(size_t)cam * dm.nq + q] != 0;
        count += seen ? 1 : 0;
        if (!__any(seen)) continue;                      // no group of this wave sees the camera
        const int cam_off = cam * cam_stride;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (r * G >= LP) break;   // uniform
            const int pt = gl + r * G;
            MsdaPoint mp = so_point_none();
            if (seen && pt < LP) {
                const int l = so_level_of(pt, dm.P, dm.L);
                const int pp = pt - l * dm.P;
                const float2 rf = *(const float2 *)(ref + 2 * (((size_t)cam * dm.nq + q) * dm.P + pp));
                mp = so_point_setup(rf.x + ox[r], rf.y + oy[r], lg[r] * iden, shapes[2 * l], shapes[2 * l + 1],
                                    cam_off + starts[l] * pix_stride, pix_stride);
            }
            if constexpr (LOGG >= 5) {
                const int steps = so_compact_points<D, LOGG>(mp);
                so_team_gather_steps<D>(vb, mp, acc, steps);
            } else {
                so_team_gather<D>(vb, mp, acc);
            }
        }
    }
    // mean over the cameras that saw the query (image_cross_attention.py:133-135: count clamped to >= 1)
    const float cnt = (float)max(count, 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = acc[c] / cnt;
    so_group_reduce_store<NJ, LOGQ>(acc, gl >> LOGQ, live, out + (size_t)gq * D + 4 * s);
}

// ---------------------------------------------------------------------------------------
// backward.  Phase 1: lane per sampling point (no cross-lane reduction): the 4 corner . g_out
// dot products give grad_attw and grad_loc (coalesced per-point stores).  Phase 2: the
// grad_value scatter is TRANSPOSED so that D consecutive lanes own the D contiguous channels
// of one point's corner: an atomic instruction then touches 64 / D segments of D * 4 bytes
// instead of 64 scattered dwords (the scalar-per-lane form measured 15 G atomics/s, 100x
// slower than the forward).  Point parameters travel from the owner lane by wavefront shuffle.
// ---------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const float *__restrict__ value,
                                                       const int32_t *__restrict__ shapes,
                                                       const int32_t *__restrict__ starts,
                                                       const float *__restrict__ loc,
                                                       const float *__restrict__ attw,
                                                       const float *__restrict__ g_out,
                                                       float *__restrict__ g_value,
                                                       float *__restrict__ g_loc,
                                                       float *__restrict__ g_attw, MsdaDims dm) {
    const int LP = dm.L * dm.P;
    const long long n_pts = (long long)dm.bs * dm.nq * dm.heads * LP;
    const long long idx = (long long)so_xcd_block() * blockDim.x + threadIdx.x;
    const bool live = idx < n_pts;
    const long long idc = live ? idx : n_pts - 1;
    long long gq;
    int pt;
    if (n_pts < (1LL << 31)) {   // uniform: 32-bit division
        const unsigned g = (unsigned)idc / (unsigned)LP;
        gq = g;
        pt = (int)((unsigned)idc - g * (unsigned)LP);
    } else {
        gq = idc / LP;
        pt = (int)(idc - gq * LP);
    }
    int h, b;
    long long bq_unused;
    so_split_group(gq, n_pts, dm.nq, dm.heads, h, b, bq_unused);
    const int l = so_level_of(pt, dm.P, dm.L);
    const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
    const int pix_stride = so_pix_stride(dm, D);
    const float2 xy = *(const float2 *)(loc + 2 * idc);
    const float aw = attw[idc];
    const Bilin bl = so_bilinear_setup(xy.x, xy.y, Hl, Wl, pix_stride);
    const int vbase = (int)so_value_base(dm, D, b, h, starts[l]);  // < 2^31 (validated)
    float ga = 0.0f, gx = 0.0f, gy = 0.0f;
    float wsc[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // weight * attention weight of each corner (0: skip)
    if (live && bl.any) {
        const float *vl = value + vbase;
        float go[D];
        const float4 *gp = (const float4 *)(g_out + (size_t)gq * D);
#pragma unroll
        for (int q = 0; q < D / 4; ++q) {
            const float4 t = gp[q];
            go[4 * q] = t.x; go[4 * q + 1] = t.y; go[4 * q + 2] = t.z; go[4 * q + 3] = t.w;
        }
        float dot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dot[k] = 0.0f;
            if (bl.valid[k]) {
                const float4 *p = (const float4 *)(vl + bl.off[k]);
                wsc[k] = bl.w[k] * aw;
#pragma unroll
                for (int q = 0; q < D / 4; ++q) {
                    const float4 t = p[q];
                    dot[k] = fmaf(t.x, go[4 * q], dot[k]);
                    dot[k] = fmaf(t.y, go[4 * q + 1], dot[k]);
                    dot[k] = fmaf(t.z, go[4 * q + 2], dot[k]);
                    dot[k] = fmaf(t.w, go[4 * q + 3], dot[k]);
                }
            }
        }
        // d out / d attw = bilinear value . g_out ;  d / d (w_im, h_im) from the weight derivatives
        ga = (bl.w[0] * dot[0] + bl.w[1] * dot[1]) + (bl.w[2] * dot[2] + bl.w[3] * dot[3]);
        const float gw = (bl.hh * (dot[1] - dot[0])) + (bl.lh * (dot[3] - dot[2]));
        const float gh = (bl.hw * (dot[2] - dot[0])) + (bl.lw * (dot[3] - dot[1]));
        gx = (float)Wl * gw * aw;
        gy = (float)Hl * gh * aw;
    }
    if (live) {
        g_attw[idx] = ga;
        *(float2 *)(g_loc + 2 * idx) = make_float2(gx, gy);
    }

    // ---- phase 2: transposed scatter of grad_value -------------------------------------------
    constexpr int PPI = 64 / D;                 // points served per atomic instruction
    const int lane = threadIdx.x & 63;
    const int sub = lane % D, grp = lane / D;
    const int gq32 = (int)gq;                    // < 2^31 (validated: bs * nq * heads * D < 2^31)
    int offs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) offs[k] = vbase + bl.off[k];
    for (int j = 0; j < 64 / PPI; ++j) {
        const int src = j * PPI + grp;          // owner lane of the point this lane now serves
        const int q_src = __shfl(gq32, src, 64);
        float w_src[4];
        int o_src[4];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w_src[k] = __shfl(wsc[k], src, 64);
            o_src[k] = __shfl(offs[k], src, 64);
            any |= (w_src[k] != 0.0f);
        }
        if (any) {
            const float goc = g_out[(size_t)q_src * D + sub];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (w_src[k] != 0.0f) unsafeAtomicAdd(g_value + o_src[k] + sub, w_src[k] * goc);
        }
    }
}


// ---------------------------------------------------------------------------------------
// backward, banded (output-stationary) form.
//
// grad_value is 64 scattered float adds per sampling point (4 corners x 16 channels): 1.6 G adds per
// call at the nuscenes_occ hw-plane shape, ~110 per output element and ~2300 per element on the
// 12x25 level.  Measured on MI355X (scripts/micro/atomics2.hip): global_atomic_add_f32 sustains
// ~330 G lane-ops/s chip-wide (one dword per L2 channel per clock; ~90 G/s when the rows of one
// instruction share a line), ds_add_f32 only ~200 G/s (~3 clocks per LANE), but ds_add_f64 runs
// at 3.7 T lane-ops/s.  So the scatter goes through LDS in double precision, and to make every
// add an LDS add the work is split by OUTPUT: a block owns a band of rows of one level's map of one
// (batch, head), keeps it in LDS as f64, and looks only at the sampling points that touch the band.
//
//   kernel 1 (points):  lane per sampling point; channel teams gather the 4 corners (as in the
//       forward), dot them with g_out -> grad_attw / grad_loc; also writes a 2-byte key per point
//       (the corner row floor(h_im), or "outside") into keys[b][h][l][q][p] and a 16-byte record
//       (bilinear fractions, attention weight, corner row / column).
//   kernels 2-4 (bins): counting sort of the point indices by (b, h, level, band of rows, class) — see
//       "binned band scatter" below.
//   kernel 5 (bands):   block = one segment of one band's index list.  For every 64 entries: lanes rebuild
//       the bilinear weights of their own point, then 16-lane rows (one channel per lane) add
//       w * attw * g_out into the band with ds_add_f64.  The band is flushed once (float atomics, so
//       that the segments of one band can share it).
// Besides the rate, f64 accumulation makes a band's sums order-independent to ~1e-16 relative (only the
// float flush of several segments into one band is order-dependent).
// ---------------------------------------------------------------------------------------
constexpr int kKeyOutside = -32768;

template <int QL>
SO_DEVFN float so_team_sum(float x) {   // sum over the QL lanes of a channel team, result in every lane
    if constexpr (QL >= 2)
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));  // [1,0,3,2]
    if constexpr (QL >= 4)
        x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));  // [2,3,0,1]
    if constexpr (QL >= 8) x += __shfl_xor(x, 4, 64);
    return x;
}

// team step I: the team computes value(corner k) . g_out for the point owned by its sub-lane I
template <int D, int I>
SO_DEVFN void so_bwd_team_step(const float *__restrict__ value, const float *__restrict__ g_out, int s,
                               const int (&goff)[4], int gq32, float (&dot)[4]) {
    constexpr int QL = D / 4;
    const int gqi = so_team_bcast<QL, I>(gq32);
    const float4 go = *(const float4 *)(g_out + (size_t)gqi * D + 4 * s);
    float part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int off = so_team_bcast<QL, I>(goff[k]);
        const float4 t = *(const float4 *)(value + off + 4 * s);
        part[k] = t.x * go.x;
        part[k] = fmaf(t.y, go.y, part[k]);
        part[k] = fmaf(t.z, go.z, part[k]);
        part[k] = fmaf(t.w, go.w, part[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d = so_team_sum<QL>(part[k]);
        if (s == I) dot[k] = d;
    }
}

template <int D>
__global__ __launch_bounds__(256) void msda_bwd_point_kernel(const float *__restrict__ value,
                                                             const int32_t *__restrict__ shapes,
                                                             const int32_t *__restrict__ starts,
                                                             const float *__restrict__ loc,
                                                             const float *__restrict__ attw,
                                                             const float *__restrict__ g_out,
                                                             float *__restrict__ g_loc, float *__restrict__ g_attw,
                                                             int16_t *__restrict__ keys, float4 *__restrict__ recs,
                                                             MsdaDims dm) {
    constexpr int QL = D / 4;
    const int LP = dm.L * dm.P;
    const long long n_pts = (long long)dm.bs * dm.nq * dm.heads * LP;
    const long long idx = (long long)so_xcd_block() * blockDim.x + threadIdx.x;
    const bool live = idx < n_pts;
    const long long idc = live ? idx : n_pts - 1;
    long long gq;
    int pt;
    if (n_pts < (1LL << 31)) {   // uniform: 32-bit division
        const unsigned g = (unsigned)idc / (unsigned)LP;
        gq = g;
        pt = (int)((unsigned)idc - g * (unsigned)LP);
    } else {
        gq = idc / LP;
        pt = (int)(idc - gq * LP);
    }
    int h, b;
    long long bq;
    so_split_group(gq, n_pts, dm.nq, dm.heads, h, b, bq);
    const int l = so_level_of(pt, dm.P, dm.L);
    const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
    const int pix_stride = so_pix_stride(dm, D);
    const float2 xy = *(const float2 *)(loc + 2 * idc);
    const float aw = attw[idc];
    const Bilin bl = so_bilinear_setup(xy.x, xy.y, Hl, Wl, pix_stride);
    const int vbase = (int)so_value_base(dm, D, b, h, starts[l]);  // < 2^31 (validated)
    int goff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) goff[k] = vbase + bl.off[k];
    const int s = threadIdx.x & (QL - 1);
    float dot[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    so_bwd_team_step<D, 0>(value, g_out, s, goff, (int)gq, dot);
    if constexpr (QL > 1) so_bwd_team_step<D, 1>(value, g_out, s, goff, (int)gq, dot);
    if constexpr (QL > 2) {
        so_bwd_team_step<D, 2>(value, g_out, s, goff, (int)gq, dot);
        so_bwd_team_step<D, 3>(value, g_out, s, goff, (int)gq, dot);
    }
    if constexpr (QL > 4) {
        so_bwd_team_step<D, 4>(value, g_out, s, goff, (int)gq, dot);
        so_bwd_team_step<D, 5>(value, g_out, s, goff, (int)gq, dot);
        so_bwd_team_step<D, 6>(value, g_out, s, goff, (int)gq, dot);
        so_bwd_team_step<D, 7>(value, g_out, s, goff, (int)gq, dot);
    }
    if (!live) return;
    float ga = 0.0f, gx = 0.0f, gy = 0.0f;
    if (bl.any) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dot[k] = bl.valid[k] ? dot[k] : 0.0f;
        // d out / d attw = bilinear value . g_out ;  d / d (w_im, h_im) from the weight derivatives
        ga = (bl.w[0] * dot[0] + bl.w[1] * dot[1]) + (bl.w[2] * dot[2] + bl.w[3] * dot[3]);
        const float gw = (bl.hh * (dot[1] - dot[0])) + (bl.lh * (dot[3] - dot[2]));
        const float gh = (bl.hw * (dot[2] - dot[0])) + (bl.lw * (dot[3] - dot[1]));
        gx = (float)Wl * gw * aw;
        gy = (float)Hl * gh * aw;
    }
    g_attw[idx] = ga;
    *(float2 *)(g_loc + 2 * idx) = make_float2(gx, gy);
    const int q = (int)(bq - (long long)b * dm.nq);
    const int p = pt - l * dm.P;
    const size_t ki = ((((size_t)b * dm.heads + h) * dm.L + l) * dm.nq + q) * dm.P + p;
    keys[ki] = (int16_t)(bl.any ? bl.h_low : kKeyOutside);
    // everything the band kernel needs of this point, in ITS order (points of one (b, h, level) contiguous)
    recs[ki] = make_float4(bl.lh, bl.lw, aw, __int_as_float((int)(((unsigned)bl.h_low << 16) | ((unsigned)bl.w_low & 0xffffu))));
}

// team step I with the g_out quarter already in registers (all lanes of a team share the group)
template <int D, int I, typename VT>
SO_DEVFN void so_bwd_team_step_g(const VT *__restrict__ value, const float4 &go, int s, const int (&goff)[4],
                                 float (&dot)[4]) {
    constexpr int QL = D / 4;
    float part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int off = so_team_bcast<QL, I>(goff[k]);
        const float4 t = so_ld4(value + off + 4 * s);
        part[k] = t.x * go.x;
        part[k] = fmaf(t.y, go.y, part[k]);
        part[k] = fmaf(t.z, go.z, part[k]);
        part[k] = fmaf(t.w, go.w, part[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d = so_team_sum<QL>(part[k]);
        if (s == I) dot[k] = d;
    }
}

// ---------------------------------------------------------------------------------------
// fused backward, point part (training): the counterpart of msda_fused_fwd_kernel.  Recomputes the
// softmax and loc = ref + off / (W_l, H_l) in registers (nothing but value / ref / off_raw / logits was
// saved by the forward), gathers the corners by channel teams, and writes the gradients of the RAW
// linear outputs directly:
//     g_off   = (d out / d loc) / (W_l, H_l)
//     g_logit = aw (g_aw - sum_group aw g_aw)                     (softmax backward, group reduction on shuffles)
// plus the keys / records of the banded grad_value scatter.  Replaces, per call, the torch kernels for
// softmax, off / normalizer, + ref and their three backward kernels, and the 100-400 MB loc / weight
// tensors with their gradients.
// ---------------------------------------------------------------------------------------
// plan of the counting sort that follows the point kernels (see msda_bin_kernel below); the training point kernels count
// their own hits with it
constexpr int kMaxBands = 1024;            // bands per level (host falls back to the atomic kernel beyond)
constexpr int kBinKeysPerBlock = 256 * 32;

// n / d for any 32-bit n without a division (Granlund & Montgomery round-up method; host computes m, l from d)
struct SoFastDiv {
    unsigned m;
    int l;
};
SO_DEVFN int so_fastdiv(int n, SoFastDiv f) {
    if (f.l == 0) return n;                      // d == 1
    const unsigned t = __umulhi(f.m, (unsigned)n);
    return (int)((t + (((unsigned)n - t) >> 1)) >> (f.l - 1));
}

struct MsdaBinPlan {
    float inv_rows[8];   // 1 / rows[l] (host-computed: the bin kernels and the counting point kernels classify with the same float)
    int rows[8];      // rows per band of level l
    int bands[8];     // bands of level l
    int band0[9];     // first band of level l within a (batch, head)
    int nbands;       // bands per (batch, head)
    int seg;          // list entries per band-kernel block
    SoFastDiv divP;   // key index -> query
};

template <int D, int LOGG, typename VT>
__global__ __launch_bounds__(256) void msda_fused_bwd_point_kernel(const VT *__restrict__ value,
                                                                   const int32_t *__restrict__ shapes,
                                                                   const int32_t *__restrict__ starts,
                                                                   const float *__restrict__ ref, int ref_kind,
                                                                   const float *__restrict__ off_raw,
                                                                   const float *__restrict__ logits,
                                                                   const float *__restrict__ g_out,
                                                                   float *__restrict__ g_off, float *__restrict__ g_logits,
                                                                   int16_t *__restrict__ keys, float4 *__restrict__ recs,
                                                                   MsdaDims dm) {
    constexpr int G = 1 << LOGG;
    constexpr int QL = D / 4;
    constexpr int MAXR = so_maxr(LOGG);
    const int LP = dm.L * dm.P;
    const int groups_per_block = 256 / G;
    const long long n_groups = (long long)dm.bs * dm.nq * dm.heads;
    const long long gid = (long long)so_xcd_block() * groups_per_block + (threadIdx.x / G);
    const int gl = threadIdx.x & (G - 1);
    const bool live = gid < n_groups;
    int h, b;
    long long bq, gq;                                   // b * nq + q; (b, q, h) index of the group
    so_split_group_head_outer(live ? gid : 0, dm, h, b, bq, gq);
    const int q = (int)(bq - (long long)b * dm.nq);
    const int pix_stride = so_pix_stride(dm, D);
    const int s = gl & (QL - 1);

    float lg[MAXR];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int pt = gl + r * G;
        lg[r] = (pt < LP) ? logits[so_lg_index(dm, bq, h, LP, pt)] : -INFINITY;
        mx = fmaxf(mx, lg[r]);
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    float den = 0.0f;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        lg[r] = (gl + r * G < LP) ? __expf(lg[r] - mx) : 0.0f;
        den += lg[r];
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) den += __shfl_xor(den, m, 64);
    const float iden = 1.0f / den;

    float4 go = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) go = *(const float4 *)(g_out + (size_t)gq * D + 4 * s);
    float ga[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) ga[r] = 0.0f;
    float sum_l = 0.0f;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r * G >= LP) break;   // uniform
        const int pt = gl + r * G;
        const bool own = live && pt < LP;
        const int ptc = own ? pt : 0;
        const int l = so_level_of(ptc, dm.P, dm.L);
        const int pp = ptc - l * dm.P;
        const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
        const size_t oidx = so_off_index(dm, bq, h, LP, ptc);
        const float2 o = *(const float2 *)(off_raw + oidx);
        size_t ri;
        if (ref_kind == 1) ri = (size_t)bq * dm.P + pp;
        else if (ref_kind == 2) ri = ((size_t)bq * dm.L + l) * dm.P + pp;
        else ri = (size_t)bq * dm.L + l;
        const float2 rf = *(const float2 *)(ref + 2 * ri);
        const float lx = rf.x + o.x / (float)Wl, ly = rf.y + o.y / (float)Hl;
        const float aw = lg[r] * iden;
        const Bilin bl = so_bilinear_setup(lx, ly, Hl, Wl, pix_stride);
        const int vbase = (int)so_value_base(dm, D, b, h, starts[l]);
        int goff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) goff[k] = own ? vbase + bl.off[k] : 0;
        float dot[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        so_bwd_team_step_g<D, 0>(value, go, s, goff, dot);
        if constexpr (QL > 1) so_bwd_team_step_g<D, 1>(value, go, s, goff, dot);
        if constexpr (QL > 2) {
            so_bwd_team_step_g<D, 2>(value, go, s, goff, dot);
            so_bwd_team_step_g<D, 3>(value, go, s, goff, dot);
        }
        if constexpr (QL > 4) {
            so_bwd_team_step_g<D, 4>(value, go, s, goff, dot);
            so_bwd_team_step_g<D, 5>(value, go, s, goff, dot);
            so_bwd_team_step_g<D, 6>(value, go, s, goff, dot);
            so_bwd_team_step_g<D, 7>(value, go, s, goff, dot);
        }
        if (own) {
            float gx = 0.0f, gy = 0.0f;
            if (bl.any) {
#pragma unroll
                for (int k = 0; k < 4; ++k) dot[k] = bl.valid[k] ? dot[k] : 0.0f;
                ga[r] = (bl.w[0] * dot[0] + bl.w[1] * dot[1]) + (bl.w[2] * dot[2] + bl.w[3] * dot[3]);
                const float gw = (bl.hh * (dot[1] - dot[0])) + (bl.lh * (dot[3] - dot[2]));
                const float gh = (bl.hw * (dot[2] - dot[0])) + (bl.lw * (dot[3] - dot[1]));
                gx = (float)Wl * gw * aw;
                gy = (float)Hl * gh * aw;
            }
            // loc = ref + off / (W, H)  =>  g_off = g_loc / (W, H)
            *(float2 *)(g_off + oidx) = make_float2(gx / (float)Wl, gy / (float)Hl);
            sum_l = fmaf(aw, ga[r], sum_l);
            const size_t ki = ((((size_t)b * dm.heads + h) * dm.L + l) * dm.nq + q) * dm.P + pp;
            keys[ki] = (int16_t)(bl.any ? bl.h_low : kKeyOutside);
            recs[ki] = make_float4(bl.lh, bl.lw, aw, __int_as_float((int)(((unsigned)bl.h_low << 16) | ((unsigned)bl.w_low & 0xffffu))));
        }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) sum_l += __shfl_xor(sum_l, m, 64);
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int pt = gl + r * G;
        if (live && pt < LP) g_logits[so_lg_index(dm, bq, h, LP, pt)] = (lg[r] * iden) * (ga[r] - sum_l);
    }
}

// ---------------------------------------------------------------------------------------
// camera-loop backward, point part (training counterpart of msda_cross_fwd_kernel):
//   out[q] = (1 / cnt_q) sum_{cam sees q} msda_cam(q)      with shared offsets / logits per query.
// g_off and g_logits are sums over the visible cameras (divided by cnt); keys / records are written per
// (cam, h, level, q, p) for the visible cameras only (the key array is pre-filled with "outside"), the
// record's weight already divided by cnt, and the band kernel reads the shared g_out row (go_shared).
// ---------------------------------------------------------------------------------------
template <int D, int LOGG, typename VT>
// (no register cap: with min-waves 4 the compiler holds the kernel at 105 / 101 registers instead of 115 / 111 and the
// iteration loses 1.3 ms — fewer corner gathers in flight per wave; 5 / 6 waves spill)
__global__ __launch_bounds__(256) void msda_cross_bwd_point_kernel(const VT *__restrict__ value,
                                                                   const int32_t *__restrict__ shapes,
                                                                   const int32_t *__restrict__ starts,
                                                                   const float *__restrict__ ref,
                                                                   const uint8_t *__restrict__ vis,
                                                                   const float *__restrict__ off_raw,
                                                                   const float *__restrict__ logits,
                                                                   const float *__restrict__ g_out,
                                                                   float *__restrict__ g_off, float *__restrict__ g_logits,
                                                                   int16_t *__restrict__ keys, float4 *__restrict__ recs,
                                                                   int cams, MsdaDims dm, int32_t *__restrict__ bin_cnt,
                                                                   MsdaBinPlan plan) {
    constexpr int G = 1 << LOGG;
    constexpr int QL = D / 4;
    constexpr int MAXR = so_maxr(LOGG);
    const int LP = dm.L * dm.P;
    const int groups_per_block = 256 / G;
    const int n_groups = dm.nq * dm.heads;
    // bin_cnt != NULL: the kernel also does the counting pass of the sort (msda_bin_kernel<false>): an LDS histogram over
    // (head of the block: at most two, camera, band, class), flushed once per block — the keys are in registers here
    extern __shared__ int hist_s[];                              // [2 heads][cams][nbands][2]
    const int blk0 = (int)so_xcd_block() * groups_per_block;
    const int h_first = min(blk0, n_groups - 1) / dm.nq;
    const int nh = bin_cnt != nullptr ? 2 * cams * plan.nbands * 2 : 0;
    for (int e = threadIdx.x; e < nh; e += 256) hist_s[e] = 0;
    if (bin_cnt != nullptr) __syncthreads();
    const int gid = blk0 + (threadIdx.x / G);
    const int gl = threadIdx.x & (G - 1);
    const bool live = gid < n_groups;
    const int gq0 = live ? gid : 0;                            // head-outer order, see so_split_group_head_outer
    const int h = gq0 / dm.nq, q = gq0 - h * dm.nq;
    const int gq = q * dm.heads + h;
    const int pix_stride = so_pix_stride(dm, D);
    const int s = gl & (QL - 1);

    float lg[MAXR], ox[MAXR], oy[MAXR];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int pt = gl + r * G;
        lg[r] = (pt < LP) ? logits[so_lg_index(dm, q, h, LP, pt)] : -INFINITY;
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
            const float2 o = *(const float2 *)(off_raw + so_off_index(dm, q, h, LP, pt));
            ox[r] = o.x / (float)shapes[2 * l + 1];
            oy[r] = o.y / (float)shapes[2 * l];
        }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) den += __shfl_xor(den, m, 64);
    const float iden = 1.0f / den;

    int count = 0;
    for (int cam = 0; cam < cams; ++cam) count += (live && vis[(size_t)cam * dm.nq + q] != 0) ? 1 : 0;
    const float cnt = (float)max(count, 1);

    float4 go = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) go = *(const float4 *)(g_out + (size_t)gq * D + 4 * s);
    float ga[MAXR], gxs[MAXR], gys[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) ga[r] = gxs[r] = gys[r] = 0.0f;
    for (int cam = 0; cam < cams; ++cam) {
        const bool seen = live && vis[(size_t)cam * dm.nq + q] != 0;
        if (!__any(seen)) continue;                      // no group of this wave sees the camera
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (r * G >= LP) break;   // uniform
            const int pt = gl + r * G;
            const bool own = seen && pt < LP;
            const int ptc = own ? pt : 0;
            const int l = so_level_of(ptc, dm.P, dm.L);
            const int pp = ptc - l * dm.P;
            const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
            const float2 rf = *(const float2 *)(ref + 2 * (((size_t)cam * dm.nq + q) * dm.P + pp));
            const float aw = lg[r] * iden;
            const Bilin bl = so_bilinear_setup(rf.x + ox[r], rf.y + oy[r], Hl, Wl, pix_stride);
            const int vbase = (int)so_value_base(dm, D, cam, h, starts[l]);
            int goff[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) goff[k] = own ? vbase + bl.off[k] : 0;
            float dot[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            // whole-wave groups: the points that touch the map move to the front (see so_compact_points), the teams
            // compute their corner dots there, and the dots travel back to the lanes that own the points
            int steps = QL, dst = 0;
            bool moved = false;
            if constexpr (LOGG >= 5) {
                steps = so_compact_plan<D, LOGG>(own && bl.any, dst, moved);
                if (moved) {   // wave-uniform
#pragma unroll
                    for (int c = 0; c < 4; ++c) goff[c] = __builtin_amdgcn_ds_permute(dst, goff[c]);
                }
            }
            if (steps > 0) so_bwd_team_step_g<D, 0>(value, go, s, goff, dot);
            if constexpr (QL > 1) { if (steps > 1) so_bwd_team_step_g<D, 1>(value, go, s, goff, dot); }
            if constexpr (QL > 2) {
                if (steps > 2) so_bwd_team_step_g<D, 2>(value, go, s, goff, dot);
                if (steps > 3) so_bwd_team_step_g<D, 3>(value, go, s, goff, dot);
            }
            if constexpr (QL > 4) {
                if (steps > 4) so_bwd_team_step_g<D, 4>(value, go, s, goff, dot);
                if (steps > 5) so_bwd_team_step_g<D, 5>(value, go, s, goff, dot);
                if (steps > 6) so_bwd_team_step_g<D, 6>(value, go, s, goff, dot);
                if (steps > 7) so_bwd_team_step_g<D, 7>(value, go, s, goff, dot);
            }
            if (moved) {   // wave-uniform
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    dot[c] = __int_as_float(__builtin_amdgcn_ds_bpermute(dst, __float_as_int(dot[c])));
            }
            const size_t ki = ((((size_t)cam * dm.heads + h) * dm.L + l) * dm.nq + q) * dm.P + pp;
            if (own && !bl.any) keys[ki] = (int16_t)kKeyOutside;   // every key of a visited (camera, query) is written
            if (own && bl.any) {
#pragma unroll
                for (int k = 0; k < 4; ++k) dot[k] = bl.valid[k] ? dot[k] : 0.0f;
                ga[r] += (bl.w[0] * dot[0] + bl.w[1] * dot[1]) + (bl.w[2] * dot[2] + bl.w[3] * dot[3]);
                const float gw = (bl.hh * (dot[1] - dot[0])) + (bl.lh * (dot[3] - dot[2]));
                const float gh = (bl.hw * (dot[2] - dot[0])) + (bl.lw * (dot[3] - dot[1]));
                gxs[r] += ((float)Wl * gw * aw) / (float)Wl;
                gys[r] += ((float)Hl * gh * aw) / (float)Hl;
                keys[ki] = (int16_t)bl.h_low;
                recs[ki] = make_float4(bl.lh, bl.lw, aw / cnt,
                                       __int_as_float((int)(((unsigned)bl.h_low << 16) | ((unsigned)bl.w_low & 0xffffu))));
                if (bin_cnt != nullptr) {      // the classification of msda_bin_kernel, on the key in the register
                    int band0 = 0;
                    float inv_rows = plan.inv_rows[0];
                    for (int k = 1; k < 8; ++k)
                        if (k == l) { band0 = plan.band0[k]; inv_rows = plan.inv_rows[k]; }
                    const int key = bl.h_low;
                    const bool va = key >= 0 && key < Hl, vb = key + 1 >= 0 && key + 1 < Hl;
                    const int ba = (int)(((float)key + 0.5f) * inv_rows), bb = (int)(((float)key + 1.5f) * inv_rows);
                    int *hb = hist_s + (((h - h_first) * cams + cam) * plan.nbands + band0) * 2;
                    if (va && vb && ba == bb) {
                        atomicAdd(hb + 2 * ba, 1);
                    } else {
                        if (va) atomicAdd(hb + 2 * ba + 1, 1);
                        if (vb) atomicAdd(hb + 2 * bb + 1, 1);
                    }
                }
            }
        }
    }
    if (bin_cnt != nullptr) {
        __syncthreads();
        for (int e = threadIdx.x; e < nh; e += 256) {
            const int c = hist_s[e];
            if (c == 0) continue;
            const int per_head = cams * plan.nbands * 2;
            const int hr = e / per_head, rem = e - hr * per_head;
            const int cam = rem / (plan.nbands * 2), bc = rem - cam * (plan.nbands * 2);
            atomicAdd(bin_cnt + ((size_t)(cam * dm.heads + h_first + hr) * plan.nbands) * 2 + bc, c);
        }
    }
    float sum_l = 0.0f;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        ga[r] = ga[r] / cnt;
        sum_l = fmaf(lg[r] * iden, ga[r], sum_l);
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) sum_l += __shfl_xor(sum_l, m, 64);
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int pt = gl + r * G;
        if (live && pt < LP) {
            g_logits[so_lg_index(dm, q, h, LP, pt)] = (lg[r] * iden) * (ga[r] - sum_l);
            *(float2 *)(g_off + so_off_index(dm, q, h, LP, pt)) = make_float2(gxs[r] / cnt, gys[r] / cnt);
        }
    }
}

// two block shapes: 512 threads + 52 KB tile (two blocks per CU: one block's zero / flush phases overlap the other's
// adds) and 1024 threads + 103 KB tile (one block per CU, 2x taller bands: fewer points straddle two bands)
constexpr int band_tile_bytes(int threads) { return threads == 512 ? 52 * 1024 : 103 * 1024; }

// ---------------------------------------------------------------------------------------
// binned band scatter.  Round 1 let every band block stream its level's keys and pick out its hits: O(bands x keys),
// and measured on the training iteration 57 % of that kernel's time was the scan alone (no hit processed), 35 % the
// per-hit work, 8 % the LDS atomics.  Now the keys are read three times in total, by a counting sort by
// (batch, head, level, band, class):
//   bin<false>: LDS histogram of a block of keys -> global counters        (class FULL: both corner rows in the band,
//   scan:       exclusive scan of the counters, work items per band                EDGE: one corner row; a point whose
//   bin<true>:  same histogram, reserve space, write the point indices             rows straddle two bands is listed in both)
// and the band kernel walks its own index list: no scan, no rings, no ballots; bands nobody touches cost nothing
// (not even the tile zero / flush), long lists are cut into segments of `seg` entries (one block each).
// Training iteration (16 calls): 15.6 ms -> 8.1 ms; the band kernel now runs at the ds_add_f64 rate measured by
// scripts/micro/atomics2.hip (~10.6 clk per 64-lane instruction).
// ---------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(256) void msda_bin_kernel(const int16_t *__restrict__ keys, const int32_t *__restrict__ shapes,
                                                       int32_t *__restrict__ cnt, int32_t *__restrict__ cursor,
                                                       const int32_t *__restrict__ off, int32_t *__restrict__ list,
                                                       const unsigned char *__restrict__ vis, int bpb, MsdaDims dm,
                                                       MsdaBinPlan plan) {
    // vis != NULL (camera loop, P >= 4): keys of (camera, query) pairs with vis == 0 were never written — not read here
    __shared__ int hist[2 * kMaxBands];
    __shared__ int base_s[FILL ? 2 * kMaxBands : 1];
    const long long bhl = blockIdx.x / bpb;
    const int chunk = blockIdx.x - (int)(bhl * bpb);
    const int l = (int)(bhl % dm.L);
    const long long bh = bhl / dm.L;
    int bands_l = plan.bands[0], band0 = 0;
    float inv_rows = plan.inv_rows[0];
    for (int k = 1; k < 8; ++k)
        if (k == l) { bands_l = plan.bands[k]; band0 = plan.band0[k]; inv_rows = plan.inv_rows[k]; }
    const int Hl = shapes[2 * l];
    const long long n = (long long)dm.nq * dm.P;                      // keys of one (b, h, l)
    const long long kbase = bhl * n;
    // the block's keys: an 8-byte aligned window of kBinKeysPerBlock keys, clipped to this (b, h, l)
    const long long win = (kbase & ~3LL) + (long long)chunk * kBinKeysPerBlock;
    const long long lo = max(kbase, win), hi = min(kbase + n, win + kBinKeysPerBlock);
    const long long bucket0 = (bh * plan.nbands + band0) * 2;         // bucket = (band id) * 2 + class
    for (int i = threadIdx.x; i < 2 * bands_l; i += 256) hist[i] = 0;

    // every lane loads its keys ONCE (4 consecutive keys = 8 aligned bytes per step, all steps in flight together);
    // keys outside [lo, hi) and keys of invisible (camera, query) pairs become "outside" in the registers
    constexpr int IT = kBinKeysPerBlock / (256 * 4);
    constexpr unsigned kNone = 0x80008000u;
    uint2 kk[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long long p = win + (long long)(it * 256 + (int)threadIdx.x) * 4;
        kk[it] = make_uint2(kNone, kNone);
        if (p + 3 < lo || p >= hi) continue;
        bool v0 = true, v1 = true;     // visibility of the queries of the first / last key (P >= 4: at most two)
        int qsplit = 0;
        if (vis != nullptr) {
            const unsigned char *vb = vis + (size_t)(bh / dm.heads) * dm.nq;
            const int e0 = (int)(max(p, lo) - kbase), e1 = (int)(min(p + 3, hi - 1) - kbase);
            const int q0 = so_fastdiv(e0, plan.divP), q1 = so_fastdiv(e1, plan.divP);
            v0 = vb[q0] != 0;
            v1 = vb[q1] != 0;
            qsplit = q1 * dm.P;        // first key of the last query
        }
        if (!v0 && !v1) continue;
        uint2 t = *(const uint2 *)(keys + p);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long gi = p + k;
            const bool dead = gi < lo || gi >= hi || !((int)(gi - kbase) < qsplit ? v0 : v1);
            if (dead) {
                unsigned &w = (k & 2) ? t.y : t.x;
                w = (k & 1) ? ((w & 0x0000ffffu) | 0x80000000u) : ((w & 0xffff0000u) | 0x00008000u);
            }
        }
        kk[it] = t;
    }
    __syncthreads();

    // pass(emit): emit(bucket-in-level, key index) per list entry.  (One LDS atomic per lane and entry: aggregating the
    // lanes of a wave that want the same bucket measured 2.5x slower — the pillar points of a query span many rows.)
    auto pass = [&](auto emit) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (kk[it].x == kNone && kk[it].y == kNone) continue;
            const int e4 = (int)(win - kbase) + (it * 256 + (int)threadIdx.x) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int key = (int)(short)(((k & 2) ? kk[it].y : kk[it].x) >> (16 * (k & 1)));
                if (key == kKeyOutside) continue;
                const bool va = key >= 0 && key < Hl, vb = key + 1 >= 0 && key + 1 < Hl;
                const int ba = (int)(((float)key + 0.5f) * inv_rows), bb = (int)(((float)key + 1.5f) * inv_rows);
                const int e = e4 + k;
                if (va && vb && ba == bb) {
                    emit(2 * ba, e);
                } else {
                    if (va) emit(2 * ba + 1, e);
                    if (vb) emit(2 * bb + 1, e);
                }
            }
        }
    };
    pass([&](int i, int) { atomicAdd(&hist[i], 1); });
    __syncthreads();
    if constexpr (!FILL) {
        for (int i = threadIdx.x; i < 2 * bands_l; i += 256) {
            const int c = hist[i];
            if (c) atomicAdd(&cnt[bucket0 + i], c);
        }
    } else {
        for (int i = threadIdx.x; i < 2 * bands_l; i += 256) {
            const int c = hist[i];
            base_s[i] = c ? off[bucket0 + i] + atomicAdd(&cursor[bucket0 + i], c) : 0;
            hist[i] = 0;
        }
        __syncthreads();
        pass([&](int i, int e) { list[base_s[i] + atomicAdd(&hist[i], 1)] = e; });
    }
}

// one block: off[bucket] = first list entry of the bucket (a band's EDGE list follows its FULL list),
// item0[band] = first work item of the band (ceil(entries / seg) items), item0[n_bands] = number of items
__global__ __launch_bounds__(1024) void msda_bin_scan_kernel(const int32_t *__restrict__ cnt, int32_t *__restrict__ off,
                                                             int32_t *__restrict__ item0, int n_bands, int seg) {
    __shared__ int se[1024], si[1024];
    const int t = threadIdx.x;
    const int per = (n_bands + 1023) / 1024;
    const int b0 = min(n_bands, t * per), b1 = min(n_bands, b0 + per);
    int ne = 0, ni = 0;
    for (int b = b0; b < b1; ++b) {
        const int c = cnt[2 * b] + cnt[2 * b + 1];
        ne += c;
        ni += (c + seg - 1) / seg;
    }
    se[t] = ne; si[t] = ni;
    __syncthreads();
    for (int m = 1; m < 1024; m <<= 1) {
        const int ae = t >= m ? se[t - m] : 0, ai = t >= m ? si[t - m] : 0;
        __syncthreads();
        se[t] += ae; si[t] += ai;
        __syncthreads();
    }
    int pe = se[t] - ne, pi = si[t] - ni;     // exclusive prefixes of this thread's run
    for (int b = b0; b < b1; ++b) {
        const int cf = cnt[2 * b], c = cf + cnt[2 * b + 1];
        off[2 * b] = pe;
        off[2 * b + 1] = pe + cf;
        item0[b] = pi;
        pe += c;
        pi += (c + seg - 1) / seg;
    }
    if (t == 1023) item0[n_bands] = si[1023];
}

template <int D, int kBandThreads>
__global__ __launch_bounds__(kBandThreads) void msda_bwd_band_list_kernel(const int32_t *__restrict__ shapes,
                                                                          const int32_t *__restrict__ starts,
                                                                          const float *__restrict__ g_out,
                                                                          float *__restrict__ g_value,
                                                                          const float4 *__restrict__ recs,
                                                                          const int32_t *__restrict__ cnt,
                                                                          const int32_t *__restrict__ off,
                                                                          const int32_t *__restrict__ item0,
                                                                          const int32_t *__restrict__ list, int n_bands,
                                                                          MsdaDims dm, MsdaBinPlan plan) {
    constexpr int ROWS = 64 / D;        // sampling points served per atomic instruction
    constexpr int NJ = 64 / ROWS;       // row steps per batch of 64 points (= D)
    constexpr int NW = kBandThreads / 64;
    extern __shared__ __attribute__((aligned(16))) double tile[];
    __shared__ int4 recP[kBandThreads];     // band-local element offsets of the 4 (or 2) corners
    __shared__ float4 recW[kBandThreads];   // corner weight x attention weight
    __shared__ int recQ[kBandThreads];      // (b, q, h) group index: row of g_out

    const int item = blockIdx.x;
    if (item >= item0[n_bands]) return;
    int lo_b = 0, hi_b = n_bands;           // largest band id with item0[id] <= item (uniform: scalar loads)
    while (hi_b - lo_b > 1) {
        const int mid = (lo_b + hi_b) >> 1;
        if (item0[mid] <= item) lo_b = mid; else hi_b = mid;
    }
    const int band_id = lo_b;
    const int segi = item - item0[band_id];
    const int bh = band_id / plan.nbands, rb = band_id - bh * plan.nbands;
    const int h = bh % dm.heads, b = bh / dm.heads;
    int l = 0;
    for (int k = 1; k < dm.L; ++k) l += (rb >= plan.band0[k]);
    int rows_l = plan.rows[0], first = 0;
    for (int k = 1; k < 8; ++k)
        if (k == l) { rows_l = plan.rows[k]; first = plan.band0[k]; }
    const int band = rb - first;
    const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
    const int y0 = band * rows_l, y1 = min(Hl, y0 + rows_l);
    const int n_tile = (y1 - y0) * Wl * D;
    const int cF = cnt[2 * band_id], cE = cnt[2 * band_id + 1];
    const int32_t *mine = list + off[2 * band_id];        // FULL entries [0, cF), EDGE entries [cF, cF + cE)
    const int s0 = segi * plan.seg, s1 = min(cF + cE, s0 + plan.seg);

    for (int e = threadIdx.x; e < n_tile; e += kBandThreads) tile[e] = 0.0;
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave0 = threadIdx.x & ~63;
    const int sub = lane % D, row = lane / D;
    const long long kbase = (((long long)b * dm.heads + h) * dm.L + l) * (long long)dm.nq * dm.P;

    // n <= 64 list entries starting at `pos` -> band
    auto process = [&](auto cls, int pos, int n) {
        constexpr int C = decltype(cls)::value;     // 0: FULL, 1: EDGE
        float4 w4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int4 p4 = make_int4(0, 0, 0, 0);
        int gq32 = 0;
        if (lane < n) {
            const int e = mine[pos + lane];
            const int q = so_fastdiv(e, plan.divP);
            const float4 rc = recs[kbase + e];
            const int hw_ = __float_as_int(rc.w);
            const int h_low = hw_ >> 16, w_low = (int)(short)(hw_ & 0xffff);
            const float lh = rc.x, lw = rc.y, aw = rc.z;
            const float hh = 1.0f - lh, hw = 1.0f - lw;
            const bool c0 = w_low >= 0, c1 = w_low + 1 <= Wl - 1;
            const int x0 = max(w_low, 0), x1 = min(w_low + 1, Wl - 1);
            if (C == 0) {
                const int r0 = (h_low - y0) * Wl, r1 = r0 + Wl;
                w4 = make_float4(c0 ? (hh * hw) * aw : 0.0f, c1 ? (hh * lw) * aw : 0.0f,
                                 c0 ? (lh * hw) * aw : 0.0f, c1 ? (lh * lw) * aw : 0.0f);
                p4 = make_int4((r0 + x0) * D, (r0 + x1) * D, (r1 + x0) * D, (r1 + x1) * D);
            } else {
                // the one corner row inside the band: the lower one (h_low + 1 == y0) or the upper one (h_low == y1 - 1)
                const bool lower = h_low < y0;
                const int ry = lower ? h_low + 1 : h_low;
                const float wy = lower ? lh : hh;
                const int r0 = (ry - y0) * Wl;
                w4 = make_float4(c0 ? (wy * hw) * aw : 0.0f, c1 ? (wy * lw) * aw : 0.0f, 0.0f, 0.0f);
                p4 = make_int4((r0 + x0) * D, (r0 + x1) * D, 0, 0);
            }
            gq32 = dm.go_shared ? q * dm.heads + h : (int)(((long long)b * dm.nq + q) * dm.heads + h);
        }
        recW[threadIdx.x] = w4;
        recP[threadIdx.x] = p4;
        recQ[threadIdx.x] = gq32;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float goc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) goc[j] = g_out[(size_t)recQ[wave0 + j * ROWS + row] * D + sub];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int src = wave0 + j * ROWS + row;
            const float4 ws = recW[src];
            const int4 ps = recP[src];
            // unconditional: a corner outside the map carries weight 0 (adds 0.0 to the band's first pixel)
            unsafeAtomicAdd(&tile[ps.x + sub], (double)(ws.x * goc[j]));
            unsafeAtomicAdd(&tile[ps.y + sub], (double)(ws.y * goc[j]));
            if (C == 0) {
                unsafeAtomicAdd(&tile[ps.z + sub], (double)(ws.z * goc[j]));
                unsafeAtomicAdd(&tile[ps.w + sub], (double)(ws.w * goc[j]));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const int f0 = s0, f1 = min(s1, cF);              // FULL part of the segment
    for (int p = f0 + wv * 64; p < f1; p += NW * 64) process(std::integral_constant<int, 0>{}, p, min(64, f1 - p));
    const int e0 = max(s0, cF), e1 = s1;              // EDGE part
    for (int p = e0 + wv * 64; p < e1; p += NW * 64) process(std::integral_constant<int, 1>{}, p, min(64, e1 - p));
    __syncthreads();

    const int pix_stride = so_pix_stride(dm, D);
    float *gl = g_value + so_value_base(dm, D, b, h, (long long)starts[l] + (long long)y0 * Wl);
    for (int e = threadIdx.x; e < n_tile; e += kBandThreads) {
        const double v = tile[e];
        if (v != 0.0) unsafeAtomicAdd(gl + (size_t)(e / D) * pix_stride + (e % D), (float)v);
    }
}

int validate(const float *value, const int32_t *shapes, const int32_t *starts, const float *loc,
             const float *attw, int bs, int nv, int nq, int heads, int d, int L, int P) {
    SO_REQUIRE(bs >= 0 && nq >= 0 && nv >= 0, "msda: negative size");
    SO_REQUIRE((bs == 0 || nq == 0) || (value && shapes && starts && loc && attw), "msda: NULL input pointer");
    SO_REQUIRE(heads >= 1 && L >= 1 && P >= 1, "msda: heads, L, P must be >= 1");
    SO_REQUIRE(d == 4 || d == 8 || d == 16 || d == 32, "msda: channels per head must be 4, 8, 16 or 32 (got %d)", d);
    SO_REQUIRE((long long)bs * nq * heads * L * P < (1LL << 40), "msda: problem too large");
    SO_REQUIRE((long long)bs * nv * heads * d < (1LL << 31) && (long long)bs * nq * heads * d < (1LL << 31),
               "msda: value / output tensors must have < 2^31 elements");
    return 0;
}

}  // namespace

// Lanes per (b, q, head) group: a power of two G >= d / 4 (whole channel teams) with rounds = ceil(LP / G)
// <= max_rounds; among those the one that wastes the fewest lane-rounds (LP = 36 on G = 64 idles 44 % of the
// lanes, on G = 16 x 3 rounds 25 %), larger G on ties (fewer rounds).  Used by the plain forward only: the fused
// kernels redo their prologue every round and measured slower with more rounds.
static void so_pick_group(int LP, int d, int max_rounds, int &G, int &logG) {
    int best_l = 6;
    double best_u = -1.0;
    for (int lg = 0; lg <= 6; ++lg) {
        const int g = 1 << lg;
        if (g < d / 4) continue;
        const int rounds = (LP + g - 1) / g;
        if (rounds > max_rounds && lg < 6) continue;
        const double u = (double)LP / ((double)rounds * g);
        if (u >= best_u - 1e-12) { best_u = u; best_l = lg; }
    }
    logG = best_l;
    G = 1 << best_l;
}

// Fused kernels: a lane holds at most so_maxr(log2 G) points; among the group sizes that allows, the best lane
// utilisation LP / (rounds * G), ties to the larger group (fewest rounds: the softmax exchanges are per group, the
// prologue per round).  L * P = 36 (the shipped self-attention) -> G = 8, 5 rounds, 90 % live lanes (64 lanes: 56 %).
static void so_pick_group_fused(int LP, int d, int &G, int &logG) {
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
    G = 1 << best_l;
}

extern "C" int selfocc_msda_fwd(const float *value, const int32_t *shapes, const int32_t *starts,
                                const float *loc, const float *attw, float *out, int32_t bs,
                                int32_t nv, int32_t nq, int32_t heads, int32_t d, int32_t L,
                                int32_t P, void *stream) {
    if (validate(value, shapes, starts, loc, attw, bs, nv, nq, heads, d, L, P)) return -1;
    const long long n_groups = (long long)bs * nq * heads;
    if (n_groups == 0) return 0;
    SO_REQUIRE(out != nullptr, "msda_fwd: out is NULL");
    if (nv == 0)   // nothing to sample: every point is outside every (empty) map
        return (int)hipMemsetAsync(out, 0, (size_t)n_groups * d * sizeof(float), (hipStream_t)stream);
    const int LP = L * P;
    int G = 1, logG = 0;
    so_pick_group(LP, d, 8, G, logG);
    const int gpb = 256 / G;
    const long long blocks = (n_groups + gpb - 1) / gpb;
    SO_REQUIRE(blocks < (1LL << 31), "msda_fwd: grid too large");
    MsdaDims dm{bs, nv, nq, heads, L, P, 0};
    hipStream_t st = (hipStream_t)stream;
#define SO_LAUNCH_G(DD, LG)                                                                       \
    hipLaunchKernelGGL((msda_fwd_kernel<DD, LG>), dim3((unsigned)blocks), dim3(256), 0, st, value, \
                       shapes, starts, loc, attw, out, dm)
    // so_pick_group returns whole channel teams (G >= D / 4): 22 of the 4 x 7 combinations can be reached
#define SO_LAUNCH(DD, LGMIN)                                                                      \
    switch (logG) {                                                                               \
        case 0: SO_LAUNCH_G(DD, (LGMIN > 0 ? LGMIN : 0)); break;                                  \
        case 1: SO_LAUNCH_G(DD, (LGMIN > 1 ? LGMIN : 1)); break;                                  \
        case 2: SO_LAUNCH_G(DD, (LGMIN > 2 ? LGMIN : 2)); break;                                  \
        case 3: SO_LAUNCH_G(DD, 3); break;                                                        \
        case 4: SO_LAUNCH_G(DD, 4); break;                                                        \
        case 5: SO_LAUNCH_G(DD, 5); break;                                                        \
        default: SO_LAUNCH_G(DD, 6); break;                                                       \
    }
    SO_REQUIRE((1 << logG) >= d / 4, "msda_fwd: group of %d lanes for %d channels", 1 << logG, d);
    switch (d) {
        case 4: SO_LAUNCH(4, 0); break;
        case 8: SO_LAUNCH(8, 1); break;
        case 16: SO_LAUNCH(16, 2); break;
        default: SO_LAUNCH(32, 3); break;
    }
#undef SO_LAUNCH
#undef SO_LAUNCH_G
    return so_launch_status();
}



// The (D, log2 G, value type) combinations of the fused / camera-loop families that can be launched: D = 8, 16, 32 channels per
// head (the shipped lifters use 16; the plain mmcv-boundary op keeps 4), whole channel teams (G >= D / 4: so_pick_group_fused
// never returns less), bfloat16 `value` for D = 16 only.  Round 4 instantiated all 4 x 7 x 2 = 56 per family (224 kernels, 6 MB).
#define SO_FUSED_LG(DD, LGMIN, VT)                                                       \
    switch (logG_) {                                                                      \
        case 1: if (LGMIN <= 1) { SO_LAUNCH_VT(DD, (LGMIN <= 1 ? 1 : LGMIN), VT); } break; \
        case 2: if (LGMIN <= 2) { SO_LAUNCH_VT(DD, (LGMIN <= 2 ? 2 : LGMIN), VT); } break; \
        case 3: SO_LAUNCH_VT(DD, 3, VT); break;                                           \
        case 4: SO_LAUNCH_VT(DD, 4, VT); break;                                           \
        case 5: SO_LAUNCH_VT(DD, 5, VT); break;                                           \
        default: SO_LAUNCH_VT(DD, 6, VT); break;                                          \
    }
#define SO_FUSED_DISPATCH(d_, lg_, bf_)                                                  \
    do {                                                                                  \
        const int logG_ = (lg_);                                                          \
        if ((d_) == 16) {                                                                 \
            SO_REQUIRE(logG_ >= 2, "msda: group of %d lanes for 16 channels", 1 << logG_);  \
            if (bf_) { SO_FUSED_LG(16, 2, uint16_t) } else { SO_FUSED_LG(16, 2, float) }  \
        } else {                                                                          \
            SO_REQUIRE(!(bf_), "msda: bfloat16 value is built for 16 channels per head only (got %d)", (int)(d_)); \
            if ((d_) == 8) { SO_REQUIRE(logG_ >= 1, "msda: bad group"); SO_FUSED_LG(8, 1, float) }          \
            else if ((d_) == 32) { SO_REQUIRE(logG_ >= 3, "msda: bad group"); SO_FUSED_LG(32, 3, float) }   \
            else SO_REQUIRE(false, "msda fused / camera-loop ops: channels per head must be 8, 16 or 32 (got %d); the plain op takes 4", (int)(d_)); \
        }                                                                                 \
    } while (0)

// off_raw / logits row strides of a launch: dense tensors, or one merged [offsets | logits] projection row per query
static inline int so_set_ol(MsdaDims &dm, int ol_stride, const float *off_raw, const float *logits, const char *who) {
    const int LP = dm.L * dm.P;
    if (ol_stride == 0) {
        dm.off_ld = dm.heads * LP * 2;
        dm.lg_ld = dm.heads * LP;
        return 0;
    }
    SO_REQUIRE(ol_stride >= 3 * dm.heads * LP && ol_stride % 2 == 0, "%s: ol_stride must be 0 (dense) or an even number >= 3 * heads * L * P = %d (got %d)",
               who, 3 * dm.heads * LP, ol_stride);
    SO_REQUIRE((((uintptr_t)off_raw) & 7) == 0, "%s: off_raw must be 8-byte aligned", who);
    SO_REQUIRE((long long)dm.bs * dm.nq * ol_stride < (1LL << 40), "%s: offsets / logits too large", who);
    dm.off_ld = dm.lg_ld = ol_stride;
    return 0;
}

extern "C" int selfocc_msda_fused_fwd(const void *value, const int32_t *shapes, const int32_t *starts,
                                      const float *ref, int32_t ref_kind, const float *off_raw, const float *logits,
                                      float *out, int32_t bs, int32_t nv, int32_t nq, int32_t heads, int32_t d,
                                      int32_t L, int32_t P, int32_t value_layout, int32_t value_dtype, int32_t ol_stride,
                                      void *stream) {
    if (validate((const float *)value, shapes, starts, off_raw, logits, bs, nv, nq, heads, d, L, P)) return -1;
    const long long n_groups = (long long)bs * nq * heads;
    if (n_groups == 0) return 0;
    SO_REQUIRE(out != nullptr && ref != nullptr, "msda_fused_fwd: NULL pointer");
    SO_REQUIRE(value_dtype == SO_DTYPE_F32 || value_dtype == SO_DTYPE_BF16, "msda_fused_fwd: bad value_dtype");
    SO_REQUIRE(value_layout == SO_VALUE_PIXEL_MAJOR || value_layout == SO_VALUE_HEAD_MAJOR, "msda_fused_fwd: bad value_layout");
    SO_REQUIRE(ref_kind >= 0 && ref_kind <= 2, "msda_fused_fwd: ref_kind must be 0, 1 or 2");
    if (nv == 0)
        return (int)hipMemsetAsync(out, 0, (size_t)n_groups * d * sizeof(float), (hipStream_t)stream);
    const int LP = L * P;
    SO_REQUIRE(LP <= 256, "msda_fused_fwd: L * P must be <= 256 (got %d); use the unfused op", LP);
    int G = 1, logG = 0;
    so_pick_group_fused(LP, d, G, logG);
    const int gpb = 256 / G;
    const long long blocks = (n_groups + gpb - 1) / gpb;
    SO_REQUIRE(blocks < (1LL << 31), "msda_fused_fwd: grid too large");
    MsdaDims dm{bs, nv, nq, heads, L, P, 0, 0, value_layout};
    if (so_set_ol(dm, ol_stride, off_raw, logits, "msda_fused_fwd")) return -1;
    hipStream_t st = (hipStream_t)stream;
#define SO_LAUNCH_VT(DD, LG, VT) \
        hipLaunchKernelGGL((msda_fused_fwd_kernel<DD, LG, VT>), dim3((unsigned)blocks), dim3(256), 0, st, \
                           (const VT *)value, shapes, starts, ref, ref_kind, off_raw, logits, out, dm)
    SO_FUSED_DISPATCH(d, logG, value_dtype == SO_DTYPE_BF16);
#undef SO_LAUNCH_VT
    return so_launch_status();
}

extern "C" int selfocc_msda_cross_fwd(const void *value, const int32_t *shapes, const int32_t *starts,
                                      const float *ref, const uint8_t *vis, const float *off_raw,
                                      const float *logits, float *out, int32_t cams, int32_t nv, int32_t nq,
                                      int32_t heads, int32_t d, int32_t L, int32_t P, int32_t value_stride,
                                      int32_t value_layout, int32_t value_dtype, int32_t ol_stride, void *stream) {
    SO_REQUIRE(cams >= 1, "msda_cross_fwd: cams must be >= 1");
    SO_REQUIRE(value_dtype == SO_DTYPE_F32 || value_dtype == SO_DTYPE_BF16, "msda_cross_fwd: bad value_dtype");
    SO_REQUIRE(value_layout == SO_VALUE_PIXEL_MAJOR || (value_layout == SO_VALUE_HEAD_MAJOR && value_stride == 0),
               "msda_cross_fwd: bad value_layout (head-major values are dense: value_stride must be 0)");
    SO_REQUIRE(value_stride == 0 || (value_stride >= heads * d && value_stride % 4 == 0),
               "msda_cross_fwd: value_stride must be 0 or a multiple of 4 >= heads * d");
    SO_REQUIRE((long long)cams * nv * (value_stride ? value_stride : heads * d) < (1LL << 31),
               "msda_cross_fwd: value must span < 2^31 floats");
    if (validate((const float *)value, shapes, starts, off_raw, logits, cams, nv, nq, heads, d, L, P)) return -1;
    const long long n_groups = (long long)nq * heads;
    if (n_groups == 0) return 0;
    SO_REQUIRE(out != nullptr && ref != nullptr && vis != nullptr, "msda_cross_fwd: NULL pointer");
    SO_REQUIRE(n_groups < (1LL << 31), "msda_cross_fwd: nq * heads must be < 2^31");
    const int LP = L * P;
    SO_REQUIRE(LP <= 256, "msda_cross_fwd: L * P must be <= 256 (got %d)", LP);
    if (nv == 0) return (int)hipMemsetAsync(out, 0, (size_t)n_groups * d * sizeof(float), (hipStream_t)stream);
    int G = 1, logG = 0;
    so_pick_group_fused(LP, d, G, logG);
    const int gpb = 256 / G;
    const long long blocks = (n_groups + gpb - 1) / gpb;
    SO_REQUIRE(blocks < (1LL << 31), "msda_cross_fwd: grid too large");
    MsdaDims dm{1, nv, nq, heads, L, P, 0, value_stride, value_layout};
    if (so_set_ol(dm, ol_stride, off_raw, logits, "msda_cross_fwd")) return -1;
    hipStream_t st = (hipStream_t)stream;
#define SO_LAUNCH_VT(DD, LG, VT) \
        hipLaunchKernelGGL((msda_cross_fwd_kernel<DD, LG, VT>), dim3((unsigned)blocks), dim3(256), 0, st, \
                           (const VT *)value, shapes, starts, ref, vis, off_raw, logits, out, cams, dm)
    SO_FUSED_DISPATCH(d, logG, value_dtype == SO_DTYPE_BF16);
#undef SO_LAUNCH_VT
    return so_launch_status();
}

extern "C" int selfocc_msda_bwd(const float *value, const int32_t *shapes, const int32_t *starts,
                                const float *loc, const float *attw, const float *g_out,
                                float *g_value, float *g_loc, float *g_attw, int32_t bs,
                                int32_t nv, int32_t nq, int32_t heads, int32_t d, int32_t L,
                                int32_t P, void *stream) {
    if (validate(value, shapes, starts, loc, attw, bs, nv, nq, heads, d, L, P)) return -1;
    const long long n_pts = (long long)bs * nq * heads * L * P;
    if (n_pts == 0) return 0;
    SO_REQUIRE(g_out && g_value && g_loc && g_attw, "msda_bwd: NULL gradient pointer");
    const long long blocks = (n_pts + 255) / 256;
    SO_REQUIRE(blocks < (1LL << 31), "msda_bwd: grid too large");
    MsdaDims dm{bs, nv, nq, heads, L, P, 0};
    hipStream_t st = (hipStream_t)stream;
#define SO_LAUNCH(DD)                                                                             \
    hipLaunchKernelGGL((msda_bwd_kernel<DD>), dim3((unsigned)blocks), dim3(256), 0, st, value,    \
                       shapes, starts, loc, attw, g_out, g_value, g_loc, g_attw, dm)
    switch (d) {
        case 4: SO_LAUNCH(4); break;
        case 8: SO_LAUNCH(8); break;
        case 16: SO_LAUNCH(16); break;
        default: SO_LAUNCH(32); break;
    }
#undef SO_LAUNCH
    return so_launch_status();
}


// ---- banded backward -------------------------------------------------------------------------
static size_t so_band_key_bytes(long long n_pts) { return (size_t)((n_pts + 8) * 2 + 15) / 16 * 16; }
// counters of the binned scatter: cnt / cursor / off per bucket (2 per band), item0 per band (+ 1)
static size_t so_bin_counter_bytes(int bs, int heads, int L) {
    const size_t nb = (size_t)bs * heads * L * kMaxBands;
    return ((7 * nb + 1) * 4 + 15) / 16 * 16;
}
// per point: 2 (key) + 16 (record) + 8 (index lists: a point is in <= 2 bands) bytes
static size_t so_band_ws_bytes(int bs, int nq, int heads, int L, int P) {
    const long long n_pts = (long long)bs * nq * heads * L * P;
    return so_band_key_bytes(n_pts) + (size_t)n_pts * 16 + so_bin_counter_bytes(bs, heads, L) + (size_t)n_pts * 8;
}

extern "C" size_t selfocc_msda_bwd_banded_workspace(int32_t bs, int32_t nq, int32_t heads, int32_t L, int32_t P) {
    if (bs < 0 || nq < 0 || heads < 1 || L < 1 || P < 1) return 0;
    return so_band_ws_bytes(bs, nq, heads, L, P);
}

namespace {
struct BandSetup {
    MsdaBinPlan bin;
    int threads;   // block shape of the band kernel (512 / 1024)
    int tile_px;   // pixels of the largest band
    bool ok;       // false: a level is wider than the LDS tile or needs > kMaxBands bands, or the index space overflows
};

int so_env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// bands: as many rows as the tile holds (the fewer bands, the fewer points straddle two)
int so_band_setup(const int32_t *host_shapes, int bs, int nq, int heads, int d, int L, int P, BandSetup &bsu) {
    // tuning switches (A/B runs): block shape 512 / 1024, list entries per block
    static const int env_threads = so_env_int("SELFOCC_BAND_THREADS", 512), env_seg = so_env_int("SELFOCC_BAND_SEG", 4096);
    MsdaBinPlan &bp = bsu.bin;
    bsu.threads = env_threads == 1024 ? 1024 : 512;
    const int cap_px = band_tile_bytes(bsu.threads) / (8 * d);
    bsu.ok = L <= 8;
    bsu.tile_px = 0;
    int nb = 0;
    for (int l = 0; l < 8; ++l) { bp.rows[l] = 1; bp.bands[l] = 0; bp.band0[l] = 0; bp.inv_rows[l] = 1.0f; }
    for (int l = 0; l < L && bsu.ok; ++l) {
        const int Hl = host_shapes[2 * l], Wl = host_shapes[2 * l + 1];
        SO_REQUIRE(Hl >= 0 && Wl >= 0 && Hl < 32767 && Wl < 32767, "msda banded: bad level shape (%d, %d)", Hl, Wl);
        bp.band0[l] = nb;
        if (Hl == 0 || Wl == 0) continue;
        if (Wl > cap_px) { bsu.ok = false; break; }
        bp.rows[l] = std::min(Hl, cap_px / Wl);
        bp.inv_rows[l] = 1.0f / (float)bp.rows[l];
        bp.bands[l] = (Hl + bp.rows[l] - 1) / bp.rows[l];
        if (bp.bands[l] > kMaxBands) { bsu.ok = false; break; }
        nb += bp.bands[l];
        bsu.tile_px = std::max(bsu.tile_px, bp.rows[l] * Wl);
    }
    for (int l = L; l <= 8; ++l) bp.band0[std::min(l, 8)] = nb;
    bp.nbands = nb;
    bp.seg = std::max(64, env_seg);
    int lg = 0;
    while ((1LL << lg) < P) ++lg;
    bp.divP.l = lg;
    bp.divP.m = (unsigned)((((1ULL << lg) - (unsigned)P) << 32) / (unsigned)P + 1);
    const long long n_pts = (long long)bs * nq * heads * L * P;
    if (nb == 0 || (long long)bs * heads * nb >= (1LL << 28) || 2 * n_pts >= (1LL << 31) ||
        (long long)nq * P >= (1LL << 31))
        bsu.ok = false;
    return 0;
}

struct BandWorkspace {
    int16_t *keys;
    float4 *recs;
    int32_t *counters;   // cnt[2 nb] cursor[2 nb] off[2 nb] item0[nb + 1], nb = bands actually used
    int32_t *list;       // 2 * n_pts entries
    long long n_pts;
};

BandWorkspace so_band_workspace(void *workspace, int bs, int nq, int heads, int L, int P) {
    const long long n_pts = (long long)bs * nq * heads * L * P;
    BandWorkspace w;
    w.keys = (int16_t *)workspace;
    w.recs = (float4 *)((char *)workspace + so_band_key_bytes(n_pts));
    w.counters = (int32_t *)((char *)w.recs + (size_t)n_pts * 16);
    w.list = (int32_t *)((char *)w.counters + so_bin_counter_bytes(bs, heads, L));
    w.n_pts = n_pts;
    return w;
}

// counting sort + band scatter (after a point kernel filled keys / recs).  vis != NULL: camera loop, the keys of
// invisible (camera, query) pairs are unwritten and must not be read (needs P >= 4, see msda_bin_kernel)
int so_band_scatter(const int32_t *shapes, const int32_t *starts, const float *g_out, float *g_value,
                    const BandWorkspace &w, const BandSetup &bsu, MsdaDims dm, int d, const unsigned char *vis,
                    hipStream_t st, bool counted = false, int gv_stride = 0) {
    const MsdaBinPlan &bp = bsu.bin;
    // g_value in its own layout (round 6): rows of gv_stride floats per pixel, this op's heads * d channels at the pointer —
    // a column block of the row-major gradient of a (stacked) value projection, whatever layout `value` itself was gathered from
    MsdaDims dm_g = dm;
    if (gv_stride > 0) { dm_g.hm = 0; dm_g.vs = gv_stride; }
    const int nb = dm.bs * dm.heads * bp.nbands;                 // bands over all (batch, head)
    int32_t *cnt = w.counters, *cursor = cnt + 2 * (size_t)nb, *off = cursor + 2 * (size_t)nb, *item0 = off + 2 * (size_t)nb;
    const long long n = (long long)dm.nq * dm.P;
    const int bpb = (int)((n + 3 + kBinKeysPerBlock - 1) / kBinKeysPerBlock);   // windows start 8-byte aligned: <= 3 keys early
    const long long bin_blocks = (long long)dm.bs * dm.heads * dm.L * bpb;
    const long long max_items = nb + (2 * w.n_pts) / bp.seg;
    SO_REQUIRE(bin_blocks < (1LL << 31) && max_items < (1LL << 31), "msda banded: grid too large");
    if (!counted) {      // counted: the point kernel zeroed-and-filled cnt (and zeroed cursor) already
        (void)hipMemsetAsync(cnt, 0, (size_t)nb * 4 * sizeof(int32_t), st);     // cnt and cursor
        hipLaunchKernelGGL(msda_bin_kernel<false>, dim3((unsigned)bin_blocks), dim3(256), 0, st, w.keys, shapes, cnt,
                           cursor, off, w.list, vis, bpb, dm, bp);
    }
    hipLaunchKernelGGL(msda_bin_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, off, item0, nb, bp.seg);
    hipLaunchKernelGGL(msda_bin_kernel<true>, dim3((unsigned)bin_blocks), dim3(256), 0, st, w.keys, shapes, cnt,
                       cursor, off, w.list, vis, bpb, dm, bp);
    const size_t shm = (size_t)bsu.tile_px * d * sizeof(double);
    // the grid is an upper bound (every point in two bands): blocks past item0[nb] return at once
#define SO_LAUNCH_T(DD, TT)                                                                                      \
    {                                                                                                            \
        /* per launch: the attribute is per device (a process may drive several GPUs) */                         \
        (void)hipFuncSetAttribute((const void *)msda_bwd_band_list_kernel<DD, TT>,                               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, band_tile_bytes(TT));              \
        hipLaunchKernelGGL((msda_bwd_band_list_kernel<DD, TT>), dim3((unsigned)max_items), dim3(TT), shm, st,    \
                           shapes, starts, g_out, g_value, w.recs, cnt, off, item0, w.list, nb, dm_g, bp);       \
    }
#define SO_LAUNCH(DD)                                                                                            \
    if (bsu.threads == 512) SO_LAUNCH_T(DD, 512) else SO_LAUNCH_T(DD, 1024)
    switch (d) {
        case 4: SO_LAUNCH(4); break;
        case 8: SO_LAUNCH(8); break;
        case 16: SO_LAUNCH(16); break;
        default: SO_LAUNCH(32); break;
    }
#undef SO_LAUNCH
#undef SO_LAUNCH_T
    return so_launch_status();
}
}  // namespace

/* 1: the banded scatter applies to these shapes (selfocc_msda_fused_bwd needs it; selfocc_msda_bwd_banded falls
 * back to selfocc_msda_bwd by itself), 0: it does not, -1: bad arguments */
extern "C" int selfocc_msda_banded_supported(const int32_t *host_shapes, int32_t bs, int32_t nq, int32_t heads,
                                             int32_t d, int32_t L, int32_t P) {
    SO_REQUIRE(host_shapes != nullptr && bs >= 0 && nq >= 0 && heads >= 1 && L >= 1 && P >= 1, "msda banded: bad arguments");
    SO_REQUIRE(d == 4 || d == 8 || d == 16 || d == 32, "msda: channels per head must be 4, 8, 16 or 32 (got %d)", d);
    if (L > 8) return 0;
    BandSetup bsu;
    if (so_band_setup(host_shapes, bs, nq, heads, d, L, P, bsu)) return -1;
    return bsu.ok ? 1 : 0;
}

extern "C" int selfocc_msda_bwd_banded(const float *value, const int32_t *shapes, const int32_t *starts,
                                       const int32_t *host_shapes, const float *loc, const float *attw,
                                       const float *g_out, float *g_value, float *g_loc, float *g_attw,
                                       int32_t bs, int32_t nv, int32_t nq, int32_t heads, int32_t d, int32_t L,
                                       int32_t P, void *workspace, size_t workspace_bytes, void *stream) {
    if (validate(value, shapes, starts, loc, attw, bs, nv, nq, heads, d, L, P)) return -1;
    const long long n_pts = (long long)bs * nq * heads * L * P;
    if (n_pts == 0) return 0;
    SO_REQUIRE(g_out && g_value && g_loc && g_attw, "msda_bwd_banded: NULL gradient pointer");
    SO_REQUIRE(host_shapes != nullptr, "msda_bwd_banded: host_shapes is NULL (host copy of the (L, 2) level shapes)");
    SO_REQUIRE(L <= 8, "msda_bwd_banded: at most 8 levels (got %d); use selfocc_msda_bwd", L);
    SO_REQUIRE(workspace != nullptr && workspace_bytes >= so_band_ws_bytes(bs, nq, heads, L, P),
               "msda_bwd_banded: workspace too small (%zu bytes, need %zu)", workspace_bytes,
               so_band_ws_bytes(bs, nq, heads, L, P));
    SO_REQUIRE(((uintptr_t)workspace & 15) == 0, "msda_bwd_banded: workspace must be 16-byte aligned");
    BandSetup bsu;
    if (so_band_setup(host_shapes, bs, nq, heads, d, L, P, bsu)) return -1;
    if (!bsu.ok)
        return selfocc_msda_bwd(value, shapes, starts, loc, attw, g_out, g_value, g_loc, g_attw, bs, nv, nq, heads,
                                d, L, P, stream);
    hipStream_t st = (hipStream_t)stream;
    MsdaDims dm{bs, nv, nq, heads, L, P, 0};
    const BandWorkspace w = so_band_workspace(workspace, bs, nq, heads, L, P);
    const long long pblocks = (n_pts + 255) / 256;
    SO_REQUIRE(pblocks < (1LL << 31), "msda_bwd_banded: grid too large");
#define SO_LAUNCH(DD)                                                                                        \
    hipLaunchKernelGGL((msda_bwd_point_kernel<DD>), dim3((unsigned)pblocks), dim3(256), 0, st, value,        \
                       shapes, starts, loc, attw, g_out, g_loc, g_attw, w.keys, w.recs, dm)
    switch (d) {
        case 4: SO_LAUNCH(4); break;
        case 8: SO_LAUNCH(8); break;
        case 16: SO_LAUNCH(16); break;
        default: SO_LAUNCH(32); break;
    }
#undef SO_LAUNCH
    return so_band_scatter(shapes, starts, g_out, g_value, w, bsu, dm, d, nullptr, st);
}

extern "C" int selfocc_msda_fused_bwd(const void *value, const int32_t *shapes, const int32_t *starts,
                                      const int32_t *host_shapes, const float *ref, int32_t ref_kind,
                                      const float *off_raw, const float *logits, const float *g_out,
                                      float *g_value, float *g_off, float *g_logits, int32_t bs, int32_t nv,
                                      int32_t nq, int32_t heads, int32_t d, int32_t L, int32_t P, int32_t value_layout,
                                      int32_t value_dtype, int32_t ol_stride, int32_t g_value_stride, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    if (validate((const float *)value, shapes, starts, off_raw, logits, bs, nv, nq, heads, d, L, P)) return -1;
    SO_REQUIRE(value_dtype == SO_DTYPE_F32 || value_dtype == SO_DTYPE_BF16, "msda_fused_bwd: bad value_dtype");
    const long long n_groups = (long long)bs * nq * heads;
    if (n_groups == 0) return 0;
    SO_REQUIRE(ref && g_out && g_value && g_off && g_logits, "msda_fused_bwd: NULL pointer");
    SO_REQUIRE(value_layout == SO_VALUE_PIXEL_MAJOR || value_layout == SO_VALUE_HEAD_MAJOR, "msda_fused_bwd: bad value_layout");
    SO_REQUIRE(ref_kind >= 0 && ref_kind <= 2, "msda_fused_bwd: ref_kind must be 0, 1 or 2");
    SO_REQUIRE(host_shapes != nullptr, "msda_fused_bwd: host_shapes is NULL (host copy of the (L, 2) level shapes)");
    const int LP = L * P;
    SO_REQUIRE(LP <= 256, "msda_fused_bwd: L * P must be <= 256 (got %d); use the unfused op", LP);
    SO_REQUIRE(workspace != nullptr && workspace_bytes >= so_band_ws_bytes(bs, nq, heads, L, P),
               "msda_fused_bwd: workspace too small (%zu bytes, need %zu)", workspace_bytes,
               so_band_ws_bytes(bs, nq, heads, L, P));
    SO_REQUIRE(((uintptr_t)workspace & 15) == 0, "msda_fused_bwd: workspace must be 16-byte aligned");
    BandSetup bsu;
    if (so_band_setup(host_shapes, bs, nq, heads, d, L, P, bsu)) return -1;
    SO_REQUIRE(bsu.ok, "msda_fused_bwd: the banded scatter does not apply to these shapes "
                       "(check selfocc_msda_banded_supported and use the unfused op)");
    hipStream_t st = (hipStream_t)stream;
    MsdaDims dm{bs, nv, nq, heads, L, P, 0, 0, value_layout};
    if (so_set_ol(dm, ol_stride, off_raw, logits, "msda_fused_bwd")) return -1;
    SO_REQUIRE(g_value_stride == 0 || (g_value_stride >= heads * d && (long long)bs * nv * g_value_stride < (1LL << 31)),
               "msda_fused_bwd: g_value_stride must be 0 or >= heads * d with bs * nv * stride < 2^31 (got %d)", g_value_stride);
    SO_REQUIRE(ol_stride == 0 || (((uintptr_t)g_off) & 7) == 0, "msda_fused_bwd: g_off must be 8-byte aligned");
    if (nv == 0) {   // every point is outside every (empty) map: all gradients are zero
        if (ol_stride)   // merged rows: [offsets | logits] of a query are adjacent
            return (int)hipMemset2DAsync(g_off, (size_t)ol_stride * sizeof(float), 0, (size_t)3 * heads * LP * sizeof(float), (size_t)bs * nq, st);
        (void)hipMemsetAsync(g_off, 0, (size_t)n_groups * LP * 2 * sizeof(float), st);
        return (int)hipMemsetAsync(g_logits, 0, (size_t)n_groups * LP * sizeof(float), st);
    }
    const BandWorkspace w = so_band_workspace(workspace, bs, nq, heads, L, P);
    int G = 1, logG = 0;
    so_pick_group_fused(LP, d, G, logG);
    const int gpb = 256 / G;
    const long long blocks = (n_groups + gpb - 1) / gpb;
    SO_REQUIRE(blocks < (1LL << 31), "msda_fused_bwd: grid too large");
    // (counting the sort's buckets inside this kernel, as selfocc_msda_cross_bwd does, measured slower here: 454 -> 570 us for
    // 58 us of msda_bin_kernel<false>: 32 groups per block contend for the few buckets of one plane)
#define SO_LAUNCH_VT(DD, LG, VT) \
        hipLaunchKernelGGL((msda_fused_bwd_point_kernel<DD, LG, VT>), dim3((unsigned)blocks), dim3(256), 0, st, \
                           (const VT *)value, shapes, starts, ref, ref_kind, off_raw, logits, g_out, g_off, \
                           g_logits, w.keys, w.recs, dm)
    SO_FUSED_DISPATCH(d, logG, value_dtype == SO_DTYPE_BF16);
#undef SO_LAUNCH_VT
    return so_band_scatter(shapes, starts, g_out, g_value, w, bsu, dm, d, nullptr, st, false, g_value_stride);
}


extern "C" int selfocc_msda_cross_bwd(const void *value, const int32_t *shapes, const int32_t *starts,
                                      const int32_t *host_shapes, const float *ref, const uint8_t *vis,
                                      const float *off_raw, const float *logits, const float *g_out,
                                      float *g_value, float *g_off, float *g_logits, int32_t cams, int32_t nv,
                                      int32_t nq, int32_t heads, int32_t d, int32_t L, int32_t P, int32_t value_layout,
                                      int32_t value_dtype, int32_t ol_stride, int32_t g_value_stride, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    SO_REQUIRE(cams >= 1, "msda_cross_bwd: cams must be >= 1");
    SO_REQUIRE(value_dtype == SO_DTYPE_F32 || value_dtype == SO_DTYPE_BF16, "msda_cross_bwd: bad value_dtype");
    SO_REQUIRE(value_layout == SO_VALUE_PIXEL_MAJOR || value_layout == SO_VALUE_HEAD_MAJOR, "msda_cross_bwd: bad value_layout");
    if (validate((const float *)value, shapes, starts, off_raw, logits, cams, nv, nq, heads, d, L, P)) return -1;
    const long long n_groups = (long long)nq * heads;
    if (n_groups == 0) return 0;
    SO_REQUIRE(ref && vis && g_out && g_value && g_off && g_logits, "msda_cross_bwd: NULL pointer");
    SO_REQUIRE(n_groups < (1LL << 31), "msda_cross_bwd: nq * heads must be < 2^31");
    SO_REQUIRE(host_shapes != nullptr, "msda_cross_bwd: host_shapes is NULL (host copy of the (L, 2) level shapes)");
    const int LP = L * P;
    SO_REQUIRE(LP <= 256, "msda_cross_bwd: L * P must be <= 256 (got %d)", LP);
    SO_REQUIRE(workspace != nullptr && workspace_bytes >= so_band_ws_bytes(cams, nq, heads, L, P),
               "msda_cross_bwd: workspace too small (%zu bytes, need %zu)", workspace_bytes,
               so_band_ws_bytes(cams, nq, heads, L, P));
    SO_REQUIRE(((uintptr_t)workspace & 15) == 0, "msda_cross_bwd: workspace must be 16-byte aligned");
    BandSetup bsu;
    if (so_band_setup(host_shapes, cams, nq, heads, d, L, P, bsu)) return -1;
    SO_REQUIRE(bsu.ok, "msda_cross_bwd: the banded scatter does not apply to these shapes "
                       "(check selfocc_msda_banded_supported with bs = cams)");
    hipStream_t st = (hipStream_t)stream;
    const long long n_pts = (long long)cams * nq * heads * LP;
    MsdaDims dm{cams, nv, nq, heads, L, P, 1, 0, value_layout};
    if (so_set_ol(dm, ol_stride, off_raw, logits, "msda_cross_bwd")) return -1;
    SO_REQUIRE(g_value_stride == 0 || (g_value_stride >= heads * d && (long long)cams * nv * g_value_stride < (1LL << 31)),
               "msda_cross_bwd: g_value_stride must be 0 or >= heads * d with cams * nv * stride < 2^31 (got %d)", g_value_stride);
    SO_REQUIRE(ol_stride == 0 || (((uintptr_t)g_off) & 7) == 0, "msda_cross_bwd: g_off must be 8-byte aligned");
    if (nv == 0) {
        if (ol_stride)
            return (int)hipMemset2DAsync(g_off, (size_t)ol_stride * sizeof(float), 0, (size_t)3 * heads * LP * sizeof(float), (size_t)nq, st);
        (void)hipMemsetAsync(g_off, 0, (size_t)n_groups * LP * 2 * sizeof(float), st);
        return (int)hipMemsetAsync(g_logits, 0, (size_t)n_groups * LP * sizeof(float), st);
    }
    const BandWorkspace w = so_band_workspace(workspace, cams, nq, heads, L, P);
    // the point kernel writes the keys of every (camera, query) pair it visits; the bin kernels skip the others by
    // `vis` (P >= 4), otherwise the unvisited keys are preset to "outside"
    const unsigned char *bin_vis = P >= 4 ? vis : nullptr;
    if (bin_vis == nullptr) (void)hipMemsetD16Async((hipDeviceptr_t)w.keys, (unsigned short)0x8000, (size_t)n_pts, st);
    int G = 1, logG = 0;
    so_pick_group_fused(LP, d, G, logG);
    const int gpb = 256 / G;
    const long long blocks = (n_groups + gpb - 1) / gpb;
    SO_REQUIRE(blocks < (1LL << 31), "msda_cross_bwd: grid too large");
    // the point kernel does the sort's counting pass itself (an LDS histogram per block, flushed once) when a block's groups
    // span at most two heads and the histogram fits: saves one pass over the keys (msda_bin_kernel<false>, 0.9 ms per iteration)
    static const bool count_env = so_env_int("SELFOCC_MSDA_COUNT_IN_POINT", 1) != 0;
    const size_t hist_need = (size_t)2 * cams * bsu.bin.nbands * 2 * sizeof(int);
    const bool count_here = count_env && nq >= gpb && hist_need <= 32 * 1024;
    const size_t hist_bytes = count_here ? hist_need : 0;
    if (count_here) {
        const size_t nb_all = (size_t)cams * heads * bsu.bin.nbands;
        (void)hipMemsetAsync(w.counters, 0, nb_all * 4 * sizeof(int32_t), st);      // cnt and cursor, before the counting kernel
    }
#define SO_LAUNCH_VT(DD, LG, VT) \
        hipLaunchKernelGGL((msda_cross_bwd_point_kernel<DD, LG, VT>), dim3((unsigned)blocks), dim3(256), hist_bytes, st, \
                           (const VT *)value, shapes, starts, ref, vis, off_raw, logits, g_out, g_off, g_logits, \
                           w.keys, w.recs, cams, dm, count_here ? w.counters : nullptr, bsu.bin)
    SO_FUSED_DISPATCH(d, logG, value_dtype == SO_DTYPE_BF16);
#undef SO_LAUNCH_VT
    return so_band_scatter(shapes, starts, g_out, g_value, w, bsu, dm, d, bin_vis, st, count_here, g_value_stride);
}
