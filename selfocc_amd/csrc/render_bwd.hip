// render_bwd.hip — backward of the fused SDF ray-march renderer for gfx950 (MI355X).
//
// What the reference gets from autograd through ~40 torch ops + cuda_gridsample_grad2's
// double-backward: d loss / d volume (SDF + colour/semantic channels) and
// d loss / d inv_s, given upstream gradients of every differentiable output of
// selfocc_render_fwd: depth, acc, rgb, sem, per-sample weights, per-sample sdf and the
// per-sample metre gradient (eikonal term).
//
// Hardware mapping: ONE WAVEFRONT PER RAY, lane l owns the M = ceil(S / 64) consecutive
// samples [l*M, (l+1)*M).  All per-sample state lives in registers; the two recurrences
//   transmittance  T_i = prod_{j<i} f_j,            f_j = 1 - alpha_j + 1e-7   (forward)
//   E_i = Gw_{i+1} alpha_{i+1} + f_{i+1} E_{i+1}                                 (reverse)
// are scans over wavefront shuffles (the reverse one composes affine maps, so there is no
// division by the tiny f_i that a "suffix sum / f_i" formulation would need).
// d L / d alpha_i = T_i (Gw_i - E_i), Gw_i = total derivative of the loss wrt weight i.
// The volume gradient is a scatter of 8 (+ 8 * n_feat) hardware float atomics per sample.
#include "so_device.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kMaxM = 8;  // samples per lane: S <= 512

struct RayGeomB {
    float ox, oy, oz, dx, dy, dz, dn;
};

SO_DEVFN RayGeomB load_ray(const so_render_args &a, int ray) {
    RayGeomB g;
    if (a.ray_mode == SO_RAYS_PIXEL_GRID) {
        const int per_cam = a.nx * a.ny;
        const int cam = ray / per_cam, rem = ray - cam * per_cam;
        const int iy = rem / a.nx, ix = rem - iy * a.nx;
        const float *M = a.img2lidar + cam * 16;
        const float u = (float)ix * a.sx + a.ox, v = (float)iy * a.sy + a.oy;
        g.ox = M[3]; g.oy = M[7]; g.oz = M[11];
        const float dx = (M[0] * u + M[1] * v) + M[2];
        const float dy = (M[4] * u + M[5] * v) + M[6];
        const float dz = (M[8] * u + M[9] * v) + M[10];
        g.dn = sqrtf((dx * dx + dy * dy) + dz * dz);
        g.dx = dx / g.dn; g.dy = dy / g.dn; g.dz = dz / g.dn;
    } else {
        g.ox = a.origins[3 * (size_t)ray]; g.oy = a.origins[3 * (size_t)ray + 1]; g.oz = a.origins[3 * (size_t)ray + 2];
        g.dx = a.dirs[3 * (size_t)ray]; g.dy = a.dirs[3 * (size_t)ray + 1]; g.dz = a.dirs[3 * (size_t)ray + 2];
        g.dn = a.dir_norm ? a.dir_norm[ray] : 1.0f;
    }
    return g;
}

SO_DEVFN float bin01(int j, int n) {
    const float step = 1.0f / (float)n;
    return (j < (n + 1) / 2) ? step * (float)j : fmaf(-step, (float)(n - j), 1.0f);
}

SO_DEVFN float edge_t(const so_render_args &a, int ray, int j, float tn, float tf) {
    const int n = a.n_samples;
    float b = bin01(j, n);
    if (a.jitter_mode != SO_JITTER_NONE) {
        const float lo = (j == 0) ? b : (b + bin01(j - 1, n)) / 2.0f;
        const float hi = (j == n) ? b : (bin01(j + 1, n) + b) / 2.0f;
        const float tr = (a.jitter_mode == SO_JITTER_SINGLE) ? a.t_rand[ray] : a.t_rand[(size_t)ray * (n + 1) + j];
        b = lo + (hi - lo) * tr;
    }
    return b * tf + (1.0f - b) * tn;
}

SO_DEVFN float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int NF, bool BF16>
SO_DEVFN void load_feat(const void *vol, size_t vox, float f[NF > 0 ? NF : 1]) {
    if constexpr (!BF16) {
        const float4 *p = (const float4 *)((const float *)vol + vox * NF);
#pragma unroll
        for (int q = 0; q < NF / 4; ++q) {
            const float4 t = p[q];
            f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
        }
    } else {
        const uint2 *p = (const uint2 *)((const uint16_t *)vol + vox * NF);
#pragma unroll
        for (int q = 0; q < NF / 4; ++q) {
            const uint2 t = p[q];
            f[4 * q] = __uint_as_float(t.x << 16); f[4 * q + 1] = __uint_as_float(t.x & 0xffff0000u);
            f[4 * q + 2] = __uint_as_float(t.y << 16); f[4 * q + 3] = __uint_as_float(t.y & 0xffff0000u);
        }
    }
}

// ---- brick-binned scatter (so_render_bwd_args::scatter_ws) -----------------------------------------------
// The volume is cut into bricks of kBH x kBW x kBD CELLS; the voxels a brick's samples touch are the
// (kBH + 1)(kBW + 1)(kBD + 1) tile that shares its upper faces with the neighbouring bricks.
//   rb_count_kernel   one wave per ray: every sample's cell (same code as the ray kernel) -> its brick; counts the samples
//                     per (brick, shard): runs of consecutive samples with one counter, summed per 16-ray block in LDS first
//   rb_scan1 / scan2  counts -> slot cursors per (brick, shard) and a list of work items (brick, <= chunk samples)
//   render_bwd_kernel<.., BIN = true>  takes the slots of its runs (one returning atomic per run, issued early) and
//                     writes ONE RECORD PER SAMPLE AT ITS SLOT, i.e. in brick order.  A record is RECF floats:
//                         [0 .. NCH)        d L / d (interpolated feature channel)
//                         [RECF - 8]        ds                (coefficient of the trilinear weights in d L / d sdf corner)
//                         [RECF - 7]        packed cell       ((h0 + 2) << 20 | (w0 + 2) << 10 | (d0 + 2), as bits; indices clamped to [-2, 1021])
//                         [RECF - 6 .. -4]  fh1, fw1, fd1     (fractions inside the cell)
//                         [RECF - 3 .. -1]  qx, qy, qz        (coefficients of the weights' axis derivatives)
//   rb_brick_kernel   one workgroup per item: streams the item's records, sums them into the brick's tile in LDS and
//                     adds the tile's non-zero rows to the gradient volumes: (items x touched rows) row atomics instead
//                     of (sample runs x 8).  The tile is DOUBLE: ds_add_f64 issues in ~15 clocks per 64-lane
//                     instruction on gfx950, ds_add_f32 in ~190 (scripts/micro/lds_atomic_types.hip; the first version
//                     of this kernel, with a float tile, took 7.7 ms).
constexpr int kBH = 4, kBW = 4, kBD = 8;
constexpr int kTH = kBH + 1, kTW = kBW + 1, kTD = kBD + 1;
constexpr int kTileVox = kTH * kTW * kTD;   // 225
constexpr int kInvsSlots = 1024;
constexpr int kShards = 32;  // counters per brick (shard = bits 8.. of the sample index): the bricks around the cameras
                             // are entered by every ray, and same-address device atomics serialise
SO_DEVFN int rb_shard(long long sample) { return (int)(sample >> 8) & (kShards - 1); }

template <int NF>
struct RbRec {
    static constexpr int NCH = NF == 4 ? 3 : NF;                      // feature channels with a gradient
    static constexpr int RECF = NF >= 20 ? 32 : (NF >= 4 ? 16 : 8);   // floats per record = lanes per sample
    static constexpr int RW = NCH + 1;                                 // tile row: features then the sdf column
};

struct RbBin {
    float *rec;        // [n_rays * n_samples][RECF], in brick order
    int *counts;       // [n_bricks][kShards] (zeroed per call, together with invs_part and n_items)
    float *invs_part;  // [kInvsSlots]     partial sums of d L / d inv_s
    int *n_items;      // [1]
    int *cursor;       // [n_bricks][kShards] next free slot
    int2 *blk_tot;     // [ceil(n_bricks / 1024)] (samples, items) per scan block
    int4 *items;       // [max_items] {brick, begin, end, -}
    int nbh, nbw, nbd, chunk;
    int dbg;           // dev switches (SELFOCC_RB_DBG): 1 = no accumulation, 2 = no flush, 4 = no run merging, 8 = the ray kernel
                       // does not write the feature part of its records, 16 = the brick kernel does not read it (timing only:
                       // 8 + 16 = the traffic of a 32-byte record, profiles/r5_c_render_bwd_record_bound.txt)
};

SO_DEVFN int rb_key(const RbBin &b, const so_cell &c, int H, int W, int D) {
    const int bh = min(max(c.h0, 0), H - 1) / kBH, bw = min(max(c.w0, 0), W - 1) / kBW, bd = min(max(c.d0, 0), D - 1) / kBD;
    return (bh * b.nbw + bw) * b.nbd + bd;
}

// entry / exit of the ray in the box collider, and the cell of sample i: ONE definition for the ray kernel and the
// counting pre-pass (they must agree on every sample's brick)
SO_DEVFN void ray_bounds(const so_render_args &a, const RayGeomB &g, float &tn, float &tf) {
    const float fx = 1.0f / (g.dx + 1e-6f), fy = 1.0f / (g.dy + 1e-6f), fz = 1.0f / (g.dz + 1e-6f);
    const float t1 = (a.aabb[0] - g.ox) * fx, t2 = (a.aabb[3] - g.ox) * fx;
    const float t3 = (a.aabb[1] - g.oy) * fy, t4 = (a.aabb[4] - g.oy) * fy;
    const float t5 = (a.aabb[2] - g.oz) * fz, t6 = (a.aabb[5] - g.oz) * fz;
    tn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    tn = fmaxf(tn, a.near_plane);
    tf = fmaxf(tf, tn + 1e-6f);
}

SO_DEVFN so_cell sample_cell(const so_render_args &a, const RayGeomB &g, float t0, float t1) {
    float px, py, pz;
    if (a.sample_pos == SO_SAMPLE_AT_START) {
        px = g.ox + g.dx * t0; py = g.oy + g.dy * t0; pz = g.oz + g.dz * t0;
    } else {
        const float tt = t0 + t1;
        px = g.ox + (g.dx * tt) / 2.0f; py = g.oy + (g.dy * tt) / 2.0f; pz = g.oz + (g.dz * tt) / 2.0f;
    }
    return so_locate(a.map, px, py, pz);
}

// run of consecutive lanes with one key: returns the run's length at its head lane (0 elsewhere) and the head's lane
SO_DEVFN int rb_run(int key, int lane, int &head_lane) {
    const int prev = __shfl_up(key, 1, 64);
    const bool head = (lane == 0) || (key != prev);
    const unsigned long long hm = __ballot(head);
    const unsigned long long upto = hm & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));   // heads at lanes <= mine
    head_lane = 63 - __clzll((long long)upto);
    if (!head) return 0;
    const unsigned long long later = (lane == 63) ? 0ull : (hm >> (lane + 1));
    return later ? __ffsll((long long)later) : 64 - lane;
}

// WPR = waves per ray.  WPR == 1: a wave owns a ray and M * 64 samples (4 rays per block).  WPR == 4: the
// block's four waves share one ray, wave w taking step (j * 4 + w) — at the shipped 256 samples per ray
// that is M = 1, which cuts the per-sample register state 4x (the M = 4 form needed > 256 VGPRs: one wave
// per SIMD, latency-bound); scan carries and per-ray sums cross the waves through a few floats of LDS.
// (Measured and dropped, round 4: forcing the 24-channel BIN instantiation to 3 / 4 waves per SIMD with
// amdgpu_waves_per_eu — 168 / 128 VGPRs, 296 / 484 B of scratch — 0.98 -> 1.67 / 1.96 ms: the spills land in the corner loops.)
template <int NF, bool BF16, int M, int WPR, bool BIN>
__global__ __launch_bounds__(256) void render_bwd_kernel(so_render_bwd_args ba, RbBin bin) {
    static_assert(WPR == 1 || WPR == 4, "waves per ray");
    constexpr int RECF = RbRec<NF>::RECF, NCH = RbRec<NF>::NCH;
    const so_render_args &a = ba.fwd;
    constexpr int NSEM = NF > 4 ? NF - 3 : 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = WPR == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    const int wstep = WPR == 1 ? 0 : wave;   // position of this wave inside a step group
    __shared__ float xch[4][8];
    // sum of v[0..N) over the waves that share the ray (block-uniform control flow when WPR == 4)
    auto ray_sum = [&](auto &v, auto nconst) {
        constexpr int N = decltype(nconst)::value;
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
        if constexpr (WPR > 1) {
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < N; ++k) xch[wave][k] = v[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = (xch[0][k] + xch[1][k]) + (xch[2][k] + xch[3][k]);
            __syncthreads();
        }
    };
    // per-wave transpose buffer of the feature-gradient scatter (phase B); odd record stride
    __shared__ __attribute__((aligned(16))) float lds_rec[BIN ? 4 : 4 * 64 * (NF > 0 ? (NF + 9 + (((NF + 9) & 1) ? 0 : 1)) : 9)];
    if (ray >= a.n_rays) return;  // wave-uniform (block-uniform when the waves share a ray)
    const int H = a.map.h.tot_len, W = a.map.w.tot_len, D = a.map.d.tot_len;
    const int S = a.n_samples;
    const RayGeomB g = load_ray(a, ray);

    float tn, tf;
    ray_bounds(a, g, tn, tf);

    // ---- phase A: per-sample forward state -------------------------------------------------
    so_cell cell[M];
    float alpha[M], fj[M], tmid[M], delta[M], Pc[M], Nc[M], sdfv[M], halfv[M];
    bool live[M], cneg[M], unclipped[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int i = (j * WPR + wstep) * 64 + lane;   // a step covers 64 CONSECUTIVE samples (one per lane)
        live[j] = i < S;
        const int ic = live[j] ? i : S - 1;
        const float t0 = edge_t(a, ray, ic, tn, tf), t1 = edge_t(a, ray, ic + 1, tn, tf);
        delta[j] = t1 - t0;
        tmid[j] = (t0 + t1) / 2.0f;
        cell[j] = sample_cell(a, g, t0, t1);
        float v[8], wk[8];
        so_gather_sdf(a.sdf_vol, H, W, D, cell[j], v);
        sdfv[j] = so_trilerp_sdf(cell[j], v, wk);
        float gx, gy, gz;
        so_trilerp_grad(cell[j], v, gx, gy, gz);
        const float cosv = (g.dx * gx + g.dy * gy) + g.dz * gz;
        cneg[j] = cosv < 0.0f;
        halfv[j] = (fminf(cosv, 0.0f) * delta[j]) * 0.5f;
        Pc[j] = so_sigmoid((sdfv[j] - halfv[j]) * so_inv_s(a));
        Nc[j] = so_sigmoid((sdfv[j] + halfv[j]) * so_inv_s(a));
        const float araw = ((Pc[j] - Nc[j]) + 1e-5f) / (Pc[j] + 1e-5f);
        unclipped[j] = (araw > 0.0f) && (araw < 1.0f);
        alpha[j] = live[j] ? fminf(fmaxf(araw, 0.0f), 1.0f) : 0.0f;
        fj[j] = live[j] ? (1.0f - alpha[j]) + 1e-7f : 1.0f;
    }
    // BIN: the slots of this wave's samples in the brick-ordered record array: one returning atomic per run of lanes
    // with one (brick, shard) counter, issued here so that its latency hides behind phase B's gathers
    int slot_base[M], slot_head[M];
    if constexpr (BIN) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
            int key = -1;
            if (live[j]) key = rb_key(bin, cell[j], H, W, D) * kShards + rb_shard((long long)ray * S + ((j * WPR + wstep) * 64 + lane));
            const int run = rb_run(key, lane, slot_head[j]);
            slot_base[j] = 0;
            if (run > 0 && key >= 0) slot_base[j] = atomicAdd(bin.cursor + key, run);
        }
    }
    // transmittance: exclusive prefix product over the samples in ray order = per step an exclusive scan
    // over the lanes (Hillis-Steele on shuffles) times the product of all earlier steps
    float T[M], w[M];
    float acc_l = 0.0f, dsum_l = 0.0f, carry = 1.0f;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        float incl = fj[j];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(incl, d, 64);
            if (lane >= d) incl *= o;
        }
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float tot = __shfl(incl, 63, 64);
        float before = carry;                     // product over all earlier steps
        if constexpr (WPR > 1) {
            if (lane == 0) xch[wave][0] = tot;
            __syncthreads();
#pragma unroll
            for (int ww = 0; ww < WPR; ++ww) {
                if (ww < wave) before *= xch[ww][0];
                carry *= xch[ww][0];
            }
            __syncthreads();
        } else {
            carry *= tot;
        }
        T[j] = before * excl;
        w[j] = alpha[j] * T[j];
        acc_l += w[j];
        dsum_l = fmaf(w[j], tmid[j], dsum_l);
    }
    float ad[2] = {acc_l, dsum_l};
    ray_sum(ad, std::integral_constant<int, 2>{});
    const float acc = ad[0], dsum = ad[1];
    const float inv_ae = 1.0f / (acc + 1e-10f);
    const float depth_raw = dsum * inv_ae;
    const float ddn = (a.flags & SO_FLAG_DEPTH_DIV_NORM) ? 1.0f / g.dn : 1.0f;

    // upstream per-ray gradients (wave-uniform)
    const float g_depth = ba.g_depth ? ba.g_depth[ray] * ddn : 0.0f;
    float g_accum = ba.g_acc ? ba.g_acc[ray] : 0.0f;
    float g_rgb[3] = {0.0f, 0.0f, 0.0f};
    if constexpr (NF > 0) {
        if (ba.g_rgb) {
            // rgb_k = clamp(sum_i w_i col_ik + bg_k (1 - acc)): the clamp and the background need the
            // forward value; recompute sum_i w_i col_ik below, so first pass: gather colours
        }
    }

    // ---- phase B: colour / semantics: Gw contributions + feature-volume scatter ---------------
    float Gw[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const size_t so = (size_t)ray * S + ((j * WPR + wstep) * 64 + lane);
        Gw[j] = (ba.g_weights && live[j]) ? ba.g_weights[so] : 0.0f;
        Gw[j] += g_depth * (tmid[j] - depth_raw) * inv_ae;
    }
    if constexpr (NF > 0) {
        float col[M][3];
        float rgb_l[3] = {0.0f, 0.0f, 0.0f};
        // pass 1: ONE gather of the 8 corners' feature rows per sample (round 6: the colours used to be gathered here and the whole
        // rows again in pass 2 — the corner gathers are 475 of the ray kernel's 994 us, profiles/r6_c_render_bwd_gather_bound.txt):
        // interpolated colour (kept: forward rgb for the clamp mask) and, with semantics, the sample's softmax probabilities (kept)
        // (the probabilities wait in lane-private LDS columns [j][k][thread], not in 21 registers: kept live across the ray
        // reduction they pushed the 24-channel kernel from 213 to 256 + 24 registers — one wave per SIMD — or into scratch)
        __shared__ float pk_s[(NSEM > 0 ? NSEM : 1) * M * 256];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float f3[3] = {0.0f, 0.0f, 0.0f};
            float lg[NSEM > 0 ? NSEM : 1];
#pragma unroll
            for (int k = 0; k < (NSEM > 0 ? NSEM : 1); ++k) lg[k] = 0.0f;
            const float fd[2] = {cell[j].fd0, cell[j].fd1}, fw[2] = {cell[j].fw0, cell[j].fw1}, fh[2] = {cell[j].fh0, cell[j].fh1};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int h = cell[j].h0 + (kk >> 2), ww = cell[j].w0 + ((kk >> 1) & 1), d = cell[j].d0 + (kk & 1);
                const bool in = (h >= 0) && (h < H) && (ww >= 0) && (ww < W) && (d >= 0) && (d < D);
                const int hc = min(max(h, 0), H - 1), wc = min(max(ww, 0), W - 1), dc = min(max(d, 0), D - 1);
                const float wgt = in ? (fd[kk & 1] * fw[(kk >> 1) & 1]) * fh[kk >> 2] : 0.0f;
                const size_t vox = ((size_t)hc * W + wc) * D + dc;
                if constexpr (NSEM > 0) {
                    float f[NF];
#ifdef SO_RB_NO_GATHER      // A/B build (scripts/build_variant.sh nogather render_bwd.hip -DSO_RB_NO_GATHER; timing only): no corner gathers —
#pragma unroll              // the bound of "keep the forward's interpolated features".  (As a RUN-TIME switch the branch cost the shipped
                    for (int k = 0; k < NF; ++k) f[k] = 0.01f * k;      // kernel 45 %: 1 040 -> 1 502 us.)
#else
                    load_feat<NF, BF16>(a.feat_vol, vox, f);
#endif
#pragma unroll
                    for (int k = 0; k < 3; ++k) f3[k] = fmaf(f[k], wgt, f3[k]);
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) lg[k] = fmaf(f[3 + k], wgt, lg[k]);
                } else {
                    float c3[3];
#ifdef SO_RB_NO_GATHER
                    c3[0] = 0.1f; c3[1] = 0.2f; c3[2] = 0.3f;
#else
                    if constexpr (!BF16) {
                        const float *p = (const float *)a.feat_vol + vox * NF;
                        c3[0] = p[0]; c3[1] = p[1]; c3[2] = p[2];
                    } else {
                        const uint16_t *p = (const uint16_t *)a.feat_vol + vox * NF;
                        c3[0] = so_bf16_to_f32(p[0]); c3[1] = so_bf16_to_f32(p[1]); c3[2] = so_bf16_to_f32(p[2]);
                    }
#endif
#pragma unroll
                    for (int k = 0; k < 3; ++k) f3[k] = fmaf(c3[k], wgt, f3[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                col[j][k] = 0.28209479177387814f * f3[k] + 0.5f;  // pre-relu
                rgb_l[k] = fmaf(w[j], fmaxf(col[j][k], 0.0f), rgb_l[k]);
            }
            if constexpr (NSEM > 0) {
                float mx = lg[0];
#pragma unroll
                for (int k = 1; k < NSEM; ++k) mx = fmaxf(mx, lg[k]);
                float den = 0.0f;
#pragma unroll
                for (int k = 0; k < NSEM; ++k) { lg[k] = so_expf(lg[k] - mx); den += lg[k]; }
                const float iden = 1.0f / den;
#pragma unroll
                for (int k = 0; k < NSEM; ++k) pk_s[(j * NSEM + k) * 256 + threadIdx.x] = lg[k] * iden;
            }
        }
        float bgk[3] = {0.0f, 0.0f, 0.0f};
        ray_sum(rgb_l, std::integral_constant<int, 3>{});
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (a.bkgd_mode == SO_BKGD_CONST) bgk[k] = a.bkgd[k];
            else if (a.bkgd_mode == SO_BKGD_PER_RAY) bgk[k] = a.bkgd_rays[3 * (size_t)ray + k];
            float r = rgb_l[k];
            if (a.bkgd_mode != SO_BKGD_NONE) r = r + bgk[k] * (1.0f - acc);
            float gk = ba.g_rgb ? ba.g_rgb[3 * (size_t)ray + k] : 0.0f;
            if ((a.flags & SO_FLAG_CLAMP_RGB) && (r < 0.0f || r > 1.0f)) gk = 0.0f;
            g_rgb[k] = gk;
            if (a.bkgd_mode != SO_BKGD_NONE) g_accum -= gk * bgk[k];
        }
        float g_semr[NSEM > 0 ? NSEM : 1];
        if constexpr (NSEM > 0) {
#pragma unroll
            for (int k = 0; k < NSEM; ++k) g_semr[k] = ba.g_sem ? ba.g_sem[(size_t)ray * NSEM + k] : 0.0f;
        }
        // pass 2: full feature vector per sample: Gw += g_rgb . col + g_sem . p; scatter d L / d feat.
        // The scatter is TRANSPOSED through LDS: each lane parks {cell, 8 corner weights, d L / d f[NF]}
        // of its sample, then GS = 2^ceil(log2 NF) consecutive lanes own the NF contiguous channels of one
        // sample's corner, so an atomic instruction touches 64 / GS segments of NF * 4 bytes instead of 64
        // scattered dwords (the lane-per-sample form cost 112 ms per nuscenes_occ iteration).
        constexpr int GS = NF <= 4 ? 4 : (NF <= 8 ? 8 : 32);       // lanes per sample in the scatter
        constexpr int REC = NF + 9 + (((NF + 9) & 1) ? 0 : 1);     // odd stride: conflict-free columns
        float *rec = lds_rec + wave * (64 * REC);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float df[NF];  // d L / d interpolated feature
#pragma unroll
            for (int k = 0; k < NF; ++k) df[k] = 0.0f;
            const float fd[2] = {cell[j].fd0, cell[j].fd1}, fw[2] = {cell[j].fw0, cell[j].fw1}, fh[2] = {cell[j].fh0, cell[j].fh1};
            if (live[j]) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Gw[j] = fmaf(g_rgb[k], fmaxf(col[j][k], 0.0f), Gw[j]);
                    df[k] = (col[j][k] > 0.0f) ? g_rgb[k] * w[j] * 0.28209479177387814f : 0.0f;
                }
                if constexpr (NSEM > 0) {
                    float pk[NSEM];
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) pk[k] = pk_s[(j * NSEM + k) * 256 + threadIdx.x];
                    float gp = 0.0f;
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) gp = fmaf(g_semr[k], pk[k], gp);
                    Gw[j] += gp;
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) df[3 + k] = w[j] * pk[k] * (g_semr[k] - gp);  // softmax backward
                }
            }
            if constexpr (BIN) {
                const int slot = __shfl(slot_base[j], slot_head[j], 64) + (lane - slot_head[j]);
                if (live[j] && !(bin.dbg & 8)) {   // the feature part of the sample's record, 16 bytes at a time
                    float4 *dst = (float4 *)(bin.rec + (size_t)slot * RECF);
#pragma unroll
                    for (int q = 0; q < (RECF - 8) / 4; ++q) {
                        float4 t;
                        t.x = (4 * q < NCH) ? df[(4 * q < NCH) ? 4 * q : 0] : 0.0f;
                        t.y = (4 * q + 1 < NCH) ? df[(4 * q + 1 < NCH) ? 4 * q + 1 : 0] : 0.0f;
                        t.z = (4 * q + 2 < NCH) ? df[(4 * q + 2 < NCH) ? 4 * q + 2 : 0] : 0.0f;
                        t.w = (4 * q + 3 < NCH) ? df[(4 * q + 3 < NCH) ? 4 * q + 3 : 0] : 0.0f;
                        dst[q] = t;
                    }
                }
            } else if (ba.g_feat_vol) {   // wave-uniform
                float *mine = rec + lane * REC;
                mine[0] = __int_as_float((cell[j].h0 * W + cell[j].w0) * D + cell[j].d0);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int h = cell[j].h0 + (kk >> 2), ww = cell[j].w0 + ((kk >> 1) & 1), d = cell[j].d0 + (kk & 1);
                    const bool in = live[j] && (h >= 0) && (h < H) && (ww >= 0) && (ww < W) && (d >= 0) && (d < D);
                    mine[1 + kk] = in ? (fd[kk & 1] * fw[(kk >> 1) & 1]) * fh[kk >> 2] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < NF; ++k) mine[9 + k] = (NF == 4 && k == 3) ? 0.0f : df[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int sub = lane % GS, grp = lane / GS;
                // Row grp serves the samples of lanes grp * GS .. + GS - 1, which are CONSECUTIVE on the ray; a
                // run of samples inside one voxel (about two at the shipped step / voxel ratio) shares its 8
                // corners, so the row sums the run's contributions and issues one set of atomics for it.
                // (The rows of one instruction are GS samples apart: different voxels, no shared L2 line.)
                for (int t = 0; t < GS; ++t) {
                    const float *r = rec + (grp * GS + t) * REC;
                    const int base = __float_as_int(r[0]);
                    if (t > 0 && __float_as_int(r[-REC]) == base) continue;   // inside a run: already added
                    float val[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    int tt = t;
                    const float *rr = r;
                    do {
                        const float dfc = (sub < NF) ? rr[9 + sub] : 0.0f;
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) val[kk] = fmaf(rr[1 + kk], dfc, val[kk]);
                        ++tt;
                        rr += REC;
                    } while (tt < GS && __float_as_int(rr[0]) == base);
                    if (sub < NF) {
#pragma unroll
                        for (int kk = 0; kk < 8; ++kk) {
                            if (val[kk] != 0.0f) {
                                const int vox = base + ((kk >> 2) * W + ((kk >> 1) & 1)) * D + (kk & 1);
                                unsafeAtomicAdd(ba.g_feat_vol + (size_t)vox * NF + sub, val[kk]);
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) Gw[j] += g_accum;

    // ---- phase C: reverse affine scan  E_i = Gw_{i+1} alpha_{i+1} + f_{i+1} E_{i+1} -----------
    // Each sample is the affine map x -> Gw_i alpha_i + f_i x; E_i is the composition of the maps of all
    // later samples applied to 0.  Per step (last step first): inclusive suffix composition over the lanes,
    // then E of lane l = (maps of lanes l+1 .. 63 of this step)(E after the step).
    float Ev[M];
    float E_end = 0.0f;   // E after the last sample of the current step
#pragma unroll
    for (int j = M - 1; j >= 0; --j) {
        float SA = Gw[j] * alpha[j], SB = fj[j];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float oa = __shfl_down(SA, d, 64), ob = __shfl_down(SB, d, 64);
            if (lane + d < 64) { SA = fmaf(SB, oa, SA); SB = SB * ob; }
        }
        const float na = __shfl_down(SA, 1, 64), nb = __shfl_down(SB, 1, 64);
        const float wa = __shfl(SA, 0, 64), wb = __shfl(SB, 0, 64);   // the whole step as one map
        float E_mine = E_end;                     // E after the last sample of THIS wave's step
        if constexpr (WPR > 1) {
            if (lane == 0) { xch[wave][0] = wa; xch[wave][1] = wb; }
            __syncthreads();
#pragma unroll
            for (int ww = WPR - 1; ww >= 0; --ww) {   // later steps are applied first
                if (ww > wave) E_mine = fmaf(xch[ww][1], E_mine, xch[ww][0]);
                E_end = fmaf(xch[ww][1], E_end, xch[ww][0]);
            }
            __syncthreads();
        } else {
            E_end = fmaf(wb, E_end, wa);
        }
        Ev[j] = (lane == 63) ? E_mine : fmaf(nb, E_mine, na);
    }

    float dinv_s_l = 0.0f;
    // SDF-volume records: {cell, 8 corner coefficients} per lane, in this wave's own slab of lds_rec (the same
    // slab as its feature records: with one wave per ray the waves of a block are not in step)
    float *srec = lds_rec + wave * (64 * (NF > 0 ? (NF + 9 + (((NF + 9) & 1) ? 0 : 1)) : 9));
#pragma unroll
    for (int j = M - 1; j >= 0; --j) {
        float dalpha = T[j] * (Gw[j] - Ev[j]);
        float coefs[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        float r_ds = 0.0f, r_qx = 0.0f, r_qy = 0.0f, r_qz = 0.0f;   // BIN: the record's sdf coefficients
        if (live[j]) {
            if (!unclipped[j]) dalpha = 0.0f;
            const float pe = Pc[j] + 1e-5f;
            const float dP = dalpha * (Nc[j] / (pe * pe));
            const float dN = -dalpha / pe;
            const float da = dP * Pc[j] * (1.0f - Pc[j]);
            const float db = dN * Nc[j] * (1.0f - Nc[j]);
            const size_t so = (size_t)ray * S + ((j * WPR + wstep) * 64 + lane);
            float ds = (da + db) * so_inv_s(a);
            const float dh = (db - da) * so_inv_s(a);
            dinv_s_l += da * (sdfv[j] - halfv[j]) + db * (sdfv[j] + halfv[j]);
            const float dc = cneg[j] ? dh * (delta[j] * 0.5f) : 0.0f;
            float dgx = dc * g.dx, dgy = dc * g.dy, dgz = dc * g.dz;
            if (ba.g_sdf) ds += ba.g_sdf[so];
            if (ba.g_grad) { dgx += ba.g_grad[3 * so]; dgy += ba.g_grad[3 * so + 1]; dgz += ba.g_grad[3 * so + 2]; }
            if (BIN && ba.g_sdf_vol) {
                r_ds = ds; r_qx = dgx * cell[j].sw; r_qy = dgy * cell[j].sh; r_qz = dgz * cell[j].sd;
            }
            if (!BIN && ba.g_sdf_vol) {
                // sdf = sum_k W_k v_k ; grad_axis = slope_axis * sum_k dW_k/d axis * v_k
                const so_cell &c = cell[j];
                const float fd[2] = {c.fd0, c.fd1}, fw[2] = {c.fw0, c.fw1}, fh[2] = {c.fh0, c.fh1};
                const float qx = dgx * c.sw, qy = dgy * c.sh, qz = dgz * c.sd;  // metre x<->w, y<->h, z<->d
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int kd = kk & 1, kw = (kk >> 1) & 1, kh = kk >> 2;
                    const int h = c.h0 + kh, ww = c.w0 + kw, d = c.d0 + kd;
                    const bool in = (h >= 0) && (h < H) && (ww >= 0) && (ww < W) && (d >= 0) && (d < D);
                    const float Wk = (fd[kd] * fw[kw]) * fh[kh];
                    const float dWd = (kd ? 1.0f : -1.0f) * (fw[kw] * fh[kh]);
                    const float dWw = (kw ? 1.0f : -1.0f) * (fd[kd] * fh[kh]);
                    const float dWh = (kh ? 1.0f : -1.0f) * (fd[kd] * fw[kw]);
                    coefs[kk] = in ? fmaf(Wk, ds, fmaf(dWd, qz, fmaf(dWw, qx, dWh * qy))) : 0.0f;
                }
            }
        }
        if constexpr (BIN) {
            const so_cell &c = cell[j];
            const int slot = __shfl(slot_base[j], slot_head[j], 64) + (lane - slot_head[j]);
            if (live[j]) {
                // so_locate does not clamp: a sample outside the box (a ray that misses it, near_plane past the exit, an
                // aabb larger than the mapping) has h0 <= -2 or h0 >= H.  The clamp keeps such an index OUTSIDE the volume
                // (-2 and 1021 >= tot_len fail every corner test of rb_brick_kernel), as the atomic path's `in` test does.
                const int pack = ((min(max(c.h0, -2), 1021) + 2) << 20) | ((min(max(c.w0, -2), 1021) + 2) << 10) |
                                 (min(max(c.d0, -2), 1021) + 2);
                float4 *dst = (float4 *)(bin.rec + (size_t)slot * RECF + (RECF - 8));
                dst[0] = make_float4(r_ds, __int_as_float(pack), c.fh1, c.fw1);
                dst[1] = make_float4(c.fd1, r_qx, r_qy, r_qz);
            }
        } else if (ba.g_sdf_vol) {   // wave-uniform
            // Scalar per-lane atomics would be one 64-byte fabric write each (8 per sample: as much traffic as
            // the whole feature scatter).  Instead 8-lane rows (one corner per lane) walk the samples in ray
            // order and add the coefficients of a run of samples inside one voxel before one atomic per corner.
            float *mine = srec + lane * 9;
            mine[0] = __int_as_float((cell[j].h0 * W + cell[j].w0) * D + cell[j].d0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) mine[1 + kk] = coefs[kk];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int sub = lane & 7, grp = lane >> 3;
            const int voff = ((sub >> 2) * W + ((sub >> 1) & 1)) * D + (sub & 1);
            for (int t = 0; t < 8; ++t) {
                const float *r = srec + (grp * 8 + t) * 9;
                const int base = __float_as_int(r[0]);
                if (t > 0 && __float_as_int(r[-9]) == base) continue;   // inside a run: already added
                float val = 0.0f;
                int tt = t;
                const float *rr = r;
                do {
                    val += rr[1 + sub];
                    ++tt;
                    rr += 9;
                } while (tt < 8 && __float_as_int(rr[0]) == base);
                // a coefficient is non-zero only for a corner inside the volume, so base + voff is a valid voxel
                if (val != 0.0f) unsafeAtomicAdd(ba.g_sdf_vol + (base + voff), val);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (ba.g_inv_s) {
        const float t = wave_sum(dinv_s_l);
        // BIN: kInvsSlots partial sums (rb_brick_kernel's first block adds them up) instead of one atomic per wave on ONE word
        if (lane == 0) unsafeAtomicAdd(BIN ? bin.invs_part + (blockIdx.x & (kInvsSlots - 1)) : ba.g_inv_s, t);
    }
}

// ---- the binned scatter's own kernels ---------------------------------------------------------------------
// one wave per ray (the ray's geometry once per lane, then S / 64 steps of 64 consecutive samples), 16 rays per block.
// Neighbouring rays cross the same bricks, and every ray starts in the bricks around the cameras: the block first sums its
// runs in a small LDS table (direct-mapped on the counter index; a collision goes to memory directly) and adds each
// occupied slot to the global counter once.  (0.24 ms with one atomic per run and 8 shards, 0.16 - 0.22 ms with 32 shards,
// with or without the table: what is left is the ~30 IEEE divisions per sample of the canonical cell, which the pass must
// repeat exactly.)
constexpr int kCountWaves = 16, kCountSlots = 1024;
__global__ __launch_bounds__(kCountWaves * 64) void rb_count_kernel(so_render_args a, RbBin b) {
    __shared__ int tkey[kCountSlots], tcnt[kCountSlots];
    for (int k = threadIdx.x; k < kCountSlots; k += kCountWaves * 64) { tkey[k] = -1; tcnt[k] = 0; }
    __syncthreads();
    const int ray = blockIdx.x * kCountWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray < a.n_rays) {   // wave-uniform
        const int S = a.n_samples;
        const RayGeomB g = load_ray(a, ray);
        float tn, tf;
        ray_bounds(a, g, tn, tf);
        const int H = a.map.h.tot_len, W = a.map.w.tot_len, D = a.map.d.tot_len;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int smp = s0 + lane;
            int key = -1;
            if (smp < S) {
                const so_cell c = sample_cell(a, g, edge_t(a, ray, smp, tn, tf), edge_t(a, ray, smp + 1, tn, tf));
                key = rb_key(b, c, H, W, D) * kShards + rb_shard((long long)ray * S + smp);
            }
            int hl;
            const int run = rb_run(key, lane, hl);
            if (run > 0 && key >= 0) {
                const int slot = (key * 0x9E3779B1u) >> 22;                  // 10 bits
                const int old = atomicCAS(&tkey[slot], -1, key);
                if (old == -1 || old == key) atomicAdd(&tcnt[slot], run);
                else atomicAdd(b.counts + key, run);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kCountSlots; k += kCountWaves * 64)
        if (tkey[k] >= 0 && tcnt[k] > 0) atomicAdd(b.counts + tkey[k], tcnt[k]);
}

// counts -> cursors + items, in two launches of ceil(n_bricks / 1024) blocks: per-block totals, then the scan proper
__global__ __launch_bounds__(1024) void rb_scan1_kernel(RbBin b) {
    __shared__ int sc[16], sn[16];
    const int nb = b.nbh * b.nbw * b.nbd;
    const int k = blockIdx.x * 1024 + threadIdx.x;
    int c = 0;
    if (k < nb) {
        const int4 *p = (const int4 *)(b.counts + (size_t)k * kShards);
#pragma unroll
        for (int q = 0; q < kShards / 4; ++q) {
            const int4 x = p[q];
            c += (x.x + x.y) + (x.z + x.w);
        }
    }
    int n = (c + b.chunk - 1) / b.chunk;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { c += __shfl_xor(c, m, 64); n += __shfl_xor(n, m, 64); }
    if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6] = c; sn[threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int cc = 0, nn = 0;
        for (int w = 0; w < 16; ++w) { cc += sc[w]; nn += sn[w]; }
        b.blk_tot[blockIdx.x] = make_int2(cc, nn);
    }
}

__global__ __launch_bounds__(1024) void rb_scan2_kernel(RbBin b) {
    __shared__ int sc[1024], sn[1024];
    static_assert(kShards % 4 == 0, "int4 loads");
    const int t = threadIdx.x;
    const int nb = b.nbh * b.nbw * b.nbd;
    const int k = blockIdx.x * 1024 + t;
    int cs[kShards];
    int c = 0;
    if (k < nb) {
        const int4 *p = (const int4 *)(b.counts + (size_t)k * kShards);
#pragma unroll
        for (int q = 0; q < kShards / 4; ++q) {
            const int4 x = p[q];
            cs[4 * q] = x.x; cs[4 * q + 1] = x.y; cs[4 * q + 2] = x.z; cs[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int sh = 0; sh < kShards; ++sh) c += cs[sh];
    }
    const int n = (c + b.chunk - 1) / b.chunk;
    int c0 = 0, n0 = 0;   // totals of the earlier blocks
    for (int q = 0; q < (int)blockIdx.x; ++q) { const int2 v = b.blk_tot[q]; c0 += v.x; n0 += v.y; }
    sc[t] = c; sn[t] = n;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {   // inclusive Hillis-Steele scans (samples, items)
        const int oc = t >= d ? sc[t - d] : 0, on = t >= d ? sn[t - d] : 0;
        __syncthreads();
        sc[t] += oc; sn[t] += on;
        __syncthreads();
    }
    if (k < nb) {
        const int coff = c0 + sc[t] - c;
        int noff = n0 + sn[t] - n;
        int run = coff;
        int4 *qv = (int4 *)(b.cursor + (size_t)k * kShards);   // a brick's shards are consecutive ranges of the sorted order
#pragma unroll
        for (int q = 0; q < kShards / 4; ++q) {
            int4 x;
            x.x = run; run += cs[4 * q]; x.y = run; run += cs[4 * q + 1]; x.z = run; run += cs[4 * q + 2]; x.w = run; run += cs[4 * q + 3];
            qv[q] = x;
        }
        for (int o = 0; o < c; o += b.chunk) b.items[noff++] = make_int4(k, coff + o, coff + min(o + b.chunk, c), 0);
    }
    if (blockIdx.x == gridDim.x - 1 && t == 1023) b.n_items[0] = n0 + sn[1023];
}

// One workgroup per item: the item's records (contiguous) summed into the brick's tile in LDS, the tile added to the volumes.
template <int NF, int NT>
__global__ __launch_bounds__(NT) void rb_brick_kernel(RbBin b, float *__restrict__ g_sdf_vol, float *__restrict__ g_feat_vol,
                                                      float *__restrict__ g_inv_s, int H, int W, int D) {
    constexpr int RECF = RbRec<NF>::RECF, NCH = RbRec<NF>::NCH, RW = RbRec<NF>::RW;
    constexpr int NG = NT / RECF, U = 4;
    extern __shared__ double tile[];   // [kTileVox][RW]
    if (blockIdx.x == 0 && g_inv_s) {   // the ray kernel's partial sums of d L / d inv_s: one atomic per wave of this block
        float t = 0.0f;
        for (int k = threadIdx.x; k < kInvsSlots; k += NT) t += b.invs_part[k];
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(g_inv_s, t);
    }
    if ((int)blockIdx.x >= b.n_items[0]) return;
    const int4 it = b.items[blockIdx.x];
    const int bd = it.x % b.nbd, bw = (it.x / b.nbd) % b.nbw, bh = it.x / (b.nbd * b.nbw);
    const int oh = bh * kBH, ow = bw * kBW, od = bd * kBD;
    for (int k = threadIdx.x; k < kTileVox * RW; k += NT) tile[k] = 0.0;
    __syncthreads();
    const int grp = threadIdx.x / RECF, sub = threadIdx.x % RECF;
    // lanes [0, NCH) of a group: one feature channel each, all 8 corners; lanes [RECF - 8, RECF): ONE corner each of the
    // sdf column (its coefficient has four terms: spreading the corners over the 8 otherwise idle tail lanes keeps the
    // per-corner loop of the feature lanes at one multiply)
    const bool sdf_lane = sub >= RECF - 8;
    const int mk = sub - (RECF - 8), mkd = mk & 1, mkw = (mk >> 1) & 1, mkh = (mk >> 2) & 1;
    const int moff = ((mkh * kTW + mkw) * kTD + mkd) * RW + NCH;
    // Group g walks the CONTIGUOUS slice [y + g * per, ...) of the item, U records at a time.  The records of an item are
    // in ray order (a run of consecutive samples of one ray takes consecutive slots), and consecutive samples of a ray share
    // their cell 2 - 6 times at the shipped step / voxel ratio: the group sums such a run in registers and issues its LDS
    // atomics once per run (the kernel sits on the ds_add_f64 issue rate: ~15 clocks per instruction, 35 M of them per launch
    // before this).  SELFOCC_RB_DBG & 4: no merging (every sample flushes).
    const int per = (it.z - it.y + NG - 1) / NG;
    const int gb = it.y + grp * per, ge = min(it.z, gb + per);
    const bool merge = !(b.dbg & 4);
    int cur = -1;                 // packed cell of the open run
    float acc[8];                 // feature lanes: the run's sum per corner; sdf lanes: acc[0] = the own corner's sum
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
    auto flush = [&]() {
        const int h0 = (cur >> 20) - 2, w0 = ((cur >> 10) & 1023) - 2, d0 = (cur & 1023) - 2;
        const int lh = h0 - oh, lw = w0 - ow, ld = d0 - od;
        // a corner counts when it is inside the volume AND inside this brick's tile (the second never fails: the
        // counting pass and the ray kernel derive the cell with the same code; it only keeps a mismatch inside the tile)
        const bool okh[2] = {((unsigned)h0 < (unsigned)H) && ((unsigned)lh < (unsigned)kTH),
                             ((unsigned)(h0 + 1) < (unsigned)H) && ((unsigned)(lh + 1) < (unsigned)kTH)};
        const bool okw[2] = {((unsigned)w0 < (unsigned)W) && ((unsigned)lw < (unsigned)kTW),
                             ((unsigned)(w0 + 1) < (unsigned)W) && ((unsigned)(lw + 1) < (unsigned)kTW)};
        const bool okd[2] = {((unsigned)d0 < (unsigned)D) && ((unsigned)ld < (unsigned)kTD),
                             ((unsigned)(d0 + 1) < (unsigned)D) && ((unsigned)(ld + 1) < (unsigned)kTD)};
        double *t0 = tile + ((lh * kTW + lw) * kTD + ld) * RW;
        if constexpr (NCH > 0) {
            if (sub < NCH) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int kd = kk & 1, kw = (kk >> 1) & 1, kh = kk >> 2;
                    if (okd[kd] && okw[kw] && okh[kh] && acc[kk] != 0.0f)
                        __hip_atomic_fetch_add(t0 + ((kh * kTW + kw) * kTD + kd) * RW + sub, (double)acc[kk], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (sdf_lane && okd[mkd] && okw[mkw] && okh[mkh] && acc[0] != 0.0f)
            __hip_atomic_fetch_add(t0 + moff, (double)acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
    };
    for (int i0 = gb; i0 < ge; i0 += U) {
        bool ok[U];
        float v[U];
        float4 ta[U], tb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            ok[u] = i < ge;
            if (ok[u]) {
                const float *r = b.rec + (size_t)i * RECF;
                v[u] = (b.dbg & 16) ? 0.0f : r[sub];
                ta[u] = *(const float4 *)(r + (RECF - 8));
                tb[u] = *(const float4 *)(r + (RECF - 4));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u] || (b.dbg & 1)) continue;
            const int pack = __float_as_int(ta[u].y);
            if (pack != cur || !merge) {
                if (cur >= 0) flush();
                cur = pack;
            }
            const float fh[2] = {1.0f - ta[u].z, ta[u].z}, fw[2] = {1.0f - ta[u].w, ta[u].w}, fd[2] = {1.0f - tb[u].x, tb[u].x};
            if constexpr (NCH > 0) {
                if (sub < NCH) {
                    const float fdfw[2][2] = {{fd[0] * fw[0], fd[0] * fw[1]}, {fd[1] * fw[0], fd[1] * fw[1]}};
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) acc[kk] = fmaf(fdfw[kk & 1][(kk >> 1) & 1] * fh[kk >> 2], v[u], acc[kk]);
                }
            }
            if (sdf_lane) {
                // d L / d sdf corner = ds W_k + qz dW_k/dd + qx dW_k/dw + qy dW_k/dh
                const float fds = fd[mkd], fws = fw[mkw], fhs = fh[mkh];
                const float Wk = (fds * fws) * fhs;
                const float dWd = (mkd ? 1.0f : -1.0f) * (fws * fhs);
                const float dWw = (mkw ? 1.0f : -1.0f) * (fds * fhs);
                const float dWh = (mkh ? 1.0f : -1.0f) * (fds * fws);
                acc[0] += fmaf(Wk, ta[u].x, fmaf(dWd, tb[u].w, fmaf(dWw, tb[u].y, dWh * tb[u].z)));
            }
        }
    }
    if (cur >= 0) flush();
    __syncthreads();
    if (b.dbg & 2) return;
    // flush.  Feature rows: RECF lanes per voxel row, consecutive groups = consecutive d (contiguous rows in HBM).
    if constexpr (NCH > 0) {
        if (g_feat_vol) {
            for (int r = grp; r < kTileVox; r += NG) {
                const int rd = r % kTD, rw = (r / kTD) % kTW, rh = r / (kTD * kTW);
                const int h = oh + rh, w = ow + rw, d = od + rd;
                if (h < H && w < W && d < D && sub < NCH) {
                    const float val = (float)tile[r * RW + sub];
                    if (val != 0.0f) unsafeAtomicAdd(g_feat_vol + ((size_t)(h * W + w) * D + d) * NF + sub, val);
                }
            }
        }
    }
    if (g_sdf_vol) {   // the sdf column: consecutive lanes = consecutive d
        for (int r = threadIdx.x; r < kTileVox; r += NT) {
            const int rd = r % kTD, rw = (r / kTD) % kTW, rh = r / (kTD * kTW);
            const int h = oh + rh, w = ow + rw, d = od + rd;
            if (h < H && w < W && d < D) {
                const float val = (float)tile[r * RW + NCH];
                if (val != 0.0f) unsafeAtomicAdd(g_sdf_vol + (size_t)(h * W + w) * D + d, val);
            }
        }
    }
}

inline size_t rb_align(size_t x) { return (x + 255) & ~(size_t)255; }

inline size_t rb_bricks(const so_render_args &a, RbBin *b) {
    const int H = a.map.h.tot_len, W = a.map.w.tot_len, D = a.map.d.tot_len;
    const int nbh = (H + kBH - 1) / kBH, nbw = (W + kBW - 1) / kBW, nbd = (D + kBD - 1) / kBD;
    if (b) { b->nbh = nbh; b->nbw = nbw; b->nbd = nbd; }
    return (size_t)nbh * nbw * nbd;
}

// workspace carve-up; returns the total size (base may be NULL to size only)
template <int NF>
size_t rb_layout(const so_render_args &a, int chunk, char *base, RbBin *out, int *max_items) {
    RbBin b;
    const size_t nb = rb_bricks(a, &b);
    b.chunk = chunk;
    b.dbg = 0;
    const size_t total = (size_t)a.n_rays * a.n_samples;
    const size_t mi = (total + chunk - 1) / chunk + nb;
    size_t off = 0;
    b.counts = (int *)(base + off); off += nb * kShards * 4;
    b.invs_part = (float *)(base + off); off += kInvsSlots * 4;
    b.n_items = (int *)(base + off); off += 4;       // [0, here) is cleared per call (rb_zeroed_bytes)
    off = rb_align(off);
    b.cursor = (int *)(base + off); off = rb_align(off + nb * kShards * 4);
    b.blk_tot = (int2 *)(base + off); off = rb_align(off + ((nb + 1023) / 1024) * 8);
    b.items = (int4 *)(base + off); off = rb_align(off + mi * 16);
    b.rec = (float *)(base + off); off = rb_align(off + total * RbRec<NF>::RECF * 4);
    if (out) *out = b;
    if (max_items) *max_items = (int)mi;
    return off;
}

inline size_t rb_zeroed_bytes(const so_render_args &a) { return rb_bricks(a, nullptr) * kShards * 4 + kInvsSlots * 4 + 4; }

inline int rb_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
inline int rb_chunk() { static const int c = max(256, rb_env_int("SELFOCC_RB_CHUNK", 4096)); return c; }

inline bool rb_in_range(const so_render_args &a) {
    return a.map.h.tot_len <= 1021 && a.map.w.tot_len <= 1021 && a.map.d.tot_len <= 1021 &&
           (size_t)a.n_rays * a.n_samples < ((size_t)1 << 31);
}

template <int NF, bool BF16, bool BIN>
int launch_ray(const so_render_bwd_args &ba, const RbBin &bin, hipStream_t st) {
    const int S = ba.fwd.n_samples;
    const int m = (S + 63) / 64;
    if (m >= 3) {   // four waves per ray: M = ceil(m / 4) steps per wave
        const int blocks = ba.fwd.n_rays;
#define SO_L(MM) hipLaunchKernelGGL((render_bwd_kernel<NF, BF16, MM, 4, BIN>), dim3(blocks), dim3(256), 0, st, ba, bin)
        if (m <= 4) SO_L(1);
        else SO_L(2);
#undef SO_L
    } else {
        const int blocks = (ba.fwd.n_rays + 3) / 4;
#define SO_L(MM) hipLaunchKernelGGL((render_bwd_kernel<NF, BF16, MM, 1, BIN>), dim3(blocks), dim3(256), 0, st, ba, bin)
        if (m <= 1) SO_L(1);
        else SO_L(2);
#undef SO_L
    }
    return so_launch_status();
}

template <int NF, bool BF16>
int launch_m(const so_render_bwd_args &ba, hipStream_t st) {
    const so_render_args &a = ba.fwd;
    const bool binned = ba.scatter_ws != nullptr && rb_in_range(a) && (ba.g_sdf_vol || ba.g_feat_vol);
    if (!binned) return launch_ray<NF, BF16, false>(ba, RbBin{}, st);
    RbBin bin;
    int max_items = 0;
    const size_t need = rb_layout<NF>(a, rb_chunk(), (char *)ba.scatter_ws, &bin, &max_items);
    bin.dbg = rb_env_int("SELFOCC_RB_DBG", 0);
    SO_REQUIRE(ba.scatter_ws_bytes >= need, "render_bwd: scatter_ws holds %llu bytes, selfocc_render_bwd_ws_bytes() asks for %llu",
               (unsigned long long)ba.scatter_ws_bytes, (unsigned long long)need);
    SO_REQUIRE(((uintptr_t)ba.scatter_ws & 255) == 0, "render_bwd: scatter_ws must be 256-byte aligned");
    hipError_t e = hipMemsetAsync(ba.scatter_ws, 0, rb_zeroed_bytes(a), st);
    SO_REQUIRE(e == hipSuccess, "render_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e));
    const unsigned sblocks = (unsigned)((rb_bricks(a, nullptr) + 1023) / 1024);
    hipLaunchKernelGGL(rb_count_kernel, dim3((unsigned)((a.n_rays + kCountWaves - 1) / kCountWaves)), dim3(kCountWaves * 64), 0, st, a, bin);
    hipLaunchKernelGGL(rb_scan1_kernel, dim3(sblocks), dim3(1024), 0, st, bin);
    hipLaunchKernelGGL(rb_scan2_kernel, dim3(sblocks), dim3(1024), 0, st, bin);
    if (int rc = launch_ray<NF, BF16, true>(ba, bin, st)) return rc;
    const int H = a.map.h.tot_len, W = a.map.w.tot_len, D = a.map.d.tot_len;
    const size_t lds = (size_t)kTileVox * RbRec<NF>::RW * 8;
    // 512 threads per item (SELFOCC_RB_THREADS chose among 256 / 512 / 1024 while the kernel was tuned: 512 won at every shape)
    static const hipError_t attr = hipFuncSetAttribute((const void *)rb_brick_kernel<NF, 512>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)attr;
    hipLaunchKernelGGL((rb_brick_kernel<NF, 512>), dim3(max_items), dim3(512), lds, st, bin, ba.g_sdf_vol, ba.g_feat_vol,
                       ba.g_inv_s, H, W, D);
    return so_launch_status();
}

}  // namespace

int so_validate_render(const so_render_args &a);

extern "C" size_t selfocc_render_bwd_ws_bytes(const so_render_bwd_args *args) {
    if (!args) return 0;
    const so_render_args &a = args->fwd;
    if (!rb_in_range(a) || a.n_rays <= 0 || a.n_samples <= 0) return 0;
    switch (a.n_rgb + a.n_sem == 3 ? 4 : a.n_rgb + a.n_sem) {
        case 0: return rb_layout<0>(a, rb_chunk(), nullptr, nullptr, nullptr);
        case 4: return rb_layout<4>(a, rb_chunk(), nullptr, nullptr, nullptr);
        case 8: return rb_layout<8>(a, rb_chunk(), nullptr, nullptr, nullptr);
        case 24: return rb_layout<24>(a, rb_chunk(), nullptr, nullptr, nullptr);
        default: return 0;
    }
}

extern "C" int selfocc_render_bwd(const so_render_bwd_args *args, void *stream) {
    SO_REQUIRE(args != nullptr, "args is NULL");
    const so_render_bwd_args &ba = *args;
    const so_render_args &a = ba.fwd;
    if (so_validate_render(a)) return -1;
    SO_REQUIRE(a.n_samples <= 64 * kMaxM, "render_bwd: n_samples must be <= %d", 64 * kMaxM);
    if (a.n_rays == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int nf = a.n_rgb + a.n_sem;
    const bool bf = a.feat_dtype == SO_DTYPE_BF16;
    if (nf == 0) return launch_m<0, false>(ba, st);
    if (nf == 3) {
        SO_REQUIRE(a.feat_stride == 4, "n_rgb=3, n_sem=0 requires feat_stride == 4");
        return bf ? launch_m<4, true>(ba, st) : launch_m<4, false>(ba, st);
    }
    SO_REQUIRE(a.feat_stride == nf, "semantic volumes require feat_stride == n_rgb + n_sem");
    switch (nf) {
        case 8:
            SO_REQUIRE(!bf, "render_bwd: bfloat16 feature volumes: n_rgb + n_sem must be 3 or 24 (got 8)");
            return launch_m<8, false>(ba, st);
        case 24: return bf ? launch_m<24, true>(ba, st) : launch_m<24, false>(ba, st);
        default: break;
    }
    SO_REQUIRE(false, "unsupported n_rgb + n_sem = %d (built: 3, 8, 24)", nf);
    return -1;
}
