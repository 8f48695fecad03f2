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

// WPR = waves per ray.  WPR == 1: a wave owns a ray and M * 64 samples (4 rays per block).  WPR == 4: the
// block's four waves share one ray, wave w taking step (j * 4 + w) — at the shipped 256 samples per ray
// that is M = 1, which cuts the per-sample register state 4x (the M = 4 form needed > 256 VGPRs: one wave
// per SIMD, latency-bound); scan carries and per-ray sums cross the waves through a few floats of LDS.
template <int NF, bool BF16, int M, int WPR>
__global__ __launch_bounds__(256) void render_bwd_kernel(so_render_bwd_args ba) {
    static_assert(WPR == 1 || WPR == 4, "waves per ray");
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
    __shared__ __attribute__((aligned(16))) float lds_rec[4 * 64 * (NF > 0 ? (NF + 9 + (((NF + 9) & 1) ? 0 : 1)) : 9)];
    if (ray >= a.n_rays) return;  // wave-uniform (block-uniform when the waves share a ray)
    const int H = a.map.h.tot_len, W = a.map.w.tot_len, D = a.map.d.tot_len;
    const int S = a.n_samples;
    const RayGeomB g = load_ray(a, ray);

    float tn, tf;
    {
        const float fx = 1.0f / (g.dx + 1e-6f), fy = 1.0f / (g.dy + 1e-6f), fz = 1.0f / (g.dz + 1e-6f);
        const float t1 = (a.aabb[0] - g.ox) * fx, t2 = (a.aabb[3] - g.ox) * fx;
        const float t3 = (a.aabb[1] - g.oy) * fy, t4 = (a.aabb[4] - g.oy) * fy;
        const float t5 = (a.aabb[2] - g.oz) * fz, t6 = (a.aabb[5] - g.oz) * fz;
        tn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
        tn = fmaxf(tn, a.near_plane);
        tf = fmaxf(tf, tn + 1e-6f);
    }

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
        float px, py, pz;
        if (a.sample_pos == SO_SAMPLE_AT_START) {
            px = g.ox + g.dx * t0; py = g.oy + g.dy * t0; pz = g.oz + g.dz * t0;
        } else {
            const float tt = t0 + t1;
            px = g.ox + (g.dx * tt) / 2.0f; py = g.oy + (g.dy * tt) / 2.0f; pz = g.oz + (g.dz * tt) / 2.0f;
        }
        cell[j] = so_locate(a.map, px, py, pz);
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
        // pass 1: interpolated colour per sample (kept), forward rgb for the clamp mask
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float f3[3] = {0.0f, 0.0f, 0.0f};
            const float fd[2] = {cell[j].fd0, cell[j].fd1}, fw[2] = {cell[j].fw0, cell[j].fw1}, fh[2] = {cell[j].fh0, cell[j].fh1};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int h = cell[j].h0 + (kk >> 2), ww = cell[j].w0 + ((kk >> 1) & 1), d = cell[j].d0 + (kk & 1);
                const bool in = (h >= 0) && (h < H) && (ww >= 0) && (ww < W) && (d >= 0) && (d < D);
                const int hc = min(max(h, 0), H - 1), wc = min(max(ww, 0), W - 1), dc = min(max(d, 0), D - 1);
                const float wgt = in ? (fd[kk & 1] * fw[(kk >> 1) & 1]) * fh[kk >> 2] : 0.0f;
                const size_t vox = ((size_t)hc * W + wc) * D + dc;
                float c3[3];
                if constexpr (!BF16) {
                    const float *p = (const float *)a.feat_vol + vox * NF;
                    c3[0] = p[0]; c3[1] = p[1]; c3[2] = p[2];
                } else {
                    const uint16_t *p = (const uint16_t *)a.feat_vol + vox * NF;
                    c3[0] = so_bf16_to_f32(p[0]); c3[1] = so_bf16_to_f32(p[1]); c3[2] = so_bf16_to_f32(p[2]);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) f3[k] = fmaf(c3[k], wgt, f3[k]);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                col[j][k] = 0.28209479177387814f * f3[k] + 0.5f;  // pre-relu
                rgb_l[k] = fmaf(w[j], fmaxf(col[j][k], 0.0f), rgb_l[k]);
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
                    float lg[NSEM];
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) lg[k] = 0.0f;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const int h = cell[j].h0 + (kk >> 2), ww = cell[j].w0 + ((kk >> 1) & 1), d = cell[j].d0 + (kk & 1);
                        const bool in = (h >= 0) && (h < H) && (ww >= 0) && (ww < W) && (d >= 0) && (d < D);
                        const int hc = min(max(h, 0), H - 1), wc = min(max(ww, 0), W - 1), dc = min(max(d, 0), D - 1);
                        const float wgt = in ? (fd[kk & 1] * fw[(kk >> 1) & 1]) * fh[kk >> 2] : 0.0f;
                        float f[NF];
                        load_feat<NF, BF16>(a.feat_vol, ((size_t)hc * W + wc) * D + dc, f);
#pragma unroll
                        for (int k = 0; k < NSEM; ++k) lg[k] = fmaf(f[3 + k], wgt, lg[k]);
                    }
                    float mx = lg[0];
#pragma unroll
                    for (int k = 1; k < NSEM; ++k) mx = fmaxf(mx, lg[k]);
                    float den = 0.0f, pk[NSEM];
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) { pk[k] = so_expf(lg[k] - mx); den += pk[k]; }
                    const float iden = 1.0f / den;
                    float gp = 0.0f;
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) { pk[k] *= iden; gp = fmaf(g_semr[k], pk[k], gp); }
                    Gw[j] += gp;
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) df[3 + k] = w[j] * pk[k] * (g_semr[k] - gp);  // softmax backward
                }
            }
            if (ba.g_feat_vol) {   // wave-uniform
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
            if (ba.g_sdf_vol) {
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
        if (ba.g_sdf_vol) {   // wave-uniform
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
        if (lane == 0) unsafeAtomicAdd(ba.g_inv_s, t);
    }
}

template <int NF, bool BF16>
int launch_m(const so_render_bwd_args &ba, hipStream_t st) {
    const int S = ba.fwd.n_samples;
    const int m = (S + 63) / 64;
    if (m >= 3) {   // four waves per ray: M = ceil(m / 4) steps per wave
        const int blocks = ba.fwd.n_rays;
#define SO_L(MM) hipLaunchKernelGGL((render_bwd_kernel<NF, BF16, MM, 4>), dim3(blocks), dim3(256), 0, st, ba)
        if (m <= 4) SO_L(1);
        else SO_L(2);
#undef SO_L
    } else {
        const int blocks = (ba.fwd.n_rays + 3) / 4;
#define SO_L(MM) hipLaunchKernelGGL((render_bwd_kernel<NF, BF16, MM, 1>), dim3(blocks), dim3(256), 0, st, ba)
        if (m <= 1) SO_L(1);
        else SO_L(2);
#undef SO_L
    }
    return so_launch_status();
}

}  // namespace

int so_validate_render(const so_render_args &a);

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
        case 8: return bf ? launch_m<8, true>(ba, st) : launch_m<8, false>(ba, st);
        case 20: return bf ? launch_m<20, true>(ba, st) : launch_m<20, false>(ba, st);
        case 24: return bf ? launch_m<24, true>(ba, st) : launch_m<24, false>(ba, st);
        default: break;
    }
    SO_REQUIRE(false, "unsupported n_rgb + n_sem = %d (built: 3, 8, 20, 24)", nf);
    return -1;
}
