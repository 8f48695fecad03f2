// render_train.hip — sample-parallel SDF ray march for the TRAINING API of selfocc_render_fwd
// (per-sample outputs weights / ts / deltas / sdf / grad requested: NeuSHead.forward,
// model/head/neus_head/neus_head.py:531-577, 640).
//
// The per-ray march of render_fwd.hip gives one ray to a lane.  That is the right shape for
// evaluation (2.16 M rays), and the wrong one for training: the shipped configs train on 48 x 100
// rays x 6 cameras = 28 800 rays x 256 samples — 450 wavefronts on a chip with 8 192 wave slots, each
// walking 256 dependent steps and writing its per-sample tensors with a stride of S floats between
// lanes (2.35 ms per iteration in the round-1 profile, almost all of it latency).
//
// Here a LANE OWNS A SAMPLE: the 64 lanes of a wave are 64 consecutive samples of one ray, the
// 1 / 2 / 4 waves of a ray group cover up to 256 samples per pass.
//   * everything per sample (bin edges incl. jitter, position, meter -> grid, trilinear value +
//     gradient, NeuS alpha, colour / semantic lookup) is the CANONICAL arithmetic of so_device.h,
//     independent of the other samples: sdf / grad / ts / deltas are bit-identical to the oracle;
//   * the transmittance T_i = prod_{j<i} (1 - alpha_j + 1e-7) is an exclusive prefix product:
//     a 6-step wave scan, the wave totals of a ray combined through LDS, a running carry across
//     passes when S > 64 * waves-per-ray.  (A scan multiplies in a different order than the
//     reference's cumprod: weights agree to ~1e-7 relative, not bit for bit.)
//   * per-ray outputs (depth, acc, rgb, sem, max-depth) are wave reductions of the per-sample terms;
//   * every per-sample store is a coalesced 256-byte row segment.
#include "so_device.h"

namespace {

struct TrainRay {
    float ox, oy, oz, dx, dy, dz, dn;
};

// RaySampler lattice + Img2LiDAR (ray_sampler.py:23-31, 58-68; img2lidar.py:58-69; neus_head.py:326)
SO_DEVFN TrainRay so_train_ray(const so_render_args &a, int ray) {
    TrainRay g;
    if (a.ray_mode == SO_RAYS_PIXEL_GRID) {
        const int per_cam = a.nx * a.ny;
        const int cam = ray / per_cam, rem = ray - cam * per_cam;
        const int iy = rem / a.nx, ix = rem - iy * a.nx;
        const float *M = a.img2lidar + cam * 16;
        const float u = (float)ix * a.sx + a.ox;
        const float v = (float)iy * a.sy + a.oy;
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

// AABBBoxCollider, canonical order (identical to render_fwd.hip so_collide)
SO_DEVFN void so_train_collide(const so_render_args &a, const TrainRay &g, float &tnear, float &tfar) {
    const float fx = 1.0f / (g.dx + 1e-6f), fy = 1.0f / (g.dy + 1e-6f), fz = 1.0f / (g.dz + 1e-6f);
    const float t1 = (a.aabb[0] - g.ox) * fx, t2 = (a.aabb[3] - g.ox) * fx;
    const float t3 = (a.aabb[1] - g.oy) * fy, t4 = (a.aabb[4] - g.oy) * fy;
    const float t5 = (a.aabb[2] - g.oz) * fz, t6 = (a.aabb[5] - g.oz) * fz;
    tnear = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    tfar = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    tnear = fmaxf(tnear, a.near_plane);
    tfar = fmaxf(tfar, tnear + 1e-6f);
}

SO_DEVFN float so_train_bin(int j, int n) {   // torch.linspace(0, 1, n + 1)[j], float32
    const float step = 1.0f / (float)n;
    return (j < (n + 1) / 2) ? step * (float)j : fmaf(-step, (float)(n - j), 1.0f);
}

SO_DEVFN float so_train_edge(const so_render_args &a, int ray, int j, float tnear, float tfar) {
    const int n = a.n_samples;
    float b = so_train_bin(j, n);
    if (a.jitter_mode != SO_JITTER_NONE) {
        const float lo = (j == 0) ? b : (b + so_train_bin(j - 1, n)) / 2.0f;
        const float hi = (j == n) ? b : (so_train_bin(j + 1, n) + b) / 2.0f;
        const float tr = (a.jitter_mode == SO_JITTER_SINGLE) ? a.t_rand[ray] : a.t_rand[(size_t)ray * (n + 1) + j];
        b = lo + (hi - lo) * tr;
    }
    return b * tfar + (1.0f - b) * tnear;
}

template <int NF, bool BF16>
SO_DEVFN void so_train_feat(const void *__restrict__ vol, int H, int W, int D, const so_cell &c, const float wk[8],
                            float f[NF > 0 ? NF : 1]) {
#pragma unroll
    for (int k = 0; k < NF; ++k) f[k] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int h = c.h0 + (kk >> 2), w = c.w0 + ((kk >> 1) & 1), d = c.d0 + (kk & 1);
        const bool in = (h >= 0) && (h < H) && (w >= 0) && (w < W) && (d >= 0) && (d < D);
        const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1), dc = min(max(d, 0), D - 1);
        const size_t vox = ((size_t)hc * W + wc) * D + dc;
        const float wgt = in ? wk[kk] : 0.0f;
        if constexpr (!BF16) {
            const float4 *p = (const float4 *)((const float *)vol + vox * NF);
#pragma unroll
            for (int q = 0; q < NF / 4; ++q) {
                const float4 t = p[q];
                f[4 * q + 0] = fmaf(t.x, wgt, f[4 * q + 0]);
                f[4 * q + 1] = fmaf(t.y, wgt, f[4 * q + 1]);
                f[4 * q + 2] = fmaf(t.z, wgt, f[4 * q + 2]);
                f[4 * q + 3] = fmaf(t.w, wgt, f[4 * q + 3]);
            }
        } else {
            const uint2 *p = (const uint2 *)((const uint16_t *)vol + vox * NF);
#pragma unroll
            for (int q = 0; q < NF / 4; ++q) {
                const uint2 t = p[q];
                f[4 * q + 0] = fmaf(__uint_as_float(t.x << 16), wgt, f[4 * q + 0]);
                f[4 * q + 1] = fmaf(__uint_as_float(t.x & 0xffff0000u), wgt, f[4 * q + 1]);
                f[4 * q + 2] = fmaf(__uint_as_float(t.y << 16), wgt, f[4 * q + 2]);
                f[4 * q + 3] = fmaf(__uint_as_float(t.y & 0xffff0000u), wgt, f[4 * q + 3]);
            }
        }
    }
}

SO_DEVFN float so_wave_sum(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// WPR = waves per ray (1, 2 or 4); a 256-thread block serves 4 / WPR rays.
template <int NF, bool BF16, int WPR>
__global__ __launch_bounds__(256) void render_fwd_samples_kernel(so_render_args a) {
    constexpr int NSEM = NF > 4 ? NF - 3 : 0;
    constexpr int RPB = 4 / WPR;                  // rays per block
    constexpr int NACC = 5 + NSEM;                // acc, dsum, rgb[3], sem[NSEM] partial sums per wave
    __shared__ float s_tot[2][4];                 // wave totals of the step factors, double-buffered over passes
    __shared__ float s_part[4][NACC + 3];         // per-wave partial sums + (best q, best t, best index)
    const int H = a.map.h.tot_len, W = a.map.w.tot_len, D = a.map.d.tot_len;
    const int S = a.n_samples;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rslot = wave / WPR, wr = wave - rslot * WPR;      // ray slot in the block, wave within the ray
    const int ray_raw = blockIdx.x * RPB + rslot;
    const bool live = ray_raw < a.n_rays;                       // wave-uniform; dead waves still join the barriers
    const int ray = live ? ray_raw : 0;
    const TrainRay g = so_train_ray(a, ray);
    float tnear, tfar;
    so_train_collide(a, g, tnear, tfar);
    const float eps32 = 1.1920928955078125e-07f;

    float carry = 1.0f;                                          // transmittance entering this pass (ray-uniform)
    float acc = 0.0f, dsum = 0.0f, rgb[3] = {0.0f, 0.0f, 0.0f};
    float sem[NSEM > 0 ? NSEM : 1];
#pragma unroll
    for (int k = 0; k < NSEM; ++k) sem[k] = 0.0f;
    float best_q = -INFINITY, best_t = 0.0f;
    int best_i = 0x7fffffff;

    const int per_pass = 64 * WPR;
    for (int base = 0, pass = 0; base < S; base += per_pass, ++pass) {
        const int i = base + wr * 64 + lane;
        const bool valid = live && i < S;
        float alpha = 0.0f, fstep = 1.0f, t_mid = 0.0f, tz = 0.0f, dz_ = 0.0f, sdf = 0.0f, gx = 0.0f, gy = 0.0f, gz = 0.0f;
        float f[NF > 0 ? NF : 1];
#pragma unroll
        for (int k = 0; k < NF; ++k) f[k] = 0.0f;
        if (valid) {
            const float t_start = so_train_edge(a, ray, i, tnear, tfar);
            const float t_end = so_train_edge(a, ray, i + 1, tnear, tfar);
            const float delta = t_end - t_start;
            t_mid = (t_start + t_end) / 2.0f;
            float px, py, pz;
            if (a.sample_pos == SO_SAMPLE_AT_START) {
                px = g.ox + g.dx * t_start; py = g.oy + g.dy * t_start; pz = g.oz + g.dz * t_start;
            } else {
                const float tt = t_start + t_end;
                px = g.ox + (g.dx * tt) / 2.0f; py = g.oy + (g.dy * tt) / 2.0f; pz = g.oz + (g.dz * tt) / 2.0f;
            }
            const so_cell c = so_locate(a.map, px, py, pz);
            float v[8], wk[8];
            so_gather_sdf(a.sdf_vol, H, W, D, c, v);
            sdf = so_trilerp_sdf(c, v, wk);
            so_trilerp_grad(c, v, gx, gy, gz);
            // NeuS alpha (sdfstudio NeuS get_alpha, cos anneal ratio 1), canonical order
            const float cosv = (g.dx * gx + g.dy * gy) + g.dz * gz;
            const float icos = fminf(cosv, 0.0f);
            const float half = (icos * delta) * 0.5f;
            const float prev_cdf = so_sigmoid((sdf - half) * so_inv_s(a));
            const float next_cdf = so_sigmoid((sdf + half) * so_inv_s(a));
            alpha = ((prev_cdf - next_cdf) + 1e-5f) / (prev_cdf + 1e-5f);
            alpha = fminf(fmaxf(alpha, 0.0f), 1.0f);
            fstep = (1.0f - alpha) + 1e-7f;
            tz = t_mid / g.dn;
            dz_ = delta / g.dn;
            if constexpr (NF > 0) so_train_feat<NF, BF16>(a.feat_vol, H, W, D, c, wk, f);
        }
        // exclusive prefix product of fstep over the samples of this pass
        float incl = fstep;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const float up = __shfl_up(incl, m, 64);
            if (lane >= m) incl = incl * up;
        }
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        float before = 1.0f, all = 1.0f;      // product of the earlier waves of this ray / of all its waves
        if constexpr (WPR > 1) {
            if (lane == 63) s_tot[pass & 1][wave] = incl;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < WPR; ++k) {
                const float t = s_tot[pass & 1][rslot * WPR + k];
                if (k < wr) before = before * t;
                all = all * t;
            }
        } else {
            all = __shfl(incl, 63, 64);
        }
        const float T = (carry * before) * excl;
        carry = carry * all;
        const float w = alpha * T;

        if (valid) {
            const size_t o = (size_t)ray * S + i;
            if (a.weights) a.weights[o] = w;
            if (a.ts) a.ts[o] = tz;
            if (a.deltas) a.deltas[o] = dz_;
            if (a.sdf) a.sdf[o] = sdf;
            if (a.grad) { a.grad[3 * o] = gx; a.grad[3 * o + 1] = gy; a.grad[3 * o + 2] = gz; }
            acc += w;
            dsum += w * t_mid;
            const float wq = (dz_ < eps32) ? 0.0f : w;                // neus_head.py:430-438
            const float q = wq / fmaxf(dz_, eps32);
            if (q > best_q) { best_q = q; best_t = tz; best_i = i; }
            if constexpr (NF > 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float col = fmaxf(0.28209479177387814f * f[k] + 0.5f, 0.0f);   // sh_render.py:84-91
                    rgb[k] = fmaf(w, col, rgb[k]);
                }
                if constexpr (NSEM > 0) {
                    float m = f[3];
#pragma unroll
                    for (int k = 1; k < NSEM; ++k) m = fmaxf(m, f[3 + k]);
                    float e[NSEM], den = 0.0f;
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) { e[k] = so_expf(f[3 + k] - m); den = den + e[k]; }
                    const float wd = w / den;
#pragma unroll
                    for (int k = 0; k < NSEM; ++k) sem[k] = fmaf(wd, e[k], sem[k]);
                }
            }
        }
    }

    // ---- per-ray outputs: wave reductions, then the waves of a ray through LDS ----------------------
    acc = so_wave_sum(acc);
    dsum = so_wave_sum(dsum);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = so_wave_sum(rgb[k]);
#pragma unroll
    for (int k = 0; k < NSEM; ++k) sem[k] = so_wave_sum(sem[k]);
    // arg-max of w / delta: the FIRST maximal sample, like torch.argmax
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float oq = __shfl_xor(best_q, m, 64), ot = __shfl_xor(best_t, m, 64);
        const int oi = __shfl_xor(best_i, m, 64);
        if (oq > best_q || (oq == best_q && oi < best_i)) { best_q = oq; best_t = ot; best_i = oi; }
    }
    if constexpr (WPR > 1) {
        __syncthreads();   // s_part is independent of s_tot, but keep the passes' barriers paired
        if (lane == 0) {
            s_part[wave][0] = acc; s_part[wave][1] = dsum;
#pragma unroll
            for (int k = 0; k < 3; ++k) s_part[wave][2 + k] = rgb[k];
#pragma unroll
            for (int k = 0; k < NSEM; ++k) s_part[wave][5 + k] = sem[k];
            s_part[wave][NACC] = best_q; s_part[wave][NACC + 1] = best_t; s_part[wave][NACC + 2] = __int_as_float(best_i);
        }
        __syncthreads();
        if (wr != 0) return;
        acc = 0.0f; dsum = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < NSEM; ++k) sem[k] = 0.0f;
        best_q = -INFINITY; best_t = 0.0f; best_i = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < WPR; ++k) {       // in sample order
            const float *p = s_part[rslot * WPR + k];
            acc += p[0]; dsum += p[1];
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] += p[2 + c];
#pragma unroll
            for (int c = 0; c < NSEM; ++c) sem[c] += p[5 + c];
            const float oq = p[NACC], ot = p[NACC + 1];
            const int oi = __float_as_int(p[NACC + 2]);
            if (oq > best_q || (oq == best_q && oi < best_i)) { best_q = oq; best_t = ot; best_i = oi; }
        }
    }
    if (!live || lane != 0) return;
    float depth = dsum / (acc + 1e-10f);
    if (a.flags & SO_FLAG_DEPTH_DIV_NORM) depth = depth / g.dn;
    if (a.depth) a.depth[ray] = depth;
    if (a.acc) a.acc[ray] = acc;
    if (a.max_depth) a.max_depth[ray] = best_t;
    if (a.nears) a.nears[ray] = tnear;
    if (a.fars) a.fars[ray] = tfar;
    if constexpr (NF > 0) {
        if (a.rgb) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float bg = 0.0f;
                if (a.bkgd_mode == SO_BKGD_CONST) bg = a.bkgd[k];
                else if (a.bkgd_mode == SO_BKGD_PER_RAY) bg = a.bkgd_rays[3 * (size_t)ray + k];
                float r = rgb[k];
                if (a.bkgd_mode != SO_BKGD_NONE) r = r + bg * (1.0f - acc);
                if (a.flags & SO_FLAG_CLAMP_RGB) r = fminf(fmaxf(r, 0.0f), 1.0f);
                a.rgb[3 * (size_t)ray + k] = r;
            }
        }
        if constexpr (NSEM > 0) {
            if (a.sem) {
#pragma unroll
                for (int k = 0; k < NSEM; ++k) a.sem[(size_t)ray * NSEM + k] = sem[k];
            }
        }
    }
}

template <int NF, bool BF16, int WPR>
int launch_samples_w(const so_render_args &a, hipStream_t st) {
    constexpr int RPB = 4 / WPR;
    hipLaunchKernelGGL((render_fwd_samples_kernel<NF, BF16, WPR>), dim3((a.n_rays + RPB - 1) / RPB), dim3(256), 0, st, a);
    return so_launch_status();
}

}  // namespace

// called by selfocc_render_fwd (render_fwd.hip) for launches that request per-sample outputs
template <int NF, bool BF16>
int so_render_fwd_samples(const so_render_args &a, hipStream_t st) {
    if (a.n_samples <= 64) return launch_samples_w<NF, BF16, 1>(a, st);
    if (a.n_samples <= 128) return launch_samples_w<NF, BF16, 2>(a, st);
    return launch_samples_w<NF, BF16, 4>(a, st);
}

template int so_render_fwd_samples<0, false>(const so_render_args &, hipStream_t);
template int so_render_fwd_samples<4, false>(const so_render_args &, hipStream_t);
template int so_render_fwd_samples<4, true>(const so_render_args &, hipStream_t);
template int so_render_fwd_samples<8, false>(const so_render_args &, hipStream_t);
template int so_render_fwd_samples<24, false>(const so_render_args &, hipStream_t);
template int so_render_fwd_samples<24, true>(const so_render_args &, hipStream_t);
