/*
 * oracle_render.c — CPU restatement of SelfOcc's SDF volume-rendering path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in selfocc_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker.  It shares no code with selfocc_amd/csrc: only the public struct definitions
 * of include/selfocc_hip.h (the interface under test) are included.
 *
 * PARITY STATUS: "parity unpinned" for the NeuS internals.  The arithmetic of
 * NeuSCustomModel / SDFCustomField lives in the un-pinned huang-yh/sdfstudio fork, which
 * is absent from /root/reference (docs/installation.md:28-39).  This file restates the
 * published upstream algorithm (autonomousvision/sdfstudio NeuS: box collider, uniform
 * spaced sampler, get_alpha, cumprod transmittance, expected-depth renderer) anchored on
 * the reference's own call sites and in-repo field code:
 *   - volume lookup: meter2grid(normalize) -> 2g-1 -> F.grid_sample(bilinear,
 *     align_corners=True) on a (1,C,H,W,D) volume with grid[..., [2,1,0]]
 *     (model/head/nerfacc_head/bev_nerf.py:103-113, neus_head.py:612-619);
 *   - grid<->metre: model/encoder/bevformer/mappings.py:97-150 (pinned against the
 *     imported reference module in tests/golden);
 *   - colour: SH degree 0, relu(C0 * raw + 0.5) (model/head/utils/sh_render.py:84-91);
 *   - semantics: per-sample softmax, weight-composited (bev_nerf.py:131-134,
 *     rendering.py:146-148);
 *   - head post-math: ts, deltas, max-depth (model/head/neus_head/neus_head.py:366-374,
 *     430-438, 571-587);
 *   - rays: RaySampler lattice + Img2LiDAR (nerfacc_head/ray_sampler.py:23-68,
 *     img2lidar.py:58-69).
 * Declared omission: upstream's expected-depth renderer clips the depth to the smallest /
 * largest sample mid-point of the whole CALL (a chunk-dependent value that only changes rays
 * with sum(w) ~ 0); this restatement, the torch port and the kernels do not (README
 * "Deviations").
 * The trilinear lookup itself IS pinned: tests compare it bit-for-bit with
 * torch.nn.functional.grid_sample (the op the reference calls).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../include/selfocc_hip.h"

/* ---- canonical expf (Cephes expf: Cody-Waite reduction + degree-5 polynomial) -------- */
static float ref_expf(float x) {
    if (x < -87.0f) x = -87.0f;
    if (x > 88.0f) x = 88.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(-n, 0.693359375f, x);
    r = fmaf(-n, -2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float z = r * r;
    float y = fmaf(p, z, r) + 1.0f;
    int e = (int)n;
    int e1 = e >> 1, e2 = e - e1;
    union { uint32_t u; float f; } s1, s2;
    s1.u = (uint32_t)(e1 + 127) << 23;
    s2.u = (uint32_t)(e2 + 127) << 23;
    return (y * s1.f) * s2.f;
}

static float ref_sigmoid(float x) { return 1.0f / (1.0f + ref_expf(-x)); }

/* ---- mappings.py:97-143, one axis ---------------------------------------------------- */
static float axis_meter2grid(const so_axis *A, float m, float *slope) {
    float ctr = m - A->start;
    float mag = fabsf(ctr);
    float gabs;
    if (A->size1 == 0.0f) {
        gabs = mag / A->range0 * A->size0;
        *slope = A->size0 / A->range0;
    } else if (mag > A->range0) {
        gabs = A->size0 + (mag - A->range0) / A->range1 * A->size1;
        *slope = A->size1 / A->range1;
    } else {
        gabs = mag / A->range0 * A->size0;
        *slope = A->size0 / A->range0;
    }
    float sgn = (ctr > 0.0f) ? 1.0f : ((ctr < 0.0f) ? -1.0f : 0.0f);
    float signed_g = sgn * gabs;
    return (signed_g + A->off0) + A->off1;
}

/* normalize (mappings.py:145-148), 2g-1 (bev_nerf.py:106), grid_sample un-normalise with
 * align_corners=True: ((c + 1) / 2) * (size - 1) */
static float sample_coord(float g, int tot_len) {
    float span = (float)(tot_len - 1);
    float gn = g / span;
    float c = 2.0f * gn - 1.0f;
    return ((c + 1.0f) / 2.0f) * span;
}

typedef struct {
    int ih, iw, id;      /* floor */
    float wh[2], ww[2], wd[2];
    float slope_h, slope_w, slope_d;
} cell_t;

static cell_t locate(const so_mapping *M, float x, float y, float z) {
    cell_t c;
    float gh = sample_coord(axis_meter2grid(&M->h, y, &c.slope_h), M->h.tot_len);
    float gw = sample_coord(axis_meter2grid(&M->w, x, &c.slope_w), M->w.tot_len);
    float gd = sample_coord(axis_meter2grid(&M->d, z, &c.slope_d), M->d.tot_len);
    float fh = floorf(gh), fw = floorf(gw), fd = floorf(gd);
    c.ih = (int)fh; c.iw = (int)fw; c.id = (int)fd;
    /* ATen grid_sampler_3d: weight of the low corner = (i_high - i), high = (i - i_low) */
    c.wh[0] = (fh + 1.0f) - gh; c.wh[1] = gh - fh;
    c.ww[0] = (fw + 1.0f) - gw; c.ww[1] = gw - fw;
    c.wd[0] = (fd + 1.0f) - gd; c.wd[1] = gd - fd;
    return c;
}

static int in_bounds(int h, int w, int d, int H, int W, int D) {
    return h >= 0 && h < H && w >= 0 && w < W && d >= 0 && d < D;
}

/* ATen grid_sampler_3d_cpu forward, one channel: corners visited tnw tne tsw tse bnw bne
 * bsw bse (x fastest = our d axis), out += value * weight, weight = (wx * wy) * wz. */
static float trilinear(const float *vol, int H, int W, int D, const cell_t *c, float corner[8],
                       float weight[8]) {
    float out = 0.0f;
    for (int k = 0; k < 8; ++k) {
        int kd = k & 1, kw = (k >> 1) & 1, kh = k >> 2;
        int h = c->ih + kh, w = c->iw + kw, d = c->id + kd;
        weight[k] = (c->wd[kd] * c->ww[kw]) * c->wh[kh];
        corner[k] = in_bounds(h, w, d, H, W, D) ? vol[((size_t)h * W + w) * D + d] : 0.0f;
        out = out + corner[k] * weight[k];
    }
    return out;
}

/* ATen grid_sampler_3d backward wrt the grid, then the chain rule to metres.  The
 * un-normalise / 2g-1 / normalize factors cancel exactly: (size-1)/2 * 2 / (tot_len-1). */
static void trilinear_grad(const cell_t *c, const float corner[8], float g[3]) {
    float gd = 0.0f, gw = 0.0f, gh = 0.0f;
    for (int k = 0; k < 8; ++k) {
        int kd = k & 1, kw = (k >> 1) & 1, kh = k >> 2;
        float td = (corner[k] * c->ww[kw]) * c->wh[kh];
        float tw = (corner[k] * c->wd[kd]) * c->wh[kh];
        float th = (corner[k] * c->wd[kd]) * c->ww[kw];
        gd = kd ? gd + td : gd - td;
        gw = kw ? gw + tw : gw - tw;
        gh = kh ? gh + th : gh - th;
    }
    g[0] = gw * c->slope_w; /* d/dx */
    g[1] = gh * c->slope_h; /* d/dy */
    g[2] = gd * c->slope_d; /* d/dz */
}

static float feat_at(const so_render_args *a, size_t vox, int k) {
    if (a->feat_dtype == SO_DTYPE_BF16) {
        union { uint32_t u; float f; } cv;
        cv.u = ((uint32_t)((const uint16_t *)a->feat_vol)[vox * a->feat_stride + k]) << 16;
        return cv.f;
    }
    return ((const float *)a->feat_vol)[vox * a->feat_stride + k];
}

/* torch.linspace(0, 1, n + 1)[j], float32 (ATen RangeFactories.cpp) */
static float unit_bin(int j, int n) {
    float step = 1.0f / (float)n;
    if (j < (n + 1) / 2) return step * (float)j;
    return fmaf(-step, (float)(n - j), 1.0f); /* ATen's vectorised kernel fuses end - step * k */
}

static float bin_edge(const so_render_args *a, int ray, int j, float tn, float tf) {
    int n = a->n_samples;
    float b = unit_bin(j, n);
    if (a->jitter_mode != SO_JITTER_NONE) {
        /* nerfstudio SpacedSampler, train_stratified */
        float lower = (j == 0) ? b : (b + unit_bin(j - 1, n)) / 2.0f;
        float upper = (j == n) ? b : (unit_bin(j + 1, n) + b) / 2.0f;
        float u = (a->jitter_mode == SO_JITTER_SINGLE) ? a->t_rand[ray]
                                                        : a->t_rand[(size_t)ray * (n + 1) + j];
        b = lower + (upper - lower) * u;
    }
    return b * tf + (1.0f - b) * tn;
}

static void one_ray(const so_render_args *a, int ray) {
    const int H = a->map.h.tot_len, W = a->map.w.tot_len, D = a->map.d.tot_len;
    const int S = a->n_samples;
    float o[3], dir[3], dn;

    if (a->ray_mode == SO_RAYS_PIXEL_GRID) {
        int per_cam = a->nx * a->ny;
        int cam = ray / per_cam, rem = ray % per_cam;
        int iy = rem / a->nx, ix = rem % a->nx;
        const float *M = a->img2lidar + 16 * cam;
        float u = (float)ix * a->sx + a->ox; /* ray_sampler.py:25-31, 66-67 */
        float v = (float)iy * a->sy + a->oy;
        for (int r = 0; r < 3; ++r) { /* img2lidar.py:51, 65-69 */
            o[r] = M[4 * r + 3];
            dir[r] = (M[4 * r + 0] * u + M[4 * r + 1] * v) + M[4 * r + 2];
        }
        dn = sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]); /* neus_head.py:326-327 */
        for (int r = 0; r < 3; ++r) dir[r] = dir[r] / dn;
    } else {
        for (int r = 0; r < 3; ++r) {
            o[r] = a->origins[3 * (size_t)ray + r];
            dir[r] = a->dirs[3 * (size_t)ray + r];
        }
        dn = a->dir_norm ? a->dir_norm[ray] : 1.0f;
    }

    /* AABBBoxCollider */
    float tlo[3], thi[3];
    for (int r = 0; r < 3; ++r) {
        float frac = 1.0f / (dir[r] + 1e-6f);
        float ta = (a->aabb[r] - o[r]) * frac, tb = (a->aabb[3 + r] - o[r]) * frac;
        tlo[r] = fminf(ta, tb);
        thi[r] = fmaxf(ta, tb);
    }
    float tn = fmaxf(fmaxf(tlo[0], tlo[1]), tlo[2]);
    float tf = fminf(fminf(thi[0], thi[1]), thi[2]);
    tn = fmaxf(tn, a->near_plane);
    tf = fmaxf(tf, tn + 1e-6f);

    const int nsem = a->n_sem;
    float trans = 1.0f, acc = 0.0f, dsum = 0.0f, rgb[3] = {0, 0, 0};
    float sem[64];
    for (int k = 0; k < nsem; ++k) sem[k] = 0.0f;
    float best = -INFINITY, best_t = 0.0f;
    const float eps32 = 1.1920928955078125e-07f;

    for (int i = 0; i < S; ++i) {
        float t0 = bin_edge(a, ray, i, tn, tf), t1 = bin_edge(a, ray, i + 1, tn, tf);
        float delta = t1 - t0;
        float tmid = (t0 + t1) / 2.0f;
        float p[3];
        for (int r = 0; r < 3; ++r) {
            if (a->sample_pos == SO_SAMPLE_AT_START) p[r] = o[r] + dir[r] * t0;
            else p[r] = o[r] + (dir[r] * (t0 + t1)) / 2.0f;
        }
        cell_t c = locate(&a->map, p[0], p[1], p[2]);
        float corner[8], weight[8], g[3];
        float sdf = trilinear(a->sdf_vol, H, W, D, &c, corner, weight);
        trilinear_grad(&c, corner, g);

        /* NeuS get_alpha (cos_anneal_ratio = 1) */
        float true_cos = (dir[0] * g[0] + dir[1] * g[1]) + dir[2] * g[2];
        float iter_cos = fminf(true_cos, 0.0f); /* -relu(-cos) */
        float half = (iter_cos * delta) * 0.5f;
        float prev_cdf = ref_sigmoid((sdf - half) * a->inv_s);
        float next_cdf = ref_sigmoid((sdf + half) * a->inv_s);
        float alpha = ((prev_cdf - next_cdf) + 1e-5f) / (prev_cdf + 1e-5f);
        alpha = fminf(fmaxf(alpha, 0.0f), 1.0f);
        float w = alpha * trans;
        trans = trans * ((1.0f - alpha) + 1e-7f);
        acc = acc + w;
        dsum = dsum + w * tmid;

        /* neus_head.py:430-438 */
        float tz = tmid / dn, dz = delta / dn;
        float wq = (dz < eps32) ? 0.0f : w;
        float q = wq / fmaxf(dz, eps32);
        if (q > best) { best = q; best_t = tz; }

        if (a->n_rgb + nsem > 0) {
            float f[64];
            for (int k = 0; k < a->n_rgb + nsem; ++k) f[k] = 0.0f;
            for (int kk = 0; kk < 8; ++kk) {
                int h = c.ih + (kk >> 2), w_ = c.iw + ((kk >> 1) & 1), d = c.id + (kk & 1);
                if (!in_bounds(h, w_, d, H, W, D)) continue;
                size_t vox = ((size_t)h * W + w_) * D + d;
                for (int k = 0; k < a->n_rgb + nsem; ++k) f[k] = fmaf(feat_at(a, vox, k), weight[kk], f[k]);
            }
            for (int k = 0; k < a->n_rgb; ++k) {
                float col = fmaxf(0.28209479177387814f * f[k] + 0.5f, 0.0f);
                rgb[k] = fmaf(w, col, rgb[k]);
            }
            if (nsem > 0) {
                const float *lg = f + a->n_rgb;
                float m = lg[0];
                for (int k = 1; k < nsem; ++k) m = fmaxf(m, lg[k]);
                float e[64], den = 0.0f;
                for (int k = 0; k < nsem; ++k) { e[k] = ref_expf(lg[k] - m); den = den + e[k]; }
                float wd = w / den;
                for (int k = 0; k < nsem; ++k) sem[k] = fmaf(wd, e[k], sem[k]);
            }
        }
        size_t so = (size_t)ray * S + i;
        if (a->weights) a->weights[so] = w;
        if (a->ts) a->ts[so] = tz;
        if (a->deltas) a->deltas[so] = dz;
        if (a->sdf) a->sdf[so] = sdf;
        if (a->grad) { a->grad[3 * so] = g[0]; a->grad[3 * so + 1] = g[1]; a->grad[3 * so + 2] = g[2]; }
    }

    float depth = dsum / (acc + 1e-10f); /* DepthRenderer 'expected' */
    if (a->flags & SO_FLAG_DEPTH_DIV_NORM) depth = depth / dn;
    if (a->depth) a->depth[ray] = depth;
    if (a->acc) a->acc[ray] = acc;
    if (a->max_depth) a->max_depth[ray] = best_t;
    if (a->nears) a->nears[ray] = tn;
    if (a->fars) a->fars[ray] = tf;
    if (a->rgb && a->n_rgb == 3) {
        for (int k = 0; k < 3; ++k) {
            float r = rgb[k];
            if (a->bkgd_mode == SO_BKGD_CONST) r = r + a->bkgd[k] * (1.0f - acc);
            else if (a->bkgd_mode == SO_BKGD_PER_RAY) r = r + a->bkgd_rays[3 * (size_t)ray + k] * (1.0f - acc);
            if (a->flags & SO_FLAG_CLAMP_RGB) r = fminf(fmaxf(r, 0.0f), 1.0f);
            a->rgb[3 * (size_t)ray + k] = r;
        }
    }
    if (a->sem && nsem > 0)
        for (int k = 0; k < nsem; ++k) a->sem[(size_t)ray * nsem + k] = sem[k];
}

/* All pointers are HOST pointers here. */
int oracle_render_fwd(const so_render_args *a) {
    if (a->n_sem > 61 || a->n_rgb + a->n_sem > 64) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int ray = 0; ray < a->n_rays; ++ray) one_ray(a, ray);
    return 0;
}

/* ---- float64 evaluation of the SAME formulas (depth / acc only) ---------------------------------
 * Not a second oracle: the parity bar is the float32 canonical order above (what the reference's
 * float32 torch ops compute).  This is the yardstick for that bar: NeuS's alpha subtracts two
 * sigmoids that differ by ~1e-5 in free space, so the canonical float32 result itself carries
 * rounding noise; tests report |f32 oracle - f64| next to |HIP - f32 oracle| so that a reader can see
 * how much of a difference is arithmetic noise of the reference's own precision.  Single-segment
 * linear axes only (every shipped config). */
static void one_ray_f64(const so_render_args *a, int ray, double *depth_out, double *acc_out) {
    const int H = a->map.h.tot_len, W = a->map.w.tot_len, D = a->map.d.tot_len;
    const int S = a->n_samples;
    double o[3], dir[3], dn;
    if (a->ray_mode == SO_RAYS_PIXEL_GRID) {
        int per_cam = a->nx * a->ny;
        int cam = ray / per_cam, rem = ray % per_cam;
        int iy = rem / a->nx, ix = rem % a->nx;
        const float *M = a->img2lidar + 16 * cam;
        double u = (double)((float)ix * a->sx + a->ox), v = (double)((float)iy * a->sy + a->oy);
        for (int r = 0; r < 3; ++r) {
            o[r] = M[4 * r + 3];
            dir[r] = (double)M[4 * r + 0] * u + (double)M[4 * r + 1] * v + (double)M[4 * r + 2];
        }
        dn = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        for (int r = 0; r < 3; ++r) dir[r] /= dn;
    } else {
        for (int r = 0; r < 3; ++r) { o[r] = a->origins[3 * (size_t)ray + r]; dir[r] = a->dirs[3 * (size_t)ray + r]; }
        dn = a->dir_norm ? a->dir_norm[ray] : 1.0;
    }
    double tn = -INFINITY, tf = INFINITY;
    for (int r = 0; r < 3; ++r) {
        double frac = 1.0 / (dir[r] + 1e-6);
        double ta = (a->aabb[r] - o[r]) * frac, tb = (a->aabb[3 + r] - o[r]) * frac;
        tn = fmax(tn, fmin(ta, tb));
        tf = fmin(tf, fmax(ta, tb));
    }
    tn = fmax(tn, a->near_plane);
    tf = fmax(tf, tn + 1e-6);
    const so_axis *ax[3] = {&a->map.w, &a->map.h, &a->map.d}; /* metre x, y, z */
    double trans = 1.0, acc = 0.0, dsum = 0.0;
    for (int i = 0; i < S; ++i) {
        double b0 = (double)i / S, b1 = (double)(i + 1) / S;
        double t0 = b0 * tf + (1.0 - b0) * tn, t1 = b1 * tf + (1.0 - b1) * tn;
        double tpos = (a->sample_pos == SO_SAMPLE_AT_START) ? t0 : 0.5 * (t0 + t1);
        double g[3], slope[3];
        for (int r = 0; r < 3; ++r) {
            const so_axis *A = ax[r];
            slope[r] = (double)A->size0 / A->range0;
            g[r] = (o[r] + dir[r] * tpos - A->start) * slope[r] + A->off0 + A->off1;
        }
        double fw = floor(g[0]), fh = floor(g[1]), fd = floor(g[2]);
        int iw = (int)fw, ih = (int)fh, id = (int)fd;
        double ww[2] = {fw + 1 - g[0], g[0] - fw}, wh[2] = {fh + 1 - g[1], g[1] - fh}, wd[2] = {fd + 1 - g[2], g[2] - fd};
        double sdf = 0, gd = 0, gw = 0, gh = 0;
        for (int k = 0; k < 8; ++k) {
            int kd = k & 1, kw = (k >> 1) & 1, kh = k >> 2;
            int h = ih + kh, w = iw + kw, d = id + kd;
            double v = in_bounds(h, w, d, H, W, D) ? a->sdf_vol[((size_t)h * W + w) * D + d] : 0.0;
            sdf += v * wd[kd] * ww[kw] * wh[kh];
            gd += (kd ? v : -v) * ww[kw] * wh[kh];
            gw += (kw ? v : -v) * wd[kd] * wh[kh];
            gh += (kh ? v : -v) * wd[kd] * ww[kw];
        }
        double cosv = dir[0] * gw * slope[0] + dir[1] * gh * slope[1] + dir[2] * gd * slope[2];
        double half = fmin(cosv, 0.0) * (t1 - t0) * 0.5;
        double pa = 1.0 / (1.0 + exp(-(sdf - half) * a->inv_s)), pb = 1.0 / (1.0 + exp(-(sdf + half) * a->inv_s));
        double alpha = fmin(fmax((pa - pb + 1e-5) / (pa + 1e-5), 0.0), 1.0);
        double w = alpha * trans;
        trans *= (1.0 - alpha) + 1e-7;
        acc += w;
        dsum += w * 0.5 * (t0 + t1);
    }
    double depth = dsum / (acc + 1e-10);
    if (a->flags & SO_FLAG_DEPTH_DIV_NORM) depth /= dn;
    *depth_out = depth;
    *acc_out = acc;
}

int oracle_render_fwd_f64(const so_render_args *a, double *depth, double *acc) {
    if (a->map.h.size1 != 0.0f || a->map.w.size1 != 0.0f || a->map.d.size1 != 0.0f) return -1;
    if (a->jitter_mode != SO_JITTER_NONE) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int ray = 0; ray < a->n_rays; ++ray) one_ray_f64(a, ray, depth + ray, acc + ray);
    return 0;
}

/* Stand-alone pieces exposed for pinning against the imported reference / torch ops. */
void oracle_meter2grid(const so_mapping *M, const float *xyz, int n, int normalize, float *hwd) {
    for (int i = 0; i < n; ++i) {
        float s;
        float h = axis_meter2grid(&M->h, xyz[3 * i + 1], &s);
        float w = axis_meter2grid(&M->w, xyz[3 * i + 0], &s);
        float d = axis_meter2grid(&M->d, xyz[3 * i + 2], &s);
        if (normalize) {
            h = h / (float)(M->h.tot_len - 1);
            w = w / (float)(M->w.tot_len - 1);
            d = d / (float)(M->d.tot_len - 1);
        }
        hwd[3 * i] = h; hwd[3 * i + 1] = w; hwd[3 * i + 2] = d;
    }
}

/* value + metre gradient of the SDF volume at metre positions (field lookup only) */
void oracle_field_sdf(const so_mapping *M, const float *vol, const float *xyz, int n, float *sdf,
                      float *grad) {
    const int H = M->h.tot_len, W = M->w.tot_len, D = M->d.tot_len;
#pragma omp parallel for
    for (int i = 0; i < n; ++i) {
        cell_t c = locate(M, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        float corner[8], weight[8];
        sdf[i] = trilinear(vol, H, W, D, &c, corner, weight);
        if (grad) trilinear_grad(&c, corner, grad + 3 * (size_t)i);
    }
}

float oracle_expf(float x) { return ref_expf(x); }
float oracle_linspace01(int j, int n) { return unit_bin(j, n); }
