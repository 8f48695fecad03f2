/*
 * oracle_msda.c — CPU restatement of multi-scale deformable attention (forward + backward).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * The arithmetic lives in mmcv==2.0.1 (pinned by the reference's docs/installation.md:22),
 * mmcv/ops/csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh, which is NOT under
 * /root/reference.  This file restates its published algorithm (ms_deformable_im2col /
 * col2im: h_im = loc_y * H - 0.5, w_im = loc_x * W - 0.5, sample skipped unless
 * -1 < h_im < H and -1 < w_im < W, zero-padded bilinear of 4 corners, * attention weight,
 * summed over levels and points), anchored on the reference call sites
 *   model/encoder/bevformer/attention/image_cross_attention.py:340-345
 *   model/encoder/tpvformer/attention/cross_view_hybrid_attention.py:111-116
 * and pinned in tests against the torch formulation the reference itself falls back to on
 * CPU (multi_scale_deformable_attn_pytorch: per-level F.grid_sample(align_corners=False,
 * padding_mode='zeros')) including its autograd gradients.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static float corner(const float *v, int H, int W, int stride, int h, int w) {
    if (h < 0 || w < 0 || h > H - 1 || w > W - 1) return 0.0f;
    return v[(size_t)(h * W + w) * stride];
}

int oracle_msda_fwd(const float *value, const int32_t *shapes, const int32_t *starts,
                    const float *loc, const float *attw, float *out, int bs, int nv, int nq,
                    int heads, int d, int L, int P) {
    const int stride = heads * d;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < bs; ++b)
        for (int q = 0; q < nq; ++q)
            for (int h = 0; h < heads; ++h)
                for (int c = 0; c < d; ++c) {
                    float col = 0.0f;
                    for (int l = 0; l < L; ++l) {
                        const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
                        const float *v = value + ((size_t)b * nv + starts[l]) * stride + h * d + c;
                        for (int p = 0; p < P; ++p) {
                            const size_t i = ((((size_t)b * nq + q) * heads + h) * L + l) * P + p;
                            const float w_im = loc[2 * i] * (float)Wl - 0.5f;
                            const float h_im = loc[2 * i + 1] * (float)Hl - 0.5f;
                            if (!(h_im > -1 && w_im > -1 && h_im < Hl && w_im < Wl)) continue;
                            const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im);
                            const float lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
                            const float v1 = corner(v, Hl, Wl, stride, h0, w0);
                            const float v2 = corner(v, Hl, Wl, stride, h0, w0 + 1);
                            const float v3 = corner(v, Hl, Wl, stride, h0 + 1, w0);
                            const float v4 = corner(v, Hl, Wl, stride, h0 + 1, w0 + 1);
                            const float val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
                            col += val * attw[i];
                        }
                    }
                    out[(((size_t)b * nq + q) * heads + h) * d + c] = col;
                }
    return 0;
}

/* g_value must be zeroed by the caller; single-threaded scatter (deterministic). */
int oracle_msda_bwd(const float *value, const int32_t *shapes, const int32_t *starts,
                    const float *loc, const float *attw, const float *g_out, float *g_value,
                    float *g_loc, float *g_attw, int bs, int nv, int nq, int heads, int d, int L,
                    int P) {
    const int stride = heads * d;
    for (int b = 0; b < bs; ++b)
        for (int q = 0; q < nq; ++q)
            for (int h = 0; h < heads; ++h)
                for (int l = 0; l < L; ++l) {
                    const int Hl = shapes[2 * l], Wl = shapes[2 * l + 1];
                    const size_t vo = ((size_t)b * nv + starts[l]) * stride + h * d;
                    for (int p = 0; p < P; ++p) {
                        const size_t i = ((((size_t)b * nq + q) * heads + h) * L + l) * P + p;
                        g_loc[2 * i] = g_loc[2 * i + 1] = 0.0f;
                        g_attw[i] = 0.0f;
                        const float w_im = loc[2 * i] * (float)Wl - 0.5f;
                        const float h_im = loc[2 * i + 1] * (float)Hl - 0.5f;
                        if (!(h_im > -1 && w_im > -1 && h_im < Hl && w_im < Wl)) continue;
                        const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im);
                        const float lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
                        const int hs[4] = {h0, h0, h0 + 1, h0 + 1}, ws[4] = {w0, w0 + 1, w0, w0 + 1};
                        const float wt[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                        double ga = 0, gw = 0, gh = 0;
                        for (int c = 0; c < d; ++c) {
                            const float top = g_out[(((size_t)b * nq + q) * heads + h) * d + c];
                            const float top_w = top * attw[i];
                            float vv[4];
                            for (int k = 0; k < 4; ++k) {
                                const int ok = hs[k] >= 0 && ws[k] >= 0 && hs[k] <= Hl - 1 && ws[k] <= Wl - 1;
                                const size_t o = vo + (size_t)(hs[k] * Wl + ws[k]) * stride + c;
                                vv[k] = ok ? value[o] : 0.0f;
                                if (ok) g_value[o] += wt[k] * top_w;
                            }
                            ga += (double)top * (wt[0] * vv[0] + wt[1] * vv[1] + wt[2] * vv[2] + wt[3] * vv[3]);
                            gw += (double)top_w * (-hh * vv[0] + hh * vv[1] - lh * vv[2] + lh * vv[3]);
                            gh += (double)top_w * (-hw * vv[0] - lw * vv[1] + hw * vv[2] + lw * vv[3]);
                        }
                        g_attw[i] = (float)ga;
                        g_loc[2 * i] = (float)(Wl * gw);
                        g_loc[2 * i + 1] = (float)(Hl * gh);
                    }
                }
    return 0;
}
