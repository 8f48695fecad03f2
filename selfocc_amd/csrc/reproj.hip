// reproj.hip — fused temporal reprojection photometric term for gfx950 (MI355X).
//
// Replaces, per camera, the per-SAMPLE part of ReprojLossMonoMultiNewCombine.reproj_loss
// (loss/reproj_loss_mono_multi_new_combine.py:108-201), which the reference runs as ~40
// elementwise / index kernels over R*S = 1.2 M samples per camera:
//   (u t, v t, t, 1) -> img2prevImg / img2nextImg -> perspective divide (eps 1e-5) -> in-image
//   mask (:118-133) -> F.grid_sample(prev / next image, bilinear, border, align_corners=True)
//   (:140-152) -> L1 vs the current colour, masked mean over the two frames (:166-176) ->
//   weights masked + renormalised per ray (index_add_/gather, :178-184) -> per-ray weighted L1
//   (:186-187) and weighted composite colour (:190-201) -> "no valid sample" ray filter (:223-225).
// SSIM, the auto-mask minimum and the mean stay in torch on the tiny (R, 3) lattice images.
//
// Hardware mapping: one wavefront per ray, lane l owns M = ceil(S / 64) consecutive samples
// (consecutive samples of a ray project onto neighbouring pixels of an epipolar line, so
// the 2 x 4 x 3 bilinear taps of a wave land in a few cache lines); the per-ray sums are
// wavefront-shuffle reductions; weights / ts are read as contiguous runs.  Nothing
// per-sample is written in the forward pass; backward recomputes the taps.
#include "so_device.h"

namespace {

struct Taps {
    float diff;      // masked mean |curr - warped| over the valid frames
    float comb[3];   // (rgb_prev * m_prev + rgb_next * m_next) / max(cnt, 1)
    bool any;        // general_mask: at least one frame valid
};

SO_DEVFN void project(const float *__restrict__ T, float u, float v, float t, float img_h, float img_w,
                      float &px, float &py, bool &ok) {
    // cal_pixel (:118-133): trans @ (u t, v t, t, 1)
    const float x = u * t, y = v * t;
    const float p0 = ((T[0] * x + T[1] * y) + T[2] * t) + T[3];
    const float p1 = ((T[4] * x + T[5] * y) + T[6] * t) + T[7];
    const float p2 = ((T[8] * x + T[9] * y) + T[10] * t) + T[11];
    const float den = fmaxf(1e-5f, p2);
    px = p0 / den;
    py = p1 / den;
    ok = (p2 > 0.0f) && (px > 0.0f) && (px < img_w) && (py > 0.0f) && (py < img_h);
}

// F.grid_sample(bilinear, padding_mode='border', align_corners=True) of a planar (3, Hi, Wi) image
SO_DEVFN void sample_rgb(const float *__restrict__ img, int Hi, int Wi, float px, float py, float img_h,
                         float img_w, float out[3]) {
    // sample_pixel (:140-152): pixel / img_size * 2 - 1, then un-normalise ((c + 1) / 2) * (size - 1)
    float x = ((((px / img_w) * 2.0f - 1.0f) + 1.0f) / 2.0f) * (float)(Wi - 1);
    float y = ((((py / img_h) * 2.0f - 1.0f) + 1.0f) / 2.0f) * (float)(Hi - 1);
    x = fminf(fmaxf(x, 0.0f), (float)(Wi - 1));  // border: clip coordinates
    y = fminf(fmaxf(y, 0.0f), (float)(Hi - 1));
    const float fx = floorf(x), fy = floorf(y);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = min(x0 + 1, Wi - 1), y1 = min(y0 + 1, Hi - 1);
    const bool x1in = (x0 + 1 <= Wi - 1), y1in = (y0 + 1 <= Hi - 1);
    const float wx1 = x - fx, wx0 = (fx + 1.0f) - x, wy1 = y - fy, wy0 = (fy + 1.0f) - y;
    const float nw = wx0 * wy0, ne = x1in ? wx1 * wy0 : 0.0f, sw = y1in ? wx0 * wy1 : 0.0f,
                se = (x1in && y1in) ? wx1 * wy1 : 0.0f;
    const size_t plane = (size_t)Hi * Wi;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *p = img + c * plane;
        float acc = p[(size_t)y0 * Wi + x0] * nw;
        acc = acc + p[(size_t)y0 * Wi + x1] * ne;
        acc = acc + p[(size_t)y1 * Wi + x0] * sw;
        acc = acc + p[(size_t)y1 * Wi + x1] * se;
        out[c] = acc;
    }
}

SO_DEVFN Taps sample_taps(const so_reproj_args &a, float u, float v, float t, const float cur[3]) {
    Taps r;
    float px, py, qx, qy;
    bool mp, mn;
    project(a.T_prev, u, v, t, a.img_h, a.img_w, px, py, mp);
    project(a.T_next, u, v, t, a.img_h, a.img_w, qx, qy, mn);
    float rp[3], rn[3];
    sample_rgb(a.img_prev, a.Hi, a.Wi, px, py, a.img_h, a.img_w, rp);
    sample_rgb(a.img_next, a.Hi, a.Wi, qx, qy, a.img_h, a.img_w, rn);
    float dp = ((fabsf(cur[0] - rp[0]) + fabsf(cur[1] - rp[1])) + fabsf(cur[2] - rp[2])) / 3.0f;
    float dn = ((fabsf(cur[0] - rn[0]) + fabsf(cur[1] - rn[1])) + fabsf(cur[2] - rn[2])) / 3.0f;
    if (!mp) dp = 0.0f;
    if (!mn) dn = 0.0f;
    const float cnt_raw = (mp ? 1.0f : 0.0f) + (mn ? 1.0f : 0.0f);
    r.any = cnt_raw > 0.0f;
    const float cnt = fmaxf(cnt_raw, 1.0f);
    r.diff = (dp + dn) / cnt;
#pragma unroll
    for (int c = 0; c < 3; ++c) r.comb[c] = ((mp ? rp[c] : 0.0f) + (mn ? rn[c] : 0.0f)) / cnt;
    return r;
}

SO_DEVFN float wsum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// effective weight of a sample: optional w / delta (:111-116), then general-mask zeroing (:178-179)
SO_DEVFN float eff_weight(const so_reproj_args &a, size_t o, bool any, float &scale) {
    float w = a.weights[o];
    scale = 1.0f;
    if (a.deltas) {
        const float eps = 1.1920928955078125e-07f;
        const float d = a.deltas[o];
        scale = (d < eps) ? 0.0f : 1.0f / fmaxf(d, eps);
        w = (d < eps) ? 0.0f : w / fmaxf(d, eps);
    }
    if (!any) { w = 0.0f; scale = 0.0f; }
    return w;
}

template <int M, bool BWD>
__global__ __launch_bounds__(256) void reproj_kernel(so_reproj_args a, const float *__restrict__ g_l1,
                                                     const float *__restrict__ g_comb,
                                                     float *__restrict__ g_weights) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= a.R) return;
    const float u = a.pix[2 * (size_t)ray], v = a.pix[2 * (size_t)ray + 1];
    const float cur[3] = {a.curr_rgb[3 * (size_t)ray], a.curr_rgb[3 * (size_t)ray + 1], a.curr_rgb[3 * (size_t)ray + 2]};
    const float eps = 1.1920928955078125e-07f;

    Taps tp[M];
    float w[M], sc[M];
    bool live[M];
    float ws_l = 0.0f, valid_l = 0.0f;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int i = lane * M + j;
        live[j] = i < a.S;
        const size_t o = (size_t)ray * a.S + (live[j] ? i : a.S - 1);
        tp[j] = sample_taps(a, u, v, a.ts[o], cur);
        w[j] = eff_weight(a, o, tp[j].any, sc[j]);
        if (!live[j]) { w[j] = 0.0f; sc[j] = 0.0f; tp[j].any = false; }
        ws_l += w[j];
        valid_l += tp[j].any ? 1.0f : 0.0f;
    }
    const float wtot_raw = wsum(ws_l);
    const float wtot = fmaxf(wtot_raw, eps);  // clamp_min(finfo.eps) (:182)
    const float inv_w = 1.0f / wtot;

    if constexpr (!BWD) {
        float l1_l = 0.0f, c_l[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const float wn = w[j] * inv_w;
            l1_l = fmaf(wn, tp[j].diff, l1_l);
#pragma unroll
            for (int c = 0; c < 3; ++c) c_l[c] = fmaf(wn, tp[j].comb[c], c_l[c]);
            if (a.wnorm && live[j]) a.wnorm[(size_t)ray * a.S + lane * M + j] = wn;
        }
        const float l1 = wsum(l1_l), c0 = wsum(c_l[0]), c1 = wsum(c_l[1]), c2 = wsum(c_l[2]);
        const float nvalid = wsum(valid_l);
        if (lane == 0) {
            if (a.l1) a.l1[ray] = l1;
            if (a.rgb_combine) {
                a.rgb_combine[3 * (size_t)ray] = c0; a.rgb_combine[3 * (size_t)ray + 1] = c1;
                a.rgb_combine[3 * (size_t)ray + 2] = c2;
            }
            if (a.any_valid) a.any_valid[ray] = nvalid > 0.0f ? 1.0f : 0.0f;
        }
    } else {
        // L = sum_s wn_s a_s,  a_s = g_l1 diff_s + g_comb . comb_s,  wn = w / max(sum w, eps)
        const float gl = g_l1 ? g_l1[ray] : 0.0f;
        float gc[3] = {0.0f, 0.0f, 0.0f};
        if (g_comb) { gc[0] = g_comb[3 * (size_t)ray]; gc[1] = g_comb[3 * (size_t)ray + 1]; gc[2] = g_comb[3 * (size_t)ray + 2]; }
        float as[M], abar_l = 0.0f;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            as[j] = fmaf(gl, tp[j].diff, (gc[0] * tp[j].comb[0] + gc[1] * tp[j].comb[1]) + gc[2] * tp[j].comb[2]);
            abar_l = fmaf(w[j] * inv_w, as[j], abar_l);
        }
        const float abar = (wtot_raw > eps) ? wsum(abar_l) : 0.0f;  // clamped denominator is a constant
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (!live[j]) continue;
            g_weights[(size_t)ray * a.S + lane * M + j] = sc[j] * (as[j] - abar) * inv_w;
        }
    }
}

int validate(const so_reproj_args &a) {
    SO_REQUIRE(a.R >= 0 && a.S >= 1 && a.S <= 512, "reproj: need R >= 0, 1 <= S <= 512");
    if (a.R == 0) return 0;
    SO_REQUIRE(a.weights && a.ts && a.pix && a.curr_rgb && a.T_prev && a.T_next && a.img_prev && a.img_next,
               "reproj: NULL input pointer");
    SO_REQUIRE(a.Hi >= 1 && a.Wi >= 1 && a.img_h > 0 && a.img_w > 0, "reproj: bad image size");
    return 0;
}

template <bool BWD>
int launch(const so_reproj_args &a, const float *g_l1, const float *g_comb, float *g_w, hipStream_t st) {
    const int m = (a.S + 63) / 64;
    const int blocks = (a.R + 3) / 4;
#define SO_L(MM) hipLaunchKernelGGL((reproj_kernel<MM, BWD>), dim3(blocks), dim3(256), 0, st, a, g_l1, g_comb, g_w)
    if (m <= 1) SO_L(1);
    else if (m <= 2) SO_L(2);
    else if (m <= 4) SO_L(4);
    else SO_L(8);
#undef SO_L
    return so_launch_status();
}

}  // namespace

extern "C" int selfocc_reproj_fwd(const so_reproj_args *args, void *stream) {
    SO_REQUIRE(args != nullptr, "args is NULL");
    if (validate(*args)) return -1;
    if (args->R == 0) return 0;
    return launch<false>(*args, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int selfocc_reproj_bwd(const so_reproj_args *args, const float *g_l1, const float *g_rgb_combine,
                                  float *g_weights, void *stream) {
    SO_REQUIRE(args != nullptr, "args is NULL");
    if (validate(*args)) return -1;
    if (args->R == 0) return 0;
    SO_REQUIRE(g_weights != nullptr, "reproj_bwd: g_weights is NULL");
    return launch<true>(*args, g_l1, g_rgb_combine, g_weights, (hipStream_t)stream);
}
