// ssim.hip — the photometric SSIM term of the reprojection / colour losses as one launch per direction (gfx950).
//
// Reference: class SSIM in loss/reproj_loss_mono_multi_new_combine.py:26-66 (monodepth2's): reflection pad 1,
// five 3x3 average pools (mu_x, mu_y, E[x^2], E[y^2], E[xy]), out = clamp((1 - n / d) / 2, 0, 1) with
//   n = (2 mu_x mu_y + C1)(2 sigma_xy + C2),  d = (mu_x^2 + mu_y^2 + C1)(sigma_x + sigma_y + C2),
// called 19 times per nuscenes_occ iteration on 48 x 100 ray-lattice images (3 channels): ~40 tiny torch kernels
// forward and ~80 backward per call, i.e. ~2 200 launches per iteration for a few microseconds of work each.
//
// Forward: a thread per output pixel sums its 3x3 window (same tap order and divisions as avg_pool2d).
// Backward: a thread per INPUT pixel gathers the derivative of every window that contains one of its (reflected)
// copies — no atomics, deterministic.  Inputs are addressed through their strides (the call sites pass
// channel-last views), so no layout copies either.
#include "so_device.h"

namespace {

struct SsimArgs {
    const float *x, *y;
    long long xs[4], ys[4];     // element strides of (n, c, h, w)
    int N, C, H, W;
    float *out;                 // (N, C, H, W) contiguous                         [forward]
    const float *g_out;         // (N, C, H, W) contiguous                         [backward]
    float *gx, *gy;             // (N, C, H, W) contiguous, either may be NULL     [backward]
};

constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

SO_DEVFN int so_refl(int q, int n) { return q < 0 ? -q : (q >= n ? 2 * n - 2 - q : q); }   // reflection pad 1

struct SsimWin {
    float mu_x, mu_y, A1, A2, B1, B2, S;
};

SO_DEVFN SsimWin so_ssim_window(const float *xp, const float *yp, const long long (&xs)[4], const long long (&ys)[4],
                                int ci, int cj, int H, int W) {
    float sx = 0.0f, sy = 0.0f, sxx = 0.0f, syy = 0.0f, sxy = 0.0f;
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
        const int ri = so_refl(ci + di, H);
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            const int rj = so_refl(cj + dj, W);
            const float a = xp[ri * xs[2] + rj * xs[3]], b = yp[ri * ys[2] + rj * ys[3]];
            sx += a; sy += b;
            sxx += a * a; syy += b * b; sxy += a * b;
        }
    }
    SsimWin w;
    w.mu_x = sx / 9.0f; w.mu_y = sy / 9.0f;
    const float sigma_x = sxx / 9.0f - w.mu_x * w.mu_x;
    const float sigma_y = syy / 9.0f - w.mu_y * w.mu_y;
    const float sigma_xy = sxy / 9.0f - w.mu_x * w.mu_y;
    w.A1 = 2.0f * w.mu_x * w.mu_y + kC1;
    w.A2 = 2.0f * sigma_xy + kC2;
    w.B1 = w.mu_x * w.mu_x + w.mu_y * w.mu_y + kC1;
    w.B2 = sigma_x + sigma_y + kC2;
    w.S = (w.A1 * w.A2) / (w.B1 * w.B2);
    return w;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(SsimArgs a) {
    const long long n_el = (long long)a.N * a.C * a.H * a.W;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_el) return;
    const int j = (int)(e % a.W), i = (int)((e / a.W) % a.H);
    const int c = (int)((e / ((long long)a.W * a.H)) % a.C), n = (int)(e / ((long long)a.W * a.H * a.C));
    const SsimWin w = so_ssim_window(a.x + n * a.xs[0] + c * a.xs[1], a.y + n * a.ys[0] + c * a.ys[1], a.xs, a.ys, i, j,
                                     a.H, a.W);
    a.out[e] = fminf(fmaxf((1.0f - w.S) / 2.0f, 0.0f), 1.0f);
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(SsimArgs a) {
    const long long n_el = (long long)a.N * a.C * a.H * a.W;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_el) return;
    const int j = (int)(e % a.W), i = (int)((e / a.W) % a.H);
    const int c = (int)((e / ((long long)a.W * a.H)) % a.C), n = (int)(e / ((long long)a.W * a.H * a.C));
    const float *xp = a.x + n * a.xs[0] + c * a.xs[1], *yp = a.y + n * a.ys[0] + c * a.ys[1];
    const float *gp = a.g_out + ((long long)n * a.C + c) * a.H * a.W;
    const float xv = xp[i * a.xs[2] + j * a.xs[3]], yv = yp[i * a.ys[2] + j * a.ys[3]];
    // padded positions that hold a copy of source row i: i itself, -1 (copy of row 1), H (copy of row H - 2)
    int qi[3], qj[3], nqi = 0, nqj = 0;
    qi[nqi++] = i;
    if (i == 1) qi[nqi++] = -1;
    if (i == a.H - 2) qi[nqi++] = a.H;
    qj[nqj++] = j;
    if (j == 1) qj[nqj++] = -1;
    if (j == a.W - 2) qj[nqj++] = a.W;
    float gx = 0.0f, gy = 0.0f;
    for (int u = 0; u < nqi; ++u) {
        for (int v = 0; v < nqj; ++v) {
            for (int ci = max(qi[u] - 1, 0); ci <= min(qi[u] + 1, a.H - 1); ++ci) {
                for (int cj = max(qj[v] - 1, 0); cj <= min(qj[v] + 1, a.W - 1); ++cj) {
                    const float g = gp[(long long)ci * a.W + cj];
                    if (g == 0.0f) continue;
                    const SsimWin w = so_ssim_window(xp, yp, a.xs, a.ys, ci, cj, a.H, a.W);
                    const float val = (1.0f - w.S) / 2.0f;
                    if (val < 0.0f || val > 1.0f) continue;          // clamp: no gradient outside [0, 1]
                    const float dS = -0.5f * g;
                    const float inv = 1.0f / (w.B1 * w.B2);
                    // d S / d x_p and d S / d y_p for one occurrence of the pixel in the window
                    const float dA2x = 2.0f * (yv - w.mu_y) / 9.0f, dB2x = 2.0f * (xv - w.mu_x) / 9.0f;
                    const float dA2y = 2.0f * (xv - w.mu_x) / 9.0f, dB2y = 2.0f * (yv - w.mu_y) / 9.0f;
                    const float dA1x = 2.0f * w.mu_y / 9.0f, dB1x = 2.0f * w.mu_x / 9.0f;
                    const float dA1y = 2.0f * w.mu_x / 9.0f, dB1y = 2.0f * w.mu_y / 9.0f;
                    gx += dS * ((dA1x * w.A2 + w.A1 * dA2x) * inv - w.S * (dB1x / w.B1 + dB2x / w.B2));
                    gy += dS * ((dA1y * w.A2 + w.A1 * dA2y) * inv - w.S * (dB1y / w.B1 + dB2y / w.B2));
                }
            }
        }
    }
    if (a.gx) a.gx[e] = gx;
    if (a.gy) a.gy[e] = gy;
}

int ssim_check(const float *x, const float *y, const int64_t *xs, const int64_t *ys, int N, int C, int H, int W) {
    SO_REQUIRE(N >= 0 && C >= 1 && H >= 2 && W >= 2, "ssim: bad shape (%d, %d, %d, %d) (reflection pad needs H, W >= 2)", N, C,
               H, W);
    SO_REQUIRE(N == 0 || (x && y && xs && ys), "ssim: NULL pointer");
    SO_REQUIRE((long long)N * C * H * W < (1LL << 31) * 256, "ssim: too many elements");
    return 0;
}

}  // namespace

extern "C" int selfocc_ssim_fwd(const float *x, const float *y, const int64_t *x_strides, const int64_t *y_strides,
                                int32_t N, int32_t C, int32_t H, int32_t W, float *out, void *stream) {
    if (ssim_check(x, y, x_strides, y_strides, N, C, H, W)) return -1;
    const long long n_el = (long long)N * C * H * W;
    if (n_el == 0) return 0;
    SO_REQUIRE(out != nullptr, "ssim_fwd: out is NULL");
    SsimArgs a{x, y, {x_strides[0], x_strides[1], x_strides[2], x_strides[3]},
               {y_strides[0], y_strides[1], y_strides[2], y_strides[3]}, N, C, H, W, out, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return so_launch_status();
}

extern "C" int selfocc_ssim_bwd(const float *x, const float *y, const int64_t *x_strides, const int64_t *y_strides,
                                int32_t N, int32_t C, int32_t H, int32_t W, const float *g_out, float *g_x, float *g_y,
                                void *stream) {
    if (ssim_check(x, y, x_strides, y_strides, N, C, H, W)) return -1;
    const long long n_el = (long long)N * C * H * W;
    if (n_el == 0) return 0;
    SO_REQUIRE(g_out != nullptr && (g_x != nullptr || g_y != nullptr), "ssim_bwd: NULL gradient pointer");
    SsimArgs a{x, y, {x_strides[0], x_strides[1], x_strides[2], x_strides[3]},
               {y_strides[0], y_strides[1], y_strides[2], y_strides[3]}, N, C, H, W, nullptr, g_out, g_x, g_y};
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return so_launch_status();
}
