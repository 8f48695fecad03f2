// so_device.h — device-side helpers shared by the gfx950 kernels of libselfocc_hip.so.
//
// Arithmetic contract ("canonical order"): this translation unit is built
// with -ffp-contract=off; a fused multiply-add only happens where fmaf() is spelled
// out.  The CPU oracle (oracle/) spells the same sequence independently, so integer /
// threshold results are bit-exact and float results agree to the last few ulp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/selfocc_hip.h"

#define SO_DEVFN __device__ __forceinline__

// ---- error plumbing -------------------------------------------------------------------
void so_set_error(const char *fmt, ...);
#define SO_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            so_set_error(__VA_ARGS__); \
            return -1;                 \
        }                              \
    } while (0)

static inline int so_launch_status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        so_set_error("HIP launch failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ---- canonical expf / sigmoid -----------------------------------------------------------
// Cody-Waite reduction + Cephes degree-5 polynomial, every step an explicit IEEE op so that
// the CPU oracle reproduces it bit for bit (the NeuS alpha subtracts two sigmoids that
// differ by ~1e-5, so a 1-ulp exp difference would show up as ~1e-3 relative in alpha).
SO_DEVFN float so_expf(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
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
    // 2^e in two factors so that e = -126 .. 127 never builds a denormal / inf scale
    int e1 = e >> 1, e2 = e - e1;
    float s1 = __int_as_float((e1 + 127) << 23);
    float s2 = __int_as_float((e2 + 127) << 23);
    return (y * s1) * s2;
}

SO_DEVFN float so_sigmoid(float x) { return 1.0f / (1.0f + so_expf(-x)); }

// inv_s of a render launch: the device copy wins when given (so_render_args::inv_s_dev)
SO_DEVFN float so_inv_s(const so_render_args &a) { return a.inv_s_dev ? a.inv_s_dev[0] : a.inv_s; }

// ---- grid <-> metre ---------------------------------------------------------------------
// LinearMapping.meter2grid, one axis (reference model/encoder/bevformer/mappings.py:97-143)
SO_DEVFN float so_axis_m2g(const so_axis &A, float m, float &slope) {
    float c = m - A.start;
    float a = fabsf(c);
    float g = a / A.range0 * A.size0;
    slope = A.size0 / A.range0;
    if (A.size1 != 0.0f && a > A.range0) {
        g = A.size0 + (a - A.range0) / A.range1 * A.size1;
        slope = A.size1 / A.range1;
    }
    float s = c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : 0.0f);
    float t = s * g;
    return (t + A.off0) + A.off1;
}

// grid index -> the coordinate F.grid_sample(align_corners=True) computes after the
// reference's normalize (/(tot_len-1)), 2g-1 and un-normalise steps
// (mappings.py:145-148, nerfacc_head/bev_nerf.py:103-113).
SO_DEVFN float so_grid_coord(float g, int tot_len) {
    float lm1 = (float)(tot_len - 1);
    float gn = g / lm1;
    float c = 2.0f * gn - 1.0f;
    return ((c + 1.0f) / 2.0f) * lm1;
}

struct so_cell {
    int h0, w0, d0;          // floor of the grid coordinate
    float fh0, fh1;          // weights of h0 / h0+1 (torch: (h1 - ih), (ih - h0))
    float fw0, fw1, fd0, fd1;
    float sh, sw, sd;        // d grid / d metre along each axis
};

SO_DEVFN so_cell so_locate(const so_mapping &M, float x, float y, float z) {
    so_cell c;
    float gh = so_grid_coord(so_axis_m2g(M.h, y, c.sh), M.h.tot_len);
    float gw = so_grid_coord(so_axis_m2g(M.w, x, c.sw), M.w.tot_len);
    float gd = so_grid_coord(so_axis_m2g(M.d, z, c.sd), M.d.tot_len);
    float fh = floorf(gh), fw = floorf(gw), fd = floorf(gd);
    c.h0 = (int)fh; c.w0 = (int)fw; c.d0 = (int)fd;
    c.fh1 = gh - fh; c.fh0 = (fh + 1.0f) - gh;
    c.fw1 = gw - fw; c.fw0 = (fw + 1.0f) - gw;
    c.fd1 = gd - fd; c.fd0 = (fd + 1.0f) - gd;
    return c;
}

struct __attribute__((packed, aligned(4))) so_f2u { float x, y; };

SO_DEVFN float so_bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// 8 corner values of the SDF cell in torch's grid_sampler_3d order:
//   k = 4*dh + 2*dw + dd  with grid_sample's (z, y, x) = (h, w, d)
//   torch names: tnw tne tsw tse bnw bne bsw bse = k 0 1 2 3 4 5 6 7
// Out-of-range corners read as 0 (padding_mode='zeros').
SO_DEVFN void so_gather_sdf(const float *__restrict__ vol, int H, int W, int D,
                            const so_cell &c, float v[8]) {
    int d0c = min(max(c.d0, 0), D - 2);
    bool dlo_in = (c.d0 >= 0) && (c.d0 <= D - 1);
    bool dhi_in = (c.d0 + 1 >= 0) && (c.d0 + 1 <= D - 1);
    bool lo_first = (c.d0 == d0c);      // value at d0 is pair.x, else pair.y
    bool hi_first = (c.d0 + 1 == d0c);  // value at d0+1 is pair.x, else pair.y
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            int h = c.h0 + dh, w = c.w0 + dw;
            bool in = (h >= 0) && (h < H) && (w >= 0) && (w < W);
            int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
            const so_f2u *p = (const so_f2u *)(vol + ((size_t)hc * W + wc) * D + d0c);
            so_f2u pr = *p;
            float lo = lo_first ? pr.x : pr.y;
            float hi = hi_first ? pr.x : pr.y;
            v[4 * dh + 2 * dw + 0] = (in && dlo_in) ? lo : 0.0f;
            v[4 * dh + 2 * dw + 1] = (in && dhi_in) ? hi : 0.0f;
        }
    }
}

// trilinear value in torch's accumulation order (out = 0; out += v_k * w_k, k = 0..7,
// w_k = (wd * ww) * wh), and the metre-space gradient in torch's backward order.
SO_DEVFN float so_trilerp_sdf(const so_cell &c, const float v[8], float wk[8]) {
    const float fd[2] = {c.fd0, c.fd1}, fw[2] = {c.fw0, c.fw1}, fh[2] = {c.fh0, c.fh1};
    float out = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        wk[k] = (fd[k & 1] * fw[(k >> 1) & 1]) * fh[k >> 2];
        out = out + v[k] * wk[k];
    }
    return out;
}

SO_DEVFN void so_trilerp_grad(const so_cell &c, const float v[8], float &gx, float &gy,
                              float &gz) {
    const float fd[2] = {c.fd0, c.fd1}, fw[2] = {c.fw0, c.fw1}, fh[2] = {c.fh0, c.fh1};
    float gd = 0.0f, gw = 0.0f, gh = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int kd = k & 1, kw = (k >> 1) & 1, kh = k >> 2;
        float td = (v[k] * fw[kw]) * fh[kh];
        float tw = (v[k] * fd[kd]) * fh[kh];
        float th = (v[k] * fd[kd]) * fw[kw];
        gd = kd ? gd + td : gd - td;
        gw = kw ? gw + tw : gw - tw;
        gh = kh ? gh + th : gh - th;
    }
    gx = gw * c.sw;  // metre x <-> grid w
    gy = gh * c.sh;  // metre y <-> grid h
    gz = gd * c.sd;  // metre z <-> grid d
}

// Explicit 2-wide FMAs.  The library is built without the compiler's vectorizers (csrc/build.sh: their half-swapping op_sel
// forms of v_pk_*_f32 are not safe beside bf16 MFMA waves on gfx950); where the packed rate matters the sources spell the
// low-half-BROADCAST form themselves (v_pk_fma_f32 ... op_sel_hi:[0,1,1], measured clean; tests/test_isa_lint.py).
typedef float so_f32x2 __attribute__((ext_vector_type(2)));
// f[0..3] += w * (t0, t1, t2, t3)
__device__ __forceinline__ void so_fma4_bcast(float &f0, float &f1, float &f2, float &f3, float t0, float t1, float t2, float t3,
                                              float w) {
    const so_f32x2 ww = {w, w};
    so_f32x2 a = {f0, f1}, b = {f2, f3};
    a = __builtin_elementwise_fma(so_f32x2{t0, t1}, ww, a);
    b = __builtin_elementwise_fma(so_f32x2{t2, t3}, ww, b);
    f0 = a[0]; f1 = a[1]; f2 = b[0]; f3 = b[1];
}
