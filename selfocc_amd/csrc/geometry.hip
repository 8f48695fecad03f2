// geometry.hip — point_sampling of the TPV / BEV encoder (model/encoder/bevformer/utils.py:114-170, called once per
// plane and frame from tpvformer_encoder.py / bevformer_encoder.py): project the pillar reference points of every
// query into every camera, normalise by the image size, mark the points that land inside the image in front of
// the camera.  The reference (and this repo's CPU path) runs it as ~25 broadcast torch kernels over (D, B, N, Q)
// tensors — 0.7 ms per nuscenes_depth frame for 1.3 M points x 6 cameras; here it is one pass: a thread owns a
// point, loops over the cameras, writes cam / mask in the (N, B, Q, D) order the attention modules read (no
// permuted views, no .contiguous() copies later) and marks the (camera, query) pairs that see any point — the
// `visible` mask the camera-loop MSDA kernels want (it was an any(-1) reduction per plane and layer).
// Arithmetic = the torch ops' order, float32, no fused multiply-adds (build flag), IEEE division.
#include "so_device.h"

namespace {

__global__ __launch_bounds__(256) void point_sampling_kernel(const float *__restrict__ ref /* (B, D, Q, 3) */,
                                                             const float *__restrict__ l2i /* (B, N, 4, 4) */,
                                                             const float *__restrict__ fx, const float *__restrict__ fy,
                                                             float *__restrict__ cam /* (N, B, Q, D, 2) */,
                                                             uint8_t *__restrict__ mask /* (N, B, Q, D) */,
                                                             uint8_t *__restrict__ visible /* (N, B, Q) */, int B, int D,
                                                             int Q, int N, float img_h, float img_w) {
    const long long n_pts = (long long)B * Q * D;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;     // (b, q, d), d fastest: the output order
    if (t >= n_pts) return;
    const int d = (int)(t % D);
    const long long bq = t / D;
    const int q = (int)(bq % Q), b = (int)(bq / Q);
    const float *p = ref + (((size_t)b * D + d) * Q + q) * 3;
    const float x = p[0], y = p[1], z = p[2];
    const float eps = 1e-5f;
    for (int n = 0; n < N; ++n) {
        const float *M = l2i + ((size_t)b * N + n) * 16;                 // wave-uniform: scalar loads
        const float c0 = ((M[0] * x + M[1] * y) + M[2] * z) + M[3] * 1.0f;
        const float c1 = ((M[4] * x + M[5] * y) + M[6] * z) + M[7] * 1.0f;
        const float c2 = ((M[8] * x + M[9] * y) + M[10] * z) + M[11] * 1.0f;
        const float den = fmaxf(c2, eps);
        float u = (c0 / den) / img_w, v = (c1 / den) / img_h;
        const bool in = (c2 > eps) && (v > 0.0f) && (v < 1.0f) && (u < 1.0f) && (u > 0.0f);
        if (fx != nullptr) { u = u * fx[n]; v = v * fy[n]; }
        const size_t o = (((size_t)n * B + b) * Q + q) * D + d;
        *(float2 *)(cam + 2 * o) = make_float2(u, v);
        mask[o] = in ? 1 : 0;
        if (in && visible != nullptr) visible[((size_t)n * B + b) * Q + q] = 1;   // same value from every writer
    }
}

}  // namespace

extern "C" int selfocc_point_sampling(const float *ref, const float *lidar2img, const float *focal_x, const float *focal_y,
                                      float *cam, uint8_t *mask, uint8_t *visible, int32_t B, int32_t D, int32_t Q,
                                      int32_t N, float img_h, float img_w, void *stream) {
    SO_REQUIRE(B >= 0 && D >= 0 && Q >= 0 && N >= 0, "point_sampling: negative size");
    const long long n_pts = (long long)B * Q * D;
    if (n_pts == 0 || N == 0) return 0;
    SO_REQUIRE(ref && lidar2img && cam && mask, "point_sampling: NULL pointer");
    SO_REQUIRE((focal_x == nullptr) == (focal_y == nullptr), "point_sampling: focal_x / focal_y both or neither");
    SO_REQUIRE(n_pts * N < (1LL << 40) && (n_pts + 255) / 256 < (1LL << 31), "point_sampling: problem too large");
    hipStream_t st = (hipStream_t)stream;
    if (visible != nullptr) (void)hipMemsetAsync(visible, 0, (size_t)N * B * Q, st);
    hipLaunchKernelGGL(point_sampling_kernel, dim3((unsigned)((n_pts + 255) / 256)), dim3(256), 0, st, ref, lidar2img, focal_x,
                       focal_y, cam, mask, visible, B, D, Q, N, img_h, img_w);
    return so_launch_status();
}
