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

// ---------------------------------------------------------------------------------------------------------------------
// flatten_feats: the image features as the encoders' `value` (tpvformer_encoder.py:261-277, bevformer_encoder.py:194-210):
// per FPN level (B, N, C, h, w) -> rows (camera, pixel, batch) x C of ONE (N, sum hw, B, C) tensor, plus the camera and the
// level embedding, (feat + cams_embeds[n]) + level_embeds[l] in that order.  torch runs it as two broadcast adds per level on
// permuted views and one concatenating, transposing copy: 12 kernels, 0.22 ms per nuscenes frame; here one pass — a block
// transposes a (C, 64 pixels) tile through LDS (reads coalesced along the pixels, writes along the channels).
struct FlattenArgs {
    const float *feat[8];
    int hw[8], start[8];     // pixels and first row of level l
    int tile0[9];            // first tile of level l (64-pixel tiles)
    const float *cams, *lvls;
    float *out;
    int L, B, N, C, S;
};

__global__ __launch_bounds__(256) void flatten_feats_kernel(FlattenArgs a) {
    extern __shared__ float tile[];      // [C][65]
    int l = 0;
    for (int k = 1; k < a.L; ++k) l += ((int)blockIdx.x >= a.tile0[k]);
    const int p0 = ((int)blockIdx.x - a.tile0[l]) * 64;
    const int n = blockIdx.y, b = blockIdx.z;
    const int hw = a.hw[l], np = min(64, hw - p0);
    const float *src = a.feat[l] + ((size_t)b * a.N + n) * a.C * hw + p0;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < a.C; c += 4)
        if (tx < np) tile[c * 65 + tx] = src[(size_t)c * hw + tx];
    __syncthreads();
    const float *ce = a.cams + (size_t)n * a.C, *le = a.lvls + (size_t)l * a.C;
    for (int e = threadIdx.x; e < np * a.C; e += 256) {
        const int p = e / a.C, c = e - p * a.C;
        a.out[(((size_t)n * a.S + a.start[l] + p0 + p) * a.B + b) * a.C + c] = (tile[c * 65 + p] + ce[c]) + le[c];
    }
}

}  // namespace

extern "C" int selfocc_flatten_feats(const float *const *feats, const int32_t *host_hw, int32_t n_levels, int32_t B, int32_t N,
                                     int32_t C, const float *cams_embeds, const float *level_embeds, float *out, void *stream) {
    SO_REQUIRE(n_levels >= 1 && n_levels <= 8, "flatten_feats: 1 .. 8 levels (got %d)", n_levels);
    SO_REQUIRE(B >= 1 && N >= 1 && C >= 1 && C <= 512 && B < 65536 && N < 65536, "flatten_feats: bad sizes (B %d, N %d, C %d)", B, N, C);
    SO_REQUIRE(feats && host_hw && cams_embeds && level_embeds && out, "flatten_feats: NULL pointer");
    FlattenArgs a;
    long long S = 0, tiles = 0;
    for (int l = 0; l < n_levels; ++l) {
        SO_REQUIRE(feats[l] != nullptr && host_hw[l] >= 1, "flatten_feats: level %d: NULL map or no pixels", l);
        a.feat[l] = feats[l]; a.hw[l] = host_hw[l]; a.start[l] = (int)S; a.tile0[l] = (int)tiles;
        S += host_hw[l];
        tiles += (host_hw[l] + 63) / 64;
    }
    SO_REQUIRE(S * B * N * C < (1LL << 40) && tiles < (1LL << 31) && S < (1LL << 31), "flatten_feats: problem too large");
    a.tile0[n_levels] = (int)tiles;
    a.cams = cams_embeds; a.lvls = level_embeds; a.out = out;
    a.L = n_levels; a.B = B; a.N = N; a.C = C; a.S = (int)S;
    const size_t shm = (size_t)C * 65 * sizeof(float);
    if (shm > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)flatten_feats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(flatten_feats_kernel, dim3((unsigned)tiles, (unsigned)N, (unsigned)B), dim3(256), shm, (hipStream_t)stream, a);
    return so_launch_status();
}

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
