// occ.hip — dense field query + Occ3D occupancy tail + integer IoU counts for gfx950.
//
// Replaces:
//   * NeuSHead.get_uniform_sdf -> field.forward_geonetwork / forward_sdfnetwork
//     (model/head/neus_head/neus_head.py:265-293): trilinear lookup of the pre-computed
//     volume at a metre lattice;
//   * eval_iou.py:211-250: F.grid_sample of the dense SDF (+ logits) at the ego-frame
//     Occ3D lattice, (sdf <= thresh), border crop, argmax, openseed2nuscenes LUT;
//   * MeanIoU._after_step (utils/metric_util.py:90-121): integer confusion counts.
//
// Everything here is the CANONICAL arithmetic (so_device.h): the trilinear value is
// bit-identical to torch's CPU grid_sampler_3d, hence the integer occupancy is bit-exact.
// The work is small (640 k lattice points) and pure gather: a team of 8 lanes per output point (one channel
// group each), D-axis neighbours of the SDF fetched as one 8-byte load.
#include "so_device.h"

namespace {

// canonical trilinear of one channel of a channels-last [H][W][D][C] volume
SO_DEVFN float so_trilerp_chan(const float *__restrict__ vol, int H, int W, int D, int C, int ch,
                               const so_cell &c) {
    const float fd[2] = {c.fd0, c.fd1}, fw[2] = {c.fw0, c.fw1}, fh[2] = {c.fh0, c.fh1};
    float out = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int h = c.h0 + (k >> 2), w = c.w0 + ((k >> 1) & 1), d = c.d0 + (k & 1);
        const bool in = (h >= 0) && (h < H) && (w >= 0) && (w < W) && (d >= 0) && (d < D);
        const float wk = (fd[k & 1] * fw[(k >> 1) & 1]) * fh[k >> 2];
        if (in) out = out + vol[(((size_t)h * W + w) * D + d) * C + ch] * wk;
    }
    return out;
}

// ---- channel teams ------------------------------------------------------------------------------------------------
// A query point is served by a TEAM of 8 adjacent lanes: lane j owns channels j, j + 8, j + 16, ... of the channels-last
// record, so one load instruction of the team reads 32 contiguous bytes of a corner (a lane-per-point loop over the 21
// semantic channels issued 168 scattered dword loads per lane and left the chip with 10 k waves for the whole 640 k-point
// lattice: 1.5 ms; the teams make it 80 k short waves).  Per channel the arithmetic is unchanged — out = sum over the
// corners k = 0..7 in order, out-of-range corners skipped (torch's grid_sampler_3d, padding_mode = 'zeros') — so values
// and the integer occupancy / arg-max stay bit-identical.  Lane 0 of the team also does the SDF channel.
constexpr int kTeam = 8, kTeamRounds = 4;          // up to 32 channels

struct TeamCorner {
    size_t vox[8];     // voxel index of corner k
    float w[8];        // its trilinear weight ((fd * fw) * fh, torch's order)
    bool in[8];
};

SO_DEVFN TeamCorner so_team_corners(const so_cell &c, int H, int W, int D) {
    TeamCorner t;
    const float fd[2] = {c.fd0, c.fd1}, fw[2] = {c.fw0, c.fw1}, fh[2] = {c.fh0, c.fh1};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int h = c.h0 + (k >> 2), w = c.w0 + ((k >> 1) & 1), d = c.d0 + (k & 1);
        t.in[k] = (h >= 0) && (h < H) && (w >= 0) && (w < W) && (d >= 0) && (d < D);
        t.w[k] = (fd[k & 1] * fw[(k >> 1) & 1]) * fh[k >> 2];
        t.vox[k] = t.in[k] ? ((size_t)h * W + w) * D + d : 0;
    }
    return t;
}

// this lane's channels (j, j + 8, ...) of the trilinear lookup; the team's first maximum over all n_ch channels
template <bool BF16>
SO_DEVFN void so_team_lookup(const void *__restrict__ vol, int stride, int ch0, int n_ch, const TeamCorner &t, int j,
                             float (&out)[kTeamRounds], float &best, int &arg) {
    best = -INFINITY;
    arg = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < kTeamRounds; ++r) {
        const int ch = j + kTeam * r;
        out[r] = 0.0f;
        if (ch < n_ch) {     // (n_ch <= 32: host)
            float v = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (t.in[k]) {
                    const size_t e = t.vox[k] * (size_t)stride + ch0 + ch;
                    const float x = BF16 ? so_bf16_to_f32(((const uint16_t *)vol)[e]) : ((const float *)vol)[e];
                    v = v + x * t.w[k];
                }
            }
            out[r] = v;
            if (v > best) { best = v; arg = ch; }      // ascending channels: the lane's first maximum
        }
    }
    if (arg == 0x7fffffff) arg = 0;                      // nothing beat -inf (or the lane has no channel): like the serial loop
    const bool has = j < n_ch;
    float b = has ? best : -INFINITY;
    int a_ = has ? arg : 0x7ffffffe;
#pragma unroll
    for (int m = 1; m < kTeam; m <<= 1) {                // first maximum across the team: larger value, then smaller channel
        const float ob = __shfl_xor(b, m, 64);
        const int oa = __shfl_xor(a_, m, 64);
        if (ob > b || (ob == b && oa < a_)) { b = ob; a_ = oa; }
    }
    best = b;
    arg = (a_ >= 0x7ffffffe) ? 0 : a_;
    // the serial loop keeps arg = 0 unless some value is > -inf
    if (!(b > -INFINITY)) arg = 0;
}

__global__ __launch_bounds__(256) void field_query_kernel(so_query_args a) {
    const long long gt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pt = gt / kTeam;
    const int j = (int)(gt - pt * kTeam);
    const bool live = pt < a.n;
    const int i = live ? (int)pt : a.n - 1;      // dead teams shadow the last point (the shuffles need every lane)
    const int H = a.map.h.tot_len, W = a.map.w.tot_len, D = a.map.d.tot_len;
    const so_cell c = so_locate(a.map, a.xyz[3 * (size_t)i], a.xyz[3 * (size_t)i + 1], a.xyz[3 * (size_t)i + 2]);
    if (a.sdf && j == 0 && live) {
        float v[8], wk[8];
        so_gather_sdf(a.sdf_vol, H, W, D, c, v);
        a.sdf[i] = so_trilerp_sdf(c, v, wk);
    }
    if (a.n_sem > 0 && (a.sem_logits || a.sem_argmax)) {
        const TeamCorner t = so_team_corners(c, H, W, D);
        float out[kTeamRounds], best;
        int arg;
        if (a.feat_dtype == SO_DTYPE_F32) so_team_lookup<false>(a.feat_vol, a.feat_stride, a.n_rgb, a.n_sem, t, j, out, best, arg);
        else so_team_lookup<true>(a.feat_vol, a.feat_stride, a.n_rgb, a.n_sem, t, j, out, best, arg);
        if (live && a.sem_logits) {
#pragma unroll
            for (int r = 0; r < kTeamRounds; ++r)
                if (j + kTeam * r < a.n_sem) a.sem_logits[(size_t)i * a.n_sem + j + kTeam * r] = out[r];
        }
        if (live && a.sem_argmax && j == 0) a.sem_argmax[i] = arg;
    }
}

// Backward of field_query_kernel with respect to the volume(s): the transpose of the trilinear gather.
// One lane per query point, 8 (x n_sem) float atomics; the lattice is small (640 k points at the shipped
// sizes) and neighbouring points hit neighbouring voxels, so the adds of a wave land in a few cache lines.
__global__ __launch_bounds__(256) void field_query_bwd_kernel(so_query_args a, const float *__restrict__ g_sdf,
                                                              const float *__restrict__ g_logits,
                                                              float *__restrict__ g_sdf_vol, float *__restrict__ g_feat_vol) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int H = a.map.h.tot_len, W = a.map.w.tot_len, D = a.map.d.tot_len;
    const so_cell c = so_locate(a.map, a.xyz[3 * (size_t)i], a.xyz[3 * (size_t)i + 1], a.xyz[3 * (size_t)i + 2]);
    const float fd[2] = {c.fd0, c.fd1}, fw[2] = {c.fw0, c.fw1}, fh[2] = {c.fh0, c.fh1};
    const float gs = (g_sdf && g_sdf_vol) ? g_sdf[i] : 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int h = c.h0 + (k >> 2), w = c.w0 + ((k >> 1) & 1), d = c.d0 + (k & 1);
        if (h < 0 || h >= H || w < 0 || w >= W || d < 0 || d >= D) continue;   // zero padding: no gradient
        const float wk = (fd[k & 1] * fw[(k >> 1) & 1]) * fh[k >> 2];
        const size_t vox = ((size_t)h * W + w) * D + d;
        if (g_sdf && g_sdf_vol) atomicAdd(g_sdf_vol + vox, gs * wk);
        if (g_logits && g_feat_vol)
            for (int q = 0; q < a.n_sem; ++q)
                atomicAdd(g_feat_vol + vox * a.feat_stride + a.n_rgb + q, g_logits[(size_t)i * a.n_sem + q] * wk);
    }
}

__global__ __launch_bounds__(256) void occ_resample_kernel(so_occ_args a) {
    const int n = a.n0 * a.n1 * a.n2;
    const long long gt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pt = gt / kTeam;
    const int j = (int)(gt - pt * kTeam);
    const bool live = pt < n;
    const int i = live ? (int)pt : n - 1;
    // F.grid_sample(align_corners=True) of coords * 2 - 1  (eval_iou.py:217-221)
    so_cell c;
    const float gh = ((((a.coords[3 * (size_t)i] * 2.0f) - 1.0f) + 1.0f) / 2.0f) * (float)(a.H - 1);
    const float gw = ((((a.coords[3 * (size_t)i + 1] * 2.0f) - 1.0f) + 1.0f) / 2.0f) * (float)(a.W - 1);
    const float gd = ((((a.coords[3 * (size_t)i + 2] * 2.0f) - 1.0f) + 1.0f) / 2.0f) * (float)(a.D - 1);
    const float fh = floorf(gh), fw = floorf(gw), fd = floorf(gd);
    c.h0 = (int)fh; c.w0 = (int)fw; c.d0 = (int)fd;
    c.fh1 = gh - fh; c.fh0 = (fh + 1.0f) - gh;
    c.fw1 = gw - fw; c.fw0 = (fw + 1.0f) - gw;
    c.fd1 = gd - fd; c.fd0 = (fd + 1.0f) - gd;
    float v[8], wk[8];
    so_gather_sdf(a.grid, a.H, a.W, a.D, c, v);       // (8 lanes read the same 8 values: one broadcast line each)
    const float s = so_trilerp_sdf(c, v, wk);
    int occ = a.density ? (s >= a.thresh) : (s <= a.thresh);
    const int i2 = i % a.n2, i1 = (i / a.n2) % a.n1, i0 = i / (a.n2 * a.n1);
    if (i0 < a.crop[0] || i0 >= a.n0 - a.crop[1] || i1 < a.crop[2] || i1 >= a.n1 - a.crop[3] ||
        i2 < a.crop[4] || i2 >= a.n2 - a.crop[5])
        occ = 0;
    if (live && j == 0) {
        if (a.sampled) a.sampled[i] = s;
        if (a.occ) a.occ[i] = occ;
    }
    if (a.sem && a.logits) {
        const TeamCorner t = so_team_corners(c, a.H, a.W, a.D);
        float out[kTeamRounds], best;
        int arg;
        so_team_lookup<false>(a.logits, a.C, 0, a.C, t, j, out, best, arg);
        if (live && j == 0) {
            const int cls = a.lut ? a.lut[arg] : arg;
            a.sem[i] = occ * cls;
        }
    }
}

// seen / correct / positive per class + the binary non-empty class; one u64 atomic per
// block and counter after an LDS reduction (Guideline 12)
__global__ __launch_bounds__(256) void iou_counts_kernel(const int32_t *__restrict__ pred,
                                                         const int32_t *__restrict__ target,
                                                         const uint8_t *__restrict__ mask, long long n,
                                                         const int32_t *__restrict__ cls, int n_cls,
                                                         int empty, unsigned long long *counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned int s_cnt[];  // 3 * (n_cls + 1)
    const int nc = 3 * (n_cls + 1);
    for (int k = threadIdx.x; k < nc; k += blockDim.x) s_cnt[k] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (mask && !mask[i]) continue;
        const int p = pred[i], t = target[i];
        for (int k = 0; k < n_cls; ++k) {
            const int c = cls[k];
            if (t == c) atomicAdd(&s_cnt[k], 1u);
            if (t == c && p == c) atomicAdd(&s_cnt[(n_cls + 1) + k], 1u);
            if (p == c) atomicAdd(&s_cnt[2 * (n_cls + 1) + k], 1u);
        }
        if (t != empty) atomicAdd(&s_cnt[n_cls], 1u);
        if (t != empty && p != empty) atomicAdd(&s_cnt[(n_cls + 1) + n_cls], 1u);
        if (p != empty) atomicAdd(&s_cnt[2 * (n_cls + 1) + n_cls], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nc; k += blockDim.x)
        if (s_cnt[k]) atomicAdd(&counts[k], (unsigned long long)s_cnt[k]);
}

}  // namespace

int so_validate_mapping(const so_mapping &m);

extern "C" int selfocc_field_query(const so_query_args *args, void *stream) {
    SO_REQUIRE(args != nullptr, "args is NULL");
    const so_query_args &a = *args;
    if (so_validate_mapping(a.map)) return -1;
    SO_REQUIRE(a.n >= 0, "n must be >= 0");
    if (a.n == 0) return 0;
    SO_REQUIRE(a.xyz != nullptr, "xyz is NULL");
    SO_REQUIRE(a.sdf == nullptr || a.sdf_vol != nullptr, "sdf requested but sdf_vol is NULL");
    if (a.sem_logits || a.sem_argmax) {
        SO_REQUIRE(a.n_sem > 0 && a.feat_vol != nullptr, "semantic query needs feat_vol and n_sem > 0");
        SO_REQUIRE(a.feat_stride >= a.n_rgb + a.n_sem, "feat_stride < n_rgb + n_sem");
        SO_REQUIRE(a.feat_dtype == SO_DTYPE_F32 || a.feat_dtype == SO_DTYPE_BF16, "bad feat_dtype");
    }
    SO_REQUIRE(!(a.sem_logits || a.sem_argmax) || a.n_sem <= kTeam * kTeamRounds, "field_query: n_sem = %d > %d", a.n_sem,
               kTeam * kTeamRounds);
    const long long nthreads = (long long)a.n * kTeam;
    SO_REQUIRE(nthreads < (1LL << 31) * 256, "field_query: too many points");
    hipLaunchKernelGGL(field_query_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return so_launch_status();
}

extern "C" int selfocc_field_query_bwd(const so_query_args *args, const float *g_sdf, const float *g_logits,
                                       float *g_sdf_vol, float *g_feat_vol, void *stream) {
    SO_REQUIRE(args != nullptr, "args is NULL");
    const so_query_args &a = *args;
    if (so_validate_mapping(a.map)) return -1;
    SO_REQUIRE(a.n >= 0, "n must be >= 0");
    if (a.n == 0) return 0;
    SO_REQUIRE(a.xyz != nullptr, "xyz is NULL");
    SO_REQUIRE((g_sdf == nullptr) == (g_sdf_vol == nullptr), "g_sdf and g_sdf_vol go together");
    SO_REQUIRE((g_logits == nullptr) == (g_feat_vol == nullptr), "g_logits and g_feat_vol go together");
    if (g_logits) {
        SO_REQUIRE(a.n_sem > 0 && a.feat_stride >= a.n_rgb + a.n_sem, "semantic gradient needs n_sem > 0 and a valid feat_stride");
    }
    if (!g_sdf && !g_logits) return 0;
    hipLaunchKernelGGL(field_query_bwd_kernel, dim3((a.n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, g_sdf,
                       g_logits, g_sdf_vol, g_feat_vol);
    return so_launch_status();
}

extern "C" int selfocc_occ_resample(const so_occ_args *args, void *stream) {
    SO_REQUIRE(args != nullptr, "args is NULL");
    const so_occ_args &a = *args;
    SO_REQUIRE(a.n0 >= 0 && a.n1 >= 0 && a.n2 >= 0, "negative lattice size");
    const long long n = (long long)a.n0 * a.n1 * a.n2;
    if (n == 0) return 0;
    SO_REQUIRE(n < (1LL << 31), "lattice too large");
    SO_REQUIRE(a.grid && a.coords, "grid / coords is NULL");
    SO_REQUIRE(a.H >= 2 && a.W >= 2 && a.D >= 2, "grid dims must be >= 2");
    SO_REQUIRE(a.sem == nullptr || (a.logits != nullptr && a.C >= 1), "sem requested but logits NULL / C < 1");
    SO_REQUIRE(a.sem == nullptr || a.C <= kTeam * kTeamRounds, "occ_resample: C = %d > %d classes", a.C, kTeam * kTeamRounds);
    hipLaunchKernelGGL(occ_resample_kernel, dim3((unsigned)((n * kTeam + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return so_launch_status();
}

extern "C" int selfocc_iou_counts(const int32_t *pred, const int32_t *target, const uint8_t *mask,
                                  int64_t n, const int32_t *class_indices, int32_t n_cls,
                                  int32_t empty_label, unsigned long long *counts, void *stream) {
    SO_REQUIRE(n >= 0 && n_cls >= 0 && n_cls <= 255, "bad n / n_cls");
    if (n == 0) return 0;
    SO_REQUIRE(pred && target && counts, "NULL pointer");
    SO_REQUIRE(n_cls == 0 || class_indices, "class_indices is NULL");
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;  // grid-stride; each block folds < 2^32 elements
    const size_t shm = sizeof(unsigned int) * 3 * (n_cls + 1);
    hipLaunchKernelGGL(iou_counts_kernel, dim3((unsigned)blocks), dim3(256), shm, (hipStream_t)stream, pred,
                       target, mask, (long long)n, class_indices, n_cls, empty_label, counts);
    return so_launch_status();
}
