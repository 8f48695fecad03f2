/*
 * selfocc_hip.h — C ABI of the MI355X-native SelfOcc hot path (libselfocc_hip.so).
 *
 * Every entry point is the drop-in replacement of one native/third-party op the
 * reference (huang-yh/SelfOcc @ 2024-10-08) calls on its hot path.  Citations are
 * relative to the reference tree.  Conventions shared by all entry points:
 *
 *   - plain pointers + sizes only; all data pointers are DEVICE pointers (HBM);
 *     the *_args structs themselves live in HOST memory and are read during the call;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call
 *     is stream-ordered, does not synchronise, does not allocate, keeps no state;
 *   - the caller allocates all outputs; output pointers that are NULL are skipped;
 *   - return value: 0 = launched, <0 = argument error (see selfocc_last_error()),
 *     >0 = the hipError_t of a failed launch;
 *   - thread-safe / re-entrant (the last-error string is thread-local).
 */
#ifndef SELFOCC_HIP_H
#define SELFOCC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SELFOCC_ABI_VERSION 32

int selfocc_abi_version(void);
const char *selfocc_last_error(void);

/* ------------------------------------------------------------------------------------
 * Grid <-> metre mapping.  Replaces GridMeterMapping / LinearMapping.meter2grid
 * (model/encoder/bevformer/mappings.py:97-150).  One piece-wise linear axis:
 *     c = m - start;  a = |c|;
 *     g_abs = size1 == 0 ? a / range0 * size0
 *           : a > range0 ? size0 + (a - range0) / range1 * size1 : a / range0 * size0;
 *     g = (sign(c) * g_abs + off0) + off1          (off0/off1 = size0/size1 unless *_half)
 * Axis order everywhere: h <-> metre y, w <-> metre x, d <-> metre z.
 * ---------------------------------------------------------------------------------- */
typedef struct so_axis {
    float size0, size1;   /* cells in the inner / outer segment                        */
    float range0, range1; /* metres covered by the inner / outer segment               */
    float off0, off1;     /* grid offset of the axis centre (0,0 for *_half and for d) */
    float start;          /* metre coordinate of grid 0 on the d axis, 0 for h / w     */
    int32_t tot_len;      /* number of grid points along the axis                      */
} so_axis;

typedef struct so_mapping {
    so_axis h, w, d;
} so_mapping;

/* ------------------------------------------------------------------------------------
 * SDF volume rendering.  Replaces the sdfstudio-fork NeuSCustomModel.__call__ that
 * NeuSHead drives (model/head/neus_head/neus_head.py:353,394,531) together with the
 * head's own post-math (:366-374, 430-438, 571-587) and, in pixel-grid mode, RaySampler
 * 'fixed'/'cellular' + Img2LiDAR (model/head/nerfacc_head/ray_sampler.py:23-68,
 * img2lidar.py:58-69).
 *
 * Volume layout in HBM (ours; the reference materialises (1, C, H, W, D)):
 *     sdf_vol  [H][W][D]               float32
 *     feat_vol [H][W][D][feat_stride]  float32 or bf16 (raw colour coefficients then
 *                                      semantic logits), feat_stride >= n_rgb + n_sem
 * ---------------------------------------------------------------------------------- */
enum {
    SO_RAYS_EXPLICIT = 0, /* origins / dirs / dir_norm arrays, one entry per ray       */
    SO_RAYS_PIXEL_GRID = 1 /* rays generated in-kernel from img2lidar + a pixel lattice */
};
enum { SO_SAMPLE_AT_START = 0, SO_SAMPLE_AT_MID = 1 };
enum { SO_BKGD_NONE = 0, SO_BKGD_CONST = 1, SO_BKGD_PER_RAY = 2 };
enum { SO_JITTER_NONE = 0, SO_JITTER_SINGLE = 1, SO_JITTER_PER_BIN = 2 };
enum {
    SO_FLAG_DEPTH_DIV_NORM = 1, /* depth /= ||K^-1 (u,v,1)|| (z-depth, as the fork does) */
    SO_FLAG_CLAMP_RGB = 2,      /* eval: clamp rgb to [0,1]                              */
    SO_FLAG_EXACT = 4,          /* canonical IEEE operation order (bit-exact with oracle/): slower.
                                   Default is the fast path: per-ray affine grid coordinates (within
                                   ~1.5 ulp of the canonical divide chain; a sample closer than that
                                   to a voxel face re-derives its cell canonically, so the SAME cell
                                   as the canonical path is used at every interpolated sample),
                                   hardware exp2 / rcp, a cancellation-free form of the NeuS alpha,
                                   exact free-space skipping (see sdf_brick).  Parity of the fast
                                   path with the canonical one is stated and measured in DESIGN.md
                                   section 4 / tests/test_render_gpu.py (whole benchmarked frame:
                                   depth within 1e-4 relative on every ray that accumulates > 0.05). */
    SO_FLAG_NO_SKIP = 8,        /* fast path: do not skip saturated free-space samples (A/B switch) */
    SO_FLAG_RAY_PER_LANE = 64,  /* accepted and IGNORED since ABI 30 (rounds 2 - 4: per-sample launches through ray-per-lane
                                   kernels, an A/B switch nobody shipped; per-sample outputs always come from the
                                   sample-parallel training kernel)                                  */
    SO_FLAG_NO_AHEAD = 32,      /* fast path: general march even where the code-ahead skip marcher applies (A/B) */
    SO_FLAG_NO_FACE_SAFE = 16   /* fast path: never re-derive cells near voxel faces: ~6 % faster on
                                   the SDF-only kernel, but ~1e-4 of the rays (those with a sample
                                   within an ulp of a face, where the trilinear GRADIENT jumps) may
                                   then differ from the canonical result by more than 1e-4          */
};
enum { SO_DTYPE_F32 = 0, SO_DTYPE_BF16 = 1 };

typedef struct so_render_args {
    /* --- field ------------------------------------------------------------------- */
    so_mapping map;
    const float *sdf_vol;
    const void *feat_vol; /* NULL when n_rgb + n_sem == 0 */
    int32_t feat_dtype;   /* SO_DTYPE_* */
    int32_t feat_stride;
    int32_t n_rgb;        /* 0 or 3 (SH degree 0: rgb = relu(C0 * raw + 0.5))          */
    int32_t n_sem;        /* semantic classes; per-sample softmax, weight-composited    */
    /* --- rays -------------------------------------------------------------------- */
    int32_t ray_mode;     /* SO_RAYS_* */
    int32_t n_rays;       /* explicit: number of rays; pixel grid: n_cams * ny * nx     */
    const float *origins;   /* (n_rays, 3)                                              */
    const float *dirs;      /* (n_rays, 3) unit length                                  */
    const float *dir_norm;  /* (n_rays)  norm of the un-normalised direction            */
    const float *img2lidar; /* (n_cams, 4, 4) row-major pixel*depth -> world            */
    int32_t n_cams, nx, ny;
    float sx, sy, ox, oy;   /* pixel (u, v) = (ix * sx + ox, iy * sy + oy)              */
    /* --- sampling ---------------------------------------------------------------- */
    float aabb[6];          /* xmin ymin zmin xmax ymax zmax (box collider)             */
    float near_plane;
    int32_t n_samples;
    int32_t sample_pos;     /* SO_SAMPLE_AT_* : where the field is evaluated            */
    int32_t jitter_mode;    /* SO_JITTER_*                                              */
    const float *t_rand;    /* (n_rays) or (n_rays, n_samples + 1) uniform [0,1)        */
    /* --- NeuS -------------------------------------------------------------------- */
    float inv_s;
    /* --- compositing ------------------------------------------------------------- */
    int32_t bkgd_mode;
    float bkgd[3];
    const float *bkgd_rays; /* (n_rays, 3) */
    int32_t flags;
    /* --- per-ray outputs --------------------------------------------------------- */
    float *depth;     /* (n_rays)                                                      */
    float *acc;       /* (n_rays)                                                      */
    float *rgb;       /* (n_rays, 3)                                                   */
    float *sem;       /* (n_rays, n_sem)                                               */
    float *max_depth; /* (n_rays)  ts[argmax_s w / delta]   (neus_head.py:430-438)     */
    float *nears;     /* (n_rays)                                                      */
    float *fars;      /* (n_rays)                                                      */
    /* --- per-sample outputs (training API, neus_head.py:567-577, 640) ------------- */
    float *weights;   /* (n_rays, n_samples)                                           */
    float *ts;        /* (n_rays, n_samples)  mid-point / dir_norm                     */
    float *deltas;    /* (n_rays, n_samples)  (end - start) / dir_norm                 */
    float *sdf;       /* (n_rays, n_samples)                                           */
    float *grad;      /* (n_rays, n_samples, 3)  d sdf / d (x, y, z) in metres         */
    /* --- optional workspace --------------------------------------------------------- */
    float *sdf_brick; /* scratch of H*W*D*33 bytes (16-B aligned) or NULL.  When given, the fast
                         path first re-packs sdf_vol so that the 8 corners of every cell are one
                         32-B record (2 x 16-B loads per sample instead of 4 x 8-B gathers),
                         followed by one "free-space skip" byte per cell: the largest ray step for
                         which every sample inside the cell has both NeuS sigmoids saturated to
                         exactly 1.0f (alpha is then the constant 1e-5 / (1 + 1e-5) in the
                         canonical float32 order too), so SDF-only per-ray launches composite
                         such samples without interpolating.  The re-pack kernel is launched by
                         selfocc_render_fwd on the same stream before the march.              */
    const float *inv_s_dev; /* optional DEVICE pointer to inv_s (1 float).  When non-NULL the kernels read
                         inv_s from it and ignore the host value above: a training loop whose inv_s is
                         a learnable parameter (exp(10 * variance), neus_head.py:631-633) never has to
                         read it back to the host (no stream sync, nothing to go stale).       */
} so_render_args;

int selfocc_render_fwd(const so_render_args *args, void *stream);

/* Backward of selfocc_render_fwd with respect to the volume(s) and inv_s.  Ray
 * geometry carries no gradient (the reference's rays come from constant matrices).
 * Upstream gradients that are NULL count as zero.  g_sdf_vol / g_feat_vol must be
 * zero-initialised by the caller (atomically accumulated). */
typedef struct so_render_bwd_args {
    so_render_args fwd;       /* same inputs as the forward call (outputs ignored)     */
    const float *g_depth;     /* (n_rays)                                              */
    const float *g_acc;       /* (n_rays)                                              */
    const float *g_rgb;       /* (n_rays, 3)                                           */
    const float *g_sem;       /* (n_rays, n_sem)                                       */
    const float *g_weights;   /* (n_rays, n_samples)                                   */
    const float *g_sdf;       /* (n_rays, n_samples)                                   */
    const float *g_grad;      /* (n_rays, n_samples, 3)                                */
    float *g_sdf_vol;         /* [H][W][D]                                             */
    float *g_feat_vol;        /* [H][W][D][feat_stride] float32                        */
    float *g_inv_s;           /* (1)                                                   */
    /* Optional scratch for the BRICK-BINNED volume-gradient scatter (round 4).  NULL: every sample adds its
     * 8 corner rows to g_sdf_vol / g_feat_vol with device-scope float atomics (on MI355X each one is a write
     * through the fabric: 3.1 GB of them for a 165 MB gradient at the nuscenes_occ training shape).  Given
     * (>= selfocc_render_bwd_ws_bytes() bytes, 256-B aligned): a counting pre-pass gives every sample a slot in
     * brick order (bricks of 4 x 4 x 8 CELLS); the ray kernel writes one record per sample at its slot (cell,
     * fractions, d L / d feature row, SDF coefficients: 32 / 64 / 128 B at 0 / 3-8 / 20-24 channels); one
     * workgroup per (brick, <= chunk samples) streams its records, sums them into the brick's 5 x 5 x 9-voxel
     * tile in LDS — a DOUBLE tile: ds_add_f64 issues ~12 x faster than ds_add_f32 on gfx950 — and adds the
     * tile's non-zero rows to the gradient once.  A sample whose cell lies outside the volume contributes
     * nothing in either mode.  Same sums, different (still unspecified) float addition order.              */
    void *scatter_ws;
    uint64_t scatter_ws_bytes;
} so_render_bwd_args;

int selfocc_render_bwd(const so_render_bwd_args *args, void *stream);
/* bytes of so_render_bwd_args::scatter_ws for this call.  0 = the call is outside the binned path's range — pass
 * NULL then (the atomic path runs): args == NULL, n_rays <= 0 or n_samples <= 0, an axis longer than 1021 grid
 * points, n_rays * n_samples >= 2^31, or a channel count other than 0 / 3 / 8 / 20 / 24. */
size_t selfocc_render_bwd_ws_bytes(const so_render_bwd_args *args);

/* ------------------------------------------------------------------------------------
 * Multi-scale deformable attention.  Replaces mmcv==2.0.1
 * MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index,
 * sampling_locations, attention_weights, im2col_step) at the reference call sites
 * model/encoder/bevformer/attention/image_cross_attention.py:340-342 and
 * model/encoder/tpvformer/attention/cross_view_hybrid_attention.py:111-113.
 *
 *   value   (bs, nv, heads, d)            float32
 *   shapes  (L, 2) int32 [H_l, W_l]       starts (L) int32
 *   loc     (bs, nq, heads, L, P, 2)      (x, y) in [0,1]
 *   attw    (bs, nq, heads, L, P)
 *   out     (bs, nq, heads * d)
 * ---------------------------------------------------------------------------------- */
int selfocc_msda_fwd(const float *value, const int32_t *shapes, const int32_t *starts,
                     const float *loc, const float *attw, float *out,
                     int32_t bs, int32_t nv, int32_t nq, int32_t heads, int32_t d,
                     int32_t L, int32_t P, void *stream);

/* Layout of `value` (and of `g_value`) for the fused / camera-loop entry points — the ones this repo's own encoder
 * modules call, so the layout is theirs to choose; selfocc_msda_fwd / _bwd / _bwd_banded keep mmcv's.
 *   SO_VALUE_PIXEL_MAJOR (bs, nv, heads, d): mmcv.  A 128-byte cache line holds one pixel of TWO heads.
 *   SO_VALUE_HEAD_MAJOR  (bs, heads, nv, d): a line holds two horizontally adjacent pixels of ONE head, i.e. usually
 *   both x-corners of a bilinear footprint; the gathers of the hw-plane cross-attention run 0.34 ms instead of 0.50. */
enum { SO_VALUE_PIXEL_MAJOR = 0, SO_VALUE_HEAD_MAJOR = 1 };

/* value_dtype of the same entry points: SO_DTYPE_F32, or SO_DTYPE_BF16 = bfloat16 STORAGE of `value` (the arithmetic
 * is float32 on the exactly widened values; g_value stays float32): halves the 64-byte corner segments the gathers move.
 * Results equal the float32 kernels run on bf16-rounded values; against unrounded float32 values the relative error is
 * the bf16 rounding of `value`, ~2^-9 (opt-in: BASELINE configs[1] allows bf16 storage, the reference computes MSDA in f32). */

/* Inference form with the reference's prologue fused in (softmax over the L*P logits of a
 * (query, head); loc = ref + off / (W_l, H_l); image_cross_attention.py:314-328,
 * cross_view_hybrid_attention.py:88-99): the sampling_locations / attention_weights tensors are
 * never materialised.   off_raw (bs,nq,heads,L,P,2)  logits (bs,nq,heads,L*P)
 *   ref_kind 0: ref (bs,nq,L,2)   1: ref (bs,nq,P,2)   2: ref (bs,nq,L,P,2)        L*P <= 256 */
int selfocc_msda_fused_fwd(const void *value, const int32_t *shapes, const int32_t *starts,
                           const float *ref, int32_t ref_kind, const float *off_raw, const float *logits,
                           float *out, int32_t bs, int32_t nv, int32_t nq, int32_t heads, int32_t d,
                           int32_t L, int32_t P, int32_t value_layout, int32_t value_dtype, int32_t ol_stride, void *stream);
/* ol_stride (ABI 32; the four fused / camera-loop entry points): floats between consecutive QUERY rows of off_raw and logits
 * (and of g_off / g_logits in the backward forms).  0 = the dense tensors described above.  3 * heads * L * P (or more, even)
 * = ONE row per query [heads*L*P*2 raw offsets | heads*L*P logits], i.e. the output of the `sampling_offsets` and
 * `attention_weights` Linears computed as ONE projection with the two weights stacked (image_cross_attention.py:296-312
 * reads the same `query` twice); `logits` = `off_raw` + 2 * heads * L * P then, and the backward forms write the gradient
 * of that merged row, which is what ONE input-gradient and ONE weight-gradient pass of the stacked Linear consume.
 * g_value_stride (ABI 32; the two backward forms): 0 = g_value has the layout of `value`.  > 0 = g_value is written PIXEL-major
 * into rows of that many floats — (bs | cams, nv) rows, this op's heads * d channels starting at the pointer —, i.e. straight
 * into a column block of the row-major gradient of the (stacked) value projection: no head-major -> row-major transposing copy
 * before its weight- / input-gradient passes.  The rows must be zero-initialised by the caller like g_value. */

/* Camera-loop inference form: BEVCrossAttention's re-batch -> offset / weight linears -> MSDA ->
 * scatter-add -> divide-by-count (bevformer/attention/image_cross_attention.py:90-136) as ONE launch.
 * The offsets and logits depend on the query only, so they are given once per query and each
 * (query, head) loops over the cameras that see it (vis != 0), in camera order:
 *   out[q] = sum_{cam: vis[cam][q]} msda(value[cam], ref[cam][q] + off[q] / (W_l, H_l), softmax(logits[q]))
 *            / max(#visible cams, 1)
 *   value (cams,nv,heads,d)  ref (cams,nq,P,2)  vis (cams,nq) u8  off_raw (nq,heads,L,P,2)
 *   logits (nq,heads,L*P)  out (nq,heads*d)            batch size 1 (as the reference's masks), L*P <= 256
 * value_stride: floats between consecutive pixels of `value` (0 = dense, heads*d): lets `value` be a column block
 * of a wider matrix, e.g. the three TPV planes' value projections computed by ONE GEMM with N = 3 * heads * d. */
int selfocc_msda_cross_fwd(const void *value, const int32_t *shapes, const int32_t *starts,
                           const float *ref, const uint8_t *vis, const float *off_raw, const float *logits,
                           float *out, int32_t cams, int32_t nv, int32_t nq, int32_t heads, int32_t d,
                           int32_t L, int32_t P, int32_t value_stride, int32_t value_layout, int32_t value_dtype,
                           int32_t ol_stride, void *stream);

/* Training counterpart of selfocc_msda_cross_fwd: g_out (nq, heads*d) is the gradient of the camera MEAN;
 * returns g_value (cams,nv,heads,d; zero-initialised by the caller), g_off (nq,heads,L,P,2) and
 * g_logits (nq,heads,L*P) — sums over the visible cameras / count.  host_shapes and workspace as for
 * selfocc_msda_bwd_banded with bs = cams (selfocc_msda_bwd_banded_workspace(cams, nq, heads, L, P));
 * requires selfocc_msda_banded_supported(host_shapes, cams, nq, heads, d, L, P) == 1 and L*P <= 256. */
int selfocc_msda_cross_bwd(const void *value, const int32_t *shapes, const int32_t *starts,
                           const int32_t *host_shapes, const float *ref, const uint8_t *vis,
                           const float *off_raw, const float *logits, const float *g_out,
                           float *g_value, float *g_off, float *g_logits, int32_t cams, int32_t nv,
                           int32_t nq, int32_t heads, int32_t d, int32_t L, int32_t P, int32_t value_layout,
                           int32_t value_dtype, int32_t ol_stride, int32_t g_value_stride, void *workspace, size_t workspace_bytes,
                           void *stream);

/* g_value must be zero-initialised by the caller (atomically accumulated). */
int selfocc_msda_bwd(const float *value, const int32_t *shapes, const int32_t *starts,
                     const float *loc, const float *attw, const float *g_out,
                     float *g_value, float *g_loc, float *g_attw,
                     int32_t bs, int32_t nv, int32_t nq, int32_t heads, int32_t d,
                     int32_t L, int32_t P, void *stream);

/* Banded (output-stationary) backward, same results as selfocc_msda_bwd up to summation order
 * (grad_value is accumulated in double precision and rounded once per list segment).  The sampling points are
 * counting-sorted by (batch, head, level, band of rows); a block owns one band of one level's map in LDS and
 * adds the points of its list with ds_add_f64 — no global atomics in the scatter, no key scan per band.
 * Needs a HOST copy of `shapes` (L, 2) for the work decomposition (L <= 8) and a 16-byte aligned device
 * workspace of selfocc_msda_bwd_banded_workspace(...) bytes (26 bytes per sampling point: a 2-byte row key, a
 * 16-byte record, two 4-byte list slots; plus 28 KB of counters per (batch, head, level); contents undefined
 * before and after).  g_value must be zero-initialised by the caller.  Falls back to selfocc_msda_bwd when a
 * level is wider than the LDS tile, needs more than 1024 bands, or the call has >= 2^30 sampling points. */
size_t selfocc_msda_bwd_banded_workspace(int32_t bs, int32_t nq, int32_t heads, int32_t L, int32_t P);
int selfocc_msda_bwd_banded(const float *value, const int32_t *shapes, const int32_t *starts,
                            const int32_t *host_shapes, const float *loc, const float *attw,
                            const float *g_out, float *g_value, float *g_loc, float *g_attw,
                            int32_t bs, int32_t nv, int32_t nq, int32_t heads, int32_t d,
                            int32_t L, int32_t P, void *workspace, size_t workspace_bytes, void *stream);

/* Training counterpart of selfocc_msda_fused_fwd: gradients w.r.t. value and the RAW linear outputs
 *   g_off (bs,nq,heads,L,P,2) = (d out / d loc) / (W_l, H_l)
 *   g_logits (bs,nq,heads,L*P) = aw (g_aw - sum aw g_aw)               (softmax backward)
 * with the softmax / sampling locations recomputed in registers (the forward saves nothing but its
 * inputs) and grad_value through the banded LDS-f64 scatter.  host_shapes / workspace as for
 * selfocc_msda_bwd_banded; requires selfocc_msda_banded_supported(...) == 1 and L*P <= 256.
 * g_value must be zero-initialised by the caller. */
int selfocc_msda_banded_supported(const int32_t *host_shapes, int32_t bs, int32_t nq, int32_t heads,
                                  int32_t d, int32_t L, int32_t P);
int selfocc_msda_fused_bwd(const void *value, const int32_t *shapes, const int32_t *starts,
                           const int32_t *host_shapes, const float *ref, int32_t ref_kind,
                           const float *off_raw, const float *logits, const float *g_out,
                           float *g_value, float *g_off, float *g_logits, int32_t bs, int32_t nv,
                           int32_t nq, int32_t heads, int32_t d, int32_t L, int32_t P, int32_t value_layout,
                           int32_t value_dtype, int32_t ol_stride, int32_t g_value_stride, void *workspace, size_t workspace_bytes,
                           void *stream);

/* ------------------------------------------------------------------------------------
 * Dense SDF / semantic query on a regular metre lattice + Occ3D occupancy tail.
 * Replaces NeuSHead.get_uniform_sdf -> field.forward_geonetwork / forward_sdfnetwork
 * (model/head/neus_head/neus_head.py:265-293) and the resample / threshold / argmax /
 * LUT of eval_iou.py:211-250.
 * ---------------------------------------------------------------------------------- */
typedef struct so_query_args {
    so_mapping map;
    const float *sdf_vol;
    const void *feat_vol;
    int32_t feat_dtype, feat_stride, n_rgb, n_sem;
    const float *xyz;     /* (n, 3) metre positions                                    */
    int32_t n;
    float *sdf;           /* (n)                                                       */
    float *sem_logits;    /* (n, n_sem) raw logits (forward_geonetwork h[..., 4:])     */
    int32_t *sem_argmax;  /* (n)                                                       */
} so_query_args;

int selfocc_field_query(const so_query_args *args, void *stream);

/* Backward of selfocc_field_query with respect to the volume(s) (the reference differentiates
 * get_uniform_sdf through F.grid_sample: model/head/neus_head/neus_head.py:532-538 feeds
 * `uniform_sdf` to SoftSparsityLoss, loss/sparsity_loss.py:67-81):
 *   g_sdf (n) -> g_sdf_vol [H][W][D];  g_logits (n, n_sem) -> g_feat_vol [H][W][D][feat_stride] float32
 * (channels n_rgb .. n_rgb + n_sem - 1).  Either pair may be NULL.  The volume gradients are
 * ACCUMULATED (atomic adds): zero-initialise them.  Outputs of `args` (sdf, sem_*) are ignored. */
int selfocc_field_query_bwd(const so_query_args *args, const float *g_sdf, const float *g_logits,
                            float *g_sdf_vol, float *g_feat_vol, void *stream);

/* ------------------------------------------------------------------------------------
 * Tri-plane -> dense volume: the head's pre_compute_density_color(representation)
 * (sdfstudio-fork SDFCustomField, driven from model/head/neus_head/neus_head.py:295-306;
 * in-repo analogue BEVNeRF, model/head/nerfacc_head/bev_nerf.py:74-95), forward only:
 *   x[h,w,d,:] = hw[h,w,:] + zh[d,h,:] + wz[w,d,:]                        hw (H*W, C)  zh (D*H, C)  wz (W*D, C)
 *   out = Linear_out(Softplus(Linear_hidden(Softplus(x))))  (n_hidden = 1; n_hidden = 0 drops the hidden layer)
 *   sdf[h,w,d] = out[0];  feat[h,w,d,0..out_dim-2] = out[1..], channels up to feat_stride zero-filled.
 * Weights in torch.nn.Linear layout: w_hidden (C, C), w_out (out_dim, C).  C in {64, 96, 128},
 * out_dim <= 32, feat_stride <= 31.  One fused MFMA-f32 kernel (exact float32 arithmetic); the
 * H*W*D x C intermediate is never materialised. */
int selfocc_field_volume_fwd(const float *hw, const float *zh, const float *wz, int32_t H, int32_t W,
                             int32_t D, int32_t C, const float *w_hidden, const float *b_hidden,
                             int32_t n_hidden, const float *w_out, const float *b_out, int32_t out_dim,
                             float *sdf, void *feat, int32_t feat_dtype, int32_t feat_stride, void *stream);

/* Backward of selfocc_field_volume_fwd (C = 96, one hidden layer, float32 feature volume): g_sdf (H*W*D) and
 * g_feat (H*W*D, feat_stride) -> gradients of the three planes and of the two linear layers (all seven
 * outputs zero-initialised by the caller, accumulated).  Nothing of the forward is needed: the kernel
 * recomputes the activations per 32-voxel tile.  g_sdf / g_feat may be NULL (treated as zero). */
int selfocc_field_volume_bwd(const float *hw, const float *zh, const float *wz, int32_t H, int32_t W,
                             int32_t D, int32_t C, const float *w_hidden, const float *b_hidden,
                             const float *w_out, int32_t out_dim, const float *g_sdf, const float *g_feat,
                             int32_t feat_stride, float *g_hw, float *g_zh, float *g_wz, float *g_w_hidden,
                             float *g_b_hidden, float *g_w_out, float *g_b_out, void *stream);

/* Occ3D evaluation tail, eval_iou.py:211-250: trilinear resample (F.grid_sample,
 * align_corners=True, zero padding — bit-exact with torch's CPU kernel) of the dense SDF
 * grid (+ optional semantic logits) at the ego-frame Occ3D lattice, threshold, border crop,
 * argmax + openseed2nuscenes LUT (utils/metric_util.py:37-64), occupancy * semantics.
 *   grid    [H][W][D] float32           logits [H][W][D][C] float32 (NULL: no semantics)
 *   coords  (n, 3) normalised [0,1] along (H, W, D) of `grid`; n = n0 * n1 * n2 lattice
 *   crop    {lo0, hi0, lo1, hi1, lo2, hi2}: output index i_k outside [lo_k, n_k - hi_k) -> 0
 *   density != 0 selects (value >= thresh) (eval_iou.py:205-206,227) instead of (<=). */
typedef struct so_occ_args {
    const float *grid;
    const float *logits;
    int32_t H, W, D, C;
    const float *coords;
    int32_t n0, n1, n2;
    int32_t crop[6];
    float thresh;
    int32_t density;
    const int32_t *lut; /* (C) class LUT applied to the argmax, NULL = identity */
    float *sampled;     /* (n) resampled scalar, optional                        */
    int32_t *occ;       /* (n) 0 / 1                                             */
    int32_t *sem;       /* (n) occ * lut[argmax_c logits], optional              */
} so_occ_args;

int selfocc_occ_resample(const so_occ_args *args, void *stream);

/* Integer confusion counts of MeanIoU._after_step (utils/metric_util.py:90-121):
 * counts (3, n_cls + 1) int64 rows = seen / correct / positive; last column = the
 * binary "non-empty" class.  mask may be NULL.  counts are accumulated (+=). */
int selfocc_iou_counts(const int32_t *pred, const int32_t *target, const uint8_t *mask,
                       int64_t n, const int32_t *class_indices, int32_t n_cls,
                       int32_t empty_label, unsigned long long *counts, void *stream);

/* The eikonal regulariser (/root/reference/loss/eikonal_loss.py:19-22): sum over the n rows of grad (n, 3) float32 of
 * (||grad_i||_2 - 1)^2 as selfocc_eikonal_partials(n) per-block partial sums (the caller adds them: a fixed order, so the
 * result is deterministic; the mean is that sum / n), and its gradient g_grad_i = 2 * scale[0] * (||grad_i|| - 1) grad_i /
 * ||grad_i|| (0 where the norm is 0, as torch's norm backward), `scale` a 1-element DEVICE tensor (upstream gradient / n:
 * no host read-back).  Returns 0 / SELFOCC_ERR_*. */
int selfocc_eikonal_partials(int64_t n);
int selfocc_eikonal_fwd(const float *grad, float *partial, int64_t n, void *stream);
int selfocc_eikonal_bwd(const float *grad, const float *scale, float *g_grad, int64_t n, void *stream);

/* `y = identity + dropout(x)` of the attention / FFN outputs (mmcv: `self.dropout(output) + identity`,
 * /root/reference/model/encoder/bevformer/attention/image_cross_attention.py:137-139,
 * tpvformer/attention/cross_view_hybrid_attention.py:119-124) in one pass, and its backward g_x = keep ? g / (1 - p) : 0.
 * Element i is kept iff a counter-based hash of (seed, i) maps to >= p (keep probability 1 - p, as torch's dropout; not
 * torch's Philox stream), so no mask is stored: pass the SAME seed to the backward call.  n elements of float32,
 * 16-byte aligned pointers; y may alias x or identity. */
int selfocc_dropout_add_fwd(const float *x, const float *identity, float *y, int64_t n, float p, uint64_t seed, void *stream);
int selfocc_dropout_bwd(const float *g, float *g_x, int64_t n, float p, uint64_t seed, void *stream);

/* Compact second differences of the SDF volume (H, W, D) along h, w, d — NeuSHead's `second_grad`, the input of
 * SecondGradLoss (/root/reference/loss/second_grad_loss.py:6-19; this repo's declared compact form):
 * out = concat over the axes of ((s[2:] - 2 s[1:-1]) + s[:-2]).flatten(), selfocc_second_diff_size(H, W, D) floats (0: bad
 * shape), bit-identical to the torch expression; backward: g_sdf (H, W, D) = the transposed stencil applied to g_out, gather
 * form (deterministic, overwrites g_sdf).  Returns 0 / SELFOCC_ERR_*. */
size_t selfocc_second_diff_size(int32_t H, int32_t W, int32_t D);
int selfocc_second_diff_fwd(const float *sdf, float *out, int32_t H, int32_t W, int32_t D, void *stream);
int selfocc_second_diff_bwd(const float *g_out, float *g_sdf, int32_t H, int32_t W, int32_t D, void *stream);

/* ------------------------------------------------------------------------------------
 * SSIM term of the photometric losses (class SSIM, loss/reproj_loss_mono_multi_new_combine.py:26-66;
 * also loss/rgb_loss_ms.py): reflection pad 1, 3x3 means, out = clamp((1 - SSIM) / 2, 0, 1), (N, C, H, W).
 * x / y are addressed through element strides (n, c, h, w) — the call sites pass channel-last views; out,
 * g_out, g_x, g_y are contiguous.  g_x or g_y may be NULL.  H, W >= 2. */
int selfocc_ssim_fwd(const float *x, const float *y, const int64_t *x_strides, const int64_t *y_strides,
                     int32_t N, int32_t C, int32_t H, int32_t W, float *out, void *stream);
int selfocc_ssim_bwd(const float *x, const float *y, const int64_t *x_strides, const int64_t *y_strides,
                     int32_t N, int32_t C, int32_t H, int32_t W, const float *g_out, float *g_x, float *g_y,
                     void *stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the last dimension of a contiguous (rows, C) float32 tensor — the `norm` steps of
 * TPVFormerLayer / BEVFormerLayer (mmcv build_norm_layer(dict(type='LN')) in the reference:
 * model/encoder/tpvformer/tpvformer_encoder_layer.py, bevformer_encoder_layer.py), biased variance,
 * y = (x - mean) / sqrt(var + eps) * gamma + beta.  C a multiple of 4 in [4, 128].
 * mean / rstd (rows) are optional outputs of the forward (both or neither) and inputs of the backward.
 * The backward writes dx (rows, C), dgamma (C), dbeta (C) (overwritten, deterministic) and needs
 * selfocc_layernorm_bwd_workspace(rows, C) bytes of device scratch.
 * ---------------------------------------------------------------------------------- */
int selfocc_layernorm_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                          float *rstd, int64_t rows, int32_t C, float eps, void *stream);
size_t selfocc_layernorm_bwd_workspace(int64_t rows, int32_t C);
int selfocc_layernorm_bwd(const float *x, const float *gamma, const float *mean, const float *rstd,
                          const float *dy, float *dx, float *dgamma, float *dbeta, int64_t rows, int32_t C,
                          void *workspace, size_t workspace_bytes, void *stream);

/* The image features as the encoders' `value`: n_levels maps (B, N, C, h_l, w_l) float32 -> out (N, sum h_l w_l, B, C) with
 * out[n][start_l + p][b][c] = (feats[l][b][n][c][p] + cams_embeds[n][c]) + level_embeds[l][c]
 * (/root/reference/model/encoder/tpvformer/tpvformer_encoder.py:261-277, bevformer/bevformer_encoder.py:194-210: two
 * broadcast adds per level + cat + permute).  `feats`: HOST array of n_levels device pointers; host_hw[l] = h_l * w_l.
 * 1 <= n_levels <= 8, C <= 512.  Returns 0 / SELFOCC_ERR_*. */
int selfocc_flatten_feats(const float *const *feats, const int32_t *host_hw, int32_t n_levels, int32_t B, int32_t N, int32_t C,
                          const float *cams_embeds, const float *level_embeds, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * point_sampling of the TPV / BEV encoders (model/encoder/bevformer/utils.py:114-170): project the pillar reference
 * points into every camera.
 *   ref (B, D, Q, 3) metres; lidar2img (B, N, 4, 4) row-major; focal_x / focal_y (N) or NULL (the reference's
 *   optional `focal_ratios_*` metas); img_h / img_w = metas['img_shape'][:2]
 *   cam (N, B, Q, D, 2): (u / img_w, v / img_h) (x focal ratios), u = c0 / max(c2, 1e-5), c = lidar2img (x, y, z, 1)
 *   mask (N, B, Q, D) u8: c2 > 1e-5 and 0 < u, v < 1 (before the focal ratios, as in the reference)
 *   visible (N, B, Q) u8 or NULL: mask.any(-1)
 * The image-augmentation branch (post_rots / post_trans) stays host-side torch.
 * ---------------------------------------------------------------------------------- */
int selfocc_point_sampling(const float *ref, const float *lidar2img, const float *focal_x, const float *focal_y,
                           float *cam, uint8_t *mask, uint8_t *visible, int32_t B, int32_t D, int32_t Q, int32_t N,
                           float img_h, float img_w, void *stream);

/* ------------------------------------------------------------------------------------
 * Weight and bias gradient of a Linear layer with very many rows (the encoder's projections: 66 k - 180 k rows,
 * K = 96 / 192 inputs, N = 96 .. 2304 outputs; torch autograd through mmcv / nn.Linear in the reference, e.g.
 * model/encoder/tpvformer/attention/image_cross_attention.py:130-160) in one pass over dy and x:
 *     dw (N, K) = dy (T, N)^T x (T, K)          db (N) = column sums of dy   (db may be NULL)
 * float32 (MFMA f32 = exact fmaf chains), deterministic (fixed summation order).  K must be 32, 64, 96, 128 or 192
 * (selfocc_linear_wgrad_supported); workspace = selfocc_linear_wgrad_workspace(T, N, K) bytes of device scratch.
 * ---------------------------------------------------------------------------------- */
int selfocc_linear_wgrad_supported(int64_t T, int32_t N, int32_t K);
size_t selfocc_linear_wgrad_workspace(int64_t T, int32_t N, int32_t K);
int selfocc_linear_wgrad(const float *dy, const float *x, float *dw, float *db, int64_t T, int32_t N, int32_t K,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Forward of the same tall-skinny projections with the elementwise steps that follow them in TPVFormerLayer /
 * BEVFormerLayer folded into the epilogue (mmcv Linear / FFN / build_norm_layer(dict(type='LN')) under torch in the
 * reference: model/encoder/bevformer/attention/image_cross_attention.py:130-136 `output_proj` + `dropout(slots) +
 * residual`, mmcv FFN `identity + layers(x)`, tpvformer/tpvformer_encoder_layer.py:150-219 `norm`):
 *     y (T, N; row stride ldy) = LN?( relu?( x (T, K) w (N, K)^T + bias ) + residual (T, N; row stride ldr) )
 *   bias, residual: NULL = absent.  flags: SO_LINEAR_RELU (applied before the residual, as FFN does).
 *   ln_gamma / ln_beta (N) non-NULL: LayerNorm over the N outputs (biased variance, eps = ln_eps), N <= 96; then the
 *   optional outputs y_pre (T, N: the LayerNorm input), mean / rstd (T) are what selfocc_layernorm_bwd consumes.
 * float32 (MFMA f32 = exact fmaf chains; only the summation order differs from a BLAS).  K must be 32, 64, 96, 128
 * or 192 (selfocc_linear_fwd_supported).
 * ---------------------------------------------------------------------------------- */
enum { SO_LINEAR_RELU = 1 };
int selfocc_linear_fwd_supported(int64_t T, int32_t N, int32_t K);
int selfocc_linear_fwd(const float *x, const float *w, const float *bias, const float *residual, int32_t ldr,
                       const float *ln_gamma, const float *ln_beta, float ln_eps, float *y, int32_t ldy, float *y_pre,
                       float *mean, float *rstd, int64_t T, int32_t N, int32_t K, uint32_t flags, void *stream);
/* The `value_proj` of the deformable attentions (image_cross_attention.py:262-266, mmcv MultiScaleDeformableAttention)
 * with the output written HEAD-MAJOR, the layout the MSDA kernels gather fastest from (SO_VALUE_HEAD_MAJOR), by the
 * projection itself:  x (T, K) = T / nv batch items (cameras) of nv pixels;  N = G * 96 output columns = G groups (one
 * attention module each: e.g. the three TPV planes' value_proj stacked) of 6 heads x 16 channels;
 *     y (G, T / nv, 6, nv, 16):   y[g][b][h][pix][c] = relu?(x[b * nv + pix] . w[96 g + 16 h + c] + bias[...])
 * nv >= 16, T a multiple of nv, T * N < 2^31, y 16-byte aligned (written as float4).  flags: SO_LINEAR_RELU. */
int selfocc_linear_fwd_heads(const float *x, const float *w, const float *bias, float *y, int64_t T, int32_t N, int32_t K,
                             int32_t nv, uint32_t flags, void *stream);

/* Input gradient of a tall Linear: dx (T, K) = dy (T, N) W (N, K), W as nn.Linear stores it (out_features N x in_features K).
 * Replaces the `grad_output @ weight` GEMM torch's autograd runs for every nn.Linear of the encoder
 * (/root/reference/model/encoder/tpvformer/tpvformer_encoder.py:257-291 -> mmcv FFN / MSDeformableAttention projections,
 * /root/reference/model/encoder/tpvformer/attention/image_cross_attention.py:296-345); same bf16 three-way split as
 * selfocc_linear_fwd (float32-level accuracy); W is split and transposed into `workspace` first (one tiny launch), the
 * reduction over N then runs in 32-wide steps with dy streamed once.
 * supported: N % 8 == 0, 8 <= N <= 4096; K in {96, 192, 288, 384}.  Returns 0 / SELFOCC_ERR_*. */
int selfocc_linear_dgrad_supported(int64_t T, int32_t N, int32_t K);
size_t selfocc_linear_dgrad_workspace(int32_t N, int32_t K);   /* bytes: the three bf16 planes of W^T, 16-byte aligned */
int selfocc_linear_dgrad(const float *dy, const float *w, float *dx, int64_t T, int32_t N, int32_t K, void *workspace,
                         int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Fused temporal reprojection photometric term.  Replaces the per-sample part of
 * ReprojLossMonoMultiNewCombine.reproj_loss (loss/reproj_loss_mono_multi_new_combine.py
 * :108-187 and the weighted colour composite :190-201) for one camera.
 *   weights, ts (R, S); pix (R, 2) pixel (u, v); T_prev / T_next (4,4) row-major;
 *   img_* (3, Hi, Wi) planar float32; curr_rgb (R, 3) = bilinear(curr image, pix)
 * outputs per ray: l1 (R) = sum_s w' * diff, rgb_combine (R, 3), any_valid (R) in {0,1}
 * (w' = masked, per-ray renormalised weights).
 * ---------------------------------------------------------------------------------- */
typedef struct so_reproj_args {
    const float *weights, *ts, *deltas; /* deltas may be NULL (:111-116)               */
    const float *pix, *curr_rgb;
    const float *T_prev, *T_next;
    const float *img_prev, *img_next;
    int32_t R, S, Hi, Wi;
    float img_h, img_w;                 /* self.img_size used for masks + normalisation */
    float *l1, *rgb_combine, *any_valid;
    float *wnorm;                       /* (R, S) renormalised weights, saved for bwd   */
} so_reproj_args;

int selfocc_reproj_fwd(const so_reproj_args *args, void *stream);

/* d loss / d weights given d loss / d l1 and d loss / d rgb_combine. */
int selfocc_reproj_bwd(const so_reproj_args *args, const float *g_l1,
                       const float *g_rgb_combine, float *g_weights, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SELFOCC_HIP_H */
