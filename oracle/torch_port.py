"""The SelfOcc hot path written with the torch ops the reference itself calls —
what the reference would execute on CPU.  TEST INFRASTRUCTURE ONLY (see __init__.py).

Used (a) to pin the C oracle (oracle_*.c) against real ``F.grid_sample`` /
``softmax`` / ``cumprod`` arithmetic and (b) as bench.py's ``cpu_baseline`` (kind
"port").  Each function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------
# field lookup: model/head/nerfacc_head/bev_nerf.py:97-113 (query_density / forward) with the
# (1, C, H, W, D) volume of pre_compute_density_color (:74-95)
# ---------------------------------------------------------------------------------------
def field_lookup(mapping, density_color, xyz):
    grid = mapping.meter2grid(xyz, True)
    grid = 2 * grid - 1
    grid = grid.reshape(1, -1, 1, 1, 3)
    out = F.grid_sample(density_color, grid[..., [2, 1, 0]], mode='bilinear', align_corners=True)
    return out.permute(0, 2, 3, 4, 1).flatten(0, 3)  # n, C


def sh0_color(raw):
    """SHRender with deg=0, act='relu' (model/head/utils/sh_render.py:84-91)."""
    C0 = 0.28209479177387814
    return torch.relu(C0 * raw + 0.5)


# ---------------------------------------------------------------------------------------
# upstream sdfstudio pieces (absent fork; restated from the published algorithm)
# ---------------------------------------------------------------------------------------
def aabb_collider(origins, dirs, aabb, near_plane=0.0):
    aabb = torch.as_tensor(aabb, dtype=origins.dtype, device=origins.device).reshape(2, 3)
    frac = 1.0 / (dirs + 1e-6)
    t1 = (aabb[0, 0] - origins[:, 0:1]) * frac[:, 0:1]
    t2 = (aabb[1, 0] - origins[:, 0:1]) * frac[:, 0:1]
    t3 = (aabb[0, 1] - origins[:, 1:2]) * frac[:, 1:2]
    t4 = (aabb[1, 1] - origins[:, 1:2]) * frac[:, 1:2]
    t5 = (aabb[0, 2] - origins[:, 2:3]) * frac[:, 2:3]
    t6 = (aabb[1, 2] - origins[:, 2:3]) * frac[:, 2:3]
    nears = torch.max(torch.cat([torch.minimum(t1, t2), torch.minimum(t3, t4), torch.minimum(t5, t6)], 1), 1).values
    fars = torch.min(torch.cat([torch.maximum(t1, t2), torch.maximum(t3, t4), torch.maximum(t5, t6)], 1), 1).values
    nears = torch.clamp(nears, min=near_plane)
    fars = torch.maximum(fars, nears + 1e-6)
    return nears[:, None], fars[:, None]


def uniform_bins(n_rays, n_samples, nears, fars, t_rand=None):
    bins = torch.linspace(0.0, 1.0, n_samples + 1, device=nears.device)[None, :]
    if t_rand is not None:
        if t_rand.dim() == 1:
            t_rand = t_rand[:, None]
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    return bins * fars + (1 - bins) * nears  # n_rays, S + 1


def render_port(mapping, density_color, n_rgb, n_sem, origins, dirs, dir_norm, cfg,
                t_rand=None, bkgd_rays=None, chunk=90000, return_samples=False):
    """density_color: (1, 1 + n_rgb + n_sem, H, W, D).  Chunked like NeuSHead.render
    (model/head/neus_head/neus_head.py:329-385)."""
    outs = []
    N = origins.shape[0]
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        outs.append(_render_chunk(mapping, density_color, n_rgb, n_sem, origins[s:e], dirs[s:e],
                                  dir_norm[s:e], cfg, None if t_rand is None else t_rand[s:e],
                                  None if bkgd_rays is None else bkgd_rays[s:e], return_samples))
    return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}


def _render_chunk(mapping, vol, n_rgb, n_sem, o, d, dn, cfg, t_rand, bkgd_rays, return_samples):
    S = cfg.n_samples
    nears, fars = aabb_collider(o, d, cfg.aabb, cfg.near_plane)
    edges = uniform_bins(o.shape[0], S, nears, fars, t_rand)
    starts, ends = edges[:, :-1], edges[:, 1:]
    deltas = ends - starts
    mids = (starts + ends) / 2
    if cfg.sample_pos == 0:
        pos = o[:, None, :] + d[:, None, :] * starts[..., None]
    else:
        pos = o[:, None, :] + d[:, None, :] * (starts + ends)[..., None] / 2
    pos = pos.detach().requires_grad_(True)
    with torch.enable_grad():
        h = field_lookup(mapping, vol, pos.reshape(-1, 3))
        sdf = h[:, 0].reshape(-1, S)
        grad = torch.autograd.grad(sdf.sum(), pos)[0]
    sdf = sdf.detach()
    h = h.detach()
    # NeuS get_alpha, cos_anneal_ratio = 1
    true_cos = (d[:, None, :] * grad).sum(-1)
    iter_cos = -F.relu(-true_cos)
    est_next = sdf + iter_cos * deltas * 0.5
    est_prev = sdf - iter_cos * deltas * 0.5
    prev_cdf = torch.sigmoid(est_prev * cfg.inv_s)
    next_cdf = torch.sigmoid(est_next * cfg.inv_s)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], 1), 1)
    weights = alpha * trans[:, :-1]
    acc = weights.sum(-1)
    depth = (weights * mids).sum(-1) / (acc + 1e-10)
    if cfg.depth_div_norm:
        depth = depth / dn
    out = {'depth': depth, 'acc': acc, 'nears': nears[:, 0], 'fars': fars[:, 0]}
    # head post-math, neus_head.py:366-374, 430-438
    ts = mids / dn[:, None]
    dz = deltas / dn[:, None]
    eps = torch.finfo(dz.dtype).eps
    w_ = weights.clone()
    w_[dz < eps] = 0.
    idx = (w_ / dz.clamp_min(eps)).argmax(dim=-1, keepdim=True)
    out['max_depth'] = torch.gather(ts, -1, idx).squeeze(-1)
    if n_rgb:
        col = sh0_color(h[:, 1:1 + n_rgb]).reshape(-1, S, 3)
        rgb = (weights[..., None] * col).sum(-2)
        if cfg.bkgd_mode == 1:
            rgb = rgb + torch.tensor(cfg.bkgd, device=rgb.device) * (1.0 - acc[:, None])
        elif cfg.bkgd_mode == 2:
            rgb = rgb + bkgd_rays * (1.0 - acc[:, None])
        if cfg.clamp_rgb:
            rgb = rgb.clamp(0.0, 1.0)
        out['rgb'] = rgb
    if n_sem:
        sm = torch.softmax(h[:, 1 + n_rgb:], dim=-1).reshape(-1, S, n_sem)
        out['sem'] = (weights[..., None] * sm).sum(-2)
    if return_samples:
        out.update(weights=weights, ts=ts, deltas=dz, sdf=sdf, grad=grad, starts=starts, ends=ends)
    return out


# ---------------------------------------------------------------------------------------
# mmcv.ops.multi_scale_deform_attn.multi_scale_deformable_attn_pytorch — the function the
# reference itself calls when value is not on CUDA
# (model/encoder/bevformer/attention/image_cross_attention.py:344-345).  mmcv==2.0.1 is not
# in /root/reference; restated from its published source.
# ---------------------------------------------------------------------------------------
def msda_port(value, spatial_shapes, sampling_locations, attention_weights):
    bs, _, num_heads, embed_dims = value.shape
    _, num_queries, num_heads, num_levels, num_points, _ = sampling_locations.shape
    value_list = value.split([int(H_ * W_) for H_, W_ in spatial_shapes], dim=1)
    sampling_grids = 2 * sampling_locations - 1
    sampling_value_list = []
    for level, (H_, W_) in enumerate(spatial_shapes):
        value_l_ = value_list[level].flatten(2).transpose(1, 2).reshape(bs * num_heads, embed_dims, int(H_), int(W_))
        sampling_grid_l_ = sampling_grids[:, :, :, level].transpose(1, 2).flatten(0, 1)
        sampling_value_l_ = F.grid_sample(value_l_, sampling_grid_l_, mode='bilinear', padding_mode='zeros',
                                          align_corners=False)
        sampling_value_list.append(sampling_value_l_)
    attention_weights = attention_weights.transpose(1, 2).reshape(bs * num_heads, 1, num_queries,
                                                                  num_levels * num_points)
    output = (torch.stack(sampling_value_list, dim=-2).flatten(-2) * attention_weights).sum(-1).view(
        bs, num_heads * embed_dims, num_queries)
    return output.transpose(1, 2).contiguous()


# ---------------------------------------------------------------------------------------
# Occ3D evaluation tail — eval_iou.py:152-163 (lattice), :211-250 (resample, threshold,
# crop, argmax, LUT) and utils/metric_util.py:37-64, 90-165 (LUT, MeanIoU), CPU torch ops.
# ---------------------------------------------------------------------------------------
def openseed2nuscenes(sem):
    lut = torch.tensor([1, 2, 3, 4, 5, 5, 6, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 15, 15, 16, 0],
                       dtype=sem.dtype, device=sem.device)
    return lut[sem.flatten()].reshape(*sem.shape)


def occ3d_lattice():
    xx = torch.linspace(-40.0, 40.0, 200)
    yy = torch.linspace(-40.0, 40.0, 200)
    zz = torch.linspace(-1.0, 5.4, 16)
    return torch.stack([xx[:, None, None].expand(-1, 200, 16), yy[None, :, None].expand(200, -1, 16),
                        zz[None, None, :].expand(200, 200, -1), torch.ones(200, 200, 16)], dim=-1)


def occ_tail_port(sdf, logits, ego2lidar, point_cloud_range, expansion, thresh, xyz=None):
    """sdf (H, W, D); logits (H, W, D, C) or None; returns pred_occ, pred_occ_miou, lidar_points."""
    xyz = occ3d_lattice() if xyz is None else xyz
    n0, n1, n2 = xyz.shape[:3]
    ego2lidar = xyz.new_tensor(ego2lidar)
    lidar_points = torch.matmul(ego2lidar.unsqueeze(0), xyz.reshape(-1, 4, 1)).squeeze(-1)[:, :3]
    lidar_points[:, 0] = (lidar_points[:, 0] - point_cloud_range[0]) / expansion[0]
    lidar_points[:, 1] = (lidar_points[:, 1] - point_cloud_range[1]) / expansion[1]
    lidar_points[:, 2] = (lidar_points[:, 2] - point_cloud_range[2]) / expansion[2]
    lidar_points = lidar_points.reshape(1, n0, n1, n2, 3)
    sampled_sdf = F.grid_sample(sdf[None, None, ...], lidar_points[..., [2, 0, 1]] * 2 - 1, mode='bilinear',
                                align_corners=True)
    pred_occ = (sampled_sdf.squeeze(0).squeeze(0) <= thresh).to(torch.int)
    pred_occ[..., 12:] = 0
    pred_occ[:6, ...] = 0
    pred_occ[-6:, ...] = 0
    pred_occ[:, :6, :] = 0
    pred_occ[:, -6:, :] = 0
    pred_miou = None
    if logits is not None:
        sem = logits.permute(3, 0, 1, 2)
        sampled_sem = F.grid_sample(sem[None, ...], lidar_points[..., [2, 0, 1]] * 2 - 1, mode='bilinear',
                                    align_corners=True)
        sampled_sem = torch.argmax(sampled_sem, dim=1).squeeze(0)
        pred_miou = pred_occ * openseed2nuscenes(sampled_sem)
    return pred_occ, pred_miou, lidar_points[0], sampled_sdf[0, 0]


def mean_iou_counts_port(outputs, targets, class_indices, empty_label, mask=None):
    """MeanIoU._after_step (utils/metric_util.py:108-121) -> (3, n_cls + 1) int64"""
    if mask is not None:
        outputs, targets = outputs[mask], targets[mask]
    n = len(class_indices)
    c = torch.zeros(3, n + 1, dtype=torch.int64)
    for i, k in enumerate(class_indices):
        c[0, i] = torch.sum(targets == k).item()
        c[1, i] = torch.sum((targets == k) & (outputs == k)).item()
        c[2, i] = torch.sum(outputs == k).item()
    c[0, -1] = torch.sum(targets != empty_label).item()
    c[1, -1] = torch.sum((targets != empty_label) & (outputs != empty_label)).item()
    c[2, -1] = torch.sum(outputs != empty_label).item()
    return c


# ---------------------------------------------------------------------------------------
# Differentiable restatement for the BACKWARD parity tests.  Stock PyTorch cannot
# double-backward 3-D grid_sample (the reference needs the external cuda_gridsample_grad2
# for that, docs/installation.md:30), so the lookup is written as an explicit 8-corner
# trilinear (value identical to F.grid_sample(align_corners=True): tested) whose analytic
# metre gradient is itself differentiable wrt the volume.  Run in float64.
# ---------------------------------------------------------------------------------------
def trilinear_explicit(mapping, vol_chw, xyz):
    """vol_chw (C, H, W, D); xyz (n, 3) -> values (n, C), d value[:, 0] / d xyz (n, 3)."""
    C, H, W, D = vol_chw.shape
    g = mapping.meter2grid(xyz)                       # (n, 3) h, w, d (un-normalised)
    with torch.no_grad():
        eps = 1e-3
        gp = mapping.meter2grid(xyz + eps)
        slope = (gp - g) / eps                         # piece-wise constant d grid / d metre (y->h, x->w, z->d)
        slope = torch.stack([slope[:, 1], slope[:, 0], slope[:, 2]], -1)   # metre order x, y, z
    g0 = torch.floor(g)
    fr = g - g0
    i0 = g0.long()
    vals = 0
    dval = [0, 0, 0]
    for kh in (0, 1):
        for kw in (0, 1):
            for kd in (0, 1):
                h, w, d = i0[:, 0] + kh, i0[:, 1] + kw, i0[:, 2] + kd
                inb = (h >= 0) & (h < H) & (w >= 0) & (w < W) & (d >= 0) & (d < D)
                v = vol_chw[:, h.clamp(0, H - 1), w.clamp(0, W - 1), d.clamp(0, D - 1)].T * inb[:, None]
                fh = fr[:, 0] if kh else 1 - fr[:, 0]
                fw = fr[:, 1] if kw else 1 - fr[:, 1]
                fd = fr[:, 2] if kd else 1 - fr[:, 2]
                vals = vals + v * (fh * fw * fd)[:, None]
                s = v[:, 0]
                dval[0] = dval[0] + s * (1 if kw else -1) * fh * fd   # d / d grid w  (metre x)
                dval[1] = dval[1] + s * (1 if kh else -1) * fw * fd   # d / d grid h  (metre y)
                dval[2] = dval[2] + s * (1 if kd else -1) * fh * fw   # d / d grid d  (metre z)
    grad = torch.stack(dval, -1) * torch.stack([slope[:, 0], slope[:, 1], slope[:, 2]], -1)
    return vals, grad


def render_port_differentiable(mapping, vol_chw, n_rgb, n_sem, o, d, dn, cfg, inv_s, t_rand=None, bkgd_rays=None):
    """Same algorithm as _render_chunk, every op differentiable wrt vol_chw and inv_s."""
    S = cfg.n_samples
    nears, fars = aabb_collider(o, d, cfg.aabb, cfg.near_plane)
    bins = torch.linspace(0.0, 1.0, S + 1, dtype=o.dtype)[None, :]
    if t_rand is not None:
        tr = t_rand[:, None] if t_rand.dim() == 1 else t_rand
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * tr
    edges = bins * fars + (1 - bins) * nears
    starts, ends = edges[:, :-1], edges[:, 1:]
    deltas = ends - starts
    mids = (starts + ends) / 2
    pos = o[:, None, :] + d[:, None, :] * (starts if cfg.sample_pos == 0 else mids)[..., None]
    h, grad = trilinear_explicit(mapping, vol_chw, pos.reshape(-1, 3))
    sdf = h[:, 0].reshape(-1, S)
    grad = grad.reshape(-1, S, 3)
    true_cos = (d[:, None, :] * grad).sum(-1)
    iter_cos = -F.relu(-true_cos)
    prev_cdf = torch.sigmoid((sdf - iter_cos * deltas * 0.5) * inv_s)
    next_cdf = torch.sigmoid((sdf + iter_cos * deltas * 0.5) * inv_s)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], 1), 1)
    weights = alpha * trans[:, :-1]
    acc = weights.sum(-1)
    depth = (weights * mids).sum(-1) / (acc + 1e-10)
    if cfg.depth_div_norm:
        depth = depth / dn
    out = dict(depth=depth, acc=acc, weights=weights, sdf=sdf, grad=grad, nears=nears[:, 0], fars=fars[:, 0],
               starts=starts, ends=ends)
    if n_rgb:
        col = sh0_color(h[:, 1:1 + n_rgb]).reshape(-1, S, 3)
        rgb = (weights[..., None] * col).sum(-2)
        if cfg.bkgd_mode == 1:
            rgb = rgb + torch.tensor(cfg.bkgd, dtype=o.dtype) * (1.0 - acc[:, None])
        elif cfg.bkgd_mode == 2:
            rgb = rgb + bkgd_rays * (1.0 - acc[:, None])
        if cfg.clamp_rgb:
            rgb = rgb.clamp(0.0, 1.0)
        out['rgb'] = rgb
    if n_sem:
        sm = torch.softmax(h[:, 1 + n_rgb:], dim=-1).reshape(-1, S, n_sem)
        out['sem'] = (weights[..., None] * sm).sum(-2)
    return out


# ---------------------------------------------------------------------------------------
# Per-camera sampling part of ReprojLossMonoMultiNewCombine.reproj_loss
# (loss/reproj_loss_mono_multi_new_combine.py:108-201, 223-225), torch ops, one camera.
# The full loss (incl. SSIM / auto-mask) is pinned against the imported reference class via
# tests/golden; this function is the oracle of the fused kernel's three outputs.
# ---------------------------------------------------------------------------------------
def reproj_sample_port(weights, ts, deltas, pix, curr_rgb, T_prev, T_next, img_prev, img_next, img_h, img_w):
    R, S = weights.shape
    ray_idx = torch.arange(R).unsqueeze(-1).repeat(1, S).flatten()
    weight, t = weights.flatten(), ts.flatten()
    rays = pix[ray_idx]
    if deltas is not None:
        delta = deltas.flatten().detach()
        eps = torch.finfo(delta.dtype).eps
        weight = weight.clone()
        weight[delta < eps] = 0.
        weight = weight / delta.clamp_min(eps)
    pixel_coords = torch.ones((1, 1, len(rays), 4), dtype=weights.dtype)
    pixel_coords[..., :2] = rays.reshape(1, 1, -1, 2)
    pixel_coords[..., :3] *= t.reshape(1, 1, -1, 1)
    pixel_coords = pixel_coords.unsqueeze(-1)

    def cal_pixel(trans, coords):
        pixel = torch.matmul(trans.reshape(1, 1, 1, 4, 4), coords).squeeze(-1)
        mask = pixel[..., 2] > 0
        pixel = pixel[..., :2] / torch.maximum(torch.ones_like(pixel[..., :1]) * 1e-5, pixel[..., 2:3])
        mask = mask & (pixel[..., 0] > 0) & (pixel[..., 0] < img_w) & (pixel[..., 1] > 0) & (pixel[..., 1] < img_h)
        return pixel, mask

    def sample_pixel(pixel, img):
        pixel = pixel.clone()
        pixel[..., 0] /= img_w
        pixel[..., 1] /= img_h
        pixel = 2 * pixel - 1
        rgb = F.grid_sample(img[None], pixel, mode='bilinear', padding_mode='border', align_corners=True)
        return rgb.reshape(1, 1, 3, rgb.shape[-1]).permute(0, 1, 3, 2)

    pixel_prev, prev_mask = cal_pixel(T_prev, pixel_coords)
    pixel_next, next_mask = cal_pixel(T_next, pixel_coords)
    rgb_prev = sample_pixel(pixel_prev, img_prev)
    rgb_next = sample_pixel(pixel_next, img_next)
    rgb_curr_ = curr_rgb[ray_idx].reshape(1, 1, -1, 3)
    diff_prev = torch.mean(torch.abs(rgb_curr_ - rgb_prev), dim=-1)
    diff_next = torch.mean(torch.abs(rgb_curr_ - rgb_next), dim=-1)
    diff_prev[~prev_mask] = 0.
    diff_next[~next_mask] = 0.
    cnt = prev_mask.to(torch.float) + next_mask.to(torch.float)
    general_mask = cnt > 0
    cnt = torch.clamp(cnt, 1.0).to(weights.dtype)
    diff = (diff_prev + diff_next) / cnt
    weight = weight.clone()
    weight[~general_mask.flatten()] = 0.
    weight_sum = torch.zeros(R, dtype=weight.dtype)
    weight_sum.index_add_(-1, ray_idx, weight)
    weight_sum = weight_sum.clamp_min(torch.finfo(torch.float32).eps)
    weight = weight / torch.gather(weight_sum, -1, ray_idx)
    l1 = torch.zeros(R, dtype=diff.dtype)
    l1 = l1.index_add(-1, ray_idx, weight * diff.flatten())
    rgb_prev = rgb_prev * prev_mask[..., None]
    rgb_next = rgb_next * next_mask[..., None]
    comb_ = (rgb_prev + rgb_next) / cnt.unsqueeze(-1)
    comb = torch.zeros(R, 3, dtype=comb_.dtype).index_add(0, ray_idx, comb_.reshape(-1, 3) * weight.unsqueeze(-1))
    ray_filter = torch.zeros(R, dtype=weight.dtype).index_add(0, ray_idx, general_mask.flatten().to(weight.dtype))
    return l1, comb, (ray_filter > 0).to(weights.dtype)
