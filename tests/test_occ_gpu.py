"""GPU parity of the occupancy evaluation tail: bit-exact integer occupancy / semantics /
IoU counts against the torch ops the reference calls (run on CPU)."""
import math

import numpy as np
import pytest
import torch

import oracle
from oracle import torch_port as tp
from selfocc_amd import synthetic as sy
from selfocc_amd.occ import field_query, uniform_lattice, occ_resample, MeanIoU, OPENSEED2NUSCENES

pytestmark = pytest.mark.gpu
D0 = torch.device("cuda:0")


@pytest.mark.parametrize("feat_dtype", [torch.float32, torch.bfloat16])
def test_field_query_bit_exact(hip, feat_dtype):
    vol = sy.make_volume("cfg5", n_rgb=3, n_sem=21, feat_dtype=feat_dtype, seed=2)
    aabb = sy.CONFIGS["cfg5"]["aabb"]
    xyz = uniform_lattice(aabb, 0.4, "cpu", shift=True).reshape(-1, 3)[::7].contiguous()
    got = field_query(vol.to(D0), xyz.to(D0), want_sdf=True, want_logits=True, want_argmax=True)
    ref_sdf, _ = oracle.field_sdf(vol.mapping, vol.sdf, xyz, want_grad=False)
    assert torch.equal(got['sdf'].cpu(), ref_sdf)
    # forward_geonetwork h[..., 4:] = grid_sample of the (1, C, H, W, D) volume (neus_head.py:284-288)
    h = tp.field_lookup(vol.mapping, vol.to_reference_layout(), xyz)
    assert torch.equal(got['sdf'].cpu(), h[:, 0])
    assert torch.equal(got['logits'].cpu(), h[:, 4:])
    assert torch.equal(got['argmax'].cpu().long(), torch.argmax(h[:, 4:], dim=-1))


def _ego2lidar(seed):
    r = np.random.RandomState(seed)
    yaw = math.radians(r.uniform(-3, 3))
    m = np.eye(4)
    m[:2, :2] = [[math.cos(yaw), -math.sin(yaw)], [math.sin(yaw), math.cos(yaw)]]
    m[:3, 3] = r.uniform(-0.5, 0.5, 3) * [1, 1, 0.2]
    return m


@pytest.mark.parametrize("seed", [0, 1])
def test_occ3d_tail_bit_exact(hip, seed):
    """eval_iou.py:198-250 with scene_size 4 (aabb -40..40, -1..5.4, res 0.4): dense SDF query ->
    ego-frame resample -> (sdf <= thresh) -> crops -> argmax -> LUT.  Integer outputs bit-exact."""
    vol = sy.make_volume("cfg5", n_rgb=3, n_sem=21, seed=seed)
    pcr, expansion = [-40.0, -40.0, -1.0, 40.0, 40.0, 5.4], [80.0, 80.0, 6.4]
    lat = uniform_lattice(pcr, 0.4, "cpu")            # (H, W, D, 3) = (200, 200, 16, 3)
    H, W, D = lat.shape[:3]
    q = field_query(vol.to(D0), lat.reshape(-1, 3).to(D0), want_sdf=True, want_logits=True)
    sdf = q['sdf'].reshape(H, W, D)
    logits = q['logits'].reshape(H, W, D, -1)
    thresh = 0.05
    pred_occ, pred_miou, lidar_points, sampled = tp.occ_tail_port(sdf.cpu(), logits.cpu(), _ego2lidar(seed), pcr,
                                                                  expansion, thresh)
    coords = lidar_points[..., [1, 0, 2]].contiguous()   # normalised along (H<->y, W<->x, D<->z)
    got = occ_resample(sdf, coords.to(D0), thresh, logits=logits, lut=OPENSEED2NUSCENES,
                       crop=(6, 6, 6, 6, 0, 4), want_sampled=True)
    assert torch.equal(got['sampled'].cpu(), sampled)
    assert torch.equal(got['occ'].cpu(), pred_occ)
    assert torch.equal(got['sem'].cpu(), pred_miou.to(torch.int32))
    assert 0.01 < pred_occ.float().mean() < 0.9


def test_iou_counts_exact(hip):
    g = torch.Generator().manual_seed(0)
    pred = torch.randint(0, 18, (200, 200, 16), generator=g, dtype=torch.int32)
    tgt = torch.randint(0, 18, (200, 200, 16), generator=g, dtype=torch.int32)
    mask = torch.rand(200, 200, 16, generator=g) > 0.4
    cls = list(range(1, 17))
    for use_mask in (False, True):
        m = MeanIoU(cls, 0, [str(c) for c in cls], True, 0)
        m.reset()
        m._after_step(pred.to(D0), tgt.to(D0), mask.to(D0) if use_mask else None)
        m._after_step(tgt.to(D0), pred.to(D0), mask.to(D0) if use_mask else None)
        ref = tp.mean_iou_counts_port(pred, tgt, cls, 0, mask if use_mask else None) + \
            tp.mean_iou_counts_port(tgt, pred, cls, 0, mask if use_mask else None)
        assert torch.equal(m.counts.cpu(), ref)
        miou, iou = m._after_epoch()
        r = ref.double()
        assert abs(iou - (r[1, -1] / (r[0, -1] + r[2, -1] - r[1, -1])).item() * 100) < 1e-9
        exp_miou = np.mean([(r[1, i] / (r[0, i] + r[2, i] - r[1, i])).item() for i in range(16)]) * 100
        assert abs(miou - exp_miou) < 1e-9


def test_iou_counts_binary_and_empty(hip):
    m = MeanIoU([1], 0, ['occupied'], True, 0)
    m.reset()
    e = torch.zeros(0, dtype=torch.int32, device=D0)
    m._after_step(e, e)
    assert int(m.counts.sum()) == 0
    p = torch.tensor([1, 1, 0, 0, 1], dtype=torch.int32, device=D0)
    t = torch.tensor([1, 0, 0, 1, 1], dtype=torch.int32, device=D0)
    m._after_step(p, t)
    assert m.counts.cpu().tolist() == [[3, 3], [2, 2], [3, 3]]
