"""Grid <-> metre mapping, host side.

Mirrors the interface of the reference's ``GridMeterMapping`` / ``LinearMapping`` /
``NonLinearMapping`` (model/encoder/bevformer/mappings.py:4-288): same constructor
arguments, ``size_h/size_w/size_d``, ``grid2meter(grid)`` and
``meter2grid(meter, normalize=False)``.  ``to_abi()`` produces the ``so_mapping`` the
HIP kernels consume (piece-wise linear only — every shipped config uses
``nonlinear_mode='linear'``).
"""
import torch

from . import abi


def _seg_fwd(a, size, rng):
    """|metre| -> |grid| for one piece-wise linear axis (mappings.py:101-110)."""
    if size[1] == 0:
        return a / rng[0] * size[0]
    return torch.where(a > rng[0], size[0] + (a - rng[0]) / rng[1] * size[1], a / rng[0] * size[0])


def _seg_inv(g, size, rng):
    """|grid| -> |metre| (mappings.py:52-60)."""
    if size[1] == 0:
        return g / size[0] * rng[0]
    return torch.where(g > size[0], rng[0] + (g - size[0]) / size[1] * rng[1], g / size[0] * rng[0])


class LinearMapping:
    def __init__(self, h_size=[128, 32], h_range=[51.2, 28.8], h_half=False,
                 w_size=[128, 32], w_range=[51.2, 28.8], w_half=False,
                 d_size=[20, 10], d_range=[-4.0, 4.0, 12.0]):
        self.h_size, self.h_range, self.h_half = list(h_size), list(h_range), h_half
        self.w_size, self.w_range, self.w_half = list(w_size), list(w_range), w_half
        self.d_size = list(d_size)
        self.d_range = [d_range[1] - d_range[0], d_range[2] - d_range[1]]
        self.d_start = d_range[0]
        self.h_tot_len = 1 + (1 if h_half else 2) * (h_size[0] + h_size[1])
        self.w_tot_len = 1 + (1 if w_half else 2) * (w_size[0] + w_size[1])
        self.d_tot_len = 1 + d_size[0] + d_size[1]

    def grid2meter(self, grid):
        h, w = grid[..., 0], grid[..., 1]
        h_ctr = h if self.h_half else h - (self.h_size[0] + self.h_size[1])
        w_ctr = w if self.w_half else w - (self.w_size[0] + self.w_size[1])
        y = torch.sign(h_ctr) * _seg_inv(torch.abs(h_ctr), self.h_size, self.h_range)
        x = torch.sign(w_ctr) * _seg_inv(torch.abs(w_ctr), self.w_size, self.w_range)
        if grid.shape[-1] == 3:
            d = grid[..., 2]
            z = torch.sign(d) * _seg_inv(torch.abs(d), self.d_size, self.d_range) + self.d_start
            return torch.stack([x, y, z], dim=-1)
        return torch.stack([x, y], dim=-1)

    def meter2grid(self, meter, normalize=False):
        x, y, z = meter[..., 0], meter[..., 1], meter[..., 2]
        w = torch.sign(x) * _seg_fwd(torch.abs(x), self.w_size, self.w_range)
        if not self.w_half:
            w = w + self.w_size[0] + self.w_size[1]
        h = torch.sign(y) * _seg_fwd(torch.abs(y), self.h_size, self.h_range)
        if not self.h_half:
            h = h + self.h_size[0] + self.h_size[1]
        zc = z - self.d_start
        d = torch.sign(zc) * _seg_fwd(torch.abs(zc), self.d_size, self.d_range)
        if normalize:
            h = h / (self.h_tot_len - 1)
            w = w / (self.w_tot_len - 1)
            d = d / (self.d_tot_len - 1)
        return torch.stack([h, w, d], dim=-1)

    def to_abi(self):
        def axis(size, rng, half, start, tot):
            a = abi.SoAxis()
            a.size0, a.size1 = float(size[0]), float(size[1])
            a.range0, a.range1 = float(rng[0]), float(rng[1])
            a.off0, a.off1 = (0.0, 0.0) if half else (float(size[0]), float(size[1]))
            a.start, a.tot_len = float(start), int(tot)
            return a
        m = abi.SoMapping()
        m.h = axis(self.h_size, self.h_range, self.h_half, 0.0, self.h_tot_len)
        m.w = axis(self.w_size, self.w_range, self.w_half, 0.0, self.w_tot_len)
        m.d = axis(self.d_size, self.d_range, True, self.d_start, self.d_tot_len)
        return m


class NonLinearMapping:
    """'linear_upscale': uniform inner cells, arithmetically growing outer cells
    (mappings.py:199-288).  Host-side only (reference points of the encoder)."""

    def __init__(self, bev_inner=128, bev_outer=32, range_inner=51.2, range_outer=51.2,
                 z_inner=20, z_outer=10, z_ranges=[-5.0, 3.0, 11.0]):
        self.bev_inner, self.bev_outer = bev_inner, bev_outer
        self.range_inner, self.range_outer = range_inner, range_outer
        self.z_inner, self.z_outer, self.z_ranges = z_inner, z_outer, z_ranges
        self.bev_size = 1 + 2 * (bev_inner + bev_outer)
        self.z_size = 1 + z_inner + z_outer
        self.hw_unit = range_inner * 1.0 / bev_inner
        self.increase_unit = (range_outer - bev_outer * self.hw_unit) * 2.0 / bev_outer / (bev_outer + 1)
        self.z_unit = (z_ranges[1] - z_ranges[0]) * 1.0 / z_inner
        self.z_increase_unit = (z_ranges[2] - z_ranges[1] - z_outer * self.z_unit) * 2.0 / z_outer / (z_outer + 1)

    @staticmethod
    def _outer_fwd(outer, inc):
        k = torch.floor(outer)
        return k * (k + 1) / 2.0 * inc + (outer - k) * (k + 1) * inc

    @staticmethod
    def _outer_inv(m_outer, unit, inc):
        c = 0.5 + unit / inc
        k = torch.floor(torch.sqrt(c ** 2 + 2 * m_outer / inc) - c)
        resi = m_outer - k * unit - inc * k * (k + 1) / 2
        return k + resi / (unit + (k + 1) * inc)

    def grid2meter(self, grid):
        hw = grid[..., :2]
        ctr = hw - (self.bev_inner + self.bev_outer)
        a = torch.abs(ctr)
        yx = torch.sign(ctr) * (a * self.hw_unit + self._outer_fwd(torch.relu(a - self.bev_inner), self.increase_unit))
        if grid.shape[-1] == 3:
            d = grid[..., 2:3]
            z = d * self.z_unit + self._outer_fwd(torch.relu(d - self.z_inner), self.z_increase_unit) + self.z_ranges[0]
            return torch.cat([yx[..., 1:2], yx[..., 0:1], z], dim=-1)
        return yx[..., [1, 0]]

    def meter2grid(self, meter, normalize=False):
        xy = meter[..., :2]
        a = torch.abs(xy)
        base = (a / self.hw_unit).clamp(max=self.bev_inner)
        wh = torch.sign(xy) * (base + self._outer_inv(torch.relu(a - self.range_inner), self.hw_unit, self.increase_unit))
        wh = wh + self.bev_inner + self.bev_outer
        za = meter[..., 2:3] - self.z_ranges[0]
        d = (za / self.z_unit).clamp(max=self.z_inner) + self._outer_inv(
            torch.relu(za - (self.z_ranges[1] - self.z_ranges[0])), self.z_unit, self.z_increase_unit)
        if normalize:
            wh = wh / (self.bev_size - 1)
            d = d / (self.z_size - 1)
        return torch.cat([wh[..., 1:2], wh[..., 0:1], d], dim=-1)

    def to_abi(self):
        raise NotImplementedError(
            "the HIP kernels implement nonlinear_mode='linear' only (every shipped SelfOcc config); "
            "'linear_upscale' is host-side only")


class GridMeterMapping:
    def __init__(self, nonlinear_mode='linear_upscale', h_size=[128, 32], h_range=[51.2, 28.8],
                 h_half=False, w_size=[128, 32], w_range=[51.2, 28.8], w_half=False,
                 d_size=[20, 10], d_range=[-4.0, 4.0, 12.0]):
        self.nonlinear_mode = nonlinear_mode
        if nonlinear_mode == 'linear_upscale':
            assert list(h_size) == list(w_size) and list(h_range) == list(w_range)
            assert (not h_half) and (not w_half)
            self.mapping = NonLinearMapping(h_size[0], h_size[1], h_range[0], h_range[1],
                                            d_size[0], d_size[1], d_range)
            self.size_h = self.size_w = self.mapping.bev_size
            self.size_d = self.mapping.z_size
        elif nonlinear_mode == 'linear':
            self.mapping = LinearMapping(h_size, h_range, h_half, w_size, w_range, w_half, d_size, d_range)
            self.size_h, self.size_w, self.size_d = (
                self.mapping.h_tot_len, self.mapping.w_tot_len, self.mapping.d_tot_len)
        else:
            raise ValueError(f"unknown nonlinear_mode {nonlinear_mode!r}")
        self.grid2meter = self.mapping.grid2meter
        self.meter2grid = self.mapping.meter2grid

    def to_abi(self):
        return self.mapping.to_abi()
