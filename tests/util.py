"""Shared helpers for the parity tests."""
import torch


def cell_margin(mapping, rays, cfg, nears, fars, chunk=200_000, skip_first=False):
    """Per ray: the smallest distance (in voxels) between any of its sample positions and a
    voxel face, in float64.  The SDF is trilinear, so its gradient — which feeds NeuS's
    alpha through cos = d . grad — is DISCONTINUOUS across voxel faces: a sample within
    float32 rounding of a face can legitimately land on either side in two correct
    implementations.  Parity on such rays is undefined for any pair of implementations
    (reference CPU vs CUDA included); the tests exclude exactly these rays and nothing else.
    ``rays`` must be explicit (origins/dirs)."""
    S = cfg.n_samples
    out = []
    for s in range(0, rays.origins.shape[0], chunk):
        o = rays.origins[s:s + chunk].double()
        d = rays.dirs[s:s + chunk].double()
        tn, tf = nears[s:s + chunk].double(), fars[s:s + chunk].double()
        b = torch.linspace(0, 1, S + 1, dtype=torch.float64, device=o.device)
        edges = b[None] * tf[:, None] + (1 - b[None]) * tn[:, None]
        t = edges[:, :-1] if cfg.sample_pos == 0 else (edges[:, :-1] + edges[:, 1:]) / 2
        pos = o[:, None, :] + d[:, None, :] * t[..., None]
        g = mapping.meter2grid(pos)
        fr = g - torch.floor(g)
        if skip_first:
            fr = fr[:, 1:]
        out.append(torch.minimum(fr, 1 - fr).amin(dim=(1, 2)))
    return torch.cat(out)
