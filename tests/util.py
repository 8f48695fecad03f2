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


def seeded_fill(module, seed):
    """Overwrite every parameter of ``module`` with seeded values, in sorted-name order (identical on the reference
    classes and ours: their state-dict keys are the same).  Lets a fixture carry gradients of a 1.8 M-parameter
    encoder without carrying the parameters: both sides regenerate them from the seed (checksums are stored)."""
    import math
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in sorted(module.named_parameters()):
            r = torch.randn(p.shape, generator=g)
            if 'norm' in n:
                v = 1 + 0.1 * r if n.endswith('weight') else 0.1 * r
            elif n.endswith('sampling_offsets.bias'):
                v = 1.5 * r                      # pixels (the shipped init is a +-P pixel fan)
            elif n.endswith('sampling_offsets.weight'):
                v = 0.03 * r
            elif p.dim() >= 2 and n.endswith('weight'):
                v = r / math.sqrt(p.shape[-1])
            elif n.endswith('bias'):
                v = 0.05 * r
            else:
                v = 0.5 * r                      # query planes
            p.copy_(v.to(p.device))


def grad_digest(t):
    """What a fixture stores of a gradient: the tensor itself up to 20 k elements, else every 8th row + all row norms
    (rows = the leading dimension of a matrix, everything but the last dimension otherwise)."""
    t = t.detach().cpu()
    if t.numel() <= 20000:
        return {'full': t}
    rows = t if t.dim() == 2 else t.reshape(-1, t.shape[-1])
    return {'rows8': rows[::8].contiguous(), 'rownorm': rows.norm(dim=1)}
