"""Glue between the module API and the HIP kernels (forward orchestration; autograd lives here)."""
import torch

from sunerf_hip import ops


def mlp_points(model, x: torch.Tensor) -> torch.Tensor:
    """NeRF.forward on arbitrary query points (M, 4) -> (M, d_out) (model.py:44-57).

    Runs the fused render kernel with one two-sample "ray" per point (o = 0, d = xyz, z = 1 => o + d*z = xyz
    exactly) and returns the raw MLP output of the first sample."""
    flat = x.reshape(-1, 4)
    m = flat.shape[0]
    o = torch.zeros(m, 3, dtype=torch.float32, device=flat.device)
    z = torch.ones(m, 2, dtype=torch.float32, device=flat.device)
    out = ops.emission_render_fwd(model.packed(), o, flat[:, :3].contiguous(), flat[:, 3].contiguous(), z,
                                  reg_radius=0., want_raw=True)
    return out['raw'][:, 0, :]
