"""Synthetic observer rays on the device (input side of the path).

Restates the geometry of the reference's ``sunerf/data/ray_sampling.py:11-35`` (helioprojective pixel grid ->
unit directions rotated by the camera pose) and ``sunerf/train/coordinate_transformation.py:36-54``
(``pose_spherical``) without sunpy/astropy.  Plain device tensor ops: this is input plumbing for benchmarks and
the smoke test, not part of the timed path.
"""
import math

import torch


def pose_spherical(theta: float, phi: float, radius: float, device=None) -> torch.Tensor:
    ct, st, cp, sp = math.cos(theta), math.sin(theta), math.cos(phi), math.sin(phi)
    trans_t = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=torch.float32)
    rot_phi = torch.tensor([[1, 0, 0, 0], [0, cp, -sp, 0], [0, sp, cp, 0], [0, 0, 0, 1]], dtype=torch.float32)
    rot_theta = torch.tensor([[ct, 0, -st, 0], [0, 1, 0, 0], [st, 0, ct, 0], [0, 0, 0, 1]], dtype=torch.float32)
    return (rot_theta @ (rot_phi @ trans_t)).to(device)


def observer_rays(resolution: int, row_start: int = 0, row_end: int = None, theta: float = -0.3, phi: float = 0.1,
                  radius: float = 215.032, fov_half_rad: float = 1.1 * 960. / 206264.806, device='cuda'):
    """Rays of image rows [row_start, row_end) of a resolution x resolution frame: (n, 3) origins, directions."""
    row_end = resolution if row_end is None else row_end
    c2w = pose_spherical(theta, phi, radius, device)
    lin = torch.linspace(-fov_half_rad, fov_half_rad, resolution, dtype=torch.float64, device=device)
    Ty, Tx = torch.meshgrid(lin[row_start:row_end], lin, indexing='ij')
    directions = torch.stack([torch.sin(Tx), -torch.sin(Ty) * torch.cos(Tx), -torch.cos(Tx) * torch.cos(Ty)], -1)
    directions = directions.to(torch.float32)
    rays_d = torch.sum(directions[..., None, :] * c2w[:3, :3], dim=-1).reshape(-1, 3).contiguous()
    rays_o = c2w[:3, -1].expand(rays_d.shape).contiguous()
    return rays_o, rays_d
