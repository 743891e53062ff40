"""Input side of the render path: observer poses and rays, and the full-frame render driver.

``pose_spherical`` restates ``sunerf/train/coordinate_transformation.py:36-54`` (host, 4x4, same torch ops);
``observer_rays`` / ``grid_rays`` replace ``get_rays`` (``sunerf/data/ray_sampling.py:7-36``) with the device kernel
``sunerf_observer_rays``; ``render_frame`` replaces the ``torch.split`` / ``ThreadPoolExecutor`` batch loops of
``SuNeRFLoader.render_observer_image`` (``sunerf/evaluation/loader.py:63-108, 159-242``): rays are generated tile by tile
on the device, every tile's outputs land in their slice of preallocated frame tensors, nothing touches the host until the
caller asks for it.
"""
import ctypes
import math
from typing import Dict, Optional, Sequence

import torch

from . import lib as _l
from .ops import _ptr, _stream


def pose_spherical(theta: float, phi: float, radius: float, shift: Optional[Sequence[float]] = None) -> torch.Tensor:
    """Camera-to-world 4x4 (fp32, host) of an observer at longitude ``-theta``, latitude ``phi``, distance ``radius``;
    coordinate_transformation.py:36-54 including the axis permutation of :50 and the optional ``shift``."""
    f = torch.float32
    c2w = torch.eye(4, dtype=f)
    c2w = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=f) @ c2w
    c2w = torch.tensor([[1, 0, 0, 0], [0, math.cos(phi), -math.sin(phi), 0], [0, math.sin(phi), math.cos(phi), 0],
                        [0, 0, 0, 1]], dtype=f) @ c2w
    c2w = torch.tensor([[math.cos(theta), 0, -math.sin(theta), 0], [0, 1, 0, 0], [math.sin(theta), 0, math.cos(theta), 0],
                        [0, 0, 0, 1]], dtype=f) @ c2w
    c2w = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=f) @ c2w
    if shift is not None:
        tx, ty, tz = shift
        c2w = torch.tensor([[1, 0, 0, tx], [0, 1, 0, ty], [0, 0, 1, tz], [0, 0, 0, 1]], dtype=f) @ c2w
    return c2w


def grid_rays(tx: torch.Tensor, ty: torch.Tensor, c2w: torch.Tensor, pix_begin: int = 0, n_pix: Optional[int] = None,
              time: Optional[float] = None):
    """Rays of pixels ``[pix_begin, pix_begin + n_pix)`` of a frame.

    ``tx`` / ``ty`` (float64, on the ROCm device): either the two axes of a regular helioprojective grid -- ``tx`` (W,)
    column angles, ``ty`` (H,) row angles -- or per-pixel angles of identical shape (H, W) (a real WCS).
    Returns ``rays_o (n,3), rays_d (n,3)`` and, if ``time`` is given, ``times (n,1)``."""
    if not tx.is_cuda or tx.dtype != torch.float64 or ty.dtype != torch.float64 or ty.device != tx.device:
        raise _l.SunerfHipError('grid_rays: tx / ty must be float64 tensors on one ROCm device (there is no CPU path)')
    per_pixel = tx.dim() == 2
    if per_pixel:
        if tx.shape != ty.shape:
            raise ValueError('per-pixel tx / ty must have the same (H, W) shape')
        height, width = tx.shape
    else:
        if tx.dim() != 1 or ty.dim() != 1:
            raise ValueError('tx / ty must be 1-d axes or 2-d per-pixel angles')
        width, height = tx.shape[0], ty.shape[0]
    total = width * height
    n_pix = total - pix_begin if n_pix is None else n_pix
    if pix_begin < 0 or n_pix < 0 or pix_begin + n_pix > total:
        raise ValueError(f'pixels [{pix_begin}, {pix_begin + n_pix}) are outside the {height} x {width} frame')
    tx, ty = tx.contiguous(), ty.contiguous()
    dev = tx.device
    rays_o = torch.empty(n_pix, 3, dtype=torch.float32, device=dev)
    rays_d = torch.empty(n_pix, 3, dtype=torch.float32, device=dev)
    times = torch.empty(n_pix, 1, dtype=torch.float32, device=dev) if time is not None else None
    m = (ctypes.c_float * 12)(*[float(v) for v in c2w[:3, :4].reshape(-1).tolist()])
    _l.call(dev, 'sunerf_observer_rays', _ptr(tx), _ptr(ty), 1 if per_pixel else 0, width, pix_begin, n_pix, m,
            float(time) if time is not None else 0.0, _ptr(rays_o), _ptr(rays_d), _ptr(times), _stream(dev))
    return (rays_o, rays_d) if time is None else (rays_o, rays_d, times)


def fov_axis(resolution: int, fov_half_rad: float, device) -> torch.Tensor:
    """Pixel-centre angles of a square frame spanning ``[-fov_half_rad, fov_half_rad]`` (fp64)."""
    return torch.linspace(-fov_half_rad, fov_half_rad, resolution, dtype=torch.float64, device=device)


def observer_rays(resolution: int, row_start: int = 0, row_end: int = None, theta: float = -0.3, phi: float = 0.1,
                  radius: float = 215.032, fov_half_rad: float = 1.1 * 960. / 206264.806, device='cuda'):
    """Synthetic benchmark rays: rows [row_start, row_end) of a resolution x resolution frame seen from
    ``pose_spherical(theta, phi, radius)`` (1 AU = 215.032 solar radii): (n, 3) origins, directions."""
    row_end = resolution if row_end is None else row_end
    axis = fov_axis(resolution, fov_half_rad, device)
    return grid_rays(axis, axis, pose_spherical(theta, phi, radius), row_start * resolution,
                     (row_end - row_start) * resolution)


@torch.no_grad()
def render_frame(rendering, tx: torch.Tensor, ty: torch.Tensor, c2w: torch.Tensor, time: float,
                 wavelengths: Optional[torch.Tensor] = None, tile_rays: int = 1 << 18,
                 keys: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
    """Full-frame inference render: ``rendering(rays_o, rays_d, times[, wavelengths])`` over all pixels of the grid, tile by
    tile, assembled on the device into tensors of shape ``(H, W, ...)`` (the reshape of loader.py:107 / :241).

    ``wavelengths``: (W,) channel list of the density-temperature model, broadcast over the rays (loader.py:220-221).
    ``keys``: outputs to keep (default: all outputs of the rendering module)."""
    if tx.dim() == 2:
        height, width = tx.shape
    else:
        width, height = tx.shape[0], ty.shape[0]
    total = width * height
    frame: Dict[str, torch.Tensor] = {}
    wl_tile = None
    for begin in range(0, total, tile_rays):
        n = min(tile_rays, total - begin)
        rays_o, rays_d, times = grid_rays(tx, ty, c2w, begin, n, time=time)
        if wavelengths is not None:
            if wl_tile is None or wl_tile.shape[0] != n:
                wl_tile = wavelengths.to(device=tx.device, dtype=torch.float32)[None, :].expand(n, -1).contiguous()
            out = rendering(rays_o, rays_d, times, wl_tile)
        else:
            out = rendering(rays_o, rays_d, times)
        for k, v in out.items():
            if keys is not None and k not in keys:
                continue
            if k not in frame:
                frame[k] = torch.empty((total,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            frame[k][begin:begin + n] = v
    return {k: v.view(height, width, *v.shape[1:]) for k, v in frame.items()}
