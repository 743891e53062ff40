"""Mirror of the reference's ``sunerf/train/sampling.py`` (class API + buffers), computed by HIP kernels.

The render module never calls these ``forward``s on its fused path (it only needs ``z_vals``; points are formed
inside the render kernel) -- they exist so that code written against the reference keeps working.
"""
import torch
from torch import nn

from sunerf_hip import ops


def _points(rays_o, rays_d, z_vals):
    # o + d*z of sampling.py:100 -- API-compat materialisation only (the fused kernel never builds this tensor)
    return rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]


class _RaySampler(nn.Module):
    kind = None
    _scalars = None      # class-level default: instances unpickled from a reference-written .snf have no such attribute

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_scalars', None)      # a cache, not state
        return state

    def __init__(self, Rs_per_ds, distance, n_samples, perturb):
        super().__init__()
        self.perturb = perturb
        # same buffers (names, dtypes, values) as sampling.py:62-66
        self.register_buffer('distance', torch.tensor(distance / Rs_per_ds, dtype=torch.float32))
        self.register_buffer('solar_R', torch.tensor(1 / Rs_per_ds, dtype=torch.float32))
        self.register_buffer('t_vals', torch.linspace(0., 1., n_samples)[None].to(torch.float32))
        self._scalars = None

    def _buffer_scalars(self):
        # host copies of the two 0-d buffers (one sync, then cached; reset by _apply / load_state_dict)
        if self._scalars is None:
            self._scalars = (float(self.distance), float(self.solar_R))
        return self._scalars

    def _apply(self, fn, *args, **kwargs):
        self._scalars = None
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._scalars = None
        return super()._load_from_state_dict(*args, **kwargs)

    def z_vals(self, rays_o, rays_d, t_rand=None):
        distance, solar_R = self._buffer_scalars()
        if self.perturb and t_rand is None:
            t_rand = torch.rand(rays_o.shape[0], self.t_vals.shape[-1], device=rays_o.device)
        if not self.perturb:
            t_rand = None
        return ops.sample_z(self.kind, rays_o, rays_d, self.t_vals, distance, solar_R, t_rand)

    def forward(self, rays_o: torch.Tensor, rays_d: torch.Tensor):
        z_vals = self.z_vals(rays_o, rays_d)
        return {'points': _points(rays_o, rays_d, z_vals), 'z_vals': z_vals}


class StratifiedSampler(_RaySampler):
    """sampling.py:56-102: slab of +-distance around the observer distance, clipped at the solar surface."""
    kind = ops.SAMPLER_STRATIFIED

    def __init__(self, Rs_per_ds, distance=1.3, n_samples=64, perturb=True):
        super().__init__(Rs_per_ds, distance, n_samples, perturb)


class SphericalSampler(_RaySampler):
    """sampling.py:4-54: between the two intersections with the sphere of radius ``distance``."""
    kind = ops.SAMPLER_SPHERICAL

    def __init__(self, Rs_per_ds, distance=2.0, n_samples=64, perturb=True):
        super().__init__(Rs_per_ds, distance, n_samples, perturb)


class HierarchicalSampler(nn.Module):
    """sampling.py:104-169: inverse-CDF resampling of the coarse weights, merged with the coarse samples."""
    _u = None            # class-level default (see _RaySampler)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_u', None)
        return state

    def __init__(self, n_samples=128, perturb=False):
        super().__init__()
        self.n_samples = n_samples
        self.perturb = perturb
        self._u = None

    def _positions(self, n_rays, device):
        """The CDF positions to invert: fresh uniform numbers per ray (perturb) or ``linspace(0, 1, n_samples)``."""
        if self.perturb:
            return torch.rand(n_rays, self.n_samples, device=device)
        if self._u is None or self._u.device != device or self._u.numel() != self.n_samples:
            self._u = torch.linspace(0., 1., self.n_samples, device=device)   # sampling.py:140
        return self._u

    def resample(self, z_vals, weights):
        return ops.hier_resample(z_vals, weights, self._positions(z_vals.shape[0], z_vals.device))

    def sample_pdf(self, bins: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
        """sampling.py:128-169 called by itself: inverse-transform samples (N, n_samples) of the piecewise-constant density
        ``weights`` (N, B-1) on ``bins`` (N, B).  (``forward`` / ``resample`` run the same kernel with the mid points and the
        merge folded in.)"""
        return ops.sample_pdf(bins, weights, self._positions(bins.shape[0], bins.device))

    def forward(self, rays_o, rays_d, z_vals, weights):
        new_z, z_comb = self.resample(z_vals, weights)
        return {'points': _points(rays_o, rays_d, z_comb), 'z_vals': z_comb, 'new_z_samples': new_z}
