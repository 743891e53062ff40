"""Mirror of the reference's ``sunerf/rendering/base_tracing.py``.

``SuNeRFRendering.forward`` keeps the reference's signature and output dict (base_tracing.py:46-111) but runs as
four launches: sampler (z only) -> fused coarse pass -> hierarchical resample -> fused fine pass with the
epilogues (absorption_map, height_map, regularization) folded in.  Defects D1-D3 of the reference at HEAD
 are resolved to their evident intent.
"""
import torch
from torch import nn

from sunerf.model.model import NeRF
from sunerf.train.sampling import SphericalSampler, HierarchicalSampler, StratifiedSampler


_SAMPLERS = {'spherical': SphericalSampler, 'stratified': StratifiedSampler}
_RESAMPLERS = {'hierarchical': HierarchicalSampler}


def _from_config(registry, config, default_type, **fixed):
    """Builds ``registry[config['type']](**fixed, **rest of config)``.  Like the reference (base_tracing.py:24, :33) the
    ``'type'`` key is POPPED from the caller's dict, and an unknown type is a ``ValueError`` with the reference's text."""
    config = {'type': default_type} if config is None else config
    kind = config.pop('type')
    if kind not in registry:
        raise ValueError(f'Unknown sampling type {kind}')
    return registry[kind](**fixed, **config)


class SuNeRFRendering(nn.Module):
    """base_tracing.py:8-132: owns the two samplers and the coarse / fine field models; subclasses supply ``forward``."""

    def __init__(self, Rs_per_ds, sampling_config=None, hierarchical_sampling_config=None, model=NeRF,
                 model_config=None):
        super().__init__()
        self.Rs_per_ds = Rs_per_ds
        self.sampler = _from_config(_SAMPLERS, sampling_config, 'stratified', Rs_per_ds=Rs_per_ds)
        self.sampler_hierarchical = _from_config(_RESAMPLERS, hierarchical_sampling_config, 'hierarchical')
        model_config = model_config or {}
        self.coarse_model, self.fine_model = model(**model_config), model(**model_config)

    def regularization(self, distance, regularizing_quantity):
        # base_tracing.py:43-44 with D2 resolved: (N, S)
        return torch.relu(distance - 1.2 / self.Rs_per_ds) * (1 - regularizing_quantity)

    def forward(self, rays_o, rays_d, times, wavelengths=None):
        raise NotImplementedError("This method should be implemented in a subclass")

    def forward_points(self, query_points):
        # base_tracing.py:113-116 with D3 resolved: the tensor, not the dict
        return self.fine_model(query_points.view(-1, 4))['inferences']

    def raw2outputs(self, **kwargs):
        raise NotImplementedError("This method should be implemented in a subclass")


def cumprod_exclusive(tensor: torch.Tensor) -> torch.Tensor:
    """Exclusive cumulative product along the last dimension: ``out[..., i] = prod(tensor[..., :i])``, ``out[..., 0] = 1``
    (base_tracing.py:135-156; same values: the inclusive products shifted by one).  Kept for callers of the module API;
    the fused kernels use a per-wavefront scan."""
    inclusive = torch.cumprod(tensor, -1)
    return torch.cat([torch.ones_like(inclusive[..., :1]), inclusive[..., :-1]], dim=-1)
