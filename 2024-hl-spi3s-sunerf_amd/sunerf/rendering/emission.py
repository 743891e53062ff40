"""Mirror of the reference's ``sunerf/rendering/emission.py`` on the fused HIP path."""
import torch

from sunerf.rendering.base_tracing import SuNeRFRendering
from sunerf.rendering.functional import emission_pass, emission_raw2outputs


class EmissionRadiativeTransfer(SuNeRFRendering):

    def __init__(self, model_config=None, **kwargs):
        model_config = {} if model_config is None else model_config
        model_config.update({'d_input': 4, 'd_output': 2, })  # emission.py:11
        super().__init__(model_config=model_config, **kwargs)

    def forward(self, rays_o, rays_d, times, wavelengths=None):
        """base_tracing.py:46-111 for the emission subclass: same 8 output keys."""
        if self._hooks_replaced(EmissionRadiativeTransfer):      # a subclass with its own raw2outputs / _render / regularization
            return SuNeRFRendering.forward(self, rays_o, rays_d, times, wavelengths)
        if wavelengths is not None:
            raise ValueError('EmissionRadiativeTransfer takes no wavelengths')
        z_vals = self.sampler.z_vals(rays_o, rays_d)
        reg_radius = 1.2 / self.Rs_per_ds
        coarse = emission_pass(self.coarse_model, rays_o, rays_d, times, z_vals, reg_radius, want_epilogues=False)
        new_z, z_comb = self.sampler_hierarchical.resample(z_vals, coarse['weights'])   # no gradient (sampling.py:120)
        fine = emission_pass(self.fine_model, rays_o, rays_d, times, z_comb, reg_radius, want_epilogues=True)
        return {'z_vals_stratified': z_vals, 'coarse_image': coarse['image'], 'z_vals_hierarchical': new_z,
                'fine_image': fine['image'], 'image': fine['image'], 'height_map': fine['height_map'],
                'absorption_map': fine['absorption_map'], 'regularization': fine['regularization']}

    def raw2outputs(self, raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor, **kwargs):
        """emission.py:14-54 on a given ``raw`` (N, S, 2): ``{'image' (N,1), 'weights' (N,S), 'regularizing_quantity' (N,S)}``,
        differentiable w.r.t. ``raw`` (sunerf_emission_integral_fwd / _bwd).  ``forward`` does not come through here: it
        runs the same arithmetic fused behind the MLP (sunerf_emission_render_fwd)."""
        return emission_raw2outputs(raw, z_vals, rays_d)
