"""Mirror of the reference's ``sunerf/rendering/density_temperature.py`` on the fused HIP path."""
import torch

from sunerf.model.model import NeRF_DT
from sunerf.rendering.base_tracing import SuNeRFRendering
from sunerf.rendering.functional import dt_pass, dt_raw2outputs
from sunerf_hip.genx import read_aia_temp_resp


class DensityTemperatureRadiativeTransfer(SuNeRFRendering):
    """density_temperature.py:78-274.  Same constructor; the AIA response table is read from
    ``sunerf/data/aia_temp_resp.genx`` relative to the working directory exactly like the reference (:131) unless
    ``response_table=(logte [7,101], tresp [7,101])`` is passed."""

    def __init__(self, model_config=None, device=None, aia_exp_time=2.9, pixel_intensity_factor=1e10,
                 response_table=None, response_path="sunerf/data/aia_temp_resp.genx", **kwargs):
        model_config = {} if model_config is None else model_config
        kwargs.setdefault('model', NeRF_DT)
        super().__init__(model_config=model_config, **kwargs)
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu") if device is None else device
        self.device = device
        self.pixel_intensity_factor = pixel_intensity_factor
        logte, tresp = read_aia_temp_resp(response_path) if response_table is None else response_table
        # density_temperature.py:137-146: response x exposure time, cast to fp32
        self.register_buffer('response_logte', torch.as_tensor(logte).float(), persistent=False)
        self.register_buffer('response_table', torch.as_tensor(tresp * aia_exp_time).float(), persistent=False)

    def regularization(self, distance, regularizing_quantity):
        return torch.relu(distance[:, :] - 1.25 / self.Rs_per_ds) * torch.relu(regularizing_quantity)

    def forward(self, rays_o, rays_d, times, wavelengths=None):
        """base_tracing.py:46-111 for the DT subclass: same 8 output keys, images are (N, W)."""
        if wavelengths is None:
            raise ValueError('DensityTemperatureRadiativeTransfer needs the wavelengths of every ray')
        tables = (self.response_logte, self.response_table)
        reg_radius = 1.25 / self.Rs_per_ds
        z_vals = self.sampler.z_vals(rays_o, rays_d)
        coarse = dt_pass(self.coarse_model, tables, self.pixel_intensity_factor, rays_o, rays_d, times, z_vals, wavelengths,
                         reg_radius, want_epilogues=False)
        new_z, z_comb = self.sampler_hierarchical.resample(z_vals, coarse['weights'])
        fine = dt_pass(self.fine_model, tables, self.pixel_intensity_factor, rays_o, rays_d, times, z_comb, wavelengths,
                       reg_radius, want_epilogues=True)
        return {'z_vals_stratified': z_vals, 'coarse_image': coarse['image'], 'z_vals_hierarchical': new_z,
                'fine_image': fine['image'], 'image': fine['image'], 'height_map': fine['height_map'],
                'absorption_map': fine['absorption_map'], 'regularization': fine['regularization']}

    def raw2outputs(self, inferences, log_abs, vol_c, z_vals, rays_d, wavelengths, **kwargs):
        """density_temperature.py:192-271 on the state ``NeRF_DT.forward`` returns (``inferences`` (N, S, 2) with the base
        offsets added, the ``log_absortpion`` ParameterDict, ``volumetric_constant``): ``{'image' (N,W), 'weights',
        'regularizing_quantity'}``; differentiable through ``image`` (sunerf_dt_integral_fwd / _bwd)."""
        return dt_raw2outputs((self.response_logte, self.response_table), self.pixel_intensity_factor, inferences, log_abs,
                              vol_c, z_vals, rays_d, wavelengths)
