"""Mirror of the reference's ``sunerf/rendering/density_temperature.py`` on the fused HIP path."""
import torch

from sunerf.model.model import NeRF_DT
from sunerf.rendering.base_tracing import SuNeRFRendering, field_on_query_points
from sunerf.rendering.functional import dt_pass, dt_raw2outputs
from sunerf_hip.genx import CHANNELS, read_aia_temp_resp


def _tensors_inside(obj, depth=3):
    """1-D tensors reachable through the attributes of ``obj`` (an interpolator object of unknown class)."""
    found = []
    if torch.is_tensor(obj):
        return [obj] if obj.ndim == 1 and obj.numel() >= 2 else []
    if depth == 0:
        return found
    for v in (vars(obj).values() if hasattr(obj, '__dict__') else (obj if isinstance(obj, (list, tuple)) else ())):
        found += _tensors_inside(v, depth - 1)
    return found


def _tables_from_interpolators(response):
    """(logte [7, n], response [7, n]) out of the ``{wavelength: Interp1D(logte, tresp * exposure)}`` dict a reference-written
    state carries (density_temperature.py:132-146): per channel the strictly increasing grid and the other tensor of its
    length.  None when the objects do not show them (then the table is read from the file again)."""
    rows_x, rows_y = [], []
    for channel in CHANNELS:
        ts = _tensors_inside(response.get(channel))
        grids = [t for t in ts if bool((t[1:] > t[:-1]).all())]
        if not grids:
            return None
        x = grids[0]
        ys = [t for t in ts if t is not x and t.numel() == x.numel() and t.data_ptr() != x.data_ptr()]
        if len(ys) != 1:
            return None
        rows_x.append(x.detach().float().cpu())
        rows_y.append(ys[0].detach().float().cpu())
    if len({r.numel() for r in rows_x}) != 1:
        return None
    return torch.stack(rows_x), torch.stack(rows_y)


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

    # ---- .snf compatibility in both directions (sunerf.py:62-74 pickles this object) ---------------------------------
    def __getstate__(self):
        """What the reference's class needs besides modules and scalars is ``self.response`` (density_temperature.py:132-146,
        read at :248): one ``xitorch`` interpolator per channel.  Where xitorch is installed -- any environment that runs the
        reference -- they are built through its public constructor exactly as the reference builds them, so that a state
        written here renders in the reference; elsewhere the file simply lacks them (and still loads here)."""
        st = self.__dict__.copy()
        try:
            from xitorch.interpolate import Interp1D
        except ImportError:
            return st
        x, y = self._buffers['response_logte'], self._buffers['response_table']
        st['response'] = {c: Interp1D(x[i].clone(), y[i].clone(), method='linear', extrap=0) for i, c in enumerate(CHANNELS)}
        return st

    def __setstate__(self, state):
        """A state written by the REFERENCE has the interpolators but not this class's table buffers: take the table out of
        them, or read the file again like the constructor."""
        self.__dict__.update(state)
        if 'response_logte' in self._buffers and 'response_table' in self._buffers:
            return
        tables = _tables_from_interpolators(state['response']) if isinstance(state.get('response'), dict) else None
        if tables is None:
            logte, tresp = read_aia_temp_resp("sunerf/data/aia_temp_resp.genx")
            tables = torch.as_tensor(logte).float(), torch.as_tensor(tresp * 2.9).float()     # the constructor's default exposure
        where = next((p.device for p in self.parameters()), torch.device('cpu'))
        self.register_buffer('response_logte', tables[0].to(where), persistent=False)
        self.register_buffer('response_table', tables[1].to(where), persistent=False)

    def regularization(self, distance, regularizing_quantity):
        return torch.relu(distance[:, :] - 1.25 / self.Rs_per_ds) * torch.relu(regularizing_quantity)

    def forward(self, rays_o, rays_d, times, wavelengths=None):
        """base_tracing.py:46-111 for the DT subclass: same 8 output keys, images are (N, W)."""
        if wavelengths is None:
            raise ValueError('DensityTemperatureRadiativeTransfer needs the wavelengths of every ray')
        if self._hooks_replaced(DensityTemperatureRadiativeTransfer):   # a subclass with its own raw2outputs / _render / regularization
            return SuNeRFRendering.forward(self, rays_o, rays_d, times, wavelengths)
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

    def _render(self, model, query_points, rays_d, rays_o, z_vals, wavelengths):
        """density_temperature.py:148-190: ``model.forward`` at the query points -- inferences with the base offsets, the
        absorption scalars, the volumetric constant -- plus ``z_vals`` / ``rays_d`` / ``wavelengths`` into ``raw2outputs``."""
        inferences, state = field_on_query_points(model, query_points, rays_o, rays_d, z_vals)
        return self.raw2outputs(inferences=inferences, z_vals=z_vals, rays_d=rays_d, wavelengths=wavelengths, **state)

    def raw2outputs(self, inferences, log_abs, vol_c, z_vals, rays_d, wavelengths, **kwargs):
        """density_temperature.py:192-271 on the state ``NeRF_DT.forward`` returns (``inferences`` (N, S, 2) with the base
        offsets added, the ``log_absortpion`` ParameterDict, ``volumetric_constant``): ``{'image' (N,W), 'weights',
        'regularizing_quantity'}``; differentiable through ``image`` (sunerf_dt_integral_fwd / _bwd)."""
        return dt_raw2outputs((self.response_logte, self.response_table), self.pixel_intensity_factor, inferences, log_abs,
                              vol_c, z_vals, rays_d, wavelengths)
