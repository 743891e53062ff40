"""Mirror of the reference's ``sunerf/model/stellar_model.py``: the analytic ``SimpleStar`` density / temperature field
that pretends to be a trained ``NeRF_DT`` when synthetic observations are rendered (evaluation/image_render.py:236-268).

Same constructor, parameters and state-dict keys; the field itself is evaluated by ``sunerf_simple_star_field`` on the
device.  ``astropy`` is optional: quantities are converted when given, plain numbers are taken in the reference's own
units (h0 [Mm], T0 [K], R_s [solar radii], t_photosphere [K], rho_0 [cm^-3])."""
import torch
from torch import nn

from sunerf_hip import ops

MM_PER_SOLAR_RADIUS = 695.7     # IAU 2015 nominal solar radius, astropy's u.solRad


def _value(q, unit_name, default_scale=1.0):
    if hasattr(q, 'to'):
        import astropy.units as u
        unit = {'solRad': u.solRad, 'K': u.K, 'cm-3': 1 / u.cm ** 3}[unit_name]
        return float(q.to(unit).value)
    return float(q) * default_scale


class SimpleStar(nn.Module):
    """stellar_model.py:5-102."""

    def __init__(self, h0=60., T0=1.4e6, R_s=1.02, t_photosphere=5777., rho_0=3.0e8):
        super().__init__()
        self.h0 = _value(h0, 'solRad', 1. / MM_PER_SOLAR_RADIUS)      # plain number: megametres, like the default 60*u.Mm
        self.T0 = _value(T0, 'K')
        self.R_s = _value(R_s, 'solRad')
        self.t_photosphere = _value(t_photosphere, 'K')
        self.rho_0 = _value(rho_0, 'cm-3')
        self.log_absortpion = nn.ParameterDict([[str(w), torch.tensor(v, dtype=torch.float32)] for w, v in
                                                zip(ops.AIA_WAVELENGTHS, (20.4, 20.2, 20.0, 19.8, 19.6, 19.4, 19.2))])
        self.stellar_parameters = nn.ParameterDict([['Rs', torch.tensor(self.R_s, dtype=torch.float32)],
                                                    ['h0', torch.tensor(self.h0, dtype=torch.float32)],
                                                    ['T0', torch.tensor(self.T0, dtype=torch.float32)],
                                                    ['rho_0', torch.tensor(self.rho_0, dtype=torch.float32)]])
        self.volumetric_constant = nn.Parameter(torch.tensor(1.0, dtype=torch.float32, requires_grad=True))
        # NeRF_DT adds these to its raw output (model.py:182-183); the analytic field is already physical
        self.base_log_density = 0.0
        self.base_log_temperature = 0.0

    def _constants(self):
        """The four stellar parameters as host floats, re-read only when a parameter changed (one device -> host copy each)."""
        sp = self.stellar_parameters
        key = tuple((sp[k].data_ptr(), sp[k]._version) for k in ('rho_0', 'h0', 'T0', 'Rs'))
        if getattr(self, '_const_key', None) != key:
            self._const_key = key
            self._const = tuple(float(sp[k]) for k in ('rho_0', 'h0', 'T0', 'Rs'))
        return self._const

    @torch.no_grad()
    def field_on_rays(self, rays_o, rays_d, z_vals):
        """(N, S, 2) = (ln rho, log10 T) at o + d z; inference only (the reference never trains a SimpleStar)."""
        rho_0, h0, T0, Rs = self._constants()
        return ops.simple_star_field(rays_o, rays_d, z_vals, rho_0, h0, T0, Rs, self.t_photosphere)

    def forward(self, query_points):
        """(M, >=3) query points -> {'inferences': (M, 2), 'log_abs', 'vol_c'} (stellar_model.py:53-102)."""
        pts = query_points.reshape(-1, query_points.shape[-1])
        o = torch.zeros(pts.shape[0], 3, dtype=torch.float32, device=pts.device)
        z = torch.ones(pts.shape[0], 1, dtype=torch.float32, device=pts.device)    # o + d * 1 = the point itself, exactly
        raw = self.field_on_rays(o, pts[:, :3].contiguous(), z)
        return {'inferences': raw[:, 0, :], 'log_abs': self.log_absortpion, 'vol_c': self.volumetric_constant}

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_const_key', None)
        state.pop('_const', None)
        return state
