"""Mirror of the reference's ``sunerf/evaluation/loader.py``: the inference-side caller of the render path
 on the device-side frame driver ``sunerf_hip.rays.render_frame``.

Kept: class names, constructor and method signatures, the dict-of-numpy-images result (``(H, W, ...)`` per output key).
Changed: no ``nn.DataParallel`` and no ``ThreadPoolExecutor`` (loader.py:37-39, :226-229) -- one process drives one GPU,
rays are generated on the device, tiles are rendered back to back on the current stream and assembled there;
``batch_size`` is the tile size in rays and defaults to what keeps the kernels busy instead of 128 / 4096.

``astropy`` / ``sunpy`` are optional.  With them, angles may be ``astropy`` quantities and the pixel grid comes from
the reference map's WCS exactly as in the reference (``all_coordinates_from_map``).  Without them, angles are plain
radians, distances plain solar radii, and the pixel grid is a linear plate scale described by a dict
``{'shape': (H, W), 'cdelt': (arcsec/pixel x, y), 'crpix': (x, y) 1-based, 'crval': (arcsec x, y)}``.
"""
from datetime import datetime, timedelta
from typing import Optional, Tuple

import numpy as np
import torch

from sunerf_hip.rays import pose_spherical, render_frame

AU_IN_SOLAR_RADII = 215.03215567054764      # (1 * u.AU).to(u.solRad), IAU 2012 / 2015 nominal values
ARCSEC = np.pi / 180. / 3600.


def _radians(x) -> float:
    if hasattr(x, 'to_value'):
        from astropy import units as u
        return float(x.to_value(u.rad))
    return float(x)


def _solar_radii(x) -> float:
    if hasattr(x, 'to_value'):
        from astropy import units as u
        return float(x.to_value(u.solRad))
    return float(x)


def normalize_datetime(date, seconds_per_dt, ref_time):
    """sunerf/data/date_util.py:4-17."""
    return (date - ref_time).total_seconds() / seconds_per_dt


def unnormalize_datetime(norm_date: float, seconds_per_dt, ref_time) -> datetime:
    """sunerf/data/date_util.py:20-31."""
    return ref_time + timedelta(seconds=norm_date * seconds_per_dt)


def linear_plate_scale_axes(grid: dict, resolution=None, device='cuda') -> Tuple[torch.Tensor, torch.Tensor]:
    """Column (Tx) and row (Ty) angles [rad, fp64] of a frame described by a linear plate scale; ``resolution`` (H, W)
    resamples it over the same field of view like ``Map.resample`` (loader.py:73-75)."""
    h, w = grid['shape']
    cdx, cdy = grid['cdelt']
    cpx, cpy = grid.get('crpix', ((w + 1) / 2., (h + 1) / 2.))
    cvx, cvy = grid.get('crval', (0., 0.))
    if resolution is not None:
        nh, nw = (resolution, resolution) if np.isscalar(resolution) else resolution
        sx, sy = w / nw, h / nh                      # pixel-size ratio; pixel centres move with the field of view
        cdx, cdy = cdx * sx, cdy * sy
        cpx, cpy = (cpx - 0.5) / sx + 0.5, (cpy - 0.5) / sy + 0.5
        h, w = nh, nw
    col = torch.arange(1, w + 1, dtype=torch.float64, device=device)
    row = torch.arange(1, h + 1, dtype=torch.float64, device=device)
    return ((col - cpx) * cdx + cvx) * ARCSEC, ((row - cpy) * cdy + cvy) * ARCSEC


def load_state_file(state_path):
    """``torch.load`` of a ``.snf`` state (sunerf.py:62-74).  A density-temperature state written by the reference holds
    ``xitorch.interpolate.Interp1D`` objects (density_temperature.py:143-146); where xitorch is not installed a state-only
    stand-in of that name is registered for the load, and ``DensityTemperatureRadiativeTransfer.__setstate__`` takes the
    response table out of whatever the objects carry."""
    try:
        return torch.load(state_path, map_location='cpu', weights_only=False)
    except ModuleNotFoundError as err:
        if not (err.name or '').split('.')[0] == 'xitorch':
            raise
    import importlib.abc
    import importlib.machinery
    import sys
    import types

    class _Anything:                      # every class the pickle names under xitorch.* becomes a plain state holder
        def __init__(self, *a, **k):
            pass

    class _StandIn(types.ModuleType):
        __path__ = []                     # a package: sub-modules of any depth resolve through the finder below

        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            cls = type(name, (_Anything,), {'__module__': self.__name__})
            setattr(self, name, cls)
            return cls

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, fullname, path=None, target=None):
            if fullname == 'xitorch' or fullname.startswith('xitorch.'):
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            return _StandIn(spec.name)

        def exec_module(self, module):
            pass
    finder = _Finder()
    sys.meta_path.insert(0, finder)
    try:
        return torch.load(state_path, map_location='cpu', weights_only=False)
    finally:
        sys.meta_path.remove(finder)
        for name in [m for m in sys.modules if m == 'xitorch' or m.startswith('xitorch.')]:
            del sys.modules[name]


class SuNeRFLoader:
    """loader.py:16-134."""

    def __init__(self, state_path, device=None):
        device = torch.device('cuda') if device is None else torch.device(device)
        self.device = device
        state = load_state_file(state_path)
        data_config = state['data_config']
        self.config = data_config
        self.wavelength = data_config.get('wavelength')
        self.times = data_config.get('times')
        self.wcs = data_config.get('wcs')
        self.resolution = data_config.get('resolution')
        self.rendering = state['rendering'].to(device)
        self.model = self.rendering.fine_model
        self.seconds_per_dt = state['seconds_per_dt']
        self.Rs_per_ds = state['Rs_per_ds']
        self.Mm_per_ds = self.Rs_per_ds * 695.7            # (1 * u.R_sun).to_value(u.Mm)
        self.ref_time = state['ref_time']
        self.ref_map = self._reference_map()

    def _reference_map(self):
        if isinstance(self.wcs, dict):                      # linear plate scale (no sunpy needed)
            return self.wcs
        from sunpy.map import Map                           # raises if sunpy is missing: a real WCS needs it
        return Map(np.zeros(self.resolution), self.wcs)

    @property
    def start_time(self):
        return np.min(self.times)

    @property
    def end_time(self):
        return np.max(self.times)

    def _pixel_angles(self, resolution):
        """Helioprojective angles of every pixel: two axes for a plate-scale dict, per-pixel arrays for a sunpy map."""
        if isinstance(self.ref_map, dict):
            return linear_plate_scale_axes(self.ref_map, resolution, self.device)
        from astropy import units as u
        from sunpy.coordinates import frames
        from sunpy.map import all_coordinates_from_map
        ref_map = self.ref_map.resample(resolution) if resolution is not None else self.ref_map
        coords = all_coordinates_from_map(ref_map).transform_to(frames.Helioprojective)
        tx = torch.from_numpy(np.ascontiguousarray(coords.Tx.to_value(u.rad), dtype=np.float64)).to(self.device)
        ty = torch.from_numpy(np.ascontiguousarray(coords.Ty.to_value(u.rad), dtype=np.float64)).to(self.device)
        return tx, ty

    def _render(self, lat, lon, time: float, distance, center, resolution, batch_size, wl=None, as_numpy=True):
        target_pose = pose_spherical(-_radians(lon), _radians(lat), _solar_radii(distance), center)
        tx, ty = self._pixel_angles(resolution)
        wavelengths = None if wl is None else torch.as_tensor(np.asarray(wl), dtype=torch.float32, device=self.device)
        frame = render_frame(self.rendering, tx, ty, target_pose, float(time), wavelengths, tile_rays=int(batch_size))
        if not as_numpy:
            return frame
        return {k: v.cpu().numpy() for k, v in frame.items()}

    @torch.no_grad()
    def render_observer_image(self, lat, lon, time: datetime, distance=AU_IN_SOLAR_RADII,
                              center: Tuple[float, float, float] = None, resolution=None, batch_size: int = 1 << 18,
                              as_numpy: bool = True):
        """loader.py:63-108: image of the observer at (lat, lon, distance) at ``time`` (a datetime)."""
        time = normalize_datetime(time, self.seconds_per_dt, self.ref_time)
        return self._render(lat, lon, time, distance, center, resolution, batch_size, None, as_numpy)

    def normalize_datetime(self, time):
        return normalize_datetime(time, self.seconds_per_dt, self.ref_time)

    def unnormalize_datetime(self, time):
        return unnormalize_datetime(time, self.seconds_per_dt, self.ref_time)

    @torch.no_grad()
    def load_coords(self, query_points_npy, batch_size=1 << 20):
        """loader.py:119-134: model output (emission / absorption or density / temperature logits) at query points
        ``(..., 4)`` = (x, y, z, t)."""
        target_shape = query_points_npy.shape[:-1]
        flat = torch.from_numpy(np.ascontiguousarray(query_points_npy)).float().reshape(-1, 4)
        parts = []
        for b in range(0, flat.shape[0], batch_size):
            answer = self.model(flat[b:b + batch_size].to(self.device))      # any field model: NeRF (fused points mode), SimpleStar, ...
            parts.append((answer['inferences'] if isinstance(answer, dict) else answer).cpu())      # D3 resolved: the tensor
        return torch.cat(parts, 0).view(*target_shape, -1).numpy()


class ModelLoader(SuNeRFLoader):
    """loader.py:137-242: loader around an in-memory rendering module (density-temperature path: ``wl`` channels)."""

    def __init__(self, rendering, model, ref_map, device=None):
        device = torch.device('cuda') if device is None else torch.device(device)
        self.device = device
        self.ref_map = ref_map
        self.rendering = rendering.to(device)
        self.model = model.to(device)
        self.seconds_per_dt = 1
        meta = ref_map.get('meta', {}) if isinstance(ref_map, dict) else ref_map.meta
        stamp = meta['t_obs'] if 't_obs' in meta else meta.get('date-obs')
        self.ref_time = datetime.strptime(stamp, '%Y-%m-%dT%H:%M:%S.%f') if stamp is not None else None

    def process_batch(self, b_rays_o, b_rays_d, b_time, b_wl):
        return self.rendering(b_rays_o, b_rays_d, b_time, b_wl)

    def process_batch_with_index(self, index, b_rays_o, b_rays_d, b_time, b_wl):
        return index, self.process_batch(b_rays_o, b_rays_d, b_time, b_wl)

    @torch.no_grad()
    def render_observer_image(self, lat, lon, time: float, distance=AU_IN_SOLAR_RADII, wl: Optional[np.ndarray] = None,
                              center: Tuple[float, float, float] = None, resolution=None, batch_size: int = 1 << 17,
                              as_numpy: bool = True):
        """loader.py:159-242: ``time`` is already normalised here (a float)."""
        return self._render(lat, lon, time, distance, center, resolution, batch_size, wl, as_numpy)
