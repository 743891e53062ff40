"""GPU parity of the input side of the path: device ray generation vs the reference's get_rays (g8),
and the full-frame driver / loader mirror vs per-batch rendering."""
import datetime

import numpy as np
import pytest
import torch

import sunerf_oracle as orc
from conftest import gate_units, load_golden

pytestmark = pytest.mark.gpu


def _ulp_close(a, b, what):
    """fp32 results of fp64 sin / cos: the device library and glibc may differ in the last fp64 bit, which can flip a
    rounding to fp32 -- at most 1 ulp, on at most 1 % of the elements; everything else bit-exact."""
    a, b = a.cpu(), b.cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    diff = (a - b).abs()
    ulp = torch.finfo(torch.float32).eps * b.abs().clamp_min(1e-30)
    assert (diff <= 2 * ulp).all(), (what, diff.max().item())
    assert (diff > 0).float().mean().item() <= 0.01, (what, (diff > 0).float().mean().item())


def test_pose_spherical_matches_reference():
    from sunerf_hip.rays import pose_spherical
    g = load_golden('g8_observer_rays')
    for name in ('a', 'b'):
        theta, phi, radius, sx, sy, sz, has_shift = [float(v) for v in g[f'pose_{name}']]
        c2w = pose_spherical(theta, phi, radius, (sx, sy, sz) if has_shift else None)
        assert torch.equal(c2w, g[f'c2w_{name}'])


@pytest.mark.parametrize('grid', ['axis', 'pix'])
def test_grid_rays_match_reference_get_rays(grid):
    from sunerf_hip.rays import grid_rays
    g = load_golden('g8_observer_rays')
    for name in ('a', 'b'):
        tx = g[f'tx_{grid}'].double().cuda()
        ty = g[f'ty_{grid}'].double().cuda()
        o, d, t = grid_rays(tx, ty, g[f'c2w_{name}'], time=0.25)
        assert torch.equal(o.cpu(), g[f'rays_o_{name}_{grid}'].reshape(-1, 3))
        _ulp_close(d, g[f'rays_d_{name}_{grid}'].reshape(-1, 3), (name, grid))
        assert t.shape == (o.shape[0], 1) and (t == 0.25).all()
        # a tile in the middle of the frame is the same slice
        o2, d2 = grid_rays(tx, ty, g[f'c2w_{name}'], pix_begin=17, n_pix=40)
        assert torch.equal(d2, d[17:57]) and torch.equal(o2, o[17:57])
    with pytest.raises(ValueError):
        grid_rays(tx, ty, g['c2w_a'], pix_begin=100, n_pix=100)
    from sunerf_hip.lib import SunerfHipError
    with pytest.raises(SunerfHipError):
        grid_rays(tx.cpu(), ty.cpu(), g['c2w_a'])


def test_observer_rays_match_oracle_synthetic_rays():
    from sunerf_hip.rays import observer_rays
    o, d = observer_rays(33, device='cuda')
    ro, rd = orc.synthetic_rays(33)
    assert torch.equal(o.cpu(), ro)
    _ulp_close(d, rd, 'synthetic')
    o2, d2 = observer_rays(33, row_start=5, row_end=9, device='cuda')
    assert torch.equal(d2, d[5 * 33:9 * 33])


def _rendering(d_filter=64):
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    torch.manual_seed(3)
    return EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 32, 'perturb': False},
                                     hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 32},
                                     model_config={'d_filter': d_filter}).cuda()


def test_render_frame_equals_one_shot_render_and_oracle():
    from sunerf_hip.rays import fov_axis, grid_rays, pose_spherical, render_frame
    rendering = _rendering()
    half = 1.1 * 960. / 206264.806
    tx, ty = fov_axis(20, half * 1.2, 'cuda'), fov_axis(14, half, 'cuda')     # 14 rows x 20 columns
    c2w = pose_spherical(0.4, -0.2, 215.032)
    frame = render_frame(rendering, tx, ty, c2w, 0.3, tile_rays=37)           # ragged tiles
    o, d, t = grid_rays(tx, ty, c2w, time=0.3)
    with torch.no_grad():
        one = rendering(o, d, t)
    assert set(frame) == set(one)
    for k, v in one.items():
        assert frame[k].shape[:2] == (14, 20)
        assert torch.equal(torch.nan_to_num(frame[k].reshape(v.shape)), torch.nan_to_num(v)), k
    sd = {k: v.cpu() for k, v in rendering.state_dict().items()}
    want = orc.render_emission(orc.params_from_state_dict(sd, 'coarse_model.'), orc.params_from_state_dict(sd, 'fine_model.'),
                               o.cpu(), d.cpu(), t.cpu(), n_coarse=32, n_fine=32, t_vals=sd['sampler.t_vals'])
    # the north-star gate, per ray (conftest.gate_units; absorption_map = sum over 64 samples of the fp32 difference 1 - a)
    for k in ('coarse_image', 'fine_image', 'height_map', 'absorption_map'):
        u = gate_units(frame[k].reshape(want[k].shape), want[k], floor=64 * 6e-8 if k == 'absorption_map' else 0.0)
        print(f'frame {k}: {u:.3f} gate units')
        assert u <= 1.0, (k, u)


def test_sunerf_loader_roundtrip(tmp_path):
    """save_state -> SuNeRFLoader (sunerf.py:62-74, loader.py:16-134): render_observer_image and load_coords."""
    from sunerf.evaluation.loader import SuNeRFLoader
    from sunerf.model.sunerf import save_state
    rendering = _rendering()

    class _Module:
        pass

    class _Data:
        config = {'wavelength': 193, 'times': [datetime.datetime(2022, 1, 1), datetime.datetime(2022, 1, 3)],
                  'resolution': (16, 16), 'wcs': {'shape': (16, 16), 'cdelt': (150., 150.)}}
        Rs_per_ds, seconds_per_dt, ref_time = 1.0, 86400., datetime.datetime(2022, 1, 1)
    mod = _Module()
    mod.rendering = rendering
    path = str(tmp_path / 'run' / 'save_state.snf')
    save_state(mod, _Data(), path)
    loader = SuNeRFLoader(path, device='cuda')
    assert loader.start_time == datetime.datetime(2022, 1, 1) and loader.Rs_per_ds == 1.0
    when = datetime.datetime(2022, 1, 2, 12)
    assert abs(loader.normalize_datetime(when) - 1.5) < 1e-12 and loader.unnormalize_datetime(1.5) == when
    out = loader.render_observer_image(lat=0.1, lon=0.3, time=when, batch_size=100)
    assert out['image'].shape == (16, 16, 1) and isinstance(out['image'], np.ndarray)
    assert np.isfinite(out['image']).all() and out['image'].max() > 0
    # same frame through the in-memory module with hand-made rays
    from sunerf_hip.rays import grid_rays, pose_spherical
    from sunerf.evaluation.loader import linear_plate_scale_axes
    tx, ty = linear_plate_scale_axes(_Data.config['wcs'], None, 'cuda')
    assert abs(tx[0].item() + 7.5 * 150. * np.pi / 180 / 3600) < 1e-15 and tx[0] == -tx[-1]
    o, d, t = grid_rays(tx, ty, pose_spherical(-0.3, 0.1, 215.03215567054764), time=1.5)
    with torch.no_grad():
        ref = rendering(o, d, t)
    assert np.array_equal(out['image'].reshape(-1), ref['image'].cpu().numpy().reshape(-1))
    low = loader.render_observer_image(lat=0.1, lon=0.3, time=when, resolution=8)
    assert low['image'].shape == (8, 8, 1)
    # point queries
    pts = np.random.default_rng(0).uniform(-1.2, 1.2, size=(5, 7, 4)).astype(np.float32)
    got = loader.load_coords(pts, batch_size=16)
    sd = {k: v.cpu() for k, v in rendering.state_dict().items()}
    want = orc.mlp_forward(orc.params_from_state_dict(sd, 'fine_model.'), torch.from_numpy(pts).reshape(-1, 4))
    assert got.shape == (5, 7, 2)
    assert np.abs(got.reshape(-1, 2) - want.numpy()).max() < 1e-4 * np.abs(want.numpy()).max()
    # ... of any field model (loader.py:119-134 calls the module): the analytic star behind a ModelLoader
    from sunerf.evaluation.loader import ModelLoader
    from sunerf.model.stellar_model import SimpleStar
    star = SimpleStar()
    field = ModelLoader(rendering=rendering, model=star, ref_map={'meta': {'t_obs': '2022-01-01T00:00:00.000'}})
    got = field.load_coords(pts, batch_size=16)
    want = orc.simple_star_field(torch.from_numpy(pts).reshape(-1, 4), *(star.stellar_parameters[k].detach().cpu() for k in ('rho_0', 'h0', 'T0', 'Rs')))
    assert got.shape == (5, 7, 2) and np.abs(got.reshape(-1, 2) - want.numpy()).max() < 1e-5 * np.abs(want.numpy()).max()


def test_reference_written_state_file_loads_into_fused_classes():
    """A .snf pickled from the REFERENCE's classes (golden g10; sunerf.py:62-74) unpickles into the mirrored classes, and
    the fused path renders from it what the reference renders from the same weights."""
    import os
    from conftest import GOLDEN as GOLDEN_DIR
    from sunerf.evaluation.loader import SuNeRFLoader
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    g = load_golden('g10_reference_state')
    loader = SuNeRFLoader(os.path.join(GOLDEN_DIR, 'g10_reference_state.snf'), device='cuda')
    assert type(loader.rendering) is EmissionRadiativeTransfer and loader.wavelength == 193
    with torch.no_grad():
        out = loader.rendering(g['rays_o'].cuda(), g['rays_d'].cuda(), g['times'].cuda())
    assert torch.equal(out['z_vals_stratified'].cpu(), g['out__z_vals_stratified'])
    n_s = g['out__regularization'].shape[-1]
    for k in ('coarse_image', 'fine_image', 'image', 'height_map', 'absorption_map'):
        u = gate_units(out[k], g['out__' + k], floor=n_s * 6e-8 if k == 'absorption_map' else 0.0)
        print(f'g10 {k}: {u:.3f} gate units')
        assert u <= 1.0, (k, u)
    # regularization = relu(|x| - r) (1 - a): not one of the gated outputs; relative to the tensor maximum
    ref = g['out__regularization']
    err = (out['regularization'].cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f'g10 regularization: {err:.2e} of the maximum (bound 2e-4)')
    assert err < 2e-4, err
    got = loader.load_coords(g['points'].numpy())
    assert np.abs(got - g['inferences'].numpy()).max() < 1e-4 * np.abs(g['inferences'].numpy()).max()
    frame = loader.render_observer_image(lat=0.0, lon=0.2, time=datetime.datetime(2022, 3, 2))
    assert frame['image'].shape == (12, 12, 1) and np.isfinite(frame['image']).all()
    # and back: the state dict of the loaded module has exactly the reference's keys
    keys = set(loader.rendering.state_dict())
    assert {'sampler.distance', 'sampler.solar_R', 'sampler.t_vals', 'coarse_model.in_layer.0.freq_bands',
            'fine_model.out_layer.bias'} <= keys
