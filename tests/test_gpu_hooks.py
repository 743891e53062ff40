"""Smaller rows of SURVEY.md section 8 on the device:

* the subclass hook ``raw2outputs`` (base_tracing.py:128-132): ``EmissionRadiativeTransfer.raw2outputs`` against the
  reference's own outputs AND its gradient w.r.t. ``raw`` through image + weights (fixture g3), and
  ``DensityTemperatureRadiativeTransfer.raw2outputs`` against the module's own fused forward;
* f-3: device batches of the resident ray pool == the file bytes; rank shards disjoint and complete (on the GPU);
* config 5 at size (8192 rays x 128 + 256 samples x 7 channels): absent channel -> exact 0, gradients additive over ray
  blocks, a 64-ray subset against the CPU oracle.
"""
import numpy as np
import pytest
import torch

import sunerf_oracle as orc
from conftest import gate_units, load_golden

pytestmark = pytest.mark.gpu


def test_emission_raw2outputs_matches_reference_and_its_gradient():
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    g = load_golden('g3_integral')
    mod = EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 32, 'perturb': False},
                                    hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 32},
                                    model_config={'d_filter': 64}).cuda()
    raw = g['raw'].cuda().requires_grad_(True)
    out = mod.raw2outputs(raw=raw, z_vals=g['z_vals'].cuda(), rays_d=g['rays_d'].cuda(), rays_o=None, query_points=None)
    assert set(out) == {'image', 'weights', 'regularizing_quantity'}
    assert out['image'].shape == g['image'].shape
    assert gate_units(out['image'], g['image']) <= 1.0
    assert (out['weights'].detach().cpu() - g['weights']).abs().max().item() <= 1e-5 * g['weights'].abs().max().item()
    # (rays with |d| > 1 run backwards in z here: a = exp(+...) up to 10 -- relative bound, a few ulp of expf)
    assert ((out['regularizing_quantity'].detach().cpu() - g['absorption']).abs() / g['absorption']).max().item() <= 1e-6
    # the probe of the fixture: d (image.sum() + (weights * gw).sum()) / d raw, computed by the reference's autograd
    (out['image'].sum() + (out['weights'] * g['grad_probe'].cuda()).sum()).backward()
    err = (raw.grad.cpu() - g['grad_raw']).abs().max().item()
    assert err <= 1e-4 * g['grad_raw'].abs().max().item(), err
    # ... and through the third output alone
    raw2 = g['raw'].cuda().requires_grad_(True)
    out2 = mod.raw2outputs(raw=raw2, z_vals=g['z_vals'].cuda(), rays_d=g['rays_d'].cuda())
    w = torch.rand(g['absorption'].shape, generator=torch.Generator().manual_seed(3))
    (out2['regularizing_quantity'] * w.cuda()).sum().backward()
    leaf = g['raw'].clone().requires_grad_(True)
    (orc.emission_integral(leaf, g['z_vals'], g['rays_d'])['regularizing_quantity'] * w).sum().backward()
    assert (raw2.grad.cpu() - leaf.grad).abs().max().item() <= 1e-5 * leaf.grad.abs().max().item() + 1e-9


@pytest.mark.parametrize('n_rays,S', [(13, 40), (1, 2), (64, 257), (0, 8)])
def test_emission_integral_ragged_shapes(n_rays, S):
    """The stand-alone integral on shapes that are no multiple of anything (rays per workgroup 8, chunk 32): vs the oracle."""
    from sunerf_hip import ops
    gen = torch.Generator().manual_seed(S)
    raw = torch.randn(n_rays, S, 2, generator=gen)
    z = torch.sort(torch.rand(n_rays, S, generator=gen) * 2.6 + 213.7, dim=-1)[0]
    d = torch.randn(n_rays, 3, generator=gen)
    image, weights, absorption = ops.emission_integral_fwd(raw.cuda(), z.cuda(), d.cuda())
    torch.cuda.synchronize()
    assert image.shape == (n_rays, 1) and weights.shape == (n_rays, S)
    if n_rays == 0:
        return
    ref = orc.emission_integral(raw, z, d)
    assert gate_units(image, ref['image']) <= 1.0
    assert (weights.cpu() - ref['weights']).abs().max().item() <= 1e-5 * ref['weights'].abs().max().item() + 1e-9
    assert ((absorption.cpu() - ref['regularizing_quantity']).abs() / ref['regularizing_quantity']).max().item() <= 1e-6
    g_raw = ops.emission_integral_bwd(raw.cuda(), z.cuda(), d.cuda(), g_image=torch.ones(n_rays).cuda())
    leaf = raw.clone().requires_grad_(True)
    orc.emission_integral(leaf, z, d)['image'].sum().backward()
    assert (g_raw.cpu() - leaf.grad).abs().max().item() <= 1e-5 * leaf.grad.abs().max().item() + 1e-12


def _dt_module(g, n_c=16, n_f=16, d_filter=64):
    from sunerf.model.model import NeRF_DT
    from sunerf.rendering.density_temperature import DensityTemperatureRadiativeTransfer
    return DensityTemperatureRadiativeTransfer(
        Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': n_c, 'perturb': False},
        hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': n_f}, model_config={'d_filter': d_filter}, model=NeRF_DT,
        pixel_intensity_factor=float(g['pixel_intensity_factor']),
        response_table=(g['aia_logte'].numpy(), g['aia_tresp'].numpy()))


def test_dt_raw2outputs_equals_the_fused_forward():
    """density_temperature.py:148-190 by hand -- model.forward on the query points, then raw2outputs(**state) -- against the
    coarse image of the module's fused forward and the reference's (g6)."""
    g = load_golden('g6_dt_e2e')
    mod = _dt_module(g)
    mod.load_state_dict({k[4:].replace('__', '.'): v for k, v in g.items() if k.startswith('sd__')}, strict=True)
    mod = mod.cuda()
    o, d, t, wl = (g[k].cuda() for k in ('rays_o', 'rays_d', 'times', 'wavelengths'))
    z = mod.sampler.z_vals(o, d)
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    query = torch.cat([pts, t[:, None].repeat(1, z.shape[1], 1)], -1)
    with torch.no_grad():
        state = mod.coarse_model(query.view(-1, 4))
    state['inferences'] = state['inferences'].reshape(*z.shape, 2)
    out = mod.raw2outputs(**state, z_vals=z, rays_d=d, wavelengths=wl)
    assert set(out) == {'image', 'weights', 'regularizing_quantity'}
    assert gate_units(out['image'], g['out__coarse_image']) <= 1.0
    fused = mod(o, d, t, wl)
    assert gate_units(out['image'], fused['coarse_image'].detach().cpu()) <= 1.0
    # gradient through image reaches the inferences and the absorption scalars
    inf = state['inferences'].detach().requires_grad_(True)
    img = mod.raw2outputs(inferences=inf, log_abs=state['log_abs'], vol_c=state['vol_c'], z_vals=z, rays_d=d, wavelengths=wl)['image']
    img.sum().backward()
    assert inf.grad is not None and torch.isfinite(inf.grad).all() and inf.grad.abs().max().item() > 0
    assert mod.coarse_model.volumetric_constant.grad is not None


def test_device_ray_pool_matches_the_files_and_shards_partition_them(tmp_path):
    """f-3 on the device: what the kernels are fed is bit for bit what the reference's batch files hold (dataset.py:22-26),
    and the rank shards of SURVEY.md 8e are disjoint and complete."""
    from sunerf_hip.feed import RayPool
    rng = np.random.default_rng(1)
    p = 10007
    arrays = {'rays': rng.normal(size=(p, 2, 3)).astype(np.float32), 'time': rng.random((p, 1), dtype=np.float32),
              'target_image': rng.random((p, 7), dtype=np.float32), 'wavelength': np.tile(np.float32([94, 131, 171, 193, 211, 304, 335]), (p, 1))}
    paths = {}
    for k, v in arrays.items():
        paths[k] = str(tmp_path / f'{k}_batches.npy')
        np.save(paths[k], v)
    world = 4
    pools = [RayPool.from_files(paths, batch_size=512, rank=r, world=world, device='cuda', shuffle=True, seed=5) for r in range(world)]
    assert sum(pl.n_rays for pl in pools) == p
    seen = np.zeros(p, dtype=np.int32)
    for pl in pools:
        assert all(v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() for v in pl.data.values())
        for i in pl.order(0):
            b = pl.batch(int(i))
            start = pl.begin + int(i) * 512
            n = b['time'].shape[0]
            for k in arrays:
                assert b[k].data_ptr() == pl.data[k][int(i) * 512:].data_ptr()                  # a view, no copy
                assert np.array_equal(b[k].cpu().numpy(), arrays[k][start:start + n]), k          # == the file bytes
            seen[start:start + n] += 1
    assert (seen == 1).all()                                                                       # disjoint and complete


def test_dt_config5_size_properties():
    """BASELINE config 5 at its size: 8192 rays, 128 coarse + 128 resampled = 256 fine samples, 7 channels, 8 x 256 NeRF_DT."""
    from sunerf_hip.rays import observer_rays
    g = load_golden('g6_dt_e2e')
    torch.manual_seed(11)
    mod = _dt_module(g, 128, 128, 256).cuda()
    with torch.no_grad():       # absorption that matters (as in g6)
        for m in (mod.coarse_model, mod.fine_model):
            for k, v in zip(m.log_absortpion.keys(), (2e-6, 4e-6, -1e-6, 3e-6, 5e-6, 1e-6, 2e-6)):
                m.log_absortpion[k].fill_(v)
            m.out_layer.weight.mul_(6.0)
    o, d = observer_rays(1024, row_start=508, row_end=516, device='cuda')        # 8 rows through the disk = 8192 rays
    n = o.shape[0]
    gen = torch.Generator().manual_seed(2)
    t = torch.rand(n, 1, generator=gen).cuda()
    wl = torch.tensor([94., 131., 171., 193., 211., 304., 335.]).repeat(n, 1)
    wl[100:300, 2] = 0.
    wl[5000:5100, 5] = 0.
    wl = wl.cuda()
    target = torch.rand(n, 7, generator=gen).cuda()
    params = list(mod.parameters())

    def grads(sl):
        for p in params:
            p.grad = None
        out = mod(o[sl], d[sl], t[sl], wl[sl])
        loss = ((out['coarse_image'] - target[sl]) ** 2).sum() + ((out['fine_image'] - target[sl]) ** 2).sum() \
            + 1e-3 * out['regularization'].sum()
        loss.backward()
        return out, [p.grad.clone() for p in params]

    out, whole = grads(slice(0, n))
    assert all(torch.isfinite(v).all() for v in out.values())
    assert out['fine_image'].shape == (n, 7) and out['regularization'].shape == (n, 256)
    # absent channel -> exactly zero (density_temperature.py:245-256 only fills the present wavelengths)
    assert (out['image'][wl == 0] == 0).all() and (out['coarse_image'][wl == 0] == 0).all()
    assert (out['image'][wl > 0] > 0).all()
    # sums over rays are additive over disjoint ray blocks
    _, a = grads(slice(0, n // 2))
    _, b = grads(slice(n // 2, n))
    for i, (w, x, y) in enumerate(zip(whole, a, b)):
        if w.abs().max() == 0:
            continue
        assert ((w - (x + y)).norm() / w.norm()).item() < 2e-3, i
    # a 64-ray subset against the CPU oracle (forward; the DT gradients are pinned by g6)
    idx = slice(4064, 4128)
    sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
    la = {m: {str(w): sd[f'{m}.log_absortpion.{w}'] for w in orc.AIA_WAVELENGTHS} for m in ('coarse_model', 'fine_model')}
    ref = orc.render_dt(orc.params_from_state_dict(sd, 'coarse_model.'), orc.params_from_state_dict(sd, 'fine_model.'),
                        la['coarse_model'], sd['coarse_model.volumetric_constant'], la['fine_model'],
                        sd['fine_model.volumetric_constant'], o[idx].cpu(), d[idx].cpu(), t[idx].cpu(), wl[idx].cpu(),
                        g['aia_logte'], (g['aia_tresp'] * 2.9).float(), n_coarse=128, n_fine=128,
                        pixel_intensity_factor=float(g['pixel_intensity_factor']), t_vals=sd['sampler.t_vals'])
    with torch.no_grad():
        sub = mod(o[idx], d[idx], t[idx], wl[idx])
    assert torch.equal(sub['z_vals_stratified'].cpu(), ref['z_vals_stratified'])
    units = {k: gate_units(sub[k], ref[k]) for k in ('coarse_image', 'fine_image', 'height_map')}
    print('config 5, 64-ray subset vs oracle:', {k: round(v, 3) for k, v in units.items()})
    assert all(v <= 1.0 for v in units.values()), units
