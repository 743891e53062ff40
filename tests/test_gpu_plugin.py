"""The reference's PLUG-IN interface (base_tracing.py:46-132): a subclass of ``SuNeRFRendering`` supplies ``raw2outputs``
(and may replace ``_render`` / ``regularization``) in ordinary torch code; the base ``forward`` orchestrates the two passes
around it.  Here the samplers and the network run on the HIP kernels (the network differentiably: ``functional.mlp_on_rays``)
and the subclass's torch code sits in between.  Also ``HierarchicalSampler.sample_pdf`` called by itself (sampling.py:128-169).
"""
import pytest
import torch

import sunerf_oracle as orc
from conftest import gate_units, load_golden, params_from_golden

pytestmark = pytest.mark.gpu

CFG = dict(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 32, 'perturb': False},
           hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 32})


def _cfg(**model_config):
    return {**{k: (dict(v) if isinstance(v, dict) else v) for k, v in CFG.items()}, 'model_config': dict(model_config)}


def _load(module, g):
    sd = {k[4:].replace('__', '.'): v for k, v in g.items() if k.startswith('sd__')}
    module.load_state_dict(sd, strict=True)
    return module.cuda()


def _torch_emission_plugin():
    """A third-party subclass as the reference's users write them: emission.py:14-54 restated in torch autograd."""
    from sunerf.rendering.base_tracing import SuNeRFRendering

    class TorchEmission(SuNeRFRendering):
        def __init__(self, **kw):
            kw['model_config'] = {**kw.get('model_config', {}), 'd_input': 4, 'd_output': 2}
            super().__init__(**kw)

        def raw2outputs(self, raw, z_vals, rays_d, **kwargs):
            return orc.emission_integral(raw, z_vals, rays_d)          # plain torch ops on the tensors' device
    return TorchEmission


def test_torch_written_subclass_renders_and_trains_like_the_fused_class():
    """Only ``raw2outputs`` supplied, in torch: outputs at the north-star gate against the REFERENCE's outputs (g5), and the
    parameter gradients of the training loss against the reference's (1e-3 relative L2 per tensor)."""
    g = load_golden('g5_emission_e2e')
    mod = _load(_torch_emission_plugin()(**_cfg(d_filter=64)), g)
    o, d, t = g['rays_o'].cuda(), g['rays_d'].cuda(), g['times'].cuda()
    out = mod(o, d, t)
    assert set(out) == {'z_vals_stratified', 'coarse_image', 'z_vals_hierarchical', 'fine_image', 'image', 'height_map',
                        'absorption_map', 'regularization'}
    for k in ('coarse_image', 'fine_image', 'image', 'height_map'):
        assert gate_units(out[k].detach(), g['out__' + k]) <= 1.0, k
    assert gate_units(out['absorption_map'].detach(), g['out__absorption_map'], floor=64 * 6e-8) <= 1.0
    # the training loss of sunerf.py:110-120, in torch (ImageAsinhScaling(vmax=1, a=0.005), lambdas 1)
    scale = lambda im: torch.asinh(im / 0.005) / torch.asinh(torch.tensor(1 / 0.005))    # noqa: E731
    target = scale(g['target'].cuda())
    mse = torch.nn.MSELoss()
    loss = mse(scale(out['coarse_image']), target) + mse(scale(out['fine_image']), target) + out['regularization'].mean()
    assert abs(loss.item() - g['loss'].item()) <= 2e-4 * abs(g['loss'].item())
    loss.backward()
    worst = 0.0
    for name, p in mod.named_parameters():
        ref = g['grad__' + name.replace('.', '__')]
        err = ((p.grad.cpu() - ref).norm() / ref.norm()).item()
        worst = max(worst, err)
        assert err <= 1e-3, (name, err)
    print(f'torch-written raw2outputs: worst parameter-gradient error {worst:.2e}')


def test_subclass_of_the_fused_class_gets_its_own_hooks():
    """A subclass of ``EmissionRadiativeTransfer`` that replaces a hook must not be served by the built-in fused physics."""
    from sunerf.rendering.emission import EmissionRadiativeTransfer

    class Doubled(EmissionRadiativeTransfer):
        def raw2outputs(self, raw, z_vals, rays_d, **kwargs):
            out = super().raw2outputs(raw, z_vals, rays_d)
            return {**out, 'image': 2 * out['image']}

    class OtherRadius(EmissionRadiativeTransfer):
        def regularization(self, distance, regularizing_quantity):
            return torch.relu(distance - 1.05) * (1 - regularizing_quantity)

    g = load_golden('g5_emission_e2e')
    plain = _load(EmissionRadiativeTransfer(**_cfg(d_filter=64)), g)
    o, d, t = g['rays_o'].cuda(), g['rays_d'].cuda(), g['times'].cuda()
    with torch.no_grad():
        ref = plain(o, d, t)
        twice = _load(Doubled(**_cfg(d_filter=64)), g)(o, d, t)
        other = _load(OtherRadius(**_cfg(d_filter=64)), g)(o, d, t)
    assert gate_units(twice['image'], 2 * ref['image'].cpu()) <= 1.0 and gate_units(twice['coarse_image'], 2 * ref['coarse_image'].cpu()) <= 1.0
    assert gate_units(other['image'], ref['image'].cpu()) <= 1.0
    # the generic path forms |o + d z| in torch, the fused one inside the kernel: same regularization up to that rounding
    dist = (o[:, None, :] + d[:, None, :] * torch.sort(torch.cat([ref['z_vals_stratified'], ref['z_vals_hierarchical']], -1), -1)[0][..., None]).norm(dim=-1)
    assert (other['regularization'] > 0).sum() > (ref['regularization'] > 0).sum()
    assert torch.all(other['regularization'][dist < 1.05 - 1e-4] == 0)


def test_density_temperature_hooks():
    """``DensityTemperatureRadiativeTransfer._render`` (density_temperature.py:148-190) by itself and through the generic
    forward of a subclass: same images as the fused forward / fixture g6."""
    from sunerf.model.model import NeRF_DT
    from sunerf.rendering.base_tracing import ray_query_points
    from sunerf.rendering.density_temperature import DensityTemperatureRadiativeTransfer as DT

    class Louder(DT):
        def regularization(self, distance, regularizing_quantity):
            return 3 * super().regularization(distance, regularizing_quantity)

    g = load_golden('g6_dt_e2e')
    kw = dict(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 16, 'perturb': False},
              hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 16}, model_config={'d_filter': 64},
              model=NeRF_DT, pixel_intensity_factor=float(g['pixel_intensity_factor']),
              response_table=(g['aia_logte'].numpy(), g['aia_tresp'].numpy()))
    fresh = lambda cls: _load(cls(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()}), g)   # noqa: E731
    plain, loud = fresh(DT), fresh(Louder)
    o, d, t, wl = (g[k].cuda() for k in ('rays_o', 'rays_d', 'times', 'wavelengths'))
    z = g['out__z_vals_stratified'].cuda()
    coarse = plain._render(plain.coarse_model, ray_query_points(o, d, t, z), d, o, z, wl)
    assert set(coarse) == {'image', 'weights', 'regularizing_quantity'}
    for c in range(7):
        assert gate_units(coarse['image'][:, c].detach(), g['out__coarse_image'][:, c]) <= 1.0, c
    out = loud(o, d, t, wl)
    for k in ('coarse_image', 'fine_image'):
        for c in range(7):
            assert gate_units(out[k][:, c].detach(), g['out__' + k][:, c]) <= 1.0, (k, c)
    ref_reg = g['out__regularization']
    assert (out['regularization'].detach().cpu() - 3 * ref_reg).abs().max().item() <= 2e-4 * 3 * ref_reg.abs().max().item() + 1e-7
    # ... and it trains: gradients reach the MLP, the absorption scalars and the volumetric constant
    target = g['target'].cuda()
    mse = torch.nn.MSELoss()
    (mse(out['coarse_image'], target) + mse(out['fine_image'], target)).backward()
    for name, p in loud.named_parameters():
        ref = g['grad__' + name.replace('.', '__')]
        if ref.norm() == 0 or 'log_absortpion' in name or 'volumetric' in name:
            continue                       # (g6's gradients include the regularization term of the scalar heads)
        assert p.grad is not None and torch.isfinite(p.grad).all(), name


def test_sample_pdf_called_by_itself():
    """``HierarchicalSampler.sample_pdf(bins, weights)`` against the oracle's restatement of sampling.py:128-169, on g4's
    bins (bit-exact apart from the reference's own threshold discontinuity, as for the fused resampling) and on ragged sizes."""
    from sunerf.train.sampling import HierarchicalSampler
    g = load_golden('g4_hierarchical')
    z, w = g['z_vals'], g['weights']
    bins, wts = .5 * (z[:, 1:] + z[:, :-1]), w[:, 1:-1]
    s = HierarchicalSampler(n_samples=32, perturb=False)
    got = s.sample_pdf(bins.cuda(), wts.cuda()).cpu()
    diff = (got - g['new_z']).abs()
    assert diff.max().item() <= (z[:, 1:] - z[:, :-1]).max().item() * 1.001 and (diff > 6.2e-5).float().mean().item() <= 1e-3
    for n, nb, sf in ((5, 2, 7), (33, 9, 1), (1, 40, 64)):
        gen = torch.Generator().manual_seed(n)
        b = torch.sort(torch.rand(n, nb, generator=gen) * 3 + 1, -1)[0]
        ww = torch.rand(n, nb - 1, generator=gen)
        u = torch.linspace(0., 1., sf)
        want = orc.sample_pdf(b, ww, u)
        got = HierarchicalSampler(n_samples=sf).sample_pdf(b.cuda(), ww.cuda()).cpu()
        assert got.shape == (n, sf) and torch.isfinite(got).all()
        assert (got >= b[:, :1] - 1e-6).all() and (got <= b[:, -1:] + 1e-6).all()
        assert (got - want).abs().max().item() <= 1e-5


def test_module_call_on_free_standing_points_is_differentiable():
    """``NeRF.forward(points)`` (model.py:44-57) under autograd: a loss on arbitrary query points reaches the parameters, as in the
    reference -- values against the reference's own outputs (g2), gradients against the oracle's autograd."""
    from sunerf.model.model import NeRF
    g = load_golden('g2_mlp')
    net = NeRF(d_input=4, d_output=2, n_layers=8, d_filter=64)
    net.load_state_dict({k[5:].replace('__', '.'): v for k, v in g.items() if k.startswith('net__')}, strict=True)
    net = net.cuda()
    x = g['x'].cuda()
    out = net(x)['inferences']
    assert out.requires_grad and (out.detach().cpu() - g['inferences']).abs().max().item() < 2e-5
    probe = torch.randn(out.shape, generator=torch.Generator().manual_seed(4)).cuda()
    (out * probe).sum().backward()
    params = [(W.clone().requires_grad_(True), b.clone().requires_grad_(True)) for W, b in params_from_golden(g, 'net__')]
    (orc.mlp_forward(params, g['x']) * probe.cpu()).sum().backward()
    lin = net.linears()
    for (W, b), layer in zip(params, lin):
        for ref, got, bound in ((W.grad, layer.weight.grad, 1e-3), (b.grad, layer.bias.grad, 1e-3)):   # (256 points: the fp32 backward)
            assert ((got.cpu() - ref).norm() / ref.norm()).item() <= bound
    with torch.no_grad():
        plain = net(x)['inferences']
        assert not plain.requires_grad and torch.equal(plain, out.detach())
        # ragged counts: padded to whole 32-point chunks inside, every point independent of its neighbours
        for m in (1, 31, 33, 250):
            assert torch.equal(net(x[:m])['inferences'], plain[:m]), m
        assert net(x[:0])['inferences'].shape == (0, 2)


@pytest.mark.parametrize('d_filter,n_layers', [(64, 3), (128, 2), (256, 8), (512, 3)])
@pytest.mark.parametrize('mode', ['fast', 'exact', 'half'])
def test_points_mode_is_the_same_arithmetic_as_a_ray_through_the_point(d_filter, n_layers, mode):
    """``sunerf_mlp_points_fwd`` against the render pass on two-sample rays o = 0, d = xyz, z = 1 (whose samples ARE the points):
    bit-identical raw outputs for every width and forward arithmetic."""
    from sunerf_hip import ops
    precision = {'fast': ops.PRECISION_FAST, 'exact': ops.PRECISION_EXACT, 'half': ops.PRECISION_HALF}[mode]
    params = orc.init_params(d_filter=d_filter, n_layers=n_layers, seed=77)
    pk = ops.PackedMLP([W.cuda() for W, _ in params], [b.cuda() for _, b in params], precision=precision)
    pts = (torch.rand(100, 4, generator=torch.Generator().manual_seed(1)) * 2.4 - 1.2).cuda()
    got = ops.mlp_points_fwd(pk, pts)['raw']
    ref = ops.emission_render_fwd(pk, torch.zeros(100, 3, device='cuda'), pts[:, :3].contiguous(), pts[:, 3].contiguous(),
                                  torch.ones(100, 2, device='cuda'), 0.0, want_raw=True)['raw'][:, 0]
    assert torch.equal(got, ref)


def test_any_module_can_be_the_field_model():
    """``model=`` takes any ``nn.Module`` answering ``{'inferences', 'log_abs', 'vol_c'}`` on (M, 4) points -- the reference renders its MHD
    cubes that way (evaluation/image_render.py:252-268).  A torch-written wrapper around the analytic star must render, through the
    generic forward, what the fused ``SimpleStar`` path renders (fixture g9's reference outputs)."""
    from torch import nn
    from sunerf.model.stellar_model import SimpleStar
    from sunerf.rendering.density_temperature import DensityTemperatureRadiativeTransfer as DT

    class Wrapped(nn.Module):                      # no `field_on_rays`, not a NeRF: unknown to the kernels
        def __init__(self):
            super().__init__()
            self.star = SimpleStar()

        def forward(self, x):
            return self.star(x)

    g = load_golden('g9_simple_star')
    mod = DT(Rs_per_ds=1, model=Wrapped, model_config={}, sampling_config={'type': 'stratified', 'n_samples': 24, 'perturb': False},
             hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 24},
             response_table=(g['aia_logte'].numpy(), g['aia_tresp'].numpy())).cuda()
    with torch.no_grad():
        for m in (mod.coarse_model, mod.fine_model):
            for w in (94, 131, 171, 193, 211, 304, 335):
                m.star.log_absortpion[str(w)].copy_(g[f'la__{w}'])
            m.star.volumetric_constant.copy_(g['vol_c'])
        got = mod(g['rays_o'].cuda(), g['rays_d'].cuda(), g['times'].cuda(), g['wavelengths'].cuda())
    for k in ('coarse_image', 'fine_image', 'image'):
        for c in range(g['out__' + k].shape[1]):
            assert gate_units(got[k][:, c], g['out__' + k][:, c]) <= 1.0, (k, c)
    assert (got['image'].cpu()[g['wavelengths'] == 0] == 0).all()
