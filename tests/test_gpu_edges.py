"""Edge cases of the hot path on MI355X: empty batches, ragged sample counts, a single ray, rays that miss the sampling
sphere (NaN z like the reference), large batches checked through size-independent properties."""
import pytest
import torch

import sunerf_oracle as orc
from conftest import gate_units

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from sunerf_hip import ops as _ops
    return _ops


def _packed(ops, d_filter=64, n_layers=3, seed=5):
    params = orc.init_params(d_filter=d_filter, n_layers=n_layers, seed=seed)
    return params, ops.PackedMLP([W.cuda() for W, _ in params], [b.cuda() for _, b in params])


def test_empty_batch(ops):
    params, packed = _packed(ops)
    o = torch.zeros(0, 3, device='cuda')
    z = ops.sample_z(ops.SAMPLER_STRATIFIED, o, o, torch.linspace(0, 1, 32).cuda(), 1.3, 1.0)
    assert z.shape == (0, 32)
    out = ops.emission_render_fwd(packed, o, o, torch.zeros(0, device='cuda'), z, 1.2, want_epilogues=True, training=True)
    assert out['image'].shape == (0, 1) and out['regularization'].shape == (0, 32)
    nz, zc = ops.hier_resample(z, out['weights'], torch.linspace(0, 1, 16).cuda())
    assert nz.shape == (0, 16) and zc.shape == (0, 48)
    gW = [torch.ones_like(W).cuda() for W, _ in params]
    gb = [torch.ones_like(b).cuda() for _, b in params]
    ops.emission_render_bwd(packed, o, o, z, out['raw'], out['stash'], torch.zeros(0, device='cuda'), None, 0.0, 1.2, gW, gb)
    assert all((g == 0).all() for g in gW + gb)          # an empty batch contributes a zero gradient


@pytest.mark.parametrize('n_rays,S', [(1, 2), (1, 33), (3, 65), (5, 31), (7, 200)])
def test_ragged_shapes_vs_oracle(ops, n_rays, S):
    params, packed = _packed(ops)
    torch.manual_seed(S)
    o, d = orc.synthetic_rays(3)
    o, d = o[:n_rays].contiguous(), d[:n_rays].contiguous()
    t = torch.rand(n_rays, 1)
    z = orc.stratified_z(o, d, orc.linspace_t_vals(S), torch.tensor(1.3), torch.tensor(1.0))
    ref = orc.render_pass(params, o, d, t, z)
    out = ops.emission_render_fwd(packed, o.cuda(), d.cuda(), t.cuda(), z.cuda(), 1.2, want_raw=True)
    assert (out['raw'].cpu() - ref['raw']).abs().max().item() < 2e-5
    assert gate_units(out['image'], ref['image']) <= 1.0            # the north-star gate, per ray
    assert ((out['weights'].cpu() - ref['weights']).abs().max() / ref['weights'].abs().max()).item() < 1e-4


def test_spherical_sampler_miss_gives_nan_like_reference(ops):
    # a ray that misses the 2 Rs sphere: the reference's quadratic has no real root -> NaN z_vals (no guard, sampling.py:28-29)
    o = torch.tensor([[0., 0., 215.]])
    d = torch.tensor([[0.05, 0., -1.]])
    d = d / d.norm()
    z_ref = orc.spherical_z(o, d, orc.linspace_t_vals(8), torch.tensor(2.0), torch.tensor(1.0))
    z = ops.sample_z(ops.SAMPLER_SPHERICAL, o.cuda(), d.cuda(), torch.linspace(0, 1, 8).cuda(), 2.0, 1.0).cpu()
    assert torch.isnan(z_ref).all() and torch.isnan(z).all()


def test_large_batch_properties(ops):
    """BASELINE-sized launch (512 x 512 rays x 128 samples, 8 x 256 MLP): size-independent properties instead of a CPU
    comparison -- weights of every ray sum to 1, absorption in (0, 1], the image equals the sum of the un-normalised
    weights, identical rays give identical pixels, and a random subset matches the oracle."""
    from sunerf_hip.rays import observer_rays
    params, packed = _packed(ops, d_filter=256, n_layers=8, seed=7)
    o, d = observer_rays(512, device='cuda')
    n = o.shape[0]
    t = torch.zeros(n, device='cuda')
    z = ops.sample_z(ops.SAMPLER_STRATIFIED, o, d, torch.linspace(0, 1, 128).cuda(), 1.3, 1.0)
    out = ops.emission_render_fwd(packed, o, d, t, z, 1.2, want_epilogues=True)
    assert torch.isfinite(out['image']).all()
    assert (out['weights'].sum(-1) - 1).abs().max().item() < 1e-4
    assert (out['absorption'] > 0).all() and (out['absorption'] <= 1).all()
    # the frame is mirror-symmetric in neither axis (theta, phi != 0), but re-rendering a permuted copy must permute pixels
    perm = torch.randperm(n, device='cuda')[:4096]
    out2 = ops.emission_render_fwd(packed, o[perm].contiguous(), d[perm].contiguous(), t[perm].contiguous(), z[perm].contiguous(), 1.2)
    assert torch.equal(out2['image'], out['image'][perm])
    idx = perm[:64].cpu()
    ref = orc.render_pass(params, o.cpu()[idx], d.cpu()[idx], torch.zeros(64, 1), z.cpu()[idx])
    assert gate_units(out['image'][idx.cuda()], ref['image']) <= 1.0       # the north-star gate, per ray


def test_training_batch_properties(ops):
    """BASELINE-sized training launch (32768 rays x 128 samples, 8 x 256 MLP) through forward-with-stash, integral backward,
    dgrad and wgrad: gradients are additive over disjoint ray blocks (second block accumulated on top of the first equals
    the whole batch), linear in the upstream gradient, and a random subset of rays reproduces the oracle's gradient."""
    from sunerf_hip.rays import observer_rays
    params, packed = _packed(ops, d_filter=256, n_layers=8, seed=7)
    o, d = observer_rays(1024, row_start=496, row_end=528, device='cuda')       # 32 rows through the disk
    n = o.shape[0]
    gen = torch.Generator().manual_seed(4)
    t = torch.rand(n, generator=gen).cuda()
    g_image = (torch.randn(n, generator=gen) * 1e-4).cuda()
    z = ops.sample_z(ops.SAMPLER_STRATIFIED, o, d, torch.linspace(0, 1, 128).cuda(), 1.3, 1.0)

    def grads(sl, scale=1.0, into=None):
        fwd = ops.emission_render_fwd(packed, o[sl], d[sl], t[sl], z[sl], 1.2, want_epilogues=True, training=True)
        gW = into[0] if into else [torch.empty(W.shape, device='cuda') for W, _ in params]
        gb = into[1] if into else [torch.empty(b.shape, device='cuda') for _, b in params]
        ops.emission_render_bwd(packed, o[sl], d[sl], z[sl], fwd['raw'], fwd['stash'], g_image[sl] * scale, None, 3e-7, 1.2,
                                gW, gb, accumulate=into is not None)
        return gW, gb

    whole = grads(slice(0, n))
    halves = grads(slice(0, n // 2))
    grads(slice(n // 2, n), into=halves)
    scaled = grads(slice(0, n), scale=8.0)       # also moves the on-device fp16 gradient scale by 3 binades
    for i, (W, Wh, Ws) in enumerate(zip(whole[0], halves[0], scaled[0])):
        assert torch.isfinite(W).all()
        assert ((W - Wh).norm() / W.norm()).item() < 1e-3, i
    # linear in g_image for the image part (the constant regularization gradient does not scale): compare differences
    eighth = grads(slice(0, n), scale=0.0)
    for i, (W, Ws, W0) in enumerate(zip(whole[0], scaled[0], eighth[0])):
        assert (((Ws - W0) - 8.0 * (W - W0)).norm() / (8.0 * (W - W0)).norm()).item() < 2e-3, i
    # a block of 96 rays against the CPU oracle's autograd
    idx = slice(n // 2 - 48, n // 2 + 48)
    sub = grads(idx)
    leaves = [(W.clone().requires_grad_(True), b.clone().requires_grad_(True)) for W, b in params]
    out = orc.render_pass(leaves, o[idx].cpu(), d[idx].cpu(), t[idx].cpu()[:, None], z[idx].cpu())
    dist_pts = out['points'].pow(2).sum(-1).pow(0.5)
    reg = torch.relu(dist_pts - 1.2) * (1 - out['regularizing_quantity'])
    ((out['image'][:, 0] * g_image[idx].cpu()).sum() + 3e-7 * reg.sum()).backward()
    for i, ((W, b), gW, gb) in enumerate(zip(leaves, sub[0], sub[1])):
        assert ((gW.cpu() - W.grad).norm() / W.grad.norm()).item() < 1e-3, i
        assert ((gb.cpu() - b.grad).norm() / b.grad.norm()).item() < 1e-3, i


def test_unsupported_width_is_a_clear_error():
    from sunerf.model.model import NeRF
    with pytest.raises(ValueError, match='d_filter'):
        NeRF(d_filter=640)
    with pytest.raises(ValueError, match='d_input'):
        NeRF(d_input=3)


@pytest.mark.parametrize('d_filter,encoding', [(100, 'positional'), (200, None), (48, 'none'), (320, 'positional')])
def test_any_width_and_no_encoding_by_exact_zero_padding(d_filter, encoding):
    """Generality of NeRF (model.py:16-17, 28-33): any d_filter, and `encoding` other than 'positional' (first layer on the raw
    coordinates).  Both run on the compiled widths by zero padding, which changes no value: forward against the oracle at the
    north-star gate, all parameter gradients at 1e-3, state-dict keys as the reference module tree."""
    from conftest import gate_units
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    torch.manual_seed(d_filter)
    mod = EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 32, 'perturb': False},
                                    hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 32},
                                    model_config={'d_filter': d_filter, 'n_layers': 4, 'encoding': encoding})
    positional = encoding == 'positional'
    keys = set(mod.coarse_model.state_dict())
    assert ('in_layer.1.weight' in keys) == positional and ('in_layer.weight' in keys) == (not positional)
    assert mod.coarse_model.linears()[0].weight.shape == (d_filter, 84 if positional else 4)
    with torch.no_grad():       # raw coordinates are O(1): keep the pre-activations of the un-encoded variant in a sane range
        for m in (mod.coarse_model, mod.fine_model):
            m.out_layer.weight.mul_(3.0)
    o, d = orc.synthetic_rays(5)
    t = torch.rand(o.shape[0], 1)
    sd = {k: v.detach().clone() for k, v in mod.state_dict().items()}

    def params_of(prefix):
        first = 'in_layer.1' if positional else 'in_layer'
        p = [(sd[f'{prefix}{first}.weight'], sd[f'{prefix}{first}.bias'])]
        p += [(sd[f'{prefix}layers.{i}.weight'], sd[f'{prefix}layers.{i}.bias']) for i in range(3)]
        return p + [(sd[f'{prefix}out_layer.weight'], sd[f'{prefix}out_layer.bias'])]
    leaves = {m: [(W.clone().requires_grad_(True), b.clone().requires_grad_(True)) for W, b in params_of(m)]
              for m in ('coarse_model.', 'fine_model.')}
    want = orc.render_emission(leaves['coarse_model.'], leaves['fine_model.'], o, d, t, n_coarse=32, n_fine=32,
                               t_vals=sd['sampler.t_vals'], encoding=positional)
    target = torch.rand(o.shape[0], 1, generator=torch.Generator().manual_seed(1))
    loss_ref = ((want['coarse_image'] - target) ** 2).mean() + ((want['fine_image'] - target) ** 2).mean() + want['regularization'].mean()
    loss_ref.backward()
    mod = mod.cuda()
    got = mod(o.cuda(), d.cuda(), t.cuda())
    assert mod.coarse_model.packed().padded and mod.coarse_model.packed().d_filter in (64, 128, 256, 512)
    for k in ('coarse_image', 'fine_image', 'height_map'):
        assert gate_units(got[k], want[k].detach()) <= 1.0, k
    loss = ((got['coarse_image'] - target.cuda()) ** 2).mean() + ((got['fine_image'] - target.cuda()) ** 2).mean() + got['regularization'].mean()
    loss.backward()
    worst = {}
    for m in ('coarse_model', 'fine_model'):
        for lin, (W, b) in zip(getattr(mod, m).linears(), leaves[m + '.']):
            assert lin.weight.grad.shape == W.shape
            eW = ((lin.weight.grad.cpu() - W.grad).norm() / W.grad.norm()).item()
            eb = ((lin.bias.grad.cpu() - b.grad).norm() / b.grad.norm()).item()
            worst[m] = max(worst.get(m, 0.0), eW, eb)
            assert eW < 1e-3 and eb < 1e-3, (m, eW, eb)       # SURVEY's gate for every tensor (measured <= 3.4e-4)
    print('padded-width module gradients, worst tensor:', {m: f'{v:.2e}' for m, v in worst.items()}, '(bound 1e-3)')


@pytest.mark.parametrize('n_layers,S', [(8, 64), (2, 32), (3, 40)])
def test_reference_default_width_512_forward(ops, n_layers, S, precision):
    """d_filter = 512 is the reference's default (model.py:16): one activation set in registers + scratch spill."""
    params, packed = _packed(ops, d_filter=512, n_layers=n_layers, seed=11)
    torch.manual_seed(S)
    o, d = orc.synthetic_rays(5)
    t = torch.rand(o.shape[0], 1) * 3
    z = orc.stratified_z(o, d, orc.linspace_t_vals(S), torch.tensor(1.3), torch.tensor(1.0))
    ref = orc.render_pass(params, o, d, t, z)
    out = ops.emission_render_fwd(packed, o.cuda(), d.cuda(), t.cuda(), z.cuda(), 1.2, want_raw=True)
    assert (out['raw'].cpu() - ref['raw']).abs().max().item() < 2e-5
    assert gate_units(out['image'], ref['image']) <= 1.0            # the north-star gate, per ray
    assert ((out['weights'].cpu() - ref['weights']).abs().max() / ref['weights'].abs().max()).item() < 1e-4


@pytest.mark.parametrize('scale', [1e-3, 1e-6, 0.0])
def test_tiny_weights_stay_finite(ops, scale, precision):
    """Layers whose weights are all tiny (max |w| / 2 pi below the fp16 normal range): the fp8 scale search must not push the
    remainders of subnormal fp16 heads past the e4m3 maximum (this used to give NaN images in the FAST arithmetic)."""
    params = [(W * scale, b * scale) for W, b in orc.init_params(d_filter=64, n_layers=3, seed=2)]
    o, d = orc.synthetic_rays(4)
    t = torch.zeros(o.shape[0], 1)
    z = orc.stratified_z(o, d, orc.linspace_t_vals(32), torch.tensor(1.3), torch.tensor(1.0))
    ref = orc.render_pass(params, o, d, t, z)
    packed = ops.PackedMLP([W.cuda() for W, _ in params], [b.cuda() for _, b in params])
    out = ops.emission_render_fwd(packed, o.cuda(), d.cuda(), t.cuda(), z.cuda(), reg_radius=1.2, want_raw=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out['image']).all() and torch.isfinite(out['raw']).all()
    assert (out['raw'].cpu() - ref['raw']).abs().max().item() < 1e-6 + 2e-5 * scale
    assert gate_units(out['image'], ref['image']) <= 1.0            # the north-star gate, per ray
