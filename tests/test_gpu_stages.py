"""Stage-wise parity of the HIP path (through the C ABI) against the oracle, on a real MI355X.

Tolerances (fp32, BASELINE.json north_star: outputs within 1e-4 rel of the reference CPU renderer):
  z_vals                         bit-exact (same rounding sequence, no FMA contraction)
  MLP raw output                 2e-5 absolute (|raw| = O(0.1..1)); measured ~1e-6
  image / weights / absorption   1e-4 relative to the tensor's max (measured ~1e-6)
"""
import pytest
import torch

import sunerf_oracle as orc
from conftest import gate_units, load_golden, params_from_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    from sunerf_hip import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def rel_err(a, b):
    return ((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def pack(ops, params):
    return ops.PackedMLP([dev(W) for W, _ in params], [dev(b) for _, b in params])


def test_sample_z_bit_exact(ops):
    g = load_golden('g1_sampler')
    z = ops.sample_z(ops.SAMPLER_STRATIFIED, dev(g['rays_o']), dev(g['rays_d']), dev(g['t_vals']), 1.3, 1.0)
    assert torch.equal(z.cpu(), g['z_vals'])
    zp = ops.sample_z(ops.SAMPLER_STRATIFIED, dev(g['rays_o']), dev(g['rays_d']), dev(g['t_vals']), 1.3, 1.0,
                      t_rand=dev(g['t_rand']))
    assert torch.equal(zp.cpu(), g['z_vals_perturb'])
    zs = ops.sample_z(ops.SAMPLER_SPHERICAL, dev(g['rays_o_sph']), dev(g['rays_d_sph']), dev(g['t_vals']), 2.0, 1.0)
    assert torch.equal(zs.cpu(), g['z_vals_sph'])
    zr = ops.sample_z(ops.SAMPLER_STRATIFIED, dev(g['rays_o_rs']), dev(g['rays_d']), dev(g['t_vals_rs']),
                      float(torch.tensor(1.3 / 0.5, dtype=torch.float32)), 2.0)
    assert torch.equal(zr.cpu(), g['z_vals_rs'])


@pytest.mark.parametrize('d_filter,n_layers', [(64, 8), (128, 3), (256, 8), (64, 1), (64, 2)])
@pytest.mark.parametrize('S', [32, 40, 96])
def test_render_pass_vs_oracle(ops, d_filter, n_layers, S, precision):
    torch.manual_seed(d_filter + S)
    params = orc.init_params(d_filter=d_filter, n_layers=n_layers, seed=3)
    o, d = orc.synthetic_rays(5)                      # 25 rays: not a multiple of 4 -> ragged last group
    d = d * (0.8 + 0.4 * torch.rand(d.shape[0], 1))
    t = torch.rand(o.shape[0], 1) * 20.
    z = orc.stratified_z(o, d, orc.linspace_t_vals(S), torch.tensor(1.3), torch.tensor(1.0))
    ref = orc.render_pass(params, o, d, t, z)
    dist_pts = ref['points'].pow(2).sum(-1).pow(0.5)
    out = ops.emission_render_fwd(pack(ops, params), dev(o), dev(d), dev(t), dev(z), reg_radius=1.2,
                                  want_raw=True, want_epilogues=True)
    torch.cuda.synchronize()
    assert (out['raw'].cpu() - ref['raw']).abs().max().item() < 2e-5
    # per-ray outputs: the north-star bound per ray; (N, S) intermediates: relative to the row-independent maximum
    assert gate_units(out['image'], ref['image']) <= 1.0
    assert gate_units(out['height_map'], (ref['weights'] * dist_pts).sum(-1)) <= 1.0
    assert gate_units(out['absorption_map'], (1 - ref['regularizing_quantity']).sum(-1), floor=S * 6e-8) <= 1.0
    assert rel_err(out['weights'], ref['weights']) < 1e-4
    assert rel_err(out['absorption'], ref['regularizing_quantity']) < 1e-4
    reg = torch.relu(dist_pts - 1.2) * (1 - ref['regularizing_quantity'])
    # these rays have |d| != 1, so the samples sit up to 40 radii from the origin and relu(|p| - 1.2) multiplies the error of
    # (1 - absorption) by up to 40: 1e-4 of the maximum needs the exact arithmetic; the fp8-correction mode (raw output
    # within 2e-5, asserted above) gets 4e-4 on this amplified quantity (measured over the 15 cases: <= 1.9e-4; exact: <= 4.0e-5)
    tol = 1e-4 if precision == 'exact' else 4e-4
    e_reg = ((out['regularization'].cpu() - reg).abs().max() / reg.abs().max()).item()
    print(f'regularization ({precision}): measured {e_reg:.2e} of its maximum, bound {tol:.0e}')
    assert (out['regularization'].cpu() - reg).abs().max().item() <= tol * reg.abs().max().item() + 1e-7


def test_render_pass_golden_mlp(ops, precision):
    """MLP output against the REFERENCE's own output (fixture g2), large time coordinates included."""
    g = load_golden('g2_mlp')
    params = params_from_golden(g, 'net__')
    x = g['x']                                   # (256, 4) arbitrary points: drive them through rays with d = x, z = 1
    n = x.shape[0]
    o = torch.zeros(n, 3)
    z = torch.ones(n, 2)                         # S = 2, both samples at z = 1 -> point = x
    out = ops.emission_render_fwd(pack(ops, params), dev(o), dev(x[:, :3].contiguous()), dev(x[:, 3:].contiguous()),
                                  dev(z), reg_radius=1.2, want_raw=True)
    raw = out['raw'].cpu()
    assert (raw[:, 0] - g['inferences']).abs().max().item() < 2e-5
    assert (raw[:, 1] - g['inferences']).abs().max().item() < 2e-5


def _resample_close(got, want, z_vals):
    """Inverse-CDF samples agree to 6.2e-5 (4 ulp of z ~ 215) except where the reference itself is
    discontinuous: sampling.py:164-165 replaces a CDF step `denom < 1e-5` by 1, and an empty bin has
    denom = 1e-5/sum(w + 1e-5) ~ 0.99994e-5, i.e. within rounding noise (6e-8) of the threshold.  A sample that lands
    in such a bin (probability ~1e-5 per sample and bin) may take either branch; both stay inside the bin.  Allow
    at most 0.1 % such samples, each within one coarse bin width."""
    diff = (got.cpu() - want).abs()
    bin_width = (z_vals[:, 1:] - z_vals[:, :-1]).max().item()
    assert diff.max().item() <= bin_width * 1.001
    assert (diff > 6.2e-5).float().mean().item() <= 1e-3


def test_hier_resample_vs_golden(ops):
    g = load_golden('g4_hierarchical')
    for sf, kz, kc in ((32, 'new_z', 'z_comb'), (48, 'new_z48', 'z_comb48')):
        u = torch.linspace(0., 1., sf)
        nz, zc = ops.hier_resample(dev(g['z_vals']), dev(g['weights']), dev(u))
        _resample_close(nz, g[kz], g['z_vals'])
        _resample_close(zc, g[kc], g['z_vals'])
        assert (zc[:, 1:] >= zc[:, :-1]).all()
        # the merged row contains every coarse z and every new sample: same multiset as sort(cat)
        both = torch.sort(torch.cat([g['z_vals'], nz.cpu()], -1), -1)[0]
        assert torch.equal(both, zc.cpu())
    nz, zc = ops.hier_resample(dev(g['z_vals'][:4].contiguous()), dev(g['weights_deg']), dev(torch.linspace(0., 1., 32)))
    _resample_close(nz, g['new_z_deg'], g['z_vals'])
    _resample_close(zc, g['z_comb_deg'], g['z_vals'])
    # perturb=True path: per-ray random u (unsorted) vs oracle
    u = torch.rand(64, 32, generator=torch.Generator().manual_seed(5))
    nz_o, zc_o = orc.hierarchical_z(g['z_vals'], g['weights'], 32, u=u)
    nz, zc = ops.hier_resample(dev(g['z_vals']), dev(g['weights']), dev(u))
    _resample_close(nz, nz_o, g['z_vals'])
    _resample_close(zc, zc_o, g['z_vals'])
    assert (zc[:, 1:] >= zc[:, :-1]).all()


def test_cpu_tensor_is_refused(ops):
    from sunerf_hip import SunerfHipError
    with pytest.raises(SunerfHipError):
        ops.sample_z(ops.SAMPLER_STRATIFIED, torch.zeros(4, 3), torch.ones(4, 3), torch.linspace(0, 1, 8), 1.3, 1.0)


@pytest.mark.parametrize('d_filter,n_layers', [(256, 8), (128, 4), (64, 2), (512, 3)])
def test_half_precision_follows_emulated_oracle(ops, d_filter, n_layers, monkeypatch):
    """SUNERF_PRECISION_HALF (opt-in; BASELINE config 3's "bf16 weights on MFMA" class, here fp16 operands): parity with the
    oracle that rounds the same operands to fp16 (emulated oracle at 1e-4), and the deviation from the fp32
    evaluation reported / bounded loosely -- this mode is not inside the 1e-4-vs-fp32 gate and never the default."""
    params = orc.init_params(d_filter=d_filter, n_layers=n_layers, seed=5)
    o, d = orc.synthetic_rays(6)
    t = torch.rand(o.shape[0], 1, generator=torch.Generator().manual_seed(2))
    z = orc.stratified_z(o, d, orc.linspace_t_vals(64), torch.tensor(1.3), torch.tensor(1.0))
    ref16 = orc.render_pass(params, o, d, t, z, half=True)
    ref32 = orc.render_pass(params, o, d, t, z)
    packed = ops.PackedMLP([dev(W) for W, _ in params], [dev(b) for _, b in params], precision=ops.PRECISION_HALF)
    out = ops.emission_render_fwd(packed, dev(o), dev(d), dev(t), dev(z), reg_radius=1.2, want_raw=True, want_epilogues=True)
    torch.cuda.synchronize()
    # a handful of fp16 roundings flip where the kernel's fp32 pre-activation and the oracle's differ in the last bits: one
    # flipped output of the last hidden layer moves a raw value by 2^-11 |w_out| ~ 4e-5
    assert (out['raw'].cpu() - ref16['raw']).abs().max().item() < 2e-4
    for k, r in (('image', ref16['image']), ('weights', ref16['weights']), ('absorption', ref16['regularizing_quantity'])):
        assert rel_err(out[k], r) < 1e-4, k
    dev32 = rel_err(out['image'], ref32['image'])
    print(f'HALF d={d_filter} L={n_layers}: image deviation from the fp32 evaluation {dev32:.2e}')
    assert 1e-6 < dev32 < 2e-2


def test_half_precision_against_the_committed_emulated_fixture(ops):
    """SURVEY G8 (fixture g12): HALF mode on the reference weights / inputs of g2 and g5b against the committed outputs of the
    emulating oracle, at 1e-4 of the tensor's scale (a flipped fp16 rounding of one hidden value moves raw by ~4e-5)."""
    g12, g2, g5b = load_golden('g12_half_emulated'), load_golden('g2_mlp'), load_golden('g5b_emission_d256')
    params = params_from_golden(g2, 'net__')
    packed = ops.PackedMLP([dev(W) for W, _ in params], [dev(b) for _, b in params], precision=ops.PRECISION_HALF)
    x = g2['x']
    out = ops.emission_render_fwd(packed, dev(torch.zeros(x.shape[0], 3)), dev(x[:, :3].contiguous()), dev(x[:, 3:].contiguous()),
                                  dev(torch.ones(x.shape[0], 2)), reg_radius=1.2, want_raw=True)
    # (g2 is a 64-wide net with outputs of order 0.05: one hidden fp16 value rounded the other way -- the kernel's fp32
    # pre-activation and the oracle's float64 one differ in the last bits -- moves an output by 2^-11 |w_out| ~ 4e-5)
    assert (out['raw'][:, 0].cpu() - g12['g2__inferences_half']).abs().max().item() < 5e-5
    params = params_from_golden(g5b, 'sd__coarse_model__')
    packed = ops.PackedMLP([dev(W) for W, _ in params], [dev(b) for _, b in params], precision=ops.PRECISION_HALF)
    out = ops.emission_render_fwd(packed, dev(g5b['rays_o']), dev(g5b['rays_d']), dev(g5b['times']),
                                  dev(g5b['out__z_vals_stratified']), reg_radius=1.2, want_raw=True)
    torch.cuda.synchronize()
    assert (out['raw'].cpu() - g12['g5b__raw_half']).abs().max().item() < 2e-4
    assert rel_err(out['image'], g12['g5b__image_half']) < 1e-4
    assert rel_err(out['weights'], g12['g5b__weights_half']) < 1e-4
    assert rel_err(out['absorption'], g12['g5b__absorption_half']) < 1e-4
