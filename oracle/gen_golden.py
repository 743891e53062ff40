"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REAL reference (container only).

Run ``python oracle/gen_golden.py`` in the build container (``/root/reference`` present).  The fixtures are
data only: seeded inputs, explicit weights and the outputs the reference's own code produced for them.
The reference has no tests / golden vectors of its own, so these are the pins for the
CPU restatement ``oracle/sunerf_oracle.py`` and, through it, for the HIP path.

Fixtures:
  g1_sampler        StratifiedSampler / SphericalSampler z_vals + points (hit and miss rays, perturb via
                    a recorded t_rand)
  g2_mlp            PositionalEncoding + NeRF(d_filter=64) on 256 points: enc, inferences
  g3_integral       EmissionRadiativeTransfer.raw2outputs on random raw: image, weights, absorption
  g4_hierarchical   HierarchicalSampler on g3's weights: new_z_samples, z_vals_combined
  g5_emission_e2e   shimmed two-pass emission forward: all 8 outputs + training loss (sunerf.py:110-120)
                    + gradients of every parameter, d_filter=64
  g5b_emission_d256 same at d_filter=256 (weights stored), forward outputs + loss only
  g6_dt_e2e         DensityTemperatureRadiativeTransfer two-pass forward + loss + gradients ("Interp1D restated")
  g7_train_step     loss section of training_step (reference ImageAsinhScaling + nn.MSELoss, sunerf.py:110-122) with
                    its image gradients, and 4 steps of clip_grad_norm_(0.5) + torch.optim.Adam(lr=1e-4) on three
                    tensors (what Lightning 1.9.3's gradient_clip_val + configure_optimizers, sunerf.py:31, execute)
  g8_observer_rays  pose_spherical (with and without shift) and get_rays on a regular and on a distorted pixel grid
  g9_simple_star    SimpleStar.forward on 400 points and DensityTemperatureRadiativeTransfer(model=SimpleStar) two-pass
                    forward ("Interp1D restated" like g6)
  g10_reference_state  a .snf written from the reference's own classes (pickled rendering module + data config) and the
                    outputs the reference renders from those weights
  g11_trained       the (shimmed) reference TRAINED on the CPU for 400 steps of its own recipe (Adam + ExponentialLR as
                    sunerf.py:30-40 with lr_config start 1e-3, clip_grad_norm_ 0.5 as run_emission.py:72, loss
                    sunerf.py:110-120) on a limb-brightened synthetic target, then rendered: weights away from their
                    initialisation (the fp8-correction arithmetic had only been proven on fresh nn.Linear inits)
  g12_half_emulated SURVEY G8 ("emulated low-precision variants of G2 / G5"): NOT reference output -- the reference has no
                    low-precision path.  The oracle's emulation of the opt-in HALF arithmetic (fp16 operands of every
                    product, exact sums: oracle/sunerf_oracle.py mlp_forward_half) applied to the reference weights and inputs
                    of the committed g2 and g5b fixtures, with its deviation from the reference's fp32 outputs recorded.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import sunerf_oracle as orc  # noqa: E402  (only for the .genx table reader and the wavelength list)

GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
OUT = GOLDEN


def npz(name, **arrays):
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    print(f'{name}: {len(arrays)} arrays')


# ---- frozen input generator ---------------------------------------------------------------------------------------
# The fixture INPUTS are defined here and nowhere else, so that `python oracle/gen_golden.py` reproduces the committed
# files bit for bit whatever happens to the oracle's own helpers (round 1: `orc.synthetic_rays` was changed after g1-g6
# had been written and the script silently stopped regenerating them).  Two camera conventions are frozen:
#   'plain'    rot_theta @ rot_phi @ trans (no axis permutation): the rays of g1, g3-g6 (any valid ray set pins a sampler /
#              renderer; these happen to be the ones the committed outputs belong to)
#   'observer' the reference's pose_spherical incl. the axis permutation (coordinate_transformation.py:36-54): g9, g10
def _pose(theta, phi, radius, convention):
    T = lambda rows: torch.Tensor(rows).float()   # noqa: E731
    trans = T([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]])
    rphi = T([[1, 0, 0, 0], [0, np.cos(phi), -np.sin(phi), 0], [0, np.sin(phi), np.cos(phi), 0], [0, 0, 0, 1]])
    rtheta = T([[np.cos(theta), 0, -np.sin(theta), 0], [0, 1, 0, 0], [np.sin(theta), 0, np.cos(theta), 0], [0, 0, 0, 1]])
    if convention == 'plain':
        return rtheta @ (rphi @ trans)
    c2w = T([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    c2w = trans @ c2w
    c2w = rphi @ c2w
    c2w = rtheta @ c2w
    return T([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w


def fixture_rays(resolution, convention, fov_half_rad=1.1 * 960. / 206264.806, theta=-0.3, phi=0.1, radius=215.032):
    """(R*R, 3) origins and directions on a square helioprojective grid (the formula of data/ray_sampling.py:11-35)."""
    c2w = _pose(theta, phi, radius, convention)
    lin = torch.linspace(-fov_half_rad, fov_half_rad, resolution, dtype=torch.float64)
    Ty, Tx = torch.meshgrid(lin, lin, indexing='ij')
    directions = torch.stack([torch.sin(Tx), -torch.sin(Ty) * torch.cos(Tx), -torch.cos(Tx) * torch.cos(Ty)], -1).to(torch.float32)
    rays_d = torch.sum(directions[..., None, :] * c2w[:3, :3], dim=-1).reshape(-1, 3)
    rays_o = c2w[:3, -1].expand(rays_d.shape).contiguous()
    return rays_o, rays_d.contiguous()


def test_rays(n_side, seed, convention='plain'):
    """A mix of rays that hit the disk and rays that traverse the full slab, non-unit directions."""
    o, d = fixture_rays(n_side, convention)
    g = torch.Generator().manual_seed(seed)
    d = d * (0.8 + 0.4 * torch.rand(d.shape[0], 1, generator=g))  # non-unit |d| exercises dists*|d|
    t = torch.rand(d.shape[0], 1, generator=g)
    return o.contiguous(), d.contiguous(), t


def state_arrays(prefix, module):
    return {prefix + k.replace('.', '__'): v for k, v in module.state_dict().items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(7)
    torch.set_num_threads(1)
    ref = ref_import.import_reference()
    S = ref.train.sampling
    M = ref.model.model
    Emission = ref_import.shimmed_emission_class()

    # ---- G1 samplers
    o, d, t = test_rays(8, 1)
    st = S.StratifiedSampler(Rs_per_ds=1.0, distance=1.3, n_samples=32, perturb=False)
    out = st(o, d)
    sp = S.SphericalSampler(Rs_per_ds=1.0, distance=2.0, n_samples=32, perturb=False)
    # spherical needs rays that cross the 2 Rs sphere: tighter field of view
    o2, d2 = fixture_rays(8, 'plain', fov_half_rad=0.4 * 960. / 206264.806 * 2.0)
    out_sp = sp(o2, d2)
    # perturb=True with a recorded t_rand: monkeypatch torch.rand for the call
    t_rand = torch.rand(64, 32, generator=torch.Generator().manual_seed(3))
    st_p = S.StratifiedSampler(Rs_per_ds=1.0, distance=1.3, n_samples=32, perturb=True)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: t_rand
    try:
        out_p = st_p(o, d)
    finally:
        torch.rand = real_rand
    st_rs = S.StratifiedSampler(Rs_per_ds=0.5, distance=1.3, n_samples=16, perturb=False)
    out_rs = st_rs(o * 2, d)  # Rs_per_ds != 1: lengths in units of 2 solar radii... origin scaled to match
    npz('g1_sampler', rays_o=o, rays_d=d, t_vals=st.t_vals, z_vals=out['z_vals'], points=out['points'],
        rays_o_sph=o2, rays_d_sph=d2, z_vals_sph=out_sp['z_vals'], points_sph=out_sp['points'],
        t_rand=t_rand, z_vals_perturb=out_p['z_vals'],
        rays_o_rs=o * 2, t_vals_rs=st_rs.t_vals, z_vals_rs=out_rs['z_vals'])

    # ---- G2 encoder + MLP
    torch.manual_seed(7)
    net = M.NeRF(d_input=4, d_output=2, n_layers=8, d_filter=64)
    x = torch.cat([torch.randn(256, 3) * 1.2, torch.rand(256, 1) * 30.], -1)
    enc = net.in_layer[0](x)
    inf = net(x)['inferences']
    npz('g2_mlp', x=x, enc=enc, inferences=inf, **state_arrays('net__', net))

    # ---- G3 emission integral on random raw
    g = torch.Generator().manual_seed(11)
    raw = torch.randn(64, 32, 2, generator=g)
    raw.requires_grad_(True)
    em = Emission(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 32, 'perturb': False},
                  hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 32},
                  model_config={'d_filter': 64})
    r = em.raw2outputs(raw=raw, z_vals=out['z_vals'], rays_d=d)
    gw = torch.randn(64, 32, generator=g)
    (r['image'].sum() + (r['weights'] * gw).sum()).backward()
    npz('g3_integral', raw=raw, z_vals=out['z_vals'], rays_d=d, image=r['image'], weights=r['weights'],
        absorption=r['regularizing_quantity'], grad_probe=gw, grad_raw=raw.grad)

    # ---- G4 hierarchical sampler
    hs = S.HierarchicalSampler(n_samples=32, perturb=False)
    h = hs(o, d, out['z_vals'], r['weights'].detach())
    hs48 = S.HierarchicalSampler(n_samples=48, perturb=False)
    h48 = hs48(o, d, out['z_vals'], r['weights'].detach())
    # degenerate: all-zero weights (uniform pdf) and a one-hot weight row
    wdeg = torch.zeros(4, 32)
    wdeg[1, 7] = 1.0
    wdeg[2, 1] = 0.5
    wdeg[2, 30] = 0.5
    wdeg[3] = 1.0 / 32
    hdeg = hs(o[:4], d[:4], out['z_vals'][:4], wdeg)
    npz('g4_hierarchical', z_vals=out['z_vals'], weights=r['weights'], new_z=h['new_z_samples'],
        z_comb=h['z_vals'], new_z48=h48['new_z_samples'], z_comb48=h48['z_vals'],
        weights_deg=wdeg, new_z_deg=hdeg['new_z_samples'], z_comb_deg=hdeg['z_vals'])

    # ---- G5 end-to-end emission (d_filter=64) with loss and grads
    def e2e(name, d_filter, n_side, n_c, n_f, with_grads):
        torch.manual_seed(7)
        mod = Emission(Rs_per_ds=1.0,
                       sampling_config={'type': 'stratified', 'n_samples': n_c, 'perturb': False},
                       hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': n_f},
                       model_config={'d_filter': d_filter})
        o, d, t = test_rays(n_side, 5)
        outputs = mod(o, d, t)
        target = torch.rand(o.shape[0], 1, generator=torch.Generator().manual_seed(1))
        scaling = ref.train.scaling.ImageAsinhScaling(vmax=1, a=0.005)
        mse = torch.nn.MSELoss()
        tgt = scaling(target)
        coarse_loss = mse(scaling(outputs['coarse_image']), tgt)
        fine_loss = mse(scaling(outputs['fine_image']), tgt)
        reg = outputs['regularization'].mean()
        loss = 1.0 * (coarse_loss + fine_loss) + 1.0 * reg
        arrays = dict(rays_o=o, rays_d=d, times=t, target=target, loss=loss, coarse_loss=coarse_loss,
                      fine_loss=fine_loss, reg_loss=reg, t_vals=mod.sampler.t_vals)
        arrays.update({'out__' + k: v for k, v in outputs.items()})
        arrays.update(state_arrays('sd__', mod))
        if with_grads:
            loss.backward()
            arrays.update({'grad__' + k.replace('.', '__'): p.grad for k, p in mod.named_parameters()})
        npz(name, **arrays)

    e2e('g5_emission_e2e', 64, 6, 32, 32, True)
    e2e('g5b_emission_d256', 256, 4, 32, 64, False)

    # ---- G6 density-temperature head (run_density_temperature.py path): NeRF_DT + DT integral, 7 wavelengths, a row
    #      block with absent channels (wavelength 0); loss = MSE (sunerf.py:187-195); Interp1D restated (parity unpinned)
    DT = ref.rendering.density_temperature.DensityTemperatureRadiativeTransfer
    torch.manual_seed(7)
    dt = DT(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 16, 'perturb': False},
            hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 16},
            model_config={'d_filter': 64}, model=M.NeRF_DT, device=torch.device('cpu'), pixel_intensity_factor=1e17)
    with torch.no_grad():      # make absorption and both relu branches matter
        for m_ in (dt.coarse_model, dt.fine_model):
            for k_, v_ in zip(m_.log_absortpion.keys(), (2e-6, 4e-6, -1e-6, 3e-6, 5e-6, 1e-6, 2e-6)):
                m_.log_absortpion[k_].fill_(v_)
            m_.volumetric_constant.fill_(1.5)
            m_.out_layer.weight.mul_(6.0)
    o, d, t = test_rays(4, 9)
    wl = torch.tensor([94., 131., 171., 193., 211., 304., 335.]).repeat(o.shape[0], 1)
    wl[3:6, 2] = 0.
    wl[5:9, 5] = 0.
    outputs = dt(o, d, t, wl)
    target = torch.rand(o.shape[0], 7, generator=torch.Generator().manual_seed(2))
    mse = torch.nn.MSELoss()
    loss = mse(outputs['coarse_image'], target) + mse(outputs['fine_image'], target) + outputs['regularization'].mean()
    loss.backward()
    logte, tresp = orc.read_aia_response_genx(os.path.join(ref_import.REFERENCE_ROOT, 'sunerf/data/aia_temp_resp.genx'))
    arrays = dict(rays_o=o, rays_d=d, times=t, wavelengths=wl, target=target, loss=loss, t_vals=dt.sampler.t_vals,
                  aia_logte=logte, aia_tresp=tresp, aia_exp_time=2.9, pixel_intensity_factor=1e17)
    arrays.update({'out__' + k: v for k, v in outputs.items()})
    arrays.update(state_arrays('sd__', dt))
    arrays.update({'grad__' + k.replace('.', '__'): p.grad for k, p in dt.named_parameters()})
    npz('g6_dt_e2e', **arrays)
    gen_g7(ref)
    gen_g8(ref)
    gen_g9(ref)
    gen_g10(ref)
    gen_g11(ref)
    gen_g12(ref)
    ref_import.release_reference()


def gen_g7(ref):
    """Loss section + optimiser of the reference training step (pytorch_lightning itself is not importable here: the
    two torch calls it makes for ``gradient_clip_val=0.5`` and ``optimizer.step()`` are issued directly)."""
    g = torch.Generator().manual_seed(11)
    n, s = 257, 48
    scaling = ref.train.scaling.ImageAsinhScaling(vmax=1, a=0.005)
    coarse = (torch.rand(n, 1, generator=g) * 1.5).requires_grad_(True)
    fine = (torch.rand(n, 1, generator=g) * 1.5).requires_grad_(True)
    target = torch.rand(n, 1, generator=g)
    reg = (torch.rand(n, s, generator=g) * 1e-2).requires_grad_(True)
    mse = torch.nn.MSELoss()
    lam_img, lam_reg = 1.0, 0.5
    ts = scaling(target)
    coarse_loss, fine_loss = mse(scaling(coarse), ts), mse(scaling(fine), ts)
    reg_loss = reg.mean()
    loss = lam_img * (coarse_loss + fine_loss) + lam_reg * reg_loss
    loss.backward()
    psnr = -10. * torch.log10(fine_loss.detach())
    arrays = dict(coarse=coarse, fine=fine, target=target, reg=reg, lambda_image=lam_img, lambda_regularization=lam_reg,
                  vmax=1.0, a=0.005, loss=loss, coarse_loss=coarse_loss, fine_loss=fine_loss, reg_loss=reg_loss, psnr=psnr,
                  g_coarse=coarse.grad, g_fine=fine.grad, g_reg=reg.grad)
    # optimiser: three tensors, four steps; gradient scales chosen so that steps 0 and 2 clip and 1 and 3 do not
    shapes = [(64, 84), (64,), (2, 64)]
    params = [torch.nn.Parameter(torch.randn(sh, generator=g) * 0.1) for sh in shapes]
    opt = torch.optim.Adam(params, lr=1e-4)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=(1e-5 / 1e-4) ** (1 / 1e6))
    for i, p in enumerate(params):
        arrays[f'p0_{i}'] = p.detach().clone()
    for step, scale in enumerate([1.0, 1e-3, 0.2, 1e-4]):
        for i, p in enumerate(params):
            p.grad = torch.randn(p.shape, generator=g) * scale
            arrays[f'grad{step}_{i}'] = p.grad.clone()
        total = torch.nn.utils.clip_grad_norm_(params, 0.5)
        opt.step()
        if sched.get_last_lr()[0] > 5e-5:
            sched.step()
        arrays[f'norm{step}'] = total
        arrays[f'lr{step + 1}'] = sched.get_last_lr()[0]
        for i, p in enumerate(params):
            arrays[f'p{step + 1}_{i}'] = p.detach().clone()
            arrays[f'clipped{step}_{i}'] = p.grad.clone()
    npz('g7_train_step', **arrays)


def gen_g8(ref):
    """Observer poses and rays: the reference's pose_spherical (coordinate_transformation.py:36-54) and get_rays
    (data/ray_sampling.py:7-36) on a helioprojective pixel grid with per-pixel distortion (stands in for a real WCS;
    sunpy is not installed, get_rays only calls ``img_coords.Tx/Ty.to_value(u.rad)``)."""
    import importlib
    ct = importlib.import_module('sunerf.train.coordinate_transformation')
    rs = importlib.import_module('sunerf.data.ray_sampling')

    class _Angle:
        def __init__(self, a):
            self.a = a

        def to_value(self, unit):
            return self.a

    class _Coords:
        def __init__(self, tx, ty):
            self.Tx, self.Ty = _Angle(tx), _Angle(ty)

    rng = np.random.default_rng(5)
    h, w = 9, 13
    half = 1.1 * 960. / 206264.806
    ty, tx = np.meshgrid(np.linspace(-half, half, h), np.linspace(-half * 1.3, half * 1.3, w), indexing='ij')
    tx_d = tx + rng.normal(0, 1e-5, tx.shape)
    ty_d = ty + rng.normal(0, 1e-5, ty.shape)
    arrays = dict(tx_axis=tx[0], ty_axis=ty[:, 0], tx_pix=tx_d, ty_pix=ty_d)
    poses = {'a': (-0.3, 0.1, 215.032, None), 'b': (2.1, -0.7, 50.0, (0.01, -0.02, 0.03))}
    for name, (theta, phi, radius, shift) in poses.items():
        c2w = ct.pose_spherical(theta, phi, radius, shift).numpy()
        arrays[f'pose_{name}'] = np.array([theta, phi, radius] + list(shift or (0., 0., 0.)) + [0. if shift is None else 1.])
        arrays[f'c2w_{name}'] = c2w
        for grid, (gx, gy) in (('axis', (tx, ty)), ('pix', (tx_d, ty_d))):
            o, d = rs.get_rays(_Coords(gx, gy), c2w)
            arrays[f'rays_o_{name}_{grid}'] = o
            arrays[f'rays_d_{name}_{grid}'] = d
    npz('g8_observer_rays', **arrays)


def gen_g9(ref):
    """SimpleStar (stellar_model.py) on points inside / in the ramp / outside, and the two-pass DT render of
    DensityTemperatureRadiativeTransfer(model=SimpleStar) as evaluation/image_render.py:266-268 builds it."""
    import importlib
    sm = importlib.import_module('sunerf.model.stellar_model')
    DT = importlib.import_module('sunerf.rendering.density_temperature').DensityTemperatureRadiativeTransfer
    star = sm.SimpleStar()
    g = torch.Generator().manual_seed(9)
    dirs = torch.randn(400, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    radii = torch.cat([torch.rand(100, generator=g), 1 + 0.02 * torch.rand(100, generator=g),
                       1.02 + 0.5 * torch.rand(198, generator=g), torch.tensor([1.0, 1.02])])
    pts = torch.cat([dirs * radii[:, None], torch.rand(400, 1, generator=g)], -1)
    with torch.no_grad():
        out = star(pts)
    arrays = dict(points=pts, inferences=out['inferences'], t_photosphere=star.t_photosphere)
    arrays.update({'sp__' + k: v for k, v in star.stellar_parameters.items()})
    arrays.update({'la__' + k: v for k, v in star.log_absortpion.items()})
    cwd = os.getcwd()
    os.chdir(ref_import.REFERENCE_ROOT)          # density_temperature.py:131 reads the response table relative to cwd
    try:
        dt = DT(Rs_per_ds=1, model=sm.SimpleStar, model_config={}, device=torch.device('cpu'),
                sampling_config={'type': 'stratified', 'n_samples': 24, 'perturb': False},
                hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 24})
    finally:
        os.chdir(cwd)
    # the default absorption scalars (~20) make the corona opaque within the first sample (every image is exactly 0);
    # optical depths of order one need kappa ~ 1 / (rho_0 * path) ~ 1e-9
    with torch.no_grad():
        for m in (dt.coarse_model, dt.fine_model):
            for i, w in enumerate(orc.AIA_WAVELENGTHS):
                m.log_absortpion[str(w)].fill_((1 + i) * 1e-9)
            m.volumetric_constant.fill_(0.7)
    o, d = fixture_rays(6, 'observer')              # unit directions: monotonic z along the line of sight
    t = torch.rand(o.shape[0], 1, generator=g)
    wl = torch.tensor([[94., 131., 171., 193., 211., 304., 335.]]).repeat(o.shape[0], 1)
    wl[2:5, 1] = 0.
    with torch.no_grad():
        outputs = dt(o, d, t, wl)
    arrays.update({'la__' + k: v for k, v in dt.fine_model.log_absortpion.items()})
    logte, tresp = orc.read_aia_response_genx(os.path.join(ref_import.REFERENCE_ROOT, 'sunerf/data/aia_temp_resp.genx'))
    arrays.update(rays_o=o, rays_d=d, times=t, wavelengths=wl, t_vals=dt.sampler.t_vals, aia_logte=logte, aia_tresp=tresp,
                  aia_exp_time=2.9, pixel_intensity_factor=1e10, vol_c=dt.fine_model.volumetric_constant)
    arrays.update({'out__' + k: v for k, v in outputs.items()})
    npz('g9_simple_star', **arrays)


def gen_g10(ref):
    """A ``.snf`` state file as the reference's save_state writes it (sunerf.py:62-74): the pickled rendering module built
    from the reference's own classes + data configuration, and what the (shimmed) reference renders from those weights.
    Lets the loader mirror prove that reference-trained states load into the fused classes."""
    import datetime
    import importlib
    Emission = importlib.import_module('sunerf.rendering.emission').EmissionRadiativeTransfer
    torch.manual_seed(21)
    cfg = dict(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 16, 'perturb': False},
               hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 16}, model_config={'d_filter': 64})
    plain = Emission(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})
    state = {'rendering': plain,
             'data_config': {'type': 'emission', 'Rs_per_ds': 1.0, 'seconds_per_dt': 86400., 'wavelength': 193,
                             'ref_time': datetime.datetime(2022, 3, 1), 'resolution': (12, 12),
                             'wcs': {'shape': (12, 12), 'cdelt': (200., 200.)},
                             'times': [datetime.datetime(2022, 3, 1), datetime.datetime(2022, 3, 5)], 'cmap': 'gray'},
             'Rs_per_ds': 1.0, 'seconds_per_dt': 86400., 'ref_time': datetime.datetime(2022, 3, 1)}
    torch.save(state, os.path.join(OUT, 'g10_reference_state.snf'))
    shim = ref_import.shimmed_emission_class()(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})
    shim.load_state_dict(plain.state_dict())
    o, d, t = test_rays(5, seed=6, convention='observer')
    with torch.no_grad():
        outputs = shim(o, d, t)
        pts = torch.rand(33, 4, generator=torch.Generator().manual_seed(1)) * 2 - 1
        inf = plain.fine_model(pts)['inferences']
    arrays = dict(rays_o=o, rays_d=d, times=t, points=pts, inferences=inf)
    arrays.update({'out__' + k: v for k, v in outputs.items()})
    npz('g10_reference_state', **arrays)


def gen_g11(ref):
    """Trained-scale weights from the reference's own training recipe (d_filter = 64 keeps it to seconds on one CPU thread)."""
    torch.manual_seed(13)
    Emission = ref_import.shimmed_emission_class()
    mod = Emission(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 16, 'perturb': False},
                   hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 16}, model_config={'d_filter': 64})
    scaling = ref.train.scaling.ImageAsinhScaling(vmax=1, a=0.005)
    mse = torch.nn.MSELoss()
    lr_config = {'start': 1e-3, 'end': 1e-4, 'iterations': 1e4}
    opt = torch.optim.Adam(mod.parameters(), lr=lr_config['start'])
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=(lr_config['end'] / lr_config['start']) ** (1 / lr_config['iterations']))
    o_all, d_all = fixture_rays(48, 'observer')
    impact = torch.linalg.cross(o_all, d_all / d_all.norm(dim=-1, keepdim=True)).norm(dim=-1)

    def target_of(b, t):      # limb-brightened disk, exponential corona, slow modulation in time
        return torch.where(b < 1, 0.3 + 0.5 * b ** 4, 0.8 * torch.exp(-(b - 1) / 0.15)) * (1 + 0.3 * torch.sin(6.28 * t + 3 * b))

    g = torch.Generator().manual_seed(17)
    init = {k: v.clone() for k, v in mod.state_dict().items()}
    losses = []
    for step in range(400):
        idx = torch.randint(0, o_all.shape[0], (192,), generator=g)
        t = torch.rand(192, 1, generator=g)
        target = target_of(impact[idx], t[:, 0])[:, None]
        out = mod(o_all[idx], d_all[idx], t)
        tgt = scaling(target)
        loss = (mse(scaling(out['coarse_image']), tgt) + mse(scaling(out['fine_image']), tgt)) + out['regularization'].mean()
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(mod.parameters(), 0.5)
        opt.step()
        if sched.get_last_lr()[0] > 5e-5:
            sched.step()
        losses.append(loss.detach())
    # unit directions (what data/ray_sampling.py produces): with |d| > 1 the parametric hit distance falls below the near
    # plane, z runs backwards and the trained absorption overflows in the reference itself
    o, d = fixture_rays(7, 'observer')
    t = torch.rand(o.shape[0], 1, generator=torch.Generator().manual_seed(23))
    with torch.no_grad():
        outputs = mod(o, d, t)
    assert all(torch.isfinite(v).all() for v in outputs.values())
    sd = mod.state_dict()
    moved = max(((sd[k] - init[k]).abs().max() / init[k].abs().max()).item() for k in sd if k.endswith('weight'))
    arrays = dict(rays_o=o, rays_d=d, times=t, t_vals=mod.sampler.t_vals, loss_first=losses[0], loss_last=losses[-1],
                  max_relative_weight_change=moved)
    arrays.update({'out__' + k: v for k, v in outputs.items()})
    arrays.update(state_arrays('sd__', mod))
    npz('g11_trained', **arrays)


def gen_g12(_ref=None):
    """Emulated-HALF variants of g2 (network outputs) and g5b (coarse render pass).  Needs no reference import: inputs and
    weights are read from the COMMITTED g2 / g5b files (reference data), outputs come from the oracle's emulation."""
    def load(name):
        with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as f:
            return {k: torch.from_numpy(f[k]) for k in f.files}

    def params(g, prefix):
        return orc.params_from_state_dict({k[len(prefix):].replace('__', '.'): v for k, v in g.items() if k.startswith(prefix)}, '')
    torch.set_num_threads(1)
    g2, g5b = load('g2_mlp'), load('g5b_emission_d256')
    inf16 = orc.mlp_forward_half(params(g2, 'net__'), g2['x'])
    z = g5b['out__z_vals_stratified']
    p16 = orc.render_pass(params(g5b, 'sd__coarse_model__'), g5b['rays_o'], g5b['rays_d'], g5b['times'], z, half=True)
    dev = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()   # noqa: E731
    npz('g12_half_emulated', g2__inferences_half=inf16, g2__deviation_from_fp32=dev(inf16, g2['inferences']),
        g5b__raw_half=p16['raw'], g5b__image_half=p16['image'], g5b__weights_half=p16['weights'],
        g5b__absorption_half=p16['regularizing_quantity'],
        g5b__image_deviation_from_fp32=dev(p16['image'], g5b['out__coarse_image']))


def compare_with_committed(out_dir):
    """Bit-compares every array of every fixture in ``out_dir`` with the committed file of the same name.
    Returns a list of human-readable differences (empty = the script reproduces the committed fixtures)."""
    diffs = []
    for name in sorted(os.listdir(GOLDEN)):
        if not name.endswith('.npz'):
            continue
        new_path = os.path.join(out_dir, name)
        if not os.path.exists(new_path):
            diffs.append(f'{name}: not regenerated')
            continue
        a, b = np.load(os.path.join(GOLDEN, name), allow_pickle=False), np.load(new_path, allow_pickle=False)
        if sorted(a.files) != sorted(b.files):
            diffs.append(f'{name}: key sets differ: {sorted(set(a.files) ^ set(b.files))}')
            continue
        for k in a.files:
            x, y = a[k], b[k]
            if x.shape != y.shape or x.dtype != y.dtype or x.tobytes() != y.tobytes():
                diffs.append(f'{name}[{k}] differs')
    return diffs


def generate(out_dir=None, only=None):
    """Regenerates the fixtures into ``out_dir`` (default: tests/golden)."""
    global OUT
    OUT = GOLDEN if out_dir is None else out_dir
    try:
        if only:
            torch.manual_seed(7)
            torch.set_num_threads(1)
            ref = ref_import.import_reference()
            for a in only:
                {'g7': gen_g7, 'g8': gen_g8, 'g9': gen_g9, 'g10': gen_g10, 'g11': gen_g11, 'g12': gen_g12}[a](ref)
            ref_import.release_reference()
        else:
            main()
    finally:
        OUT = GOLDEN


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--out', default=None, help='write the fixtures here instead of tests/golden')
    ap.add_argument('--check', action='store_true',
                    help='regenerate into a scratch directory and bit-compare with the committed fixtures')
    ap.add_argument('only', nargs='*', help='regenerate only these (g7 g8 g9 g10 g11 g12)')
    args = ap.parse_args()
    if any(a not in ('g7', 'g8', 'g9', 'g10', 'g11', 'g12') for a in args.only):
        ap.error('only g7 g8 g9 g10 g11 g12 can be regenerated on their own')
    if args.check:
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            generate(tmp)
            diffs = compare_with_committed(tmp)
        print('\n'.join(diffs) if diffs else 'all committed fixtures reproduce bit for bit')
        sys.exit(1 if diffs else 0)
    generate(args.out, args.only)
