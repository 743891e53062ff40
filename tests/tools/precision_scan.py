"""Where do the forward arithmetics (FAST = fp16 head + fp8 corrections, EXACT = three fp16 products) sit against the
north-star gate as the weights leave their initial scale?  

For every case the SAME function is evaluated four ways on the same fp32 query points:
  f64   the oracle's aten sequence in float64 (what the reference computes, without its own rounding noise)
  ref   the oracle in float32 on the CPU (the reference arithmetic: its distance from f64 is the reference's own noise floor)
  fast  the HIP render pass, SUNERF_PRECISION_FAST
  exact the HIP render pass, SUNERF_PRECISION_EXACT
and the error of the last three against f64 is reported as the north-star per-ray bound
    max_ray |err| / (1e-4 |f64| + 1e-6 max|f64|)          (<= 1 passes)
for image / height_map / absorption_map and per element for weights.

Cases: default nn.Linear init with all hidden weights x s, s in {1, 2, 4, 8}; and the d = 256 module after K fused training
steps on a synthetic limb-brightened target (the weights a user will actually render with).

Usage (GPU box): python tools/precision_scan.py [--steps 2000] > gpurun_out/precision_scan.txt
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'), os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import sunerf_oracle as orc  # noqa: E402  (the checker)
from sunerf_hip import ops  # noqa: E402


def oracle_pass(params, o, d, t, z, dtype):
    """orc.render_pass on the reference's fp32 query points, arithmetic in ``dtype``."""
    pts = orc.points_on_rays(o, d, z)                     # fp32, as the reference forms them
    query = torch.cat([pts, t[:, None].repeat(1, pts.shape[1], 1)], -1).to(dtype)
    p = [(W.to(dtype), b.to(dtype)) for W, b in params]
    raw = orc.mlp_forward(p, query.view(-1, 4)).reshape(*query.shape[:-1], -1)
    out = orc.emission_integral(raw, z.to(dtype), d.to(dtype))
    dist = pts.to(dtype).pow(2).sum(-1).pow(0.5)
    return {'raw': raw, 'image': out['image'][:, 0], 'weights': out['weights'],
            'height_map': (out['weights'] * dist).sum(-1), 'absorption_map': (1 - out['regularizing_quantity']).sum(-1)}


def gate(err, ref):
    """max over elements of |err| / (1e-4 |ref| + 1e-6 max|ref|)"""
    return (err.abs() / (1e-4 * ref.abs() + 1e-6 * ref.abs().max())).max().item()


def hip_pass(params, o, d, t, z, precision):
    packed = ops.PackedMLP([W.cuda() for W, _ in params], [b.cuda() for _, b in params], precision=precision)
    out = ops.emission_render_fwd(packed, o.cuda(), d.cuda(), t.cuda(), z.cuda(), reg_radius=1.2, want_raw=True,
                                  want_epilogues=True)
    torch.cuda.synchronize()
    return {'raw': out['raw'].cpu(), 'image': out['image'][:, 0].cpu(), 'weights': out['weights'].cpu(),
            'height_map': out['height_map'].cpu(), 'absorption_map': out['absorption_map'].cpu()}


def report(tag, params, o, d, t, z):
    f64 = oracle_pass(params, o, d, t, z, torch.float64)
    rows = {'ref': oracle_pass(params, o, d, t, z, torch.float32),
            'fast': hip_pass(params, o, d, t, z, ops.PRECISION_FAST),
            'exact': hip_pass(params, o, d, t, z, ops.PRECISION_EXACT)}
    wmax = max(W.abs().max().item() for W, _ in params[1:-1]) if len(params) > 2 else 0.
    rowsum = max(W.abs().sum(1).max().item() for W, _ in params[1:-1]) if len(params) > 2 else 0.
    print(f'== {tag}: max|w_hidden| {wmax:.3f}, max row sum |w| {rowsum:.2f}, max|raw| {f64["raw"].abs().max().item():.2f}')
    for name, r in rows.items():
        raw_err = (r['raw'].double() - f64['raw']).abs().max().item()
        g = {k: gate(r[k].double() - f64[k], f64[k]) for k in ('image', 'weights', 'height_map', 'absorption_map')}
        print(f'   {name:5s} raw abs err {raw_err:.2e} | gate units (<= 1 passes): ' +
              '  '.join(f'{k} {v:.3f}' for k, v in g.items()))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--d', type=int, default=256)
    args = ap.parse_args()
    torch.manual_seed(0)
    o, d = orc.synthetic_rays(16)                          # 256 rays
    t = torch.rand(o.shape[0], 1) * 5.
    z = orc.stratified_z(o, d, orc.linspace_t_vals(128), torch.tensor(1.3), torch.tensor(1.0))
    for s in (1., 2., 4., 8.):
        params = orc.init_params(d_filter=args.d, n_layers=8, seed=3)
        params = [(W * s, b) if 0 < i < len(params) - 1 else (W, b) for i, (W, b) in enumerate(params)]
        report(f'init, hidden weights x {s:g}', params, o, d, t, z)
    for s in (2., 4.):
        params = orc.init_params(d_filter=args.d, n_layers=8, seed=3)
        params = [(W * s, b * s) if i < len(params) - 1 else (W, b) for i, (W, b) in enumerate(params)]
        report(f'init, in + hidden weights and biases x {s:g}', params, o, d, t, z)

    # ---- trained weights: the fused training step on a structured synthetic target ----
    from sunerf.model.sunerf import EmissionSuNeRFModule, fit_steps
    torch.manual_seed(7)
    mod = EmissionSuNeRFModule(Rs_per_ds=1.0, seconds_per_dt=86400., image_scaling_config={'vmax': 1, 'a': 0.005},
                               sampling_config={'type': 'stratified', 'n_samples': 64, 'perturb': False},
                               hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 64},
                               model_config={'d_filter': args.d}, lr_config={'start': 5e-4, 'end': 5e-5, 'iterations': 1e5}).cuda()
    mod.strict_finite_check = False
    oo, dd = orc.synthetic_rays(96)
    # target: limb-brightened disk + off-limb exponential fall-off of the ray's impact parameter, time-modulated
    b_imp = torch.linalg.cross(oo, dd / dd.norm(dim=-1, keepdim=True)).norm(dim=-1)
    n = oo.shape[0]

    def batch(i):
        g = torch.Generator().manual_seed(i)
        idx = torch.randint(0, n, (4096,), generator=g)
        tt = torch.rand(4096, 1, generator=g)
        b = b_imp[idx]
        tgt = torch.where(b < 1, 0.3 + 0.5 * b ** 4, 0.8 * torch.exp(-(b - 1) / 0.15)) * (1 + 0.3 * torch.sin(6.28 * tt[:, 0] + 3 * b))
        return {'tracing': {'rays': torch.stack([oo[idx], dd[idx]], 1).cuda(), 'time': tt.cuda(), 'target_image': tgt[:, None].cuda()}}

    done = 0
    for upto in sorted({args.steps // 10, args.steps // 3, args.steps}):
        if upto <= done:
            continue
        losses = fit_steps_resume(mod, (batch(i) for i in range(done, upto)))
        done = upto
        sd = {k: v.detach().cpu() for k, v in mod.rendering.state_dict().items()}
        for which in ('coarse_model.', 'fine_model.'):
            params = orc.params_from_state_dict(sd, which)
            report(f'{which[:-7]} model after {done} fused steps (loss {losses[-1].item():.4f})', params, o, d, t, z)


_opt = {}


def fit_steps_resume(module, batches):
    """fit_steps that keeps its optimiser between calls"""
    if 'o' not in _opt:
        (optimizer,), _ = module.configure_optimizers()
        optimizer.max_norm = 0.5
        _opt['o'] = optimizer
    optimizer = _opt['o']
    losses = []
    for i, b in enumerate(batches):
        optimizer.zero_grad()
        loss = module.training_step(b, i)
        loss.backward()
        optimizer.step(skip_if_positive=module.last_stats[5:6])
        module.on_train_batch_end()
        losses.append(loss.detach())
    return losses


if __name__ == '__main__':
    main()
