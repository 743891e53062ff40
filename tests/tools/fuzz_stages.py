"""Randomised sweep of the non-MLP stages against the CPU oracle: sample placement (both samplers, with and without jitter),
hierarchical resampling (perturb on / off, degenerate weights), the emission integral forward and backward on random raw, and the
density-temperature integral forward (random channel subsets incl. absent ones).  Development aid, run on the GPU box."""
import os
import random
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, '2024-hl-spi3s-sunerf_amd')); sys.path.insert(0, os.path.join(R, 'oracle')); sys.path.insert(0, os.path.join(R, 'tests'))
import sunerf_oracle as orc   # noqa: E402
from conftest import load_golden   # noqa: E402
from sunerf_hip import ops    # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g6 = load_golden('g6_dt_e2e')
logte, resp = g6['aia_logte'].float(), (g6['aia_tresp'] * 2.9).float()
bad = 0


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


for case in range(n_cases):
    gen = torch.Generator().manual_seed(case)
    n = rng.choice([1, 2, 7, 33, 64, 129, 300])
    S = rng.choice([3, 4, 8, 31, 32, 33, 64, 100, 128, 200])
    side = int(n ** 0.5) + 1
    o, d = orc.synthetic_rays(side)
    o, d = o[:n].contiguous(), d[:n].contiguous()
    if rng.random() < 0.3:
        d = d * (0.5 + torch.rand(n, 1, generator=gen))                 # non-unit directions
    msgs = []
    # --- samplers
    kind = rng.choice(['stratified', 'spherical'])
    t_rand = torch.rand(n, S, generator=gen) if rng.random() < 0.5 else None
    dist = 1.3 if kind == 'stratified' else 2.0
    zfn = orc.stratified_z if kind == 'stratified' else orc.spherical_z
    z_ref = zfn(o, d, orc.linspace_t_vals(S), torch.tensor(dist), torch.tensor(1.0), t_rand)
    z = ops.sample_z(ops.SAMPLER_STRATIFIED if kind == 'stratified' else ops.SAMPLER_SPHERICAL, o.cuda(), d.cuda(),
                     torch.linspace(0., 1., S).cuda(), dist, 1.0, t_rand=None if t_rand is None else t_rand.cuda()).cpu()
    finite = torch.isfinite(z_ref)
    if not torch.equal(torch.isfinite(z), finite) or (finite.any() and (z[finite] - z_ref[finite]).abs().max().item() > 4e-5):
        msgs.append(f'sampler {kind}')
    z_ref = torch.nan_to_num(z_ref, nan=215.0)
    z_ref, _ = torch.sort(z_ref, -1)
    # --- emission integral forward / backward on random raw
    raw = torch.randn(n, S, 2, generator=gen) * rng.choice([0.1, 1.0, 3.0])
    raw_l = raw.clone().requires_grad_(True)
    ref = orc.emission_integral(raw_l, z_ref, d)
    g_img, g_w, g_a = torch.randn(n, generator=gen), torch.randn(n, S, generator=gen), torch.randn(n, S, generator=gen)
    ((ref['image'][:, 0] * g_img).sum() + (ref['weights'] * g_w).sum() + (ref['regularizing_quantity'] * g_a).sum()).backward()
    img, w, ab = ops.emission_integral_fwd(raw.cuda(), z_ref.cuda(), d.cuda())
    if rel(img.cpu(), ref['image'].detach()) > 2e-5 or rel(w.cpu(), ref['weights'].detach()) > 2e-5 or rel(ab.cpu(), ref['regularizing_quantity'].detach()) > 2e-5:
        msgs.append('integral fwd')
    g_raw = ops.emission_integral_bwd(raw.cuda(), z_ref.cuda(), d.cuda(), g_img.cuda(), g_w.cuda(), g_a.cuda()).cpu()
    if rel(g_raw, raw_l.grad) > 1e-4:
        msgs.append(f'integral bwd {rel(g_raw, raw_l.grad):.1e}')
    # --- hierarchical resampling
    wts = ref['weights'].detach().clone()
    mode = rng.choice(['plain', 'zeros', 'spike', 'perturb'])
    if mode == 'zeros':
        wts[: max(1, n // 2)] = 0.
    if mode == 'spike':
        wts.zero_(); wts[:, S // 2] = 1.
    nf = rng.choice([1, 2, 16, 33, 64, 128])
    u = torch.rand(n, nf, generator=gen) if mode == 'perturb' else torch.linspace(0., 1., nf)
    nz_ref, zc_ref = orc.hierarchical_z(z_ref, wts, nf, u=u if mode == 'perturb' else None)
    nz, zc = ops.hier_resample(z_ref.cuda(), wts.cuda(), u.cuda())
    bw = (z_ref[:, 1:] - z_ref[:, :-1]).abs().max().item()
    dd = (nz.cpu() - nz_ref).abs()
    if dd.max().item() > bw * 1.001 + 1e-4 or (dd > 1e-4).float().mean().item() > 5e-3 or not bool((zc.cpu()[:, 1:] >= zc.cpu()[:, :-1]).all()):
        msgs.append(f'resample {mode} max {dd.max().item():.1e} frac {(dd > 1e-4).float().mean().item():.1e}')
    # --- density-temperature integral forward
    W = 7
    wl = torch.tensor([94., 131., 171., 193., 211., 304., 335.]).repeat(n, 1)
    wl[torch.rand(n, W, generator=gen) < 0.2] = 0.
    inf = torch.stack([torch.rand(n, S, generator=gen) * 3 + 17.5, torch.rand(n, S, generator=gen) * 2.5 + 4.2], -1)
    la = {str(w_): torch.tensor(rng.choice([-1e-9, 1e-9, 3e-9])) for w_ in orc.AIA_WAVELENGTHS}
    vc = torch.tensor(1.3)
    ref_dt = orc.dt_integral(inf, la, vc, z_ref, wl, logte, resp, 1e-10)
    out = ops.dt_integral_fwd(inf.cuda() - torch.tensor([10.0, 5.0]).cuda(), z_ref.cuda(), o.cuda(), d.cuda(), wl.cuda(), logte.cuda(), resp.cuda(),
                              torch.stack([la[str(w_)] for w_ in orc.AIA_WAVELENGTHS]).cuda(), vc.cuda(), 10.0, 5.0, 1e-10, 1.25)
    ri, gi = ref_dt['image'], out['image'].cpu()
    if not torch.equal(gi[wl == 0], torch.zeros_like(gi[wl == 0])) or (ri.abs().max() > 0 and rel(gi, ri) > 3e-4):
        msgs.append(f'dt fwd {rel(gi, ri):.1e}')
    bad += bool(msgs)
    print(('BAD ' if msgs else 'ok  ') + f'case {case:3d}: rays={n:3d} S={S:3d} {kind:10s} jitter={t_rand is not None} resample={mode}/{nf}' + (' -> ' + '; '.join(msgs) if msgs else ''), flush=True)
print(f'{n_cases - bad} of {n_cases} cases inside the gates')
sys.exit(1 if bad else 0)
