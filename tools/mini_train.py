"""End-to-end sanity run of the training path on one MI355X: the emission module (8 x 256, 64 coarse samples + 128 resampled ones: 64 + 192 network evaluations per ray) trained with
``fit_steps`` -- fused loss, overlapped bucket, clip + Adam kernels, ExponentialLR, AUTO forward arithmetic re-probed every 64
parameter versions -- on an analytic target (limb-darkened disk + exponential corona seen from 8 longitudes), batches of 3072 rays
like config/sunerfs_simple_star.yaml:8.  Prints loss / PSNR every 100 steps and the rate.  python tools/mini_train.py [steps] [d_filter]"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, '2024-hl-spi3s-sunerf_amd'))
from sunerf.model.sunerf import EmissionSuNeRFModule, fit_steps   # noqa: E402
from sunerf_hip import ops                                        # noqa: E402
from sunerf_hip.rays import observer_rays                          # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 800
d_filter = int(sys.argv[2]) if len(sys.argv) > 2 else 256          # 512 = the reference's default width (model.py:16)
torch.manual_seed(0)
views = [observer_rays(96, theta=-0.3 + 0.785 * k, phi=0.1 * (k % 3 - 1)) for k in range(8)]
rays_o, rays_d = torch.cat([v[0] for v in views]), torch.cat([v[1] for v in views])
b = torch.linalg.cross(rays_o, rays_d).norm(dim=-1) / rays_d.norm(dim=-1)              # impact parameter in solar radii
target = torch.where(b < 1, 0.25 * torch.sqrt((1 - b * b).clamp_min(0)) + 0.06, 0.06 * torch.exp(-(b - 1) / 0.12))[:, None]
times = torch.zeros(rays_o.shape[0], 1, device=rays_o.device)
mod = EmissionSuNeRFModule(Rs_per_ds=1.0, seconds_per_dt=1.0, image_scaling_config={'vmax': 1, 'a': 0.005},
                           sampling_config={'type': 'stratified', 'n_samples': 64, 'perturb': True},
                           hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 128, 'perturb': True},
                           model_config={'d_filter': d_filter}, lr_config={'start': 5e-4, 'end': 5e-5, 'iterations': steps}).cuda()
mod.strict_finite_check = False                      # no host read inside the step; checked at the end
n, B = rays_o.shape[0], 3072


def batches(k):
    for _ in range(k):
        idx = torch.randint(0, n, (B,), device=rays_o.device)
        yield {'tracing': {'rays': torch.stack([rays_o[idx], rays_d[idx]], 1), 'time': times[idx], 'target_image': target[idx]}}


done, t_all = 0, 0.0
while done < steps:
    k = min(100, steps - done)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    losses = fit_steps(mod, batches(k)) if done == 0 else None
    if losses is None:                                # keep ONE optimiser / schedule across the blocks
        losses = []
        for i, batch in enumerate(batches(k)):
            mod.optimizer.zero_grad()
            loss = mod.training_step(batch, done + i)
            loss.backward()
            mod.optimizer.step(skip_if_positive=mod.last_stats[5:6])
            mod.on_train_batch_end()
            losses.append(loss.detach())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    done += k; t_all += dt
    st = mod.last_stats
    modes = [ops.PRECISION_NAMES[m.packed().precision] for m in (mod.rendering.coarse_model, mod.rendering.fine_model)]
    print(f'step {done:5d}: loss {torch.stack(losses).mean().item():.5f}  psnr {st[4].item():6.2f} dB  lr {mod.scheduler.get_last_lr()[0]:.2e}  '
          f'{k * B * (64 + 192) / dt:.3e} MLP evaluations/s  forward arithmetic {modes}', flush=True)
mod.check_finite(mod.optimizer)
print(f'{steps} steps of {B} rays x (64 coarse + 192 fine) evaluations in {t_all:.2f} s; optimiser steps applied: {mod.optimizer.step_count}')
