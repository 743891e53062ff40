"""End-to-end sanity run of the density-temperature path (BASELINE config 0 / 4 class): the target is rendered from the analytic
``SimpleStar`` field through the same DT integral (the way the reference produces its synthetic data, evaluation/image_render.py:266),
then ``DensityTemperatureSuNeRFModule`` (NeRF_DT 8 x 256, 64 + 128 samples, 7 channels) is trained on it with ``fit_steps``.
The AIA response table comes from the data fixture tests/golden/g9_simple_star.npz.   python tools/mini_train_dt.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, '2024-hl-spi3s-sunerf_amd'))
from sunerf.model.model import NeRF_DT                                            # noqa: E402
from sunerf.model.stellar_model import SimpleStar                                 # noqa: E402
from sunerf.model.sunerf import DensityTemperatureSuNeRFModule, fit_steps         # noqa: E402
from sunerf.rendering.density_temperature import DensityTemperatureRadiativeTransfer   # noqa: E402
from sunerf_hip import ops                                                        # noqa: E402
from sunerf_hip.rays import observer_rays                                         # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
fx = np.load(os.path.join(R, 'tests', 'golden', 'g9_simple_star.npz'))
table = (fx['aia_logte'], fx['aia_tresp'])
views = [observer_rays(64, theta=-0.3 + 0.785 * k, phi=0.1 * (k % 3 - 1)) for k in range(8)]
rays_o, rays_d = torch.cat([v[0] for v in views]), torch.cat([v[1] for v in views])
n = rays_o.shape[0]
times = torch.zeros(n, 1, device='cuda')
wl = torch.tensor([94., 131., 171., 193., 211., 304., 335.], device='cuda').repeat(n, 1)
cfg = dict(sampling_config={'type': 'stratified', 'n_samples': 64, 'perturb': True},
           hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 128, 'perturb': False})
star = DensityTemperatureRadiativeTransfer(Rs_per_ds=1, model=SimpleStar, model_config={}, response_table=table,
                                           **{k: dict(v) for k, v in cfg.items()}).cuda()
with torch.no_grad():
    for m in (star.coarse_model, star.fine_model):
        for w in ops.AIA_WAVELENGTHS:
            m.log_absortpion[str(w)].copy_(torch.from_numpy(fx[f'la__{w}']))
        m.volumetric_constant.copy_(torch.from_numpy(fx['vol_c']))
    target = star(rays_o, rays_d, times, wl)['image']
scale = target.abs().max().item()
print(f'target: {n} rays x 7 channels from SimpleStar, max {scale:.3e}', flush=True)
mod = DensityTemperatureSuNeRFModule(Rs_per_ds=1.0, seconds_per_dt=1.0, image_scaling_config={}, model=NeRF_DT,
                                     pixel_intensity_factor=1e10, response_table=table,
                                     model_config={'d_filter': 256}, lr_config={'start': 5e-4, 'end': 5e-5, 'iterations': steps},
                                     **{k: dict(v) for k, v in cfg.items()}).cuda()
mod.strict_finite_check = False
target = target / scale                                         # images of order one, as the reference's loaders normalise them
B = 2048


def batches(k):
    for _ in range(k):
        idx = torch.randint(0, n, (B,), device='cuda')
        yield {'tracing': {'rays': torch.stack([rays_o[idx], rays_d[idx]], 1), 'time': times[idx], 'target_image': target[idx],
                           'wavelength': wl[idx]}}


torch.cuda.synchronize(); t0 = time.perf_counter()
losses = fit_steps(mod, batches(steps))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ls = torch.stack(losses)
print('loss: first 10 steps %.4e, last 10 steps %.4e' % (ls[:10].mean().item(), ls[-10:].mean().item()))
mod.check_finite(mod.optimizer)
print(f'{steps} steps of {B} rays x (64 + 192) evaluations x 7 channels in {dt:.2f} s = {steps * B * 256 / dt:.3e} evaluations/s; '
      f'steps applied {mod.optimizer.step_count}; forward arithmetic {[ops.PRECISION_NAMES[m.packed().precision] for m in (mod.rendering.coarse_model, mod.rendering.fine_model)]}')
