"""Is there anything to gain from running the forward of one micro-batch beside the backward of another?

The forward is bound by matrix issue / power (its clock drops while it writes the stash), the backward kernels by the
memory system: side by side on disjoint CUs (SUNERF_GRID_CAP_* knobs) they might add up to less than their sum.
Measures, on one box and in one process: forward alone (all CUs), backward alone (all CUs), each alone on its share of the
CUs, and both at once on two streams.  Synergy = 1 - t_both / (t_fwd_all + t_bwd_all).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', '2024-hl-spi3s-sunerf_amd'))
from sunerf.model.model import NeRF                      # noqa: E402
from sunerf.rendering.functional import emission_pass    # noqa: E402
from sunerf_hip import ops                               # noqa: E402
from sunerf_hip.rays import observer_rays                # noqa: E402
from sunerf_hip.train import training_loss               # noqa: E402

dev = torch.device('cuda', 0)
N, S, K = 16384, 128, 6
torch.manual_seed(7)
model = NeRF(d_input=4, d_output=2, n_layers=8, d_filter=256).to(dev)
rays_o, rays_d = observer_rays(256, theta=-0.3, device=dev)
times = torch.zeros(rays_o.shape[0], device=dev)
t_vals = torch.linspace(0., 1., S, device=dev)
target = torch.rand(rays_o.shape[0], 1, device=dev)


def caps(fwd=0, dgrad=0, wgrad=0):
    for k, v in (('FWD', fwd), ('DGRAD', dgrad), ('WGRAD', wgrad)):
        if v:
            os.environ['SUNERF_GRID_CAP_' + k] = str(v)
        else:
            os.environ.pop('SUNERF_GRID_CAP_' + k, None)


def forward(sl):
    z = ops.sample_z(ops.SAMPLER_STRATIFIED, rays_o[sl], rays_d[sl], t_vals, 1.3, 1.0)
    out = emission_pass(model, rays_o[sl], rays_d[sl], times[sl], z, 1.2, want_epilogues=True)
    loss, stats = training_loss(out['image'], out['image'], target[sl], out['regularization'], 0.5, 1.0,
                                asinh_scaling=(1.0, 0.005), finite_check=[out['height_map'], out['absorption_map']])
    return loss


A, B = slice(0, N), slice(N, 2 * N)


def timeit(fn, k=K):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


res = {}
caps()
loss_a = forward(A)
res['fwd_all'] = timeit(lambda: forward(B))
res['bwd_all'] = timeit(lambda: loss_a.backward(retain_graph=True))
for f_cus in (80, 100, 120):
    b_cus = 256 - f_cus
    caps(fwd=f_cus)
    res[f'fwd_{f_cus}'] = timeit(lambda: forward(B))
    caps(dgrad=b_cus, wgrad=b_cus)
    res[f'bwd_{b_cus}'] = timeit(lambda: loss_a.backward(retain_graph=True))
    caps(fwd=f_cus, dgrad=b_cus, wgrad=b_cus)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sa):
        loss_s = forward(A)            # graph whose backward runs on stream sa
    torch.cuda.synchronize()

    def both():
        with torch.cuda.stream(sa):
            loss_s.backward(retain_graph=True)
        with torch.cuda.stream(sb):
            forward(B)
    res[f'both_{f_cus}+{b_cus}'] = timeit(both)
    del loss_s
caps()
seq = res['fwd_all'] + res['bwd_all']
for k, v in res.items():
    print(f'{k:>16s}: {v:7.2f} ms' + (f'   synergy {1 - v / seq:+.1%} against fwd_all + bwd_all = {seq:.2f} ms' if k.startswith('both') else ''))
