#!/usr/bin/env python3
"""fp32 backward for small batches (csrc/bwd_exact.hip) against the fp16 kernels on the same batch: ms per backward call as
launched from Python (HIP events over 50 calls).  usage: exact_backward_time.py [n_rays n_samples] ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import sunerf_oracle as orc   # noqa: E402  (initial weights / synthetic rays only)
from sunerf_hip import ops    # noqa: E402

dev = torch.device('cuda')


def run(n_rays, S):
    params = orc.init_params(d_filter=256, n_layers=8, seed=3)
    o, d = orc.synthetic_rays(64)
    o, d = o[:n_rays].to(dev), d[:n_rays].to(dev)
    t = torch.rand(n_rays, 1, device=dev)
    z = orc.stratified_z(o.cpu(), d.cpu(), orc.linspace_t_vals(S), torch.tensor(1.3), torch.tensor(1.0)).to(dev)
    Ws, bs = [W.to(dev) for W, _ in params], [b.to(dev) for _, b in params]
    packed = ops.PackedMLP(Ws, bs)
    g_image = torch.randn(n_rays, device=dev) * 1e-3
    line = f'{n_rays} rays x {S} samples:'
    for name, limit in (('fp32 chain', str(1 << 24)), ('fp16 kernels', '0')):
        os.environ['SUNERF_EXACT_BACKWARD_SAMPLES'] = limit
        fwd = ops.emission_render_fwd(packed, o, d, t, z, reg_radius=1.2, training=True)
        gW, gb = [torch.empty_like(W) for W in Ws], [torch.empty_like(b) for b in bs]
        call = lambda: ops.emission_render_bwd(packed, o, d, z, fwd['raw'], fwd['stash'], g_image, None, 2e-5, 1.2, gW, gb, times=t)
        for _ in range(5):
            call()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(50):
            call()
        ev[1].record()
        torch.cuda.synchronize()
        line += f'  {name} {ev[0].elapsed_time(ev[1]) / 50:.3f} ms'
    print(line, flush=True)


if __name__ == '__main__':
    a = [int(v) for v in sys.argv[1:]]
    for n_rays, S in ([tuple(a[i:i + 2]) for i in range(0, len(a), 2)] or [(2, 17), (16, 64), (32, 128), (64, 128), (128, 128)]):
        run(n_rays, S)
