"""Round 4: how far is the pipelined backward with a SINGLE fp16 W^T (SUNERF_PIPE_HI_ONLY=1: 16 instead of 32 matrix instructions
per chunk on the data-gradient waves, -5.7 % kernel time) from the fp32 gradients at TRAINING size, and does a cheap probe -- the
relative difference between the two arithmetics on the first rays of the batch -- predict it?  Reference: the fp32 backward of
csrc/bwd_exact.hip on the SAME batch (1 M samples).  Usage on the GPU box: python tools/experiments/r4_hi_only_accuracy.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import sunerf_oracle as orc   # noqa: E402  (initial weights only)
from sunerf_hip import ops    # noqa: E402
from sunerf_hip.rays import observer_rays   # noqa: E402

dev = torch.device('cuda')


def grads(packed, o, d, t, z, fwd, g_image, mode, n=None):
    n = o.shape[0] if n is None else n
    Ws, bs = packed._keepalive
    gW, gb = [torch.empty_like(W) for W in Ws], [torch.empty_like(b) for b in bs]
    os.environ['SUNERF_PIPE_HI_ONLY'] = '0'
    os.environ['SUNERF_EXACT_BACKWARD_SAMPLES'] = '0'
    times = None
    if mode == 'exact':
        os.environ['SUNERF_EXACT_BACKWARD_SAMPLES'] = str(1 << 24)
        times = t[:n]
    elif mode == 'hi':
        os.environ['SUNERF_PIPE_HI_ONLY'] = '1'
    S = z.shape[1]
    stash = fwd['stash']
    ops.emission_render_bwd(packed, o[:n], d[:n], z[:n], fwd['raw'][:n], stash, g_image[:n], None, 0.0, 1.2, gW, gb, times=times)
    torch.cuda.synchronize()
    return gW, gb


def rel(a, b):
    return [((x - y).norm() / y.norm()).item() for x, y in zip(a, b)]


def run(name, scale, steps=0):
    torch.manual_seed(0)
    params = orc.init_params(d_filter=256, n_layers=8, seed=5)
    params = [((W * scale) if 0 < i < len(params) - 1 else W, b) for i, (W, b) in enumerate(params)]
    Ws, bs = [W.to(dev) for W, _ in params], [b.to(dev) for _, b in params]
    packed = ops.PackedMLP(Ws, bs, precision=ops.PRECISION_EXACT)
    o, d = observer_rays(1024, row_start=508, row_end=516, device=dev)      # 8192 rays through the disk
    n = o.shape[0]
    g = torch.Generator().manual_seed(4)
    t = torch.rand(n, generator=g).to(dev)
    g_image = (torch.randn(n, generator=g) * 1e-4).to(dev)
    z = ops.sample_z(ops.SAMPLER_STRATIFIED, o, d, torch.linspace(0, 1, 128, device=dev), 1.3, 1.0)
    fwd = ops.emission_render_fwd(packed, o, d, t, z, 1.2, training=True)
    ex = grads(packed, o, d, t, z, fwd, g_image, 'exact')
    hl = grads(packed, o, d, t, z, fwd, g_image, 'hilo')
    hi = grads(packed, o, d, t, z, fwd, g_image, 'hi')
    e_hl, e_hi = rel(hl[0], ex[0]) + rel(hl[1], ex[1]), rel(hi[0], ex[0]) + rel(hi[1], ex[1])
    line = f'{name:16s} vs fp32 at {n} x 128: hi+lo worst {max(e_hl):.2e}  hi-only worst {max(e_hi):.2e} (weights {max(e_hi[:9]):.2e})'
    for k in (64, 256, 1024):
        a = grads(packed, o, d, t, z, fwd, g_image, 'hilo', n=k)
        b = grads(packed, o, d, t, z, fwd, g_image, 'hi', n=k)
        line += f' | probe {k} rays: W {max(rel(b[0], a[0])):.2e} b {max(rel(b[1], a[1])):.2e}'
    full = max(rel(hi[0], hl[0]))
    print(line + f' | full batch hi vs hi+lo W {full:.2e}', flush=True)


if __name__ == '__main__':
    for name, scale in (('default init', 1.0), ('hidden x 0.5', 0.5), ('hidden x 1.5', 1.5), ('hidden x 2', 2.0), ('hidden x 3', 3.0), ('hidden x 4', 4.0)):
        run(name, scale)
