import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('2024-hl-spi3s-sunerf_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import sunerf_oracle as orc
from sunerf_hip import ops
from test_gpu_backward import _case, _oracle_grads
d_filter, n_layers, S = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (64, 3, 32)
params, o, d, t, z = _case(d_filter, n_layers, S)
n = o.shape[0]
g_image = torch.randn(n) * 1e-3
ref_out, ref_grads, ref_graw = _oracle_grads(params, o, d, t, z, g_image, 2e-5)
dev = torch.device('cuda')
Ws = [W.to(dev) for W, _ in params]; bs = [b.to(dev) for _, b in params]
packed = ops.PackedMLP(Ws, bs)
fwd = ops.emission_render_fwd(packed, o.to(dev), d.to(dev), t.to(dev), z.to(dev), reg_radius=1.2, want_epilogues=True, training=True)
gW = [torch.zeros_like(W) for W in Ws]; gb = [torch.zeros_like(b) for b in bs]
ops.emission_render_bwd(packed, o.to(dev), d.to(dev), z.to(dev), fwd['raw'], fwd['stash'], g_image.to(dev), None, 2e-5, 1.2, gW, gb)
torch.cuda.synchronize()
for i, ((rW, rb), W, b) in enumerate(zip(ref_grads, gW, gb)):
    print(i, 'W err %.3e (norm ref %.3e got %.3e)' % (((W.cpu()-rW).norm()/rW.norm()).item(), rW.norm().item(), W.norm().item()),
          'b err %.3e (norm ref %.3e got %.3e)' % (((b.cpu()-rb).norm()/rb.norm()).item(), rb.norm().item(), b.norm().item()))
