"""MLP raw-output error of the HIP forward against the fp64-evaluated reference MLP (development aid)."""
import sys, torch
sys.path.insert(0, '/root/repo/2024-hl-spi3s-sunerf_amd'); sys.path.insert(0, '/root/repo/oracle')
import sunerf_oracle as orc
from sunerf_hip import ops
for d, nl in ((64, 8), (128, 8), (256, 8), (256, 2), (256, 1)):
    params = orc.init_params(d_filter=d, n_layers=nl, seed=3)
    packed = ops.PackedMLP([W.cuda() for W, _ in params], [b.cuda() for _, b in params])
    g = torch.Generator().manual_seed(1)
    m = 4096
    pts = (torch.rand(m, 3, generator=g) * 2 - 1) * 1.2
    t = torch.rand(m, generator=g)
    o = torch.zeros(m, 3); z = torch.ones(m, 2)
    out = ops.emission_render_fwd(packed, o.cuda(), pts.cuda(), t.cuda(), z.cuda(), 0.0, want_raw=True)['raw'][:, 0, :].cpu()
    x = torch.cat([pts, t[:, None]], -1)
    ref32 = orc.mlp_forward(params, x)
    p64 = [(W.double(), b.double()) for W, b in params]
    ref64 = orc.mlp_forward(p64, x.double())
    print(f'd={d} layers={nl}: max|hip - ref64| = {(out.double() - ref64).abs().max().item():.3e}   max|ref32 - ref64| = {(ref32.double() - ref64).abs().max().item():.3e}   max|ref| = {ref64.abs().max().item():.3f}')
