import os, sys, torch
sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0,'/root/repo/2024-hl-spi3s-sunerf_amd'); sys.path.insert(0,'/root/repo/oracle')
import sunerf_oracle as orc
from sunerf_hip import ops
torch.manual_seed(0)
dev = torch.device('cuda')
params = orc.init_params(d_filter=256, n_layers=8, seed=3)
o, d = orc.synthetic_rays(6); n = o.shape[0]
t = torch.rand(n, 1) * 5.
z = orc.stratified_z(o, d, orc.linspace_t_vals(40), torch.tensor(1.3), torch.tensor(1.0))
Ws = [W.to(dev) for W, _ in params]; bs = [b.to(dev) for _, b in params]
packed = ops.PackedMLP(Ws, bs)
o, d, t, z = o.to(dev), d.to(dev), t.to(dev), z.to(dev)
fwd = ops.emission_render_fwd(packed, o, d, t, z, reg_radius=1.2, training=True)
g_image = torch.randn(n, device=dev) * 1e-3
res = {}
for mode in ('classic', 'pipe'):
    ops._backward_forced = mode
    gW = [torch.full_like(W, float('nan')) for W in Ws]; gb = [torch.full_like(b, float('nan')) for b in bs]
    ops.emission_render_bwd(packed, o, d, z, fwd['raw'], fwd['stash'], g_image, None, 2e-5, 1.2, gW, gb)
    torch.cuda.synchronize(); res[mode] = gW
for l in (1, 4, 7):
    a, b = res['classic'][l].cpu(), res['pipe'][l].cpu()
    bad = (~torch.isfinite(b)) | ((a - b).abs() > 1e-3 * a.abs().max())
    print('layer', l, 'bad', int(bad.sum()), 'nan', int((~torch.isfinite(b)).sum()))
    # rows = out features j (dZ side), cols = in features k (H side): tile pattern 8x8 of 32x32
    pat = bad.reshape(8, 32, 8, 32).any(1).any(-1).int()
    print(pat)
    cols = bad.any(0).nonzero().flatten().tolist()
    print('bad cols', cols[:64])
