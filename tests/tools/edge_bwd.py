import sys, torch
sys.path.insert(0,'/root/repo/2024-hl-spi3s-sunerf_amd'); sys.path.insert(0,'/root/repo/oracle')
import sunerf_oracle as orc
from sunerf_hip import ops
dev=torch.device('cuda')
for scale, gs in ((1.0, 0.0), (1e-3, 1e-3), (0.0, 1e-3), (1.0, 1e30), (1.0, 1e-30)):
    params=[(W*scale,b*scale) for W,b in orc.init_params(d_filter=64,n_layers=3,seed=2)]
    o,d=orc.synthetic_rays(4); t=torch.zeros(o.shape[0],1)
    z=orc.stratified_z(o,d,orc.linspace_t_vals(32),torch.tensor(1.3),torch.tensor(1.0))
    Ws=[W.to(dev) for W,_ in params]; bs=[b.to(dev) for _,b in params]
    packed=ops.PackedMLP(Ws,bs)
    fwd=ops.emission_render_fwd(packed,o.to(dev),d.to(dev),t.to(dev),z.to(dev),reg_radius=1.2,want_epilogues=True,training=True)
    g_image=torch.randn(o.shape[0],device=dev)*gs
    gW=[torch.full_like(W,float('nan')) for W in Ws]; gb=[torch.full_like(b,float('nan')) for b in bs]
    g_raw=ops.emission_render_bwd(packed,o.to(dev),d.to(dev),z.to(dev),fwd['raw'],fwd['stash'],g_image,None,0.0,1.2,gW,gb)
    torch.cuda.synchronize()
    fin=all(bool(torch.isfinite(x).all()) for x in gW+gb+[g_raw])
    print(f'weights x{scale:g}, grad scale {gs:g}: finite={fin}, max|gW0|={gW[0].abs().max().item():.3e}, max|g_raw|={g_raw.abs().max().item():.3e}')
