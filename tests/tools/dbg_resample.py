import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('2024-hl-spi3s-sunerf_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
from conftest import load_golden
from sunerf_hip import ops
g = load_golden('g4_hierarchical')
nz, zc = ops.hier_resample(g['z_vals'][:4].contiguous().cuda(), g['weights_deg'].cuda(), torch.linspace(0., 1., 32).cuda())
torch.set_printoptions(precision=4, linewidth=200)
for r in range(4):
    d = (zc[r].cpu() - g['z_comb_deg'][r]).abs()
    print(r, 'nz diff', (nz[r].cpu() - g['new_z_deg'][r]).abs().max().item(), 'zc diff', d.max().item(), d.argmax().item())
    if d.max() > 1e-3:
        print(zc[r].cpu()); print(g['z_comb_deg'][r]); print(nz[r].cpu())
import sunerf_oracle as orc
u = torch.rand(64, 32, generator=torch.Generator().manual_seed(5))
nz_o, zc_o = orc.hierarchical_z(g['z_vals'], g['weights'], 32, u=u)
nz, zc = ops.hier_resample(g['z_vals'].cuda(), g['weights'].cuda(), u.cuda())
dn = (nz.cpu() - nz_o).abs(); dz = (zc.cpu() - zc_o).abs()
print('perturb: nz diff', dn.max().item(), 'zc diff', dz.max().item())
r = dz.max(-1)[0].argmax().item()
print('row', r); print(zc[r].cpu()); print(zc_o[r]); print(nz[r].cpu()); print(nz_o[r]); print(g['z_vals'][r])
