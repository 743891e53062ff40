import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('2024-hl-spi3s-sunerf_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
from conftest import load_golden
from sunerf_hip import ops
from sunerf.model.model import NeRF_DT
from sunerf.rendering.density_temperature import DensityTemperatureRadiativeTransfer
def log(*a):
    print(*a, flush=True)
g = load_golden('g6_dt_e2e')
mod = DensityTemperatureRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 16, 'perturb': False},
    hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 16}, model_config={'d_filter': 64}, model=NeRF_DT,
    pixel_intensity_factor=float(g['pixel_intensity_factor']), response_table=(g['aia_logte'].numpy(), g['aia_tresp'].numpy()))
mod.load_state_dict({k[4:].replace('__', '.'): v for k, v in g.items() if k.startswith('sd__')}, strict=True)
mod = mod.cuda()
o, d, t, wl = g['rays_o'].cuda(), g['rays_d'].cuda(), g['times'].cuda(), g['wavelengths'].cuda()
m = mod.coarse_model
z = mod.sampler.z_vals(o, d)
packed = m.packed()
fw = ops.emission_render_fwd(packed, o, d, t, z, 0.0, want_raw=True, training=True); torch.cuda.synchronize(); log('fwd ok', fw['raw'].shape)
la = m.log_abs_vector().detach(); vc = m.volumetric_constant.detach()
out = ops.dt_integral_fwd(fw['raw'], z, o, d, wl, mod.response_logte, mod.response_table, la, vc, 10., 5., 1e17, 1.25); torch.cuda.synchronize(); log('dt fwd ok')
gi = torch.randn(16, 7, device='cuda') * 1e-3
r = ops.dt_integral_bwd(fw['raw'], z, o, d, wl, mod.response_logte, mod.response_table, la, vc, 10., 5., 1e17, 1.25, gi, None); torch.cuda.synchronize(); log('dt bwd ok', r[0].abs().max().item(), r[1], r[2])
gW = [torch.zeros_like(l.weight) for l in m.linears()]; gb = [torch.zeros_like(l.bias) for l in m.linears()]
lib = __import__('sunerf_hip').load()
import ctypes
n, s = z.shape
dz = torch.empty(lib.sunerf_dz_stash_bytes(n, s, 64, 9), dtype=torch.uint8, device='cuda')
st = lib.sunerf_mlp_dgrad(ctypes.c_void_p(packed.transposed().data_ptr()), 64, 9, ctypes.c_void_p(r[0].data_ptr()), ctypes.c_void_p(r[3].data_ptr()), ctypes.c_void_p(fw['stash'].data_ptr()), ctypes.c_void_p(dz.data_ptr()), n, s, None)
torch.cuda.synchronize(); log('dgrad ok', st)
ops.mlp_backward(packed, r[0], r[3], fw['stash'], gW, gb); torch.cuda.synchronize(); log('mlp bwd ok', gW[0].norm().item())
