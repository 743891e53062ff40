"""Reads the per-wave wait counters of a -DSUNERF_DBG_BARRIER=1 build of the forward kernel (development aid; the kernel
writes them over the first floats of the `weights` output)."""
import sys, torch
sys.path.insert(0, '/root/repo/2024-hl-spi3s-sunerf_amd')
from sunerf_hip import ops
from sunerf.model.model import NeRF
from sunerf_hip.rays import observer_rays
torch.manual_seed(7)
dev = torch.device('cuda')
model = NeRF(d_input=4, d_output=2, n_layers=8, d_filter=256).to(dev)
o, d = observer_rays(512, device=dev)
n = o.shape[0]
t = torch.zeros(n, device=dev)
tv = torch.linspace(0., 1., 128, device=dev)
z = ops.sample_z(ops.SAMPLER_STRATIFIED, o, d, tv, 1.3, 1.0)
out = ops.emission_render_fwd(model.packed(), o, d, t, z, 1.2, want_epilogues=True)
torch.cuda.synchronize()
w = out['weights'].flatten()[:64 * 4 * 16].reshape(64, 4, 16).cpu()
vm, bar, cnt = w[..., 0], w[..., 1], w[..., 2]
print(f'acquires per wave {cnt.mean().item():.0f}; ticks per acquire: vmcnt wait {vm.sum().item() / cnt.sum().item():.1f}, '
      f'barrier wait {bar.sum().item() / cnt.sum().item():.1f}')
print('per-wave barrier wait, workgroup 0:', [round(x, 1) for x in (bar[0] / cnt[0]).tolist()])
