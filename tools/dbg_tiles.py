"""Reads the region timers of a -DSUNERF_DBG_TILES=1 -DSUNERF_DBG_KIND=k build of the forward kernel (development aid)."""
import sys, torch
sys.path.insert(0, '/root/repo/2024-hl-spi3s-sunerf_amd')
from sunerf_hip import ops
from sunerf.model.model import NeRF
from sunerf_hip.rays import observer_rays
torch.manual_seed(7)
dev = torch.device('cuda')
model = NeRF(d_input=4, d_output=2, n_layers=8, d_filter=256).to(dev)
o, d = observer_rays(512, device=dev)
t = torch.zeros(o.shape[0], device=dev)
z = ops.sample_z(ops.SAMPLER_STRATIFIED, o, d, torch.linspace(0., 1., 128, device=dev), 1.3, 1.0)
out = ops.emission_render_fwd(model.packed(), o, d, t, z, 1.2, want_epilogues=True)
torch.cuda.synchronize()
w = out['weights'].flatten()[:64 * 4 * 16].reshape(64, 4, 16).cpu()
cyc, cnt = w[..., 0].sum().item(), w[..., 2].sum().item()
print(f'{sys.argv[1] if len(sys.argv) > 1 else ""}: {cnt / 256:.0f} regions per wave, {cyc / max(cnt, 1):.0f} cycles each, {cyc / 256 / 1e6:.2f} M cycles per wave')
