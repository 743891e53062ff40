"""Is the forward's clock governed by power?  Same kernel, same instruction stream, three operand sets: the random-init
network, the same weights x 1e-3 (activations ~0: few toggling operand bits), all-zero weights (development aid)."""
import sys, torch
sys.path.insert(0, '/root/repo/2024-hl-spi3s-sunerf_amd')
from sunerf_hip import ops
from sunerf.model.model import NeRF
from sunerf_hip.rays import observer_rays
dev = torch.device('cuda')
o, d = observer_rays(512, device=dev)
t = torch.zeros(o.shape[0], device=dev)
z = ops.sample_z(ops.SAMPLER_STRATIFIED, o, d, torch.linspace(0., 1., 128, device=dev), 1.3, 1.0)
for name, scale in (('random init', 1.0), ('weights x 1e-3', 1e-3), ('zero weights', 0.0), ('random init', 1.0)):
    torch.manual_seed(7)
    model = NeRF(d_input=4, d_output=2, n_layers=8, d_filter=256).to(dev)
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(scale)
    packed = model.packed()
    for _ in range(2):
        out = ops.emission_render_fwd(packed, o, d, t, z, 1.2, want_epilogues=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        out = ops.emission_render_fwd(packed, o, d, t, z, 1.2, want_epilogues=True)
    e1.record(); torch.cuda.synchronize()
    fin = bool(torch.isfinite(out['image']).all())
    print(f'{name:16s}: {e0.elapsed_time(e1) / 6:.2f} ms per 512 x 512 x 128 frame, finite={fin}, max image {out["image"].abs().max().item():.3e}')
