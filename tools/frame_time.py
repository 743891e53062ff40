"""Times the full-frame two-pass inference driver (development aid): python tools/frame_time.py [res] [tile]"""
import sys, time, torch
sys.path.insert(0, '/root/repo/2024-hl-spi3s-sunerf_amd')
from sunerf.rendering.emission import EmissionRadiativeTransfer
from sunerf_hip.rays import fov_axis, pose_spherical, render_frame
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
torch.manual_seed(7)
r = EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 64, 'perturb': False},
                              hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 64},
                              model_config={'d_filter': 256}).cuda()
ax = fov_axis(res, 1.1 * 960. / 206264.806, 'cuda')
c2w = pose_spherical(-0.3, 0.1, 215.032)
for keys in (None, ('image', 'height_map', 'absorption_map')):
    render_frame(r, ax, ax, c2w, 0.2, tile_rays=tile, keys=keys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        f = render_frame(r, ax, ax, c2w, 0.2, tile_rays=tile, keys=keys)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'{res}x{res} two-pass frame, tile {tile}, keys={keys}: {dt * 1e3:.1f} ms = {res * res * 192 / dt:.3e} MLP evaluations/s')
