"""Throughput of free-standing point queries (NeRF.forward / SuNeRFLoader.load_coords: volume cubes): python tools/points_rate.py [log2 n]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '2024-hl-spi3s-sunerf_amd'))
from sunerf.model.model import NeRF
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
torch.manual_seed(0)
net = NeRF(d_input=4, d_output=2, n_layers=8, d_filter=256).cuda()
x = torch.rand(n, 4, device='cuda') * 2.6 - 1.3
with torch.no_grad():
    net(x[:4096]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        y = net(x)['inferences']
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
print(f'{n} points through the 8 x 256 network: {dt * 1e3:.1f} ms = {n / dt:.3e} points/s')
