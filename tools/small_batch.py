#!/usr/bin/env python3
"""Reference-shaped two-pass training step at the reference's batch sizes (config yaml: 1024 ... 4096 rays per GPU):
ms per step as launched from Python.  usage: small_batch.py [batch ...]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
from sunerf_hip.rays import observer_rays  # noqa: E402
res = 256
rays_o, rays_d = observer_rays(res, theta=-0.3, device=dev)
n = rays_o.shape[0]
times = torch.rand(n, device=dev)
target = torch.rand(n, 1, generator=torch.Generator().manual_seed(1)).to(dev)
for b in [int(x) for x in sys.argv[1:]] or [1024, 3072, 8192]:
    r = bench.two_pass_rate(dev, 1, rays_o, rays_d, times, target, b, 128, steps=30, warmup=5)
    print(f"batch {b}: {r['ms_per_step']:.3f} ms/step, {r['value']:.3e} evals/s", flush=True)
