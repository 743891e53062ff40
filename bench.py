#!/usr/bin/env python3
"""Headline benchmark: ray-samples/s of the fused emission renderer on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank/GPU)

One step = one pass of the hot path over this rank's batch of synthetic rays: a
``--res`` x ``--res`` observer image (default 1024 x 1024), ``--samples`` samples per ray (default 128), 8 x 256 sine
MLP, rays resident in HBM before the timed region.  Rays are sharded over ranks by image rows (weak scaling: every
rank renders a full-size frame of its own at N > 1, so per-GPU work is fixed); value = ray-samples of all ranks /
max-over-ranks wall time.

The JSON line carries
  roofline     : algorithmic GEMM FLOPs of the dominant kernel (sunerf_emission_render_fwd) per launch / its average
                 duration measured with HIP events on the launch stream, against the dense f16 MFMA peak
  cpu_baseline : the CPU oracle (port of the reference's aten op sequence) timed on this host's cores on a bounded
                 sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

D_FILTER, N_LAYERS, ENC = 256, 8, 84
FLOPS_FWD_PER_SAMPLE = 2 * (ENC * D_FILTER + (N_LAYERS - 1) * D_FILTER * D_FILTER + D_FILTER * 2)   # 961 536
PEAK_F16_DENSE_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense BF16/F16 MFMA peak
PEAK_F32_MFMA_TFLOPS = 157.3


def cpu_baseline(res, samples, seconds_budget=20.0):
    """Times oracle/sunerf_oracle.py (kind 'port') on a bounded sample: a strip of rows of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import sunerf_oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get('SUNERF_CPU_THREADS', 16)))   # the GPU box grants 16 cores per GPU
    torch.set_num_threads(cores)
    params = orc.init_params(D_FILTER, N_LAYERS, seed=7)
    o, d = orc.synthetic_rays(res)
    n = 2048
    start = (res // 2) * res                      # rows through the disk centre
    o, d = o[start:start + n].contiguous(), d[start:start + n].contiguous()
    t = torch.zeros(n, 1)
    z = orc.stratified_z(o, d, orc.linspace_t_vals(samples), torch.tensor(1.3), torch.tensor(1.0))
    with torch.no_grad():
        orc.render_pass(params, o[:256], d[:256], t[:256], z[:256])   # warm-up
        best, spent, reps = float('inf'), 0.0, 0
        while spent < seconds_budget and reps < 5:
            t0 = time.perf_counter()
            orc.render_pass(params, o, d, t, z)
            dt = time.perf_counter() - t0
            best = min(best, dt)
            spent += dt
            reps += 1
    return {'value': n * samples / best, 'unit': 'ray-samples/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} rays x {samples} samples (rows through disk centre of the {res}x{res} frame), '
                      f'forward render pass, best of {reps}'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--samples', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)

    from sunerf.model.model import NeRF
    from sunerf_hip import ops
    from sunerf_hip.rays import observer_rays

    torch.manual_seed(7)
    model = NeRF(d_input=4, d_output=2, n_layers=N_LAYERS, d_filter=D_FILTER).to(dev)
    packed = model.packed()
    # every rank renders a full frame seen from its own longitude (weak scaling)
    rays_o, rays_d = observer_rays(args.res, theta=-0.3 + 0.05 * rank, device=dev)
    n_rays = rays_o.shape[0]
    times = torch.zeros(n_rays, device=dev)
    t_vals = torch.linspace(0., 1., args.samples, device=dev)
    z_vals = ops.sample_z(ops.SAMPLER_STRATIFIED, rays_o, rays_d, t_vals, 1.3, 1.0)
    torch.cuda.synchronize()

    def step():
        return ops.emission_render_fwd(packed, rays_o, rays_d, times, z_vals, reg_radius=1.2, want_epilogues=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()          # on the current stream == the stream the kernel is launched on
        out = step()
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    assert os.environ.get('SUNERF_DEBUG') or torch.isfinite(out['image']).all()

    el = torch.tensor([elapsed], device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    samples_per_step = n_rays * args.samples * world
    value = samples_per_step * args.steps / elapsed

    if rank == 0:
        achieved = n_rays * args.samples * FLOPS_FWD_PER_SAMPLE / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get('render_fwd_bytes_per_launch')
        line = {
            'metric': 'ray-samples/sec (fused emission render, fwd)', 'value': value, 'unit': 'ray-samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (fp16 hi/lo split, 3 MFMA per product, fp32 accumulate)', 'data': 'synthetic',
            'config': {'workload': f'emission render fwd, {args.res}x{args.res} rays x {args.samples} samples/ray, '
                                   f'single fused pass, {N_LAYERS}x{D_FILTER} sine MLP, per GPU',
                       'rays_per_gpu': n_rays, 'samples_per_ray': args.samples},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F16_DENSE_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F16_DENSE_TFLOPS, 'traffic': traffic,
                         'kernel': 'render_fwd_kernel<256>', 'kernel_ms': kernel_ms,
                         'flops_per_sample': FLOPS_FWD_PER_SAMPLE, 'executed_over_algorithmic': 3.0,
                         'frac_of_f32_mfma_peak': achieved / PEAK_F32_MFMA_TFLOPS},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.res, args.samples)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
