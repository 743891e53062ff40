#!/usr/bin/env python3
"""Headline benchmark: ray-samples/s (forward + backward) of the fused emission renderer on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank/GPU)

Workload (BASELINE.json metric: ray-samples/sec/GPU (fwd+bwd), 1024^2 image x 128 samples): a synthetic 1024 x 1024
observer frame, 128 samples per ray, 8 x 256 sine MLP, fp32 parameters.  One step = one optimiser
step of the hot path on this rank's next batch of ``--batch`` rays (default 32768 = 1/32 of the frame) that are
resident in HBM before the timed region: fused forward render (with activation stash), loss of sunerf.py:110-120
(asinh-scaled MSE + regularization mean), backward (integral, dgrad, wgrad), gradient all-reduce over the ranks
(RCCL), clip_grad_norm_(0.5) and Adam.  Rays are sharded over ranks by image rows, every rank works on equally sized
batches of its own rows (weak scaling); value = ray-samples of all ranks / max-over-ranks wall time.
``--mode fwd`` times the inference render of the whole frame instead (configs[1] of BASELINE.json).

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
def flops_fwd(d):
    return 2 * (ENC * d + (N_LAYERS - 1) * d * d + d * 2)            # 961 536 at d = 256


def flops_bwd(d):
    return 2 * ((N_LAYERS - 1) * d * d + d * 2) + flops_fwd(d)        # dgrad + wgrad = 1 880 064 at d = 256
PEAK_F16_DENSE_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense BF16/F16 MFMA peak
PEAK_F32_MFMA_TFLOPS = 157.3


def cpu_baseline(res, samples, mode, seconds_budget=20.0):
    """Times oracle/sunerf_oracle.py (kind 'port': the reference's aten op sequence) on a bounded sample of the same
    workload: a strip of rays through the disk centre of the same frame; forward only or forward + backward."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import sunerf_oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get('SUNERF_CPU_THREADS', 16)))   # the GPU box grants 16 cores per GPU
    torch.set_num_threads(cores)
    params = orc.init_params(D_FILTER, N_LAYERS, seed=7)
    o, d = orc.synthetic_rays(res)
    n = 2048 if mode == 'fwd' else 1024
    start = (res // 2) * res                      # rows through the disk centre
    o, d = o[start:start + n].contiguous(), d[start:start + n].contiguous()
    t = torch.zeros(n, 1)
    z = orc.stratified_z(o, d, orc.linspace_t_vals(samples), torch.tensor(1.3), torch.tensor(1.0))
    target = torch.rand(n, 1, generator=torch.Generator().manual_seed(1))
    if mode == 'train':
        for W, b in params:
            W.requires_grad_(True)
            b.requires_grad_(True)

    def run(nn_):
        if mode == 'fwd':
            with torch.no_grad():
                orc.render_pass(params, o[:nn_], d[:nn_], t[:nn_], z[:nn_])
        else:
            out = orc.render_pass(params, o[:nn_], d[:nn_], t[:nn_], z[:nn_])
            dist_pts = out['points'].pow(2).sum(-1).pow(0.5)
            reg = torch.relu(dist_pts - 1.2) * (1 - out['regularizing_quantity'])
            loss = torch.nn.functional.mse_loss(orc.asinh_scaling(out['image']), orc.asinh_scaling(target[:nn_])) + reg.mean()
            loss.backward()

    run(128)   # warm-up
    best, spent, reps = float('inf'), 0.0, 0
    while spent < seconds_budget and reps < 5:
        t0 = time.perf_counter()
        run(n)
        dt = time.perf_counter() - t0
        best = min(best, dt)
        spent += dt
        reps += 1
    return {'value': n * samples / best, 'unit': 'ray-samples/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} rays x {samples} samples (rows through disk centre of the {res}x{res} frame), '
                      + ('forward render pass' if mode == 'fwd' else 'forward + loss + backward (autograd)')
                      + f', best of {reps}'}




def two_pass_rate(dev, world, rays_o, rays_d, times, target, batch, samples, steps=3, warmup=1, density_temperature=False):
    """Reference-shaped training step beside the single-pass headline: StratifiedSampler (S/2 jittered
    samples) -> coarse pass -> HierarchicalSampler (S/2 more) -> fine pass over S samples -> loss on both images ->
    backward through both models -> all-reduce + clip + Adam.  MLP evaluations per ray: S/2 + S."""
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    from sunerf_hip.train import ClipAdam, training_loss
    torch.manual_seed(7)
    cfg = dict(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': samples // 2},
               hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': samples // 2},
               model_config={'d_filter': D_FILTER, 'n_layers': N_LAYERS})
    wavelengths = None
    if density_temperature:
        # BASELINE config 5: NeRF_DT dual head, 7 AIA channels on every ray (run_density_temperature.py path).  The AIA
        # response is a synthetic log-normal bump per channel (the .genx table is data, not part of the timed work).
        import numpy as np
        from sunerf.model.model import NeRF_DT
        from sunerf.rendering.density_temperature import DensityTemperatureRadiativeTransfer
        logte = np.tile(np.linspace(4.0, 9.0, 101, dtype=np.float32), (7, 1))
        peaks = np.array([6.8, 5.6, 5.9, 6.2, 6.3, 4.9, 5.4], dtype=np.float64)[:, None]
        tresp = 1e-25 * np.exp(-0.5 * ((logte - peaks) / 0.15) ** 2)
        rendering = DensityTemperatureRadiativeTransfer(model=NeRF_DT, response_table=(logte, tresp), **cfg).to(dev)
        wavelengths = torch.tensor([[94., 131., 171., 193., 211., 304., 335.]], device=dev).repeat(batch, 1)
        target = torch.rand(target.shape[0], 7, generator=torch.Generator().manual_seed(1)).to(dev)
    else:
        rendering = EmissionRadiativeTransfer(**cfg).to(dev)
    opt = ClipAdam(rendering.parameters(), lr=1e-4, max_norm=0.5)
    n_batches = max(1, rays_o.shape[0] // batch)

    def step(i):
        b = (i % n_batches) * batch
        sl = slice(b, b + batch)
        opt.zero_grad()
        if density_temperature:
            out = rendering(rays_o[sl], rays_d[sl], times[sl].reshape(-1, 1), wavelengths)
        else:
            out = rendering(rays_o[sl], rays_d[sl], times[sl].reshape(-1, 1))
        loss, stats = training_loss(out['coarse_image'], out['fine_image'], target[sl], out['regularization'], 1.0, 1.0,
                                    asinh_scaling=None if density_temperature else (1.0, 0.005),
                                    finite_check=[out['height_map'], out['absorption_map'], out['z_vals_stratified'],
                                                  out['z_vals_hierarchical']])
        loss.backward()
        opt.step(skip_if_positive=stats[5:6])
        return loss

    for i in range(warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    assert torch.isfinite(loss).all()
    evals = samples // 2 + samples
    return {'value': batch * evals * world * steps / el.item(), 'unit': 'ray-samples/s',
            'mlp_evaluations_per_ray': evals, 'ms_per_step': el.item() / steps * 1e3, 'steps': steps,
            'what': ('density-temperature head (7 channels): ' if density_temperature else '')
                    + f'coarse pass {samples // 2} + hierarchical resampling + fine pass {samples} samples per ray, two models, '
                    'loss + backward + all-reduce + clip + Adam'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--samples', type=int, default=128)
    ap.add_argument('--batch', type=int, default=32768, help='rays per rank and optimiser step (train mode)')
    ap.add_argument('--mode', choices=['train', 'fwd'], default='train')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-half', action='store_true', help='skip the HALF-precision figure reported beside the headline')
    ap.add_argument('--no-two-pass', action='store_true', help='skip the reference-shaped two-pass figure (train mode)')
    ap.add_argument('--dt', action='store_true', help='also time BASELINE config 5: density-temperature head, two-pass, '
                                                      '256 samples per ray in the fine pass (reported under "dt_two_pass")')
    ap.add_argument('--d-filter', type=int, default=D_FILTER, help='MLP width (headline: 256; 512 = reference default)')
    args = ap.parse_args()
    globals()['D_FILTER'] = args.d_filter
    precision = os.environ.get('SUNERF_FORWARD_PRECISION', 'fast').lower()
    fast, half = precision == 'fast', precision == 'half'
    FP8C = ('true' if fast else 'false') + (', true' if half else '')

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    # one rank per GPU; SUNERF_DIST_BACKEND=gloo + several ranks on one card is only for rehearsing the code path
    local_dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    if world > 1:
        backend = os.environ.get('SUNERF_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from sunerf.model.model import NeRF
    from sunerf.rendering.functional import emission_pass
    from sunerf_hip import ops
    from sunerf_hip.dist import shard_range
    from sunerf_hip.train import ClipAdam, training_loss
    from sunerf_hip.rays import observer_rays

    torch.manual_seed(7)
    model = NeRF(d_input=4, d_output=2, n_layers=N_LAYERS, d_filter=D_FILTER).to(dev)
    t_vals = torch.linspace(0., 1., args.samples, device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == 'fwd':
        # every rank renders a full frame seen from its own longitude
        rays_o, rays_d = observer_rays(args.res, theta=-0.3 + 0.05 * rank, device=dev)
        n_rays = rays_o.shape[0]
        times = torch.zeros(n_rays, device=dev)
        z_vals = ops.sample_z(ops.SAMPLER_STRATIFIED, rays_o, rays_d, t_vals, 1.3, 1.0)
        rays_per_step = n_rays

        def step(i):
            return ops.emission_render_fwd(model.packed(), rays_o, rays_d, times, z_vals, reg_radius=1.2,
                                           want_epilogues=True)['image']
        flops_per_sample, kernel_name = flops_fwd(D_FILTER), f'render_fwd_kernel<{D_FILTER}, false, {FP8C}>'
    else:
        # this rank's rows of the frame; batches of --batch rays cycle through them
        r0, r1 = shard_range(args.res, rank, world)
        rays_o, rays_d = observer_rays(args.res, row_start=r0, row_end=r1, device=dev)
        n_local = rays_o.shape[0]
        perm = torch.randperm(n_local, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
        rays_o, rays_d = rays_o[perm].contiguous(), rays_d[perm].contiguous()      # pre-shuffled ray pool
        times = torch.rand(n_local, generator=torch.Generator().manual_seed(0)).to(dev)
        target = torch.rand(n_local, 1, generator=torch.Generator().manual_seed(1)).to(dev)
        z_all = ops.sample_z(ops.SAMPLER_STRATIFIED, rays_o, rays_d, t_vals, 1.3, 1.0)
        B = min(args.batch, n_local)
        n_batches = n_local // B
        rays_per_step = B
        # sunerf.py:31 Adam(lr 1e-4) + run_emission.py:72 gradient_clip_val 0.5, on flat buffers; step() all-reduces the
        # gradient bucket over the ranks (RCCL), then norm -> clip -> Adam in two kernels, no host synchronisation
        opt = ClipAdam(model.parameters(), lr=1e-4, max_norm=0.5)

        def step(i):
            b = (i % n_batches) * B
            sl = slice(b, b + B)
            opt.zero_grad()
            out = emission_pass(model, rays_o[sl], rays_d[sl], times[sl], z_all[sl], 1.2, want_epilogues=True)
            # single-pass workload: the one image plays both roles of sunerf.py:112-119 (0.5 * (mse + mse) = mse)
            loss, stats = training_loss(out['image'], out['image'], target[sl], out['regularization'], 0.5, 1.0,
                                        asinh_scaling=(1.0, 0.005), finite_check=[out['height_map'], out['absorption_map']])
            loss.backward()
            opt.step(skip_if_positive=stats[5:6])
            return loss
        flops_per_sample, kernel_name = flops_fwd(D_FILTER) + flops_bwd(D_FILTER), f'render_fwd_kernel<{D_FILTER}, true, {FP8C}> + dgrad + wgrad'

    for i in range(args.warmup):
        step(i)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()          # on the current stream == the stream the kernels are launched on
        out = step(args.warmup + i)
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    assert torch.isfinite(out).all()

    el = torch.tensor([elapsed], device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    half_line = None
    if precision == 'fast' and not args.no_half:
        # beside the headline: the same step in the opt-in HALF arithmetic (single fp16 MFMA operands, fp32 accumulate --
        # the "bf16 MLP weights on MFMA" class of BASELINE config 3).  Never the headline: it follows an fp16-emulating
        # oracle to 1e-4, not the fp32 reference.
        os.environ['SUNERF_FORWARD_PRECISION'] = 'half'
        model._packed = None
        hs = max(2, args.steps // 2)
        for i in range(2):
            step(i)
        barrier()
        th = time.perf_counter()
        for i in range(hs):
            step(2 + i)
        barrier()
        eh = torch.tensor([time.perf_counter() - th], device=dev)
        if world > 1:
            dist.all_reduce(eh, op=dist.ReduceOp.MAX)
        os.environ['SUNERF_FORWARD_PRECISION'] = precision
        model._packed = None
        half_line = {'value': rays_per_step * args.samples * world * hs / eh.item(), 'unit': 'ray-samples/s',
                     'ms_per_step': eh.item() / hs * 1e3, 'steps': hs,
                     'what': 'the same step with SUNERF_FORWARD_PRECISION=half: single fp16 MFMA operands, fp32 accumulate '
                             '(arithmetic class of BASELINE config 3); parity gate: fp16-emulating oracle at 1e-4, NOT the '
                             'fp32 reference -- reported beside the headline, never as it'}
    two_pass = dt_two_pass = None
    if args.mode == 'train' and not args.no_two_pass:      # after the timed region of the headline metric
        del opt, model
        torch.cuda.empty_cache()
        two_pass = two_pass_rate(dev, world, rays_o, rays_d, times, target, B, args.samples)
        if args.dt:
            torch.cuda.empty_cache()
            dt_two_pass = two_pass_rate(dev, world, rays_o, rays_d, times, target, min(B, 8192), 256, density_temperature=True)
    samples_per_step = rays_per_step * args.samples * world
    value = samples_per_step * args.steps / elapsed

    if rank == 0:
        achieved = rays_per_step * args.samples * flops_per_sample / (step_ms * 1e-3) / 1e12
        # matrix-pipe work of the forward per algorithmic flop: EXACT 3 fp16 products; FAST 1 fp16 product + two 64-deep
        # fp8 instructions per four 16-deep steps (measured 84 cycles each against 4 x 32: tools/probes/bench_mfma_mix.hip)
        fwd_factor = 1.0 if half else (1.0 + 2.0 * 84.0 / 128.0 if fast else 3.0)
        if args.mode == 'fwd':
            executed_factor = fwd_factor
        else:
            dgrad = flops_bwd(D_FILTER) - flops_fwd(D_FILTER)
            executed_factor = (fwd_factor * flops_fwd(D_FILTER) + 2.0 * dgrad + flops_fwd(D_FILTER)) / flops_per_sample
        traffic = None
        for tname in ('hbm_traffic.json', 'hbm_traffic_d512.json'):
            tpath = os.path.join(ROOT, 'profiles', tname)
            if not os.path.exists(tpath):
                continue
            with open(tpath) as f:
                t = json.load(f)
            measured_on = t.get(f'{args.mode}_config', {})
            here = {'rays': rays_per_step, 'samples': args.samples, 'd_filter': D_FILTER}
            if all(measured_on.get(k) == v for k, v in here.items()):
                traffic = t.get(f'{args.mode}_bytes_per_step')
        what = 'fwd+bwd' if args.mode == 'train' else 'fwd'
        line = {
            'metric': f'ray-samples/sec ({what}, fused emission renderer)', 'value': value, 'unit': 'ray-samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16 MFMA operands, fp32 accumulate and parameters (opt-in HALF mode: the bf16-class arithmetic of BASELINE '
                     'config 3; follows an fp16-emulating oracle to 1e-4, NOT the fp32 reference)' if half else
                     ('f32 (forward: every operand split into an fp16 head and an exact remainder, head products on the fp16 '
                      'matrix cores + ' + ('block-scaled fp8 products for the two cross terms' if fast else 'fp16 products for the two cross terms')
                      + '; backward: fp16 MFMA, W^T hi + lo; fp32 accumulate and parameters)'),
            'data': 'synthetic',
            'config': {'workload': (f'emission render {what}, {args.res}x{args.res} frame x {args.samples} samples/ray, '
                                    f'{N_LAYERS}x{D_FILTER} sine MLP, '
                                    + (f'{rays_per_step} rays per optimiser step and GPU (render+loss+backward+all-reduce+clip+Adam)'
                                       if args.mode == 'train' else 'whole frame per step and GPU')),
                       'rays_per_step_per_gpu': rays_per_step, 'samples_per_ray': args.samples, 'mode': args.mode},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F16_DENSE_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F16_DENSE_TFLOPS, 'traffic': traffic,
                         'kernel': kernel_name, 'step_ms_hip_events': step_ms,
                         'flops_per_sample': flops_per_sample,
                         'frac_of_f32_mfma_peak': achieved / PEAK_F32_MFMA_TFLOPS,
                         # matrix-pipe time actually spent, in fp16-MFMA equivalents: forward see fwd_factor, the data
                         # gradient 2 (hi + lo weights), the weight gradient 1
                         'executed_frac': achieved * executed_factor / PEAK_F16_DENSE_TFLOPS,
                         # average HBM rate of the step against the 8 TB/s peak (PMC traffic of this configuration)
                         'hbm_frac': (traffic / (step_ms * 1e-3) / 8e12) if traffic else None},
        }
        if half_line is not None:
            line['half_precision'] = half_line
        if two_pass is not None:
            line['two_pass'] = two_pass
        if dt_two_pass is not None:
            line['dt_two_pass'] = dt_two_pass
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.res, args.samples, args.mode)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
