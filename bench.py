#!/usr/bin/env python3
"""Headline benchmark: ray-samples/s (forward + backward) of the fused emission renderer on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank/GPU)

Workload (BASELINE.json metric: ray-samples/sec/GPU (fwd+bwd), 1024^2 image x 128 samples): a synthetic 1024 x 1024
observer frame, 128 samples per ray, 8 x 256 sine MLP, fp32 parameters.  One step = one optimiser
step of the hot path on this rank's next batch of ``--batch`` rays (default 32768 = 1/32 of the frame) that are
resident in HBM before the timed region: sample placement, fused forward render (with activation stash), loss of sunerf.py:110-120
(asinh-scaled MSE + regularization mean), backward (integral, dgrad, wgrad), gradient all-reduce over the ranks
(RCCL), clip_grad_norm_(0.5) and Adam.  Rays are sharded over ranks by image rows, every rank works on equally sized
batches of its own rows (weak scaling); value = ray-samples of all ranks / max-over-ranks wall time.
``--mode fwd`` times the inference render of the whole frame instead (configs[1] of BASELINE.json).

The JSON line carries
  roofline     : algorithmic GEMM FLOPs of the dominant kernel (the fused render pass, sunerf_emission_render_fwd) per launch
                 / its average duration between HIP events recorded around its launches in the timed region, on the launch
                 stream, against the dense f16 MFMA peak; "step" inside it: the same for all kernels of a step together
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




def two_pass_rate(dev, world, rays_o, rays_d, times, target, batch, samples, steps=10, warmup=2, density_temperature=False):
    """Reference-shaped training step beside the single-pass headline: StratifiedSampler (S/2 jittered
    samples) -> coarse pass -> HierarchicalSampler (S/2 more) -> fine pass over S samples -> loss on both images ->
    backward through both models -> all-reduce + clip + Adam.  MLP evaluations per ray: S/2 + S."""
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    from sunerf_hip.train import ClipAdam, env_flag, training_loss
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
    # overlap (the fine model's slice of the bucket all-reduced while the coarse model's backward runs): the SAME
    # switch and default as the product's fit_steps (sunerf/model/sunerf.py) -- off unless SUNERF_OVERLAP=1, until a 2..8-GPU
    # RCCL record of the early collectives exists -- so this line measures what the product does
    opt = ClipAdam(rendering.parameters(), lr=1e-4, max_norm=0.5, overlap=env_flag('SUNERF_OVERLAP'))
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
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default: 40 in train mode = 1 s of timed work, visible to a '
                                                            '1 Hz utilisation sampler; 8 frames in fwd mode = 2 s)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed steps before (default 4 / 2)')
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--samples', type=int, default=128)
    ap.add_argument('--batch', type=int, default=32768, help='rays per rank and optimiser step (train mode)')
    ap.add_argument('--mode', choices=['train', 'fwd'], default='train')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-half', action='store_true', help='skip BASELINE config 3 (the same step with single fp16 MFMA operands in '
                                                           'the forward, reported under "half_precision", never as the headline)')
    ap.add_argument('--no-exact', action='store_true', help='skip the sub-line in the unconditional EXACT arithmetic')
    ap.add_argument('--no-two-pass', action='store_true', help='skip the reference-shaped two-pass figure (train mode)')
    ap.add_argument('--no-small-batch', action='store_true', help="skip the figure at the reference's own batch of 3072 rays")
    ap.add_argument('--no-dt', action='store_true', help='skip BASELINE config 5: density-temperature head, two-pass, 128 + 256 '
                                                         'samples per ray (reported under "dt_two_pass")')
    ap.add_argument('--d-filter', type=int, default=D_FILTER, help='MLP width (headline: 256; 512 = reference default)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 40 if args.mode == 'train' else 8
    if args.warmup is None:
        args.warmup = 4 if args.mode == 'train' else 2
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` started directly: become the launcher.  A child process, started before anything here
        # has touched a GPU (never an exec: see the GPU-box rules), one rank per GPU over RCCL; its output is relayed.
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29577'),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    globals()['D_FILTER'] = args.d_filter
    precision = os.environ.get('SUNERF_FORWARD_PRECISION', 'auto').lower()
    half = precision == 'half'

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

    # HIP events around every launch of the dominant kernel (the fused render pass) inside the timed region, recorded on
    # the stream the kernel is launched on (torch's current stream: ops._stream)
    render_events, recording = [], {'on': False}
    real_render = ops.emission_render_fwd

    def timed_render(*a, **k):
        if not recording['on']:
            return real_render(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = real_render(*a, **k)
        e1.record()
        render_events.append((e0, e1))
        return out
    ops.emission_render_fwd = timed_render
    # timing-only ablation builds (tools/experiments): their outputs are wrong by construction
    check_finite = os.environ.get('SUNERF_BENCH_ABLATION', '') in ('', '0')

    def timed_loop(step, steps, warmup):
        """``warmup`` untimed steps, then exactly ``steps`` steps between two barrier + synchronize pairs; max over ranks."""
        for i in range(warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            out = step(warmup + i)
        barrier()
        el = torch.tensor([time.perf_counter() - t0], device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        assert not check_finite or torch.isfinite(out).all()
        return el.item()

    if args.mode == 'fwd':
        # every rank renders a full frame seen from its own longitude; a step = sample placement + fused render of the frame
        rays_o, rays_d = observer_rays(args.res, theta=-0.3 + 0.05 * rank, device=dev)
        n_rays = rays_o.shape[0]
        times = torch.zeros(n_rays, device=dev)
        rays_per_step = n_rays

        def step(i):
            z_vals = ops.sample_z(ops.SAMPLER_STRATIFIED, rays_o, rays_d, t_vals, 1.3, 1.0)
            return ops.emission_render_fwd(model.packed(), rays_o, rays_d, times, z_vals, reg_radius=1.2,
                                           want_epilogues=True)['image']
        flops_step = flops_fwd(D_FILTER)
    else:
        # this rank's rows of the frame; batches of --batch rays cycle through them
        r0, r1 = shard_range(args.res, rank, world)
        rays_o, rays_d = observer_rays(args.res, row_start=r0, row_end=r1, device=dev)
        n_local = rays_o.shape[0]
        perm = torch.randperm(n_local, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
        rays_o, rays_d = rays_o[perm].contiguous(), rays_d[perm].contiguous()      # pre-shuffled ray pool
        times = torch.rand(n_local, generator=torch.Generator().manual_seed(0)).to(dev)
        target = torch.rand(n_local, 1, generator=torch.Generator().manual_seed(1)).to(dev)
        B = min(args.batch, n_local)
        rays_per_step = B
        # sunerf.py:31 Adam(lr 1e-4) + run_emission.py:72 gradient_clip_val 0.5, on flat buffers; step() all-reduces the
        # gradient bucket (+ the non-finite count at its tail) over the ranks (RCCL), then norm -> clip -> Adam in two
        # kernels, no host synchronisation
        opt = ClipAdam(model.parameters(), lr=1e-4, max_norm=0.5)

        def make_step(batch):
            n_batches = n_local // batch

            def step(i):
                b = (i % n_batches) * batch
                sl = slice(b, b + batch)
                opt.zero_grad()
                # a-1 (sampling.py:68-91) is part of the step: sample placement for this batch
                z = ops.sample_z(ops.SAMPLER_STRATIFIED, rays_o[sl], rays_d[sl], t_vals, 1.3, 1.0)
                out = emission_pass(model, rays_o[sl], rays_d[sl], times[sl], z, 1.2, want_epilogues=True)
                # single-pass workload: the one image plays both roles of sunerf.py:112-119 (0.5 * (mse + mse) = mse)
                loss, stats = training_loss(out['image'], out['image'], target[sl], out['regularization'], 0.5, 1.0,
                                            asinh_scaling=(1.0, 0.005), finite_check=[out['height_map'], out['absorption_map']])
                loss.backward()
                opt.step(skip_if_positive=stats[5:6])
                return loss
            return step
        step = make_step(B)
        flops_step = flops_fwd(D_FILTER) + flops_bwd(D_FILTER)

    # ---- the headline: W warm-up steps, exactly K timed steps ----
    for i in range(args.warmup):
        step(i)
    if getattr(model.packed(), 'auto', False):
        model.packed().wait_probe()          # the AUTO probe of the warm-up is read outside the timed region
    if args.mode == 'train':
        torch.cuda.synchronize()
        ops.pipe_status(raise_on_failure=False)   # a pipelined backward that gave up in the warm-up: the timed steps use the two-kernel one
    barrier()
    recording['on'] = True
    # the pipelined backward kernel of every step is bracketed by HIP events INSIDE the C ABI (flags bit 7 of
    # sunerf_mlp_backward_pipe, on the launch stream): the call sequence stays the product's -- one call per backward
    ops.pipe_kernel_time()          # (forget launches timed earlier)
    ops.pipe_timing = args.mode == 'train'
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    recording['on'] = False
    ops.pipe_timing = False
    render_ms = sum(a.elapsed_time(b) for a, b in render_events) / max(1, len(render_events))
    bwd_total_ms, bwd_launches = ops.pipe_kernel_time()
    bwd_ms = bwd_total_ms / bwd_launches if bwd_launches else None
    backward_used = 'pipe' if bwd_launches else ('classic' if args.mode == 'train' else None)
    pipe_failed = ops.pipe_status(raise_on_failure=False) if args.mode == 'train' else 0
    assert not check_finite or torch.isfinite(out).all()
    el = torch.tensor([elapsed], device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    mode_used = ops.PRECISION_NAMES[model.packed().precision]       # what AUTO settled on (fast unless the probe objected)
    probe_units = model.packed().last_probe
    # W^T of the pipelined backward's data gradient: a single fp16 image while the measured probe allows it (ops._pipe_w_probe)
    pipe_single_w = bool(getattr(model.packed(), 'pipe_hi_only', False)) or ops.pipe_w_mode() == 'hi'
    pipe_w_probe = getattr(model.packed(), 'pipe_w_probe', None)

    extras = {}
    if args.mode == 'train' and not args.no_small_batch and n_local >= 3072:
        # the reference's own batch (config/sunerfs_simple_star.yaml:8: 3072 rays per GPU): launch-bound regime
        ss = 50
        e = timed_loop(make_step(3072), ss, 5)
        extras['small_batch'] = {'value': 3072 * args.samples * world * ss / e, 'unit': 'ray-samples/s', 'rays_per_step_per_gpu': 3072,
                                 'ms_per_step': e / ss * 1e3, 'steps': ss,
                                 'what': "the same step at the reference's own batch size (config/sunerfs_simple_star.yaml:8)"}
    if not args.no_half and not half:
        # BASELINE config 3 beside the headline: the same step in the HALF arithmetic (single fp16 MFMA operands, fp32
        # accumulate -- the "bf16 MLP weights on MFMA" class, with fp16's three extra mantissa bits).  Never the headline: its
        # forward outputs follow an fp16-emulating oracle to 1e-4 and are ~1e-3 from the fp32 reference; its training
        # gradients are within 1e-3 of the fp32 oracle (tests/test_gpu_backward.py::test_half_mode_gradients_...: 8.5e-4).
        os.environ['SUNERF_FORWARD_PRECISION'] = 'half'
        model._packed = None
        hs = max(2, args.steps // 2)
        e = timed_loop(step, hs, 2)
        os.environ['SUNERF_FORWARD_PRECISION'] = precision
        model._packed = None
        extras['half_precision'] = {'value': rays_per_step * args.samples * world * hs / e, 'unit': 'ray-samples/s',
                                    'ms_per_step': e / hs * 1e3, 'steps': hs,
                                    'what': 'BASELINE config 3 class: the same step with SUNERF_FORWARD_PRECISION=half (opt-in mode): single '
                                            'fp16 MFMA operands, fp32 accumulate; forward outputs follow an fp16-emulating oracle at 1e-4 '
                                            '(~1e-3 from the fp32 reference), training gradients within 1e-3 of the fp32 oracle (measured '
                                            '8.5e-4) -- reported beside the headline, never as it'}
    if not args.no_exact and precision == 'auto':
        # the arithmetic that holds the 1e-4 gate unconditionally (three fp16 products per term), beside the AUTO headline
        os.environ['SUNERF_FORWARD_PRECISION'] = 'exact'
        model._packed = None
        xs = max(2, args.steps // 2)
        e = timed_loop(step, xs, 2)
        os.environ['SUNERF_FORWARD_PRECISION'] = precision
        model._packed = None
        extras['exact'] = {'value': rays_per_step * args.samples * world * xs / e, 'unit': 'ray-samples/s',
                           'ms_per_step': e / xs * 1e3, 'steps': xs,
                           'what': 'the same step with SUNERF_FORWARD_PRECISION=exact: every product from three fp16 MFMA products (fp32-class '
                                   'results whatever the weights); the headline runs the AUTO policy (fast, guarded by a measured probe)'}
    if args.mode == 'train' and not args.no_two_pass:      # after the timed region of the headline metric
        del opt, model
        torch.cuda.empty_cache()
        extras['two_pass'] = two_pass_rate(dev, world, rays_o, rays_d, times, target, B, args.samples)
        if not args.no_dt:
            torch.cuda.empty_cache()
            extras['dt_two_pass'] = two_pass_rate(dev, world, rays_o, rays_d, times, target, min(B, 8192), 256,
                                                  density_temperature=True)
    samples_per_step = rays_per_step * args.samples * world
    value = samples_per_step * args.steps / elapsed

    if rank == 0:
        fast = mode_used == 'fast'
        # template arguments as rocprofv3 prints them: <d_filter, stash (0 none, 1 fp16 sin + cos, 2 16-bit phases), fp8 cross terms, half>
        stash_id = 0 if args.mode != 'train' else (2 if bwd_ms and D_FILTER == 256 else 1)
        kernel_name = f"render_fwd_kernel<{D_FILTER}, {stash_id}, {'true' if fast else 'false'}, {'true' if half else 'false'}>"
        # dominant kernel = the fused render pass: algorithmic GEMM FLOPs of one launch / its average duration between the
        # HIP events recorded around its launches in the timed region
        achieved = rays_per_step * args.samples * flops_fwd(D_FILTER) / (render_ms * 1e-3) / 1e12
        # the pipelined backward (csrc/bwd_pipe.hip) is ONE kernel with the data AND weight gradients of every layer: when it
        # runs it is the longest kernel of the step, i.e. the dominant one
        bwd_kernel = None
        if bwd_ms:
            bwd_achieved = rays_per_step * args.samples * flops_bwd(D_FILTER) / (bwd_ms * 1e-3) / 1e12
            bwd_kernel = {'kernel': 'bwd_pipe_kernel<true>' if pipe_single_w else 'bwd_pipe_kernel<false>',
                          'kernel_ms_hip_events': bwd_ms, 'launches_timed': bwd_launches, 'achieved': bwd_achieved,
                          'frac': bwd_achieved / PEAK_F16_DENSE_TFLOPS, 'flops_per_sample': flops_bwd(D_FILTER),
                          'executed_frac': bwd_achieved * ((2 if pipe_single_w else 3) * 7 * D_FILTER * D_FILTER + 2 * ENC * D_FILTER) * 2
                                           / flops_bwd(D_FILTER) / PEAK_F16_DENSE_TFLOPS,
                          'what': ('dgrad (W^T as ' + ('a single fp16 image: the measured policy of ops._pipe_w_probe allows it' if pipe_single_w
                                                      else 'fp16 head + remainder: two products')
                                   + ') + wgrad of all layers but the out layer in one persistent launch; the out layer and dZ of the '
                                   'last activation layer come from a 0.9 ms prologue kernel'),
                          'weight_precision': f"{ops.pipe_w_mode()} -> {'single fp16' if pipe_single_w else 'fp16 head + remainder'}"
                                              + (f' (probe: worst weight tensor differs by {pipe_w_probe:.2e}, limit {ops.PIPE_W_LIMIT:.0e})'
                                                 if pipe_w_probe is not None else '')}
        step_ms = elapsed / args.steps * 1e3
        step_achieved = rays_per_step * args.samples * flops_step / (step_ms * 1e-3) / 1e12
        # matrix-pipe work of the forward per algorithmic flop: EXACT 3 fp16 products; FAST 1 fp16 product + two 64-deep
        # fp8 instructions per four 16-deep steps (64 cycles each against 4 x 32: tools/probes/probe_mfma_i8.hip)
        fwd_factor = 1.0 if half else (1.0 + 2.0 * 64.0 / 128.0 if fast else 3.0)
        # What the arithmetic itself allows on this chip (tools/probes/mfma_sustained.hip, profiles/r2_mfma_sustained.txt: nothing
        # but the matrix instructions of the mode, operands in registers, every pipe busy, random operands): the board's power
        # cap holds a pure fp16 MFMA stream at 0.72 of the nominal peak, the FAST group (4 fp16 + 2 fp8 per 64-deep product) at
        # 0.395 in useful flops, three fp16 products (EXACT) at 0.72 / 3.  No kernel computing these results can exceed it.
        arithmetic_ceiling = 0.72 if half else (0.395 if fast else 0.24)
        traffic = traffic_source = bwd_traffic = None
        for tname in ('hbm_traffic.json', 'hbm_traffic_d512.json'):
            tpath = os.path.join(ROOT, 'profiles', tname)
            if not os.path.exists(tpath):
                continue
            with open(tpath) as f:
                t = json.load(f)
            measured_on = t.get(f'{args.mode}_config', {})
            here = {'rays': rays_per_step, 'samples': args.samples, 'd_filter': D_FILTER}
            if all(measured_on.get(k) == v for k, v in here.items()):
                traffic = t.get(f'{args.mode}_kernel_bytes', t.get(f'{args.mode}_bytes_per_step'))
                bwd_traffic = t.get(f'{args.mode}_bwd_kernel_bytes')
                traffic_source = (f'profiles/{tname}: rocprofv3 PMC (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) of this '
                                  'configuration, ' + ('the render kernel alone' if f'{args.mode}_kernel_bytes' in t else
                                                       'whole step') + ' -- not measured in this run')
        what = 'fwd+bwd' if args.mode == 'train' else 'fwd'
        line = {
            'metric': f'ray-samples/sec ({what}, fused emission renderer)', 'value': value, 'unit': 'ray-samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': step_ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16 MFMA operands, fp32 accumulate and parameters (opt-in HALF mode: the bf16-class arithmetic of BASELINE '
                     'config 3; follows an fp16-emulating oracle to 1e-4, NOT the fp32 reference)' if half else
                     ('f32 (forward: every operand split into an fp16 head and an exact remainder, head products on the fp16 '
                      'matrix cores + ' + ('block-scaled fp8 products for the two cross terms' if fast else 'fp16 products for the two cross terms')
                      + '; backward: fp16 MFMA, W^T ' + ('as a single fp16 image (measured policy)' if pipe_single_w else 'hi + lo')
                      + '; fp32 accumulate and parameters)'),
            'data': 'synthetic',
            'config': {'workload': (f'emission render {what}, {args.res}x{args.res} frame x {args.samples} samples/ray, '
                                    f'{N_LAYERS}x{D_FILTER} sine MLP, '
                                    + (f'{rays_per_step} rays per optimiser step and GPU (sample placement+render+loss+backward+'
                                       'all-reduce+clip+Adam)' if args.mode == 'train' else 'whole frame per step and GPU (sample '
                                       'placement + render)')),
                       'rays_per_step_per_gpu': rays_per_step, 'samples_per_ray': args.samples, 'mode': args.mode,
                       'forward_precision': f'{precision} -> {mode_used}' + (f' (probe: {probe_units:.3f} gate units)'
                                                                             if probe_units is not None else '')},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F16_DENSE_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F16_DENSE_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': kernel_name, 'kernel_ms_hip_events': render_ms, 'launches_timed': len(render_events),
                         'flops_per_sample': flops_fwd(D_FILTER),
                         'frac_of_f32_mfma_peak': achieved / PEAK_F32_MFMA_TFLOPS,
                         # matrix-pipe time actually spent by this kernel, in fp16-MFMA equivalents
                         'executed_frac': achieved * fwd_factor / PEAK_F16_DENSE_TFLOPS,
                         # ceiling of the forward arithmetic under the power cap (fraction of `peak`, measured) and our share of it
                         'arithmetic_ceiling_frac': arithmetic_ceiling,
                         'frac_of_arithmetic_ceiling': achieved / PEAK_F16_DENSE_TFLOPS / arithmetic_ceiling,
                         # the whole step (all kernels) on the same scale: algorithmic FLOPs of the step / ms_per_step
                         'step': {'achieved': step_achieved, 'frac': step_achieved / PEAK_F16_DENSE_TFLOPS,
                                  'flops_per_sample': flops_step}},
        }
        line['config']['backward'] = backward_used
        if pipe_failed:
            line['config']['backward'] = f'pipe GAVE UP (status {pipe_failed}): steps were skipped, the record is invalid'
        if bwd_kernel:
            if bwd_ms > render_ms:
                # dominant kernel of the step = the pipelined backward; the render kernel's figures stay beside it
                fwd_view = {k: line['roofline'][k] for k in ('kernel', 'kernel_ms_hip_events', 'launches_timed', 'achieved', 'frac',
                                                             'flops_per_sample', 'executed_frac', 'arithmetic_ceiling_frac',
                                                             'frac_of_arithmetic_ceiling', 'traffic', 'traffic_source')}
                line['roofline'].update({k: bwd_kernel[k] for k in ('kernel', 'kernel_ms_hip_events', 'launches_timed', 'achieved', 'frac',
                                                                    'flops_per_sample', 'executed_frac', 'weight_precision')})
                line['roofline'].pop('frac_of_f32_mfma_peak')      # (an fp16-operand kernel: only the f16 peak is its scale)
                line['roofline']['what'] = bwd_kernel['what']
                line['roofline'].pop('arithmetic_ceiling_frac'); line['roofline'].pop('frac_of_arithmetic_ceiling')
                line['roofline']['traffic'] = bwd_traffic
                line['roofline']['traffic_source'] = traffic_source.replace('the render kernel alone', 'the pipelined backward kernel alone') if bwd_traffic and traffic_source else None
                line['roofline']['render_kernel'] = fwd_view
            else:
                line['roofline']['backward_kernel'] = bwd_kernel
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.res, args.samples, args.mode)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
