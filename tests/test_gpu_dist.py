"""Two ranks sharing one MI355X (gloo for the exchange -- a rehearsal of the one-process-per-GPU RCCL path): the
sharded, fused training step (render -> loss -> backward -> all-reduce -> clip -> Adam) reproduces the single-process step on
the whole batch, and both ranks end with identical parameters."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 3


def _setup_paths():
    for p in (os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _module(d_filter=64):
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    torch.manual_seed(5)
    return EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 32, 'perturb': False},
                                     hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 32},
                                     model_config={'d_filter': d_filter}).cuda()


def _batch():
    from sunerf_hip.rays import observer_rays
    o, d = observer_rays(8, device='cuda')                      # 64 rays
    gen = torch.Generator().manual_seed(9)
    return o, d, torch.rand(64, 1, generator=gen).cuda(), torch.rand(64, 1, generator=gen).cuda()


def _train(rendering, o, d, t, target, group_active, steps=STEPS, overlap=False):
    from sunerf_hip.train import ClipAdam, training_loss
    opt = ClipAdam(rendering.parameters(), lr=1e-3, max_norm=0.5, overlap=overlap)
    opt.reduce_single_rank = True
    early = []
    real_ready = opt.segment_ready

    def spy(params):
        real_ready(params)
        early.append(len(opt._early))
    opt.segment_ready = spy
    _train.early = early
    for _ in range(steps):
        opt.zero_grad()
        out = rendering(o, d, t)
        loss, stats = training_loss(out['coarse_image'], out['fine_image'], target, out['regularization'], 1.0, 1.0,
                                    asinh_scaling=(1.0, 0.005))
        loss.backward()
        opt.step(skip_if_positive=stats[5:6])
    return [p.detach().cpu().clone() for p in rendering.parameters()], opt.norm.cpu().clone()


def _worker(rank, world, port, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sunerf_hip.dist import shard_range
    o, d, t, target = _batch()
    b, e = shard_range(o.shape[0], rank, world)
    params, norm = _train(_module(), o[b:e].contiguous(), d[b:e].contiguous(), t[b:e].contiguous(), target[b:e].contiguous(), True)
    # the same with the fine model's slice of the bucket all-reduced as soon as its backward has finished (overlap=True)
    params_o, norm_o = _train(_module(), o[b:e].contiguous(), d[b:e].contiguous(), t[b:e].contiguous(), target[b:e].contiguous(),
                              True, overlap=True)
    torch.save({'params': params, 'norm': norm, 'params_overlap': params_o, 'norm_overlap': norm_o, 'early': _train.early},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def test_two_rank_fused_step_equals_single_process(tmp_path):
    world, port = 2, 29541
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    for a, b in zip(r0['params'], r1['params']):
        assert torch.equal(a, b)                                 # replicas stay bit-identical
    assert torch.equal(r0['norm'], r1['norm'])
    # overlapped reduction: two slices went out early in every step (fine model first, then coarse), same sums bit for bit
    assert r0['early'] == [1, 2] * STEPS, r0['early']
    assert torch.equal(r0['norm_overlap'], r0['norm'])
    for a, b in zip(r0['params_overlap'], r0['params']):
        assert torch.equal(a, b)
    _setup_paths()
    o, d, t, target = _batch()
    init = [p.detach().cpu().clone() for p in _module().parameters()]
    ref, ref_norm = _train(_module(), o, d, t, target, False)
    # the global gradient norm (before clipping) is the same whether the batch is sharded or not
    assert abs(r0['norm'][0].item() - ref_norm[0].item()) <= 1e-4 * ref_norm[0].item()
    for a, b, p0 in zip(r0['params'], ref, init):
        moved = (b - p0).abs().max().item()                     # ~3e-3 after three Adam steps of lr 1e-3
        assert (a - b).abs().max().item() <= 0.05 * moved + 1e-7, ((a - b).abs().max().item(), moved)


def _worker_nan(rank, world, port, out_dir):
    """Step 0 and 2 are clean, in step 1 the target of rank 1 holds a NaN (-> its loss, its image gradients and therefore
    its parameter gradients are NaN, and its non-finite counter is 1 while rank 0's is 0)."""
    _setup_paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sunerf_hip.dist import shard_range
    from sunerf_hip.train import ClipAdam, training_loss
    o, d, t, target = _batch()
    b, e = shard_range(o.shape[0], rank, world)
    o, d, t, target = (x[b:e].contiguous() for x in (o, d, t, target))
    rendering = _module()
    opt = ClipAdam(rendering.parameters(), lr=1e-3, max_norm=0.5)
    log = []
    for step in range(3):
        tgt = target.clone()
        if step == 1 and rank == 1:
            tgt[3, 0] = float('nan')
        opt.zero_grad()
        out = rendering(o, d, t)
        loss, stats = training_loss(out['coarse_image'], out['fine_image'], tgt, out['regularization'], 1.0, 1.0,
                                    asinh_scaling=(1.0, 0.005), finite_check=[tgt])
        loss.backward()
        opt.step(skip_if_positive=stats[5:6])
        log.append({'skipped': opt.skipped_last_step(), 'count': opt.nonfinite.item(), 'steps': opt.step_count,
                    'params': [p.detach().cpu().clone() for p in rendering.parameters()]})
    torch.save(log, os.path.join(out_dir, f'nan_rank{rank}.pt'))
    dist.destroy_process_group()


def test_non_finite_on_one_rank_skips_the_step_on_every_rank(tmp_path):
    """SURVEY.md 8e / sunerf.py:105-107: the non-finite counter rides at the tail of the all-reduced gradient bucket, so the
    rank that did NOT see the NaN skips too -- replicas stay bit-identical and finite, and the step counter of the bias
    correction does not advance on either."""
    world, port = 2, 29547
    mp.spawn(_worker_nan, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'nan_rank0.pt')
    r1 = torch.load(tmp_path / 'nan_rank1.pt')
    assert [x['skipped'] for x in r0] == [False, True, False] == [x['skipped'] for x in r1]
    assert [x['steps'] for x in r0] == [1, 1, 2] == [x['steps'] for x in r1]
    assert r0[1]['count'] == r1[1]['count'] and r0[1]['count'] >= 1          # the GLOBAL count, on both ranks
    for s0, s1 in zip(r0, r1):
        for a, b in zip(s0['params'], s1['params']):
            assert torch.equal(a, b) and torch.isfinite(a).all()
    for a, b in zip(r0[0]['params'], r0[1]['params']):
        assert torch.equal(a, b)                                             # the skipped step changed nothing
    assert any(not torch.equal(a, b) for a, b in zip(r0[1]['params'], r0[2]['params']))


def _worker_rccl(rank, world, port, out_dir):
    _setup_paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world)
    o, d, t, target = _batch()
    params, norm = _train(_module(), o, d, t, target, True, steps=2, overlap=True)     # async slices + remainder, through RCCL
    torch.save({'params': params, 'norm': norm, 'early': _train.early}, os.path.join(out_dir, 'rccl.pt'))
    dist.destroy_process_group()


def test_rccl_backend_runs_the_bucket_all_reduce(tmp_path):
    """backend='nccl' IS RCCL on ROCm.  One GPU cannot host two RCCL ranks, so this is the world-size-1 case through the real
    library (communicator set-up + the all-reduce call on the flat bucket inside ClipAdam.step): it must equal the
    process-group-free step.  The N > 1 exchange itself is covered with gloo above and run by the driver on 8 GPUs."""
    mp.spawn(_worker_rccl, args=(1, 29549, str(tmp_path)), nprocs=1, join=True)
    got = torch.load(tmp_path / 'rccl.pt')
    assert got['early'] == [1, 2] * 2          # both models' slices went out through the asynchronous path
    _setup_paths()
    o, d, t, target = _batch()
    ref, ref_norm = _train(_module(), o, d, t, target, False, steps=2)
    assert torch.equal(got['norm'], ref_norm)
    for a, b in zip(got['params'], ref):
        assert torch.equal(a, b)


def _worker_rccl_pipe(rank, world, port, out_dir):
    """d_filter = 256: the backward of both models is the PERSISTENT layer-pipelined launch (256 workgroups that must all become
    resident, csrc/bwd_pipe.hip), next to RCCL kernels on the collective's stream -- with the early slice all-reduce
    (overlap=True: an RCCL kernel is in flight while the coarse model's pipelined backward is launched) and without."""
    _setup_paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.pop('SUNERF_BACKWARD', None)
    os.environ['SUNERF_EXACT_BACKWARD_SAMPLES'] = '0'      # 64 rays x 32 / 64 samples would take the small-batch fp32 backward
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world)
    from sunerf_hip import ops
    o, d, t, target = _batch()
    res = {}
    for overlap in (False, True):
        assert ops.backward_mode() == 'pipe'
        ops.pipe_kernel_time()
        ops.pipe_timing = True
        params, norm = _train(_module(256), o, d, t, target, True, steps=2, overlap=overlap)
        ops.pipe_timing = False
        torch.cuda.synchronize()
        res[overlap] = {'params': params, 'norm': norm, 'early': list(_train.early), 'status': ops.pipe_status(raise_on_failure=False),
                        'pipelined_launches': ops.pipe_kernel_time()[1], 'mode_after': ops.backward_mode()}
    torch.save(res, os.path.join(out_dir, 'rccl_pipe.pt'))
    dist.destroy_process_group()


def test_rccl_next_to_the_persistent_pipelined_backward(tmp_path, monkeypatch):
    """VERDICT r3 / ADVICE r3: the only way on a 1-GPU lease to see an RCCL kernel and the 256-workgroup persistent launch on one
    device.  World size 1 through the real library; status 0 (no start-up or hand-off time-out), every backward really was the
    pipelined kernel (2 steps x 2 models), parameters and norm equal the group-free step bit for bit, overlap on and off."""
    monkeypatch.setenv('SUNERF_EXACT_BACKWARD_SAMPLES', '0')
    mp.spawn(_worker_rccl_pipe, args=(1, 29551, str(tmp_path)), nprocs=1, join=True)
    got = torch.load(tmp_path / 'rccl_pipe.pt')
    _setup_paths()
    from sunerf_hip import ops
    o, d, t, target = _batch()
    assert ops.backward_mode() == 'pipe'
    ref, ref_norm = _train(_module(256), o, d, t, target, False, steps=2)
    torch.cuda.synchronize()
    assert ops.pipe_status(raise_on_failure=False) == 0
    for overlap in (False, True):
        r = got[overlap]
        assert r['status'] == 0 and r['mode_after'] == 'pipe', (overlap, r['status'], r['mode_after'])
        assert r['pipelined_launches'] == 4, r['pipelined_launches']
        assert r['early'] == ([1, 2] * 2 if overlap else [0, 0] * 2), r['early']
        assert torch.equal(r['norm'], ref_norm)
        for a, b in zip(r['params'], ref):
            assert torch.equal(a, b)

