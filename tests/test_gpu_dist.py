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


def _module():
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    torch.manual_seed(5)
    return EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 32, 'perturb': False},
                                     hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 32},
                                     model_config={'d_filter': 64}).cuda()


def _batch():
    from sunerf_hip.rays import observer_rays
    o, d = observer_rays(8, device='cuda')                      # 64 rays
    gen = torch.Generator().manual_seed(9)
    return o, d, torch.rand(64, 1, generator=gen).cuda(), torch.rand(64, 1, generator=gen).cuda()


def _train(rendering, o, d, t, target, group_active):
    from sunerf_hip.train import ClipAdam, training_loss
    opt = ClipAdam(rendering.parameters(), lr=1e-3, max_norm=0.5)
    for _ in range(STEPS):
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
    torch.save({'params': params, 'norm': norm}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def test_two_rank_fused_step_equals_single_process(tmp_path):
    world, port = 2, 29541
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    for a, b in zip(r0['params'], r1['params']):
        assert torch.equal(a, b)                                 # replicas stay bit-identical
    assert torch.equal(r0['norm'], r1['norm'])
    _setup_paths()
    o, d, t, target = _batch()
    init = [p.detach().cpu().clone() for p in _module().parameters()]
    ref, ref_norm = _train(_module(), o, d, t, target, False)
    # the global gradient norm (before clipping) is the same whether the batch is sharded or not
    assert abs(r0['norm'][0].item() - ref_norm[0].item()) <= 1e-4 * ref_norm[0].item()
    for a, b, p0 in zip(r0['params'], ref, init):
        moved = (b - p0).abs().max().item()                     # ~3e-3 after three Adam steps of lr 1e-3
        assert (a - b).abs().max().item() <= 0.05 * moved + 1e-7, ((a - b).abs().max().item(), moved)
