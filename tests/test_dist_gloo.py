"""world_size-2 data-parallel step on CPU (gloo): the flat gradient bucket all-reduce + clip reproduces the gradient
of the global batch, and both ranks end with identical parameters."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sunerf_hip.dist import GradBucket, shard_range
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))   # replicated parameters
    x = torch.randn(40, 6, generator=torch.Generator().manual_seed(1))
    y = torch.randn(40, 2, generator=torch.Generator().manual_seed(2))
    bucket = GradBucket(net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    b, e = shard_range(40, rank, world)
    for _ in range(3):
        bucket.zero()
        loss = torch.nn.functional.mse_loss(net(x[b:e]), y[b:e])     # mean over the rank's own block
        loss.backward()
        bucket.gather_grads()
        bucket.all_reduce_mean()
        bucket.clip_grad_norm_(0.5)
        opt.step()
    torch.save({'params': [p.detach().clone() for p in net.parameters()], 'grad': bucket.flat.clone()},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def test_two_rank_step_equals_global_batch(tmp_path):
    world, port = 2, 29517
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    for a, b in zip(r0['params'], r1['params']):
        assert torch.equal(a, b)
    # single-process reference on the concatenated batch
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
    x = torch.randn(40, 6, generator=torch.Generator().manual_seed(1))
    y = torch.randn(40, 2, generator=torch.Generator().manual_seed(2))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(x), y).backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 0.5)
        opt.step()
    for a, b in zip(r0['params'], net.parameters()):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-7)
