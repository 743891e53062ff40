"""GPU parity of the output side of the path: fused training loss and clip + Adam step vs the golden
vectors of the torch calls the reference makes (g7) and vs the CPU oracle on larger random inputs."""
import numpy as np
import pytest
import torch

import sunerf_oracle as orc
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def train():
    from sunerf_hip import train as t
    return t


def _T(g, k, dev='cuda'):
    return torch.as_tensor(g[k]).to(dev)


def _rel(a, b):
    return ((a.detach().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_training_loss_golden(train):
    g = load_golden('g7_train_step')
    coarse, fine, reg = (_T(g, k).requires_grad_(True) for k in ('coarse', 'fine', 'reg'))
    loss, stats = train.training_loss(coarse, fine, _T(g, 'target'), reg, float(g['lambda_image']),
                                      float(g['lambda_regularization']), asinh_scaling=(float(g['vmax']), float(g['a'])))
    loss.backward()
    s = stats.cpu()
    for i, k in enumerate(('loss', 'coarse_loss', 'fine_loss', 'reg_loss', 'psnr')):
        assert abs(s[i].item() - float(g[k])) <= 2e-6 * abs(float(g[k])), (k, s[i].item(), float(g[k]))
    assert s[5].item() == 0
    # tolerance 1e-5 of the largest gradient: asinhf / sqrtf of the device library vs glibc (1-2 ulp) through the chain rule
    assert _rel(coarse.grad, g['g_coarse']) < 1e-5
    assert _rel(fine.grad, g['g_fine']) < 1e-5
    assert _rel(reg.grad, g['g_reg']) < 1e-6


@pytest.mark.parametrize('shape,scaled', [((32768, 1), True), ((4099, 7), False)])
def test_training_loss_vs_oracle(train, shape, scaled):
    gen = torch.Generator().manual_seed(3)
    coarse = (torch.rand(shape, generator=gen) * 2).requires_grad_(True)
    fine = (torch.rand(shape, generator=gen) * 2).requires_grad_(True)
    target = torch.rand(shape, generator=gen)
    reg = (torch.rand(shape[0], 96, generator=gen) * 1e-3).requires_grad_(True)
    if scaled:
        ref = orc.emission_training_loss({'coarse_image': coarse, 'fine_image': fine, 'regularization': reg}, target, 1.0, 2.0)
        ref_loss = ref['loss']
    else:   # DensityTemperatureSuNeRFModule.training_step, sunerf.py:185-190: nn.MSELoss on the unscaled images
        mse = torch.nn.MSELoss()
        ref_loss = 1.0 * (mse(coarse, target) + mse(fine, target)) + 2.0 * reg.mean()
    ref_loss.backward()
    c, f, r = (t.detach().cuda().requires_grad_(True) for t in (coarse, fine, reg))
    loss, stats = train.training_loss(c, f, target.cuda(), r, 1.0, 2.0, asinh_scaling=(1.0, 0.005) if scaled else None)
    (3.0 * loss).backward()         # an upstream factor must reach all three gradients
    assert abs(loss.item() - ref_loss.item()) <= 2e-6 * abs(ref_loss.item())
    assert _rel(c.grad, 3.0 * coarse.grad) < 1e-5
    assert _rel(f.grad, 3.0 * fine.grad) < 1e-5
    assert _rel(r.grad, 3.0 * reg.grad) < 1e-6


def test_training_loss_counts_non_finite(train):
    n = 1000
    coarse, fine, target = torch.rand(n, 1).cuda(), torch.rand(n, 1).cuda(), torch.rand(n, 1).cuda()
    reg = torch.rand(n, 8).cuda()
    z = torch.rand(n, 8).cuda()
    z[3, 2] = float('nan')
    z[7, 0] = float('inf')
    reg[5, 5] = float('-inf')
    fine[9, 0] = float('nan')
    _, stats = train.training_loss(coarse, fine, target, reg, asinh_scaling=(1.0, 0.005), finite_check=[z, coarse])
    assert stats[5].item() == 4
    # the workspace is left ready for the next call
    _, stats = train.training_loss(coarse, coarse, target, reg[:3], asinh_scaling=(1.0, 0.005))
    assert stats[5].item() == 0


def test_clip_adam_golden(train):
    """4 steps of clip_grad_norm_(0.5) + Adam(lr 1e-4) + ExponentialLR: parameters, clipped gradients and norms."""
    g = load_golden('g7_train_step')
    params = [torch.nn.Parameter(_T(g, f'p0_{i}')) for i in range(3)]
    opt = train.ClipAdam(params, lr=1e-4, max_norm=0.5)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=(1e-5 / 1e-4) ** (1 / 1e6))
    for step in range(4):
        opt.zero_grad()
        for i, p in enumerate(params):
            if i == 1:
                p.grad = _T(g, f'grad{step}_{i}')          # a gradient autograd allocated outside the bucket
            else:
                p.grad.copy_(_T(g, f'grad{step}_{i}'))     # written in place, like the backward kernels do
        v0 = params[0]._version
        opt.step()
        if sched.get_last_lr()[0] > 5e-5:
            sched.step()
        assert params[0]._version > v0                     # the packed-weights cache keys on it
        assert abs(opt.norm[0].item() - float(g[f'norm{step}'])) <= 1e-6 * float(g[f'norm{step}'])
        assert abs(sched.get_last_lr()[0] - float(g[f'lr{step + 1}'])) < 1e-15
        for i, p in enumerate(params):
            assert _rel(p.grad, g[f'clipped{step}_{i}']) < 1e-6, (step, i)
            assert _rel(p, g[f'p{step + 1}_{i}']) < 1e-6, (step, i)
            # the update itself (1e-4 per step against parameters of 0.1) must be right, not just the parameter
            upd = p.detach().cpu() - g[f'p{step}_{i}']
            ref = g[f'p{step + 1}_{i}'] - g[f'p{step}_{i}']
            assert ((upd - ref).abs().max() / ref.abs().max()).item() < 1e-3, (step, i)


def test_clip_adam_vs_torch_adam_and_state_dict(train):
    gen = torch.Generator().manual_seed(5)
    shapes = [(256, 84), (256,), (256, 256), (2, 256)]
    init = [torch.randn(s, generator=gen) * 0.05 for s in shapes]
    ref_p = [torch.nn.Parameter(t.clone()) for t in init]
    ref_opt = torch.optim.Adam(ref_p, lr=3e-4, betas=(0.8, 0.99), eps=1e-7)
    params = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    opt = train.ClipAdam(params, lr=3e-4, betas=(0.8, 0.99), eps=1e-7, max_norm=None)      # clipping off
    for step in range(6):
        if step == 3:       # round trip through a torch.optim.Adam state dict (a reference checkpoint)
            sd = ref_opt.state_dict()
            params = [torch.nn.Parameter(p.detach().clone()) for p in params]
            opt = train.ClipAdam(params, lr=1.0, max_norm=None)
            opt.load_state_dict(sd)
            assert opt.step_count == 3 and opt.param_groups[0]['lr'] == 3e-4
        grads = [torch.randn(s, generator=gen) for s in shapes]
        for p, q, gr in zip(ref_p, params, grads):
            p.grad = gr.clone()
            q.grad = gr.cuda()
        ref_opt.step()
        opt.step()
    for p, q in zip(ref_p, params):
        assert _rel(q, p.detach()) < 2e-6
    sd = opt.state_dict()
    assert len(sd['state']) == 4 and sd['state'][0]['exp_avg'].shape == (256, 84)


def test_clip_adam_skips_on_flag(train):
    p = torch.nn.Parameter(torch.ones(1000).cuda())
    opt = train.ClipAdam([p], lr=0.1, max_norm=1.0)
    p.grad.fill_(1.0)
    flag = torch.tensor([2.0]).cuda()
    opt.step(skip_if_positive=flag)
    assert torch.equal(p.detach().cpu(), torch.ones(1000))
    assert opt.skipped_last_step() and opt.step_count == 0          # a skipped step does not advance the bias correction
    assert torch.equal(opt.exp_avg.cpu(), torch.zeros(1000))
    flag.zero_()
    p.grad.fill_(1.0)
    opt.step(skip_if_positive=flag)
    assert not opt.skipped_last_step() and opt.step_count == 1
    # first APPLIED step: m / (1 - beta1) = g, sqrt(v / (1 - beta2)) = |g|  ->  the update is lr (0.0744 with step = 2)
    assert (p.detach().cpu() - 0.9).abs().max().item() < 1e-5


def test_clip_adam_skips_on_non_finite_gradient(train):
    """A NaN / Inf that only shows up in the gradients (e.g. an fp16 overflow in the backward pass) must not reach the
    parameters: the total norm is not finite -> the update is skipped, with or without clipping."""
    for max_norm in (0.5, None):
        for bad in (float('nan'), float('inf')):
            p = torch.nn.Parameter(torch.ones(3000).cuda())
            opt = train.ClipAdam([p], lr=0.1, max_norm=max_norm)
            p.grad.fill_(0.5)
            p.grad[1234] = bad
            opt.step()
            assert opt.skipped_last_step() and opt.step_count == 0
            assert torch.equal(p.detach().cpu(), torch.ones(3000))
            assert torch.equal(opt.exp_avg.cpu(), torch.zeros(3000)) and torch.equal(opt.exp_avg_sq.cpu(), torch.zeros(3000))
            opt.zero_grad()
            p.grad.fill_(0.5)
            opt.step()
            assert not opt.skipped_last_step() and opt.step_count == 1 and (p.detach() < 1.0).all()
