import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd')
for p in (ROOT, PKG, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope='session')
def golden():
    return load_golden


def params_from_golden(g, prefix):
    """[(W, b)] from the '__'-flattened reference state-dict keys stored in a fixture."""
    sd = {k[len(prefix):].replace('__', '.'): v for k, v in g.items() if k.startswith(prefix)}
    import sunerf_oracle as orc
    return orc.params_from_state_dict(sd, '')


@pytest.fixture(params=['fast', 'exact'])
def precision(request, monkeypatch):
    """Runs a test under both forward arithmetics (include/sunerf_hip.h: SUNERF_PRECISION_*): packed models pick the mode
    up from the environment when they are created."""
    monkeypatch.setenv('SUNERF_FORWARD_PRECISION', request.param)
    return request.param


def gate_units(got, ref, floor=0.0):
    """The north-star bound, per ray and purely relative (not relative to the tensor's maximum; VERDICT r2: the
    1e-6 max|ref| term it still had let a pixel at 1 % of the frame maximum pass at 2e-4 -- removed, every GPU test passes without):
        |got - ref| <= 1e-4 |ref| (+ floor)
    as max over elements of |got - ref| / bound; <= 1 passes.  ``floor``: absolute rounding noise the REFERENCE's own fp32
    evaluation carries, e.g. S * 2^-24 for absorption_map = sum over S samples of the fp32 difference (1 - a)."""
    got, ref = got.detach().cpu().double(), ref.double()
    assert got.shape == ref.shape or got.numel() == ref.numel(), (got.shape, ref.shape)
    got = got.reshape(ref.shape)
    bound = 1e-4 * ref.abs() + floor
    err = (got - ref).abs()
    if not bool((bound > 0).all()):               # an all-zero reference channel (absent wavelength): must be exactly zero too
        assert bool((err[bound == 0] == 0).all())
        err, bound = err[bound > 0], bound[bound > 0]
        if err.numel() == 0:
            return 0.0
    return (err / bound).max().item()


def fp16_chain_bounds(params, rays_o, rays_d, times, z_vals, g_raw):
    """What single-fp16-operand arithmetic CAN deliver for the gradients of a given batch (used by the tests of the fp16
    backward kernels; the product's default path takes the fp32 backward for batches this small, csrc/bwd_exact.hip).

    db_l = sum over samples of t_n, t_n = dZ_l[:, n] (dW_l: t_n = dZ_l[:, n] x X_l[:, n]).  Every term reaches the sum with ~2^-12 of
    relative rounding error from each fp16 operand it passed through on the way down the chain -- g_raw, then (cos_k, dZ_k) for
    every layer k >= l: 2 (L - l) + 1 independent sources, one more (X_l) for the weights -- so the sum carries
    2^-12 sqrt(sources) kappa  with the condition number
        kappa = || sqrt(sum_n t_n^2) || / || sum_n t_n ||      (norms over the tensor's elements),
    ~ 1 / sqrt(N) when the terms share a sign, >> 1 when a small batch cancels (tests/tools/bias_conditioning.py: measured error /
    this model = 0.3 ... 0.9 for kappa from 0.1 to 65 on the CPU emulation, up to 1.5 on the kernels, whose operands also carry the
    per-layer powers of two).  Returns ([(kappa, bound)] for the weights, the same for the biases), bound = max(1e-3, 1.6 x the
    model): 1e-3 is SURVEY 8d's gate, which the kernels must hold wherever the arithmetic allows it."""
    import sunerf_oracle as orc
    S = z_vals.shape[1]
    pts = orc.points_on_rays(rays_o, rays_d, z_vals)
    x = torch.cat([pts, times.reshape(-1, 1)[:, None, :].expand(-1, S, -1)], -1).reshape(-1, 4).double()
    h = orc.positional_encoding(x.float()).double()
    cos, X = [], [h]
    for W, b in params[:-1]:
        zz = h @ W.double().T + b.double()
        cos.append(torch.cos(zz))
        h = torch.sin(zz)
        X.append(h)
    n_lin = len(params)
    dz = g_raw.reshape(-1, g_raw.shape[-1])[:, :params[-1][0].shape[0]].double()
    w_out, b_out = [None] * n_lin, [None] * n_lin
    for l in range(n_lin - 1, -1, -1):
        if l < n_lin - 1:
            dz = (dz @ params[l + 1][0].double()) * cos[l]
        sources = 2 * (n_lin - 1 - l) + 1
        kb = (dz.pow(2).sum(0).sqrt().norm() / dz.sum(0).norm().clamp_min(1e-300)).item()
        kw = ((dz.pow(2).T @ X[l].pow(2)).sum().sqrt() / (dz.T @ X[l]).norm().clamp_min(1e-300)).item()
        b_out[l] = (kb, max(1e-3, 1.6 * 2.0 ** -12 * sources ** 0.5 * kb))
        w_out[l] = (kw, max(1e-3, 1.6 * 2.0 ** -12 * (sources + 1) ** 0.5 * kw))
    return w_out, b_out


def fp16_chain_bias_bounds(params, rays_o, rays_d, times, z_vals, g_raw):
    """The bias half of :func:`fp16_chain_bounds`."""
    return fp16_chain_bounds(params, rays_o, rays_d, times, z_vals, g_raw)[1]
