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
