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
