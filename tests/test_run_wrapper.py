"""CPU-only: the multi-GPU entry for the reference's unchanged run scripts (sunerf/run_mi355x.py), the refusal of
``nn.DataParallel`` replicas (run_emission.py:69, run_density_temperature.py:84 choose strategy='dp' on a multi-GPU node) and the
shared-device rule of the pipelined backward."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd')


def test_visible_device_selection():
    from sunerf.run_mi355x import visible_device_for
    assert visible_device_for(3, {}) == ('HIP_VISIBLE_DEVICES', '3')
    assert visible_device_for(1, {'HIP_VISIBLE_DEVICES': '4,6'}) == ('HIP_VISIBLE_DEVICES', '6')
    assert visible_device_for(0, {'CUDA_VISIBLE_DEVICES': '2, 3'}) == ('CUDA_VISIBLE_DEVICES', '2')
    with pytest.raises(RuntimeError, match='lists only 2'):
        visible_device_for(2, {'HIP_VISIBLE_DEVICES': '4,6'})


def test_data_parallel_replicas_are_refused_loudly():
    """What torch.nn.parallel.replicate() calls on every submodule of a DataParallel-wrapped module: the renderer and its field
    models answer with an error that names the supported launch instead of computing on another device's packed weights."""
    from sunerf.model.model import NeRF
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    from sunerf_hip.lib import SunerfHipError
    mod = EmissionRadiativeTransfer(Rs_per_ds=1.0, model_config={'d_filter': 64})
    for m in (mod, mod.fine_model, NeRF(d_filter=64)):
        with pytest.raises(SunerfHipError, match='torch.distributed.run.*sunerf.run_mi355x'):
            m._replicate_for_data_parallel()
    # the path DataParallel takes (replicate walks network.modules()); plain modules still replicate
    plain = torch.nn.Linear(2, 2)
    assert plain._replicate_for_data_parallel() is not plain
    wrapped = torch.nn.Sequential(plain, mod)
    with pytest.raises(SunerfHipError):
        for m in wrapped.modules():
            m._replicate_for_data_parallel()


def test_shared_device_rule():
    from sunerf_hip import dist as sd
    assert not sd.any_shared([('a', 'gpu0'), ('a', 'gpu1'), ('b', 'gpu0')])
    assert sd.any_shared([('a', 'gpu0'), ('a', 'gpu1'), ('a', 'gpu0')])
    assert not sd.ranks_share_a_device('cuda:0') and not sd.shared_device_known('cuda:0')          # no process group: never
    words = sd.identity_words(('a', 'gpu0'))
    assert words == sd.identity_words(('a', 'gpu0')) != sd.identity_words(('a', 'gpu1')) and all(-2 ** 63 <= w < 2 ** 63 for w in words)


def test_wrapper_runs_an_unchanged_script_in_every_rank(tmp_path):
    """Two ranks over gloo (the CPU rehearsal of the RCCL launch): the script sees an initialised process group, its own argv,
    __name__ == '__main__' and a per-rank seed."""
    script = tmp_path / 'run_stub.py'
    script.write_text(textwrap.dedent('''
        import argparse, json, os, sys
        import torch
        import torch.distributed as dist
        if __name__ == '__main__':
            p = argparse.ArgumentParser(); p.add_argument('--config'); a = p.parse_args()
            t = torch.tensor([float(dist.get_rank() + 1)])
            dist.all_reduce(t)
            out = {'rank': dist.get_rank(), 'world': dist.get_world_size(), 'config': a.config, 'sum': t.item(),
                   'draw': torch.rand(1).item(), 'argv0': sys.argv[0]}
            json.dump(out, open(os.path.join(os.path.dirname(a.config), f"out{dist.get_rank()}.json"), 'w'))
    '''))
    cfg = tmp_path / 'cfg.yaml'
    cfg.write_text('x: 1\n')
    env = dict(os.environ, SUNERF_DIST_BACKEND='gloo', PYTHONPATH=PKG + os.pathsep + os.environ.get('PYTHONPATH', ''))
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                    '--master-port', '29571', '-m', 'sunerf.run_mi355x', str(script), '--config', str(cfg)],
                   check=True, env=env, timeout=300, cwd=str(tmp_path))
    import json
    outs = [json.load(open(tmp_path / f'out{r}.json')) for r in (0, 1)]
    assert [o['rank'] for o in outs] == [0, 1] and all(o['world'] == 2 and o['sum'] == 3.0 for o in outs)
    assert all(o['config'] == str(cfg) and o['argv0'] == str(script) for o in outs)
    assert outs[0]['draw'] != outs[1]['draw']            # ranks draw different batches


def test_wrapper_refuses_a_missing_script():
    from sunerf import run_mi355x
    with pytest.raises(SystemExit, match='is not a file'):
        run_mi355x.main(['/nonexistent/run_emission.py', '--config', 'x'])
