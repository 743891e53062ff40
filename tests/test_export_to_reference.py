"""SURVEY 8f-4 "export back": a ``.snf`` state written by THIS package's ``save_state`` from the mirrored classes unpickles
into the REFERENCE's own classes and the reference's own code renders from it -- what a user does who trains here and then
runs the reference's evaluation scripts (``evaluation/image_render.py`` / ``video.py`` are not rebuilt).

Container only (needs ``/root/reference``; skipped on the GPU box).  The reference side runs in a child interpreter whose
``sunerf`` package is the reference's: the pickle resolves every class by module path, so the child proves that the mirrored
classes carry exactly the attributes, sub-module names, parameters and buffers the reference's forward reads.  The child
runs the reference's forward (with the two one-line shims of ``oracle/ref_import.py`` for the HEAD defects D1 / D2) and its
``NeRF.forward`` on the unpickled object; the parent compares with the oracle evaluated on the same weights.
"""
import datetime
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import sunerf_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='needs the reference checkout (container only)')

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])                      # oracle/ (ref_import)
import ref_import
ref = ref_import.import_reference()
import sunerf.rendering.emission as em
assert em.__file__.startswith('/root/reference/'), em.__file__
state = torch.load(sys.argv[2], weights_only=False)
r = state['rendering']
assert type(r) is em.EmissionRadiativeTransfer and type(r).__module__ == 'sunerf.rendering.emission'
assert type(r.fine_model).__module__ == 'sunerf.model.model' and sys.modules['sunerf.model.model'].__file__.startswith('/root/reference/')
inp = np.load(sys.argv[3])
o, d, t, pts = (torch.from_numpy(inp[k]) for k in ('rays_o', 'rays_d', 'times', 'points'))
out = {}
with torch.no_grad():
    out['points_inferences'] = r.fine_model(pts)['inferences']
    r.__class__ = ref_import.shimmed_emission_class()          # the reference's forward + the D1 / D2 one-liners
    for k, v in r(o, d, t).items():
        out['out__' + k] = v
out['keys'] = np.array(sorted(r.state_dict().keys()))
np.savez(sys.argv[4], **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
print('data_config', sorted(state['data_config'].keys()), state['Rs_per_ds'], state['seconds_per_dt'], state['ref_time'])
'''


def test_state_file_written_here_renders_in_the_reference(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'))
    from sunerf.model.sunerf import save_state
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    torch.manual_seed(5)
    rendering = EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 16, 'perturb': False},
                                          hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 16},
                                          model_config={'d_filter': 64})

    class _Module:
        pass

    class _Data:
        config = {'type': 'emission', 'wavelength': 193, 'times': [datetime.datetime(2022, 1, 1), datetime.datetime(2022, 1, 3)],
                  'resolution': (16, 16), 'wcs': {'shape': (16, 16), 'cdelt': (150., 150.)}}
        Rs_per_ds, seconds_per_dt, ref_time = 1.0, 86400., datetime.datetime(2022, 1, 1)
    mod = _Module()
    mod.rendering = rendering
    path = str(tmp_path / 'run' / 'save_state.snf')
    save_state(mod, _Data(), path)

    o, d = orc.synthetic_rays(5)
    t = torch.rand(o.shape[0], 1, generator=torch.Generator().manual_seed(3)) * 2.
    pts = torch.rand(40, 4, generator=torch.Generator().manual_seed(1)) * 2 - 1
    inputs, outputs = str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')
    np.savez(inputs, rays_o=o.numpy(), rays_d=d.numpy(), times=t.numpy(), points=pts.numpy())
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    res = subprocess.run([sys.executable, '-c', CHILD, os.path.join(ROOT, 'oracle'), path, inputs, outputs], env=env,
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-3000:]
    assert "data_config ['resolution', 'times', 'type', 'wavelength', 'wcs'] 1.0 86400.0 2022-01-01 00:00:00" in res.stdout

    got = np.load(outputs)
    sd = {k: v.detach() for k, v in rendering.state_dict().items()}
    assert list(got['keys']) == sorted(sd.keys())                       # same parameter / buffer names on both sides
    coarse, fine = orc.params_from_state_dict(sd, 'coarse_model.'), orc.params_from_state_dict(sd, 'fine_model.')
    want_pts = orc.mlp_forward(fine, pts)
    assert np.abs(got['points_inferences'] - want_pts.numpy()).max() <= 1e-6
    want = orc.render_emission(coarse, fine, o, d, t, n_coarse=16, n_fine=16)
    for k in ('coarse_image', 'fine_image', 'image', 'height_map', 'absorption_map', 'z_vals_stratified'):
        a, b = got['out__' + k].reshape(-1), want[k].numpy().reshape(-1)
        assert np.abs(a - b).max() <= 2e-6 * max(1e-30, np.abs(b).max()), k


DT_CFG = dict(Rs_per_ds=1.0, sampling_config={'type': 'stratified', 'n_samples': 16, 'perturb': False},
              hierarchical_sampling_config={'type': 'hierarchical', 'n_samples': 16}, model_config={'d_filter': 64})

DT_CHILD = r'''
import sys, datetime, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import ref_import
ref = ref_import.import_reference()
import sunerf.rendering.density_temperature as dtm
import sunerf.model.model as M
assert dtm.__file__.startswith('/root/reference/')
g = np.load(sys.argv[3])
o, d, t, wl = (torch.from_numpy(g[k]) for k in ('rays_o', 'rays_d', 'times', 'wavelengths'))
# (1) the state written by the drop-in package, in the reference's own class, rendered by the reference's own code
state = torch.load(sys.argv[2], weights_only=False)
r = state['rendering']
assert type(r) is dtm.DensityTemperatureRadiativeTransfer and type(r.fine_model) is M.NeRF_DT
assert sorted(r.response.keys()) == [94, 131, 171, 193, 211, 304, 335]
with torch.no_grad():
    out = r(o, d, t, wl)
np.savez(sys.argv[4], **{k: v.numpy() for k, v in out.items()})
# (2) a state written from the reference's own class, for the other direction
cfg = eval(sys.argv[6])
own = dtm.DensityTemperatureRadiativeTransfer(model=M.NeRF_DT, device=torch.device('cpu'), pixel_intensity_factor=1e17,
                                              **{k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})
own.load_state_dict({k[4:].replace('__', '.'): torch.from_numpy(g[k]) for k in g.files if k.startswith('sd__')})
torch.save({'rendering': own, 'data_config': {'type': 'dt'}, 'Rs_per_ds': 1.0, 'seconds_per_dt': 86400.,
            'ref_time': datetime.datetime(2022, 3, 1)}, sys.argv[5])
'''


def test_density_temperature_state_files_travel_in_both_directions(tmp_path):
    """The DT rendering module carries one xitorch interpolator per channel in the reference (density_temperature.py:132-146)
    and a table buffer here.  (1) written here (xitorch importable -> interpolators built through its constructor), the file
    renders in the reference and gives fixture g6's outputs bit for bit; (2) written by the reference, it loads here without
    xitorch and the table comes out of the pickled interpolators."""
    sys.path.insert(0, os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'))
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ref_import
    from conftest import load_golden
    from sunerf.evaluation.loader import load_state_file
    from sunerf.model.model import NeRF_DT
    from sunerf.rendering.density_temperature import DensityTemperatureRadiativeTransfer
    g = load_golden('g6_dt_e2e')
    mod = DensityTemperatureRadiativeTransfer(model=NeRF_DT, device=torch.device('cpu'), pixel_intensity_factor=1e17,
                                              response_table=(g['aia_logte'].numpy(), g['aia_tresp'].numpy()),
                                              **{k: (dict(v) if isinstance(v, dict) else v) for k, v in DT_CFG.items()})
    sd = {k[4:].replace('__', '.'): v for k, v in g.items() if k.startswith('sd__')}
    mod.load_state_dict(sd, strict=True)
    ours, theirs = str(tmp_path / 'ours.snf'), str(tmp_path / 'theirs.snf')
    had = {k: sys.modules.get(k) for k in ('xitorch', 'xitorch.interpolate')}
    ref_import.install_xitorch_stub()                 # stands for an environment with xitorch (any that runs the reference)
    try:
        torch.save({'rendering': mod, 'data_config': {'type': 'dt'}, 'Rs_per_ds': 1.0, 'seconds_per_dt': 86400.,
                    'ref_time': datetime.datetime(2022, 3, 1)}, ours)
    finally:
        for k, v in had.items():
            if v is None:
                sys.modules.pop(k, None)
    inputs, outputs = os.path.join(ROOT, 'tests', 'golden', 'g6_dt_e2e.npz'), str(tmp_path / 'out.npz')
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    res = subprocess.run([sys.executable, '-c', DT_CHILD, os.path.join(ROOT, 'oracle'), ours, inputs, outputs, theirs, repr(DT_CFG)],
                         env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-3000:]
    got = np.load(outputs)
    for k in ('z_vals_stratified', 'coarse_image', 'z_vals_hierarchical', 'fine_image', 'image', 'height_map',
              'absorption_map', 'regularization'):
        assert np.array_equal(got[k], g['out__' + k].numpy()), k          # the reference's code on our file == fixture g6

    assert 'xitorch' not in sys.modules                                    # direction (2) runs without it
    state = load_state_file(theirs)
    assert 'xitorch' not in sys.modules
    back = state['rendering']
    assert type(back) is DensityTemperatureRadiativeTransfer and type(back.fine_model) is NeRF_DT
    assert torch.equal(back.response_logte, mod.response_logte) and torch.equal(back.response_table, mod.response_table)
    assert back.pixel_intensity_factor == 1e17
    for k, v in back.state_dict().items():
        assert torch.equal(v, sd[k]), k


HELPERS_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import ref_import
ref_import.import_reference()
import sunerf.model.model as M
x = torch.from_numpy(np.load(sys.argv[2])['x'])
torch.manual_seed(0)
tpe = M.TrainablePositionalEncoding(4, n_freqs=6)
em = M.EmissionModel(d_filter=32, n_layers=3)
np.savez(sys.argv[3], tpe=tpe(x).detach().numpy(), tpe_freq=tpe.frequencies.detach().numpy(), sine=M.Sine(1.7)(x).numpy(),
         pe=M.PositionalEncoding(4, 10)(x).numpy(), em_keys=np.array(sorted(em.state_dict().keys())),
         em_shapes=np.array([tuple(v.shape) + (0,) * (2 - v.ndim) for _, v in sorted(em.state_dict().items())]))
'''


def test_model_module_helpers_equal_the_reference(tmp_path):
    """The small public classes of model.py beside ``NeRF`` -- ``Sine``, ``PositionalEncoding`` applied by itself,
    ``TrainablePositionalEncoding``, ``EmissionModel`` -- against the reference's own, on the same input."""
    sys.path.insert(0, os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'))
    from sunerf.model import model as M
    x = torch.rand(50, 4, generator=torch.Generator().manual_seed(9)) * 4 - 2
    inp, out = str(tmp_path / 'x.npz'), str(tmp_path / 'o.npz')
    np.savez(inp, x=x.numpy())
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    res = subprocess.run([sys.executable, '-c', HELPERS_CHILD, os.path.join(ROOT, 'oracle'), inp, out], env=env,
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-3000:]
    want = np.load(out)
    tpe = M.TrainablePositionalEncoding(4, n_freqs=6)
    assert np.array_equal(tpe.frequencies.detach().numpy(), want['tpe_freq']) and tpe.d_output == 48
    assert np.abs(tpe(x).detach().numpy() - want['tpe']).max() <= 1e-6 * np.abs(want['tpe']).max()
    assert np.array_equal(M.Sine(1.7)(x).numpy(), want['sine'])
    assert np.array_equal(M.PositionalEncoding(4, 10)(x).numpy(), want['pe'])
    em = M.EmissionModel(d_filter=32, n_layers=3)
    assert sorted(em.state_dict().keys()) == list(want['em_keys'])
    assert [tuple(v.shape) + (0,) * (2 - v.ndim) for _, v in sorted(em.state_dict().items())] == [tuple(r) for r in want['em_shapes']]
