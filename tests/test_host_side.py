"""CPU-only checks (no GPU): the C-ABI library builds, loads and exports every symbol the header declares; host-side
size / layout helpers; the module mirror's construction semantics; the product path refuses to run without a GPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='session')
def lib():
    import sunerf_hip
    if not os.path.exists(sunerf_hip.LIB_PATH):
        import subprocess
        subprocess.check_call(['bash', os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd', 'csrc', 'build.sh')])
    return sunerf_hip.load()


def test_library_exports_every_declared_symbol(lib):
    import sunerf_hip
    header = open(os.path.join(ROOT, 'include', 'sunerf_hip.h')).read()
    declared = set(re.findall(r'\b(sunerf_\w+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(sunerf_hip.EXPORTED_SYMBOLS), declared ^ set(sunerf_hip.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.sunerf_abi_version() == 9


def test_size_helpers(lib):
    # packed image: in layer 8 tiles x 6 k-steps, 7 hidden layers x 8 tiles x 16 k-steps, out 16 k-steps, 2 KiB each
    steps = 8 * 6 + 7 * 8 * 16 + 16
    # + biases + 16 per-layer scale exponents + 16 max|w| scratch words (fp8c stream format)
    assert lib.sunerf_packed_mlp_bytes(256, 9) == steps * 2048 + (8 * 256 + 32) * 4 + 128
    assert lib.sunerf_packed_mlp_bytes(250, 9) == 0 and lib.sunerf_packed_mlp_bytes(256, 1) == 0
    # out^T hi only; hidden hi + lo; + the 16 per-layer sums of squares the backward boosts are chosen from
    assert lib.sunerf_packed_mlp_t_bytes(256, 9) == 8 * 1024 + 7 * 8 * 16 * 2048 + 16 * 4
    # stash: (6 enc + 8 layers x 2 x 16) fragments of 1 KiB per 32-sample chunk, + 1 spare chunk
    assert lib.sunerf_act_stash_bytes(10, 128, 256, 9, 0) == (10 * 4 + 1) * (6 + 8 * 32) * 1024
    assert lib.sunerf_act_stash_bytes(10, 130, 256, 9, 0) == (10 * 5 + 1) * (6 + 8 * 32) * 1024      # ragged last chunk
    # 16-bit phase format (what the pipelined backward reads): one fragment set per layer instead of sin + cos; d_filter 256 only
    assert lib.sunerf_act_stash_bytes(10, 128, 256, 9, 1) == (10 * 4 + 1) * (6 + 8 * 16) * 1024
    assert lib.sunerf_act_stash_bytes(10, 128, 128, 9, 1) == 0 and lib.sunerf_act_stash_bytes(10, 128, 256, 9, 2) == 0
    assert lib.sunerf_dz_stash_bytes(10, 128, 256, 9) == (10 * 4 + 1) * 8 * 16 * 1024
    assert lib.sunerf_wgrad_workspace_bytes(256, 9, 32) == 9 * 32 * 72 * 1024 * 4   # 8 x (8 + bias column) tiles
    assert lib.sunerf_wgrad_workspace_bytes(512, 9, 7) == 9 * 7 * 16 * 17 * 1024 * 4


def test_argument_errors_without_gpu(lib):
    # null pointers / bad sizes are rejected before anything touches a device
    assert lib.sunerf_sample_z(0, None, None, None, None, 4, 8, 1.3, 1.0, None, None) == -1
    assert lib.sunerf_sample_z(0, None, None, None, None, 0, 8, 1.3, 1.0, None, None) == 0       # empty batch: nothing to do
    assert lib.sunerf_sample_z(7, None, None, None, None, 0, 8, 1.3, 1.0, None, None) == -2      # unknown sampler kind
    assert lib.sunerf_hier_resample(None, None, None, 0, 4, 8, 8, None, None, None) == -1
    assert lib.sunerf_emission_render_fwd(None, 256, 9, 0, None, None, None, None, 4, 8, None, None, None, None, None, None,
                                          None, 1.2, None, 0, None, 0, None) == -1
    # precision: an unknown mode is a bad argument
    assert lib.sunerf_emission_render_fwd(None, 256, 9, 5, None, None, None, None, 4, 8, None, None, None, None, None, None,
                                          None, 1.2, None, 0, None, 0, None) == -1
    assert lib.sunerf_emission_render_fwd(None, 384, 9, 0, None, None, None, None, 4, 8, None, None, None, None, None, None,
                                          None, 1.2, None, 0, None, 0, None) == -1        # null pointers are checked first
    assert lib.sunerf_render_workspace_bytes(256) == 0 and lib.sunerf_render_workspace_bytes(512) == 1024 * 4 * 32 * 2048


def test_pipelined_backward_entry_points_without_gpu(lib):
    """include/sunerf_hip.h: the pipelined backward is for a 256-CU device -- without one the workspace query says 0 (= use
    sunerf_mlp_dgrad + sunerf_mlp_wgrad) and bad arguments are refused before anything touches a device."""
    import ctypes
    assert lib.sunerf_bwd_pipe_workspace_bytes(100, 128, 256, 9) == 0
    assert lib.sunerf_bwd_pipe_workspace_bytes(100, 128, 512, 9) == 0
    GW, GB = (ctypes.c_void_p * 9)(), (ctypes.c_void_p * 9)()
    assert lib.sunerf_mlp_backward_pipe(256, 9, 2, None, None, None, None, 4, 128, None, 0, None, None, 0, 0, None) == -1
    assert lib.sunerf_mlp_backward_pipe(256, 9, 2, None, None, None, None, 4, 128, None, 0, GW, GB, 0, 0, None) == -1
    assert lib.sunerf_mlp_backward_pipe(256, 9, 2, None, None, None, None, 0, 128, None, 0, GW, GB, 0, 0, None) == -1


def test_environment_switches(monkeypatch):
    """One parser for every on / off switch ('0', 'false', 'no', 'off', '' are OFF: VERDICT r2 found bool('0') in two places);
    SUNERF_BACKWARD accepts exactly its two values."""
    from sunerf_hip import ops, train
    for v, want in (('', False), ('0', False), ('false', False), ('No', False), (' off ', False), ('1', True), ('yes', True), ('on', True)):
        monkeypatch.setenv('SUNERF_OVERLAP', v)
        assert train.env_flag('SUNERF_OVERLAP') is want, v
    monkeypatch.delenv('SUNERF_OVERLAP')
    assert train.env_flag('SUNERF_OVERLAP') is False
    monkeypatch.setattr(ops, '_backward_forced', None)
    monkeypatch.delenv('SUNERF_BACKWARD', raising=False)
    assert ops.backward_mode() == 'pipe'
    monkeypatch.setenv('SUNERF_BACKWARD', 'Classic')
    assert ops.backward_mode() == 'classic'
    monkeypatch.setenv('SUNERF_BACKWARD', 'fused')
    with pytest.raises(ValueError):
        ops.backward_mode()
    monkeypatch.setenv('SUNERF_PIPE_DEBUG', '0')
    monkeypatch.setenv('SUNERF_PIPE_HI_ONLY', 'off')
    assert ops._pipe_flags() == 0
    monkeypatch.setenv('SUNERF_PIPE_DEBUG', '1')
    assert ops._pipe_flags() == 2
    # W^T precision of the pipelined backward: unset = measured policy (no flag bit: the policy sets it per launch), 1 / 0 force
    monkeypatch.delenv('SUNERF_PIPE_HI_ONLY')
    assert ops.pipe_w_mode() == 'auto' and ops._pipe_flags() == 2
    monkeypatch.setenv('SUNERF_PIPE_HI_ONLY', '1')
    assert ops.pipe_w_mode() == 'hi' and ops._pipe_flags() == 3
    monkeypatch.setenv('SUNERF_PIPE_HI_ONLY', '0')
    assert ops.pipe_w_mode() == 'hilo' and ops._pipe_flags() == 2


def test_bucket_registry_is_keyed_by_identity_and_holds_no_strong_references():
    """The optimiser finds its flat bucket through an identity-keyed weak registry, not through an attribute on the Parameter
    (the attribute pickled the optimiser into every checkpoint)."""
    import copy
    import gc
    import pickle
    import weakref
    from sunerf_hip import train

    class Owner:
        pass

    owner = Owner()
    p = torch.nn.Parameter(torch.zeros(3))
    q = torch.nn.Parameter(torch.zeros(3))          # equal VALUES must not collide (identity, not __eq__)
    assert train.bucket_of(p) is None
    train._BUCKETS[p] = (weakref.ref(owner), 5, 3)
    got = train.bucket_of(p)
    assert got is not None and got[0] is owner and tuple(got[1:]) == (5, 3)
    assert train.bucket_of(q) is None and train.bucket_of(copy.deepcopy(p)) is None
    assert b'Owner' not in pickle.dumps(p)           # nothing rides on the Parameter
    del owner, got
    gc.collect()
    assert train.bucket_of(p) is None                # a dead optimiser's tag is no tag
    n = len(train._BUCKETS)
    del p
    gc.collect()
    assert len(train._BUCKETS) == n - 1              # and a dead Parameter leaves no entry behind


def test_cpu_tensors_are_refused():
    from sunerf_hip import ops, SunerfHipError
    with pytest.raises(SunerfHipError):
        ops.sample_z(ops.SAMPLER_STRATIFIED, torch.zeros(4, 3), torch.ones(4, 3), torch.linspace(0, 1, 8), 1.3, 1.0)
    with pytest.raises(SunerfHipError):
        ops.hier_resample(torch.zeros(4, 8), torch.zeros(4, 8), torch.linspace(0, 1, 8))


def test_module_mirror_construction_semantics():
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    from sunerf.rendering.base_tracing import SuNeRFRendering
    from sunerf.train.sampling import SphericalSampler, StratifiedSampler
    sc = {'type': 'stratified', 'n_samples': 16, 'perturb': False}
    mc = {'d_filter': 64, 'n_layers': 3}
    mod = EmissionRadiativeTransfer(Rs_per_ds=2.0, sampling_config=sc, model_config=mc)
    assert 'type' not in sc                                  # popped from the caller's dict like the reference
    assert mc['d_input'] == 4 and mc['d_output'] == 2        # emission.py:11 updates the caller's dict
    assert isinstance(mod.sampler, StratifiedSampler) and mod.sampler_hierarchical.n_samples == 128
    assert abs(float(mod.sampler.distance) - 1.3 / 2.0) < 1e-7 and float(mod.sampler.solar_R) == 0.5
    assert torch.equal(mod.sampler.t_vals, torch.linspace(0., 1., 16)[None])
    keys = set(mod.state_dict().keys())
    assert {'sampler.distance', 'sampler.solar_R', 'sampler.t_vals', 'coarse_model.in_layer.0.freq_bands',
            'coarse_model.in_layer.1.weight', 'coarse_model.layers.1.bias', 'fine_model.out_layer.weight'} <= keys
    assert len(mod.coarse_model.layers) == 2
    assert isinstance(EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'spherical'},
                                                model_config={'d_filter': 64}).sampler, SphericalSampler)
    with pytest.raises(ValueError):
        EmissionRadiativeTransfer(Rs_per_ds=1.0, sampling_config={'type': 'nope'}, model_config={'d_filter': 64})
    with pytest.raises(NotImplementedError):
        SuNeRFRendering(Rs_per_ds=1.0, model_config={'d_filter': 64}).raw2outputs()
    # default nn.Linear initialisation in the reference's creation order => same weights as the oracle's generator
    torch.manual_seed(7)
    a = EmissionRadiativeTransfer(Rs_per_ds=1.0, model_config={'d_filter': 64})
    torch.manual_seed(7)
    b = EmissionRadiativeTransfer(Rs_per_ds=1.0, model_config={'d_filter': 64})
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))


def test_lightning_module_surface():
    from sunerf.model.sunerf import EmissionSuNeRFModule, BaseSuNeRFModule, save_state
    m = EmissionSuNeRFModule(Rs_per_ds=1.0, seconds_per_dt=1.0, image_scaling_config={'vmax': 1, 'a': 0.005},
                             model_config={'d_filter': 64})
    # the optimiser is the fused clip + Adam kernel on flat device buffers: a CPU module must fail loudly, not fall back
    from sunerf_hip.lib import SunerfHipError
    with pytest.raises(SunerfHipError):
        m.configure_optimizers()
    for hook in ('training_step', 'validation_step', 'validation_epoch_end', 'on_train_batch_end', 'on_load_checkpoint',
                 'configure_optimizers'):
        assert callable(getattr(m, hook))
    assert issubclass(EmissionSuNeRFModule, BaseSuNeRFModule) and callable(save_state)


def test_rendering_module_pickles_without_device_buffers(tmp_path):
    """save_state (sunerf.py:62-74) pickles the rendering module: the packed-weight cache must not travel."""
    from sunerf.rendering.emission import EmissionRadiativeTransfer
    mod = EmissionRadiativeTransfer(Rs_per_ds=1.0, model_config={'d_filter': 64})
    mod.coarse_model._packed = object()          # stand-in for a device buffer
    path = tmp_path / 'state.snf'
    torch.save({'rendering': mod}, path)
    back = torch.load(path, weights_only=False)['rendering']
    assert back.coarse_model._packed is None
    assert set(back.state_dict()) == set(mod.state_dict())


def test_shard_range_partitions():
    from sunerf_hip.dist import shard_range
    for n, w in ((1048576, 8), (10, 3), (5, 8)):
        blocks = [shard_range(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert max(e - b for b, e in blocks) - min(e - b for b, e in blocks) <= 1


def test_package_merges_with_a_reference_checkout(tmp_path):
    """with our package root first and a reference checkout later on PYTHONPATH, mirrored modules
    come from here and everything else (data loaders, run scripts) from the checkout."""
    import subprocess
    import sys
    ref = tmp_path / 'checkout'
    (ref / 'sunerf' / 'train').mkdir(parents=True)
    (ref / 'sunerf' / 'data').mkdir(parents=True)
    for d in ('sunerf', 'sunerf/train', 'sunerf/data'):
        (ref / d / '__init__.py').write_text('')
    (ref / 'sunerf' / 'train' / 'sampling.py').write_text('ORIGIN = "checkout"\n')
    (ref / 'sunerf' / 'train' / 'callback.py').write_text('ORIGIN = "checkout"\n')
    (ref / 'sunerf' / 'data' / 'dataset.py').write_text('ORIGIN = "checkout"\n')
    (ref / 'sunerf' / 'run_emission.py').write_text('ORIGIN = "checkout"\n')
    code = ('import sunerf.train.sampling as s, sunerf.train.callback as c, sunerf.data.dataset as d, sunerf.run_emission as r;'
            'assert hasattr(s, "StratifiedSampler") and not hasattr(s, "ORIGIN");'
            'assert c.ORIGIN == d.ORIGIN == r.ORIGIN == "checkout"; print("ok")')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, '2024-hl-spi3s-sunerf_amd'), str(ref)]))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr
