"""ctypes binding of libsunerf_hip.so (C ABI: include/sunerf_hip.h).

The library is built in-tree by ``csrc/build.sh`` (``__graft_entry__.build()``).  There is no CPU fallback: if
the library is missing or a tensor is not on a ROCm device the call raises.
"""
import ctypes
import os

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be mapped
# BEFORE libsunerf_hip.so so that our library binds to the same runtime instance (streams and device pointers are
# shared with torch); loading ours first would pull in /opt/rocm's copy and every launch fails with
# hipErrorNoDevice (100).
import torch  # noqa: F401  (side effect: loads torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SUNERF_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'libsunerf_hip.so')

_lib = None

c_f32p = ctypes.c_void_p   # device pointers travel as raw addresses
c_void = ctypes.c_void_p

_SIGNATURES = {
    'sunerf_abi_version': (ctypes.c_int, []),
    'sunerf_packed_mlp_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'sunerf_pack_mlp': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, c_void]),
    'sunerf_sample_z': (ctypes.c_int, [ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_float, ctypes.c_float, c_f32p, c_void]),
    'sunerf_act_stash_bytes': (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'sunerf_render_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int]),
    'sunerf_emission_render_fwd': (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                                                   ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                                                   c_f32p, c_f32p, c_f32p, ctypes.c_float, c_void, ctypes.c_int, c_void, ctypes.c_size_t,
                                                   c_void]),
    'sunerf_packed_mlp_t_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'sunerf_pack_mlp_t': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          c_void, c_void]),
    'sunerf_dz_stash_bytes': (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'sunerf_wgrad_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'sunerf_emission_integral_fwd': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, c_f32p,
                                                     c_void]),
    'sunerf_emission_integral_bwd': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_float,
                                                     ctypes.c_float, ctypes.c_int64, ctypes.c_int, c_f32p, c_void, c_void]),
    'sunerf_mlp_dgrad': (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, c_f32p, c_void, c_void, c_void,
                                         ctypes.c_int64, ctypes.c_int, c_void]),
    'sunerf_mlp_wgrad': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_f32p, c_void,
                                         ctypes.c_int64, ctypes.c_int, c_void, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                         c_void]),
    'sunerf_bwd_pipe_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'sunerf_mlp_backward_pipe': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void, c_void, c_f32p, c_void,
                                                 ctypes.c_int64, ctypes.c_int, c_void, ctypes.c_size_t,
                                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                                 ctypes.c_int, c_void]),
    'sunerf_bwd_pipe_kernel_time': (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    'sunerf_mlp_backward_exact_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    'sunerf_mlp_backward_exact': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                  ctypes.c_int64, ctypes.c_int, c_f32p, c_void, ctypes.c_size_t,
                                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                                  c_void]),
    'sunerf_dt_integral_fwd': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p, c_f32p,
                                               c_f32p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                               c_void]),
    'sunerf_dt_integral_bwd': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p, c_f32p,
                                               c_f32p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_void,
                                               c_void]),
    'sunerf_hier_resample': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_int, c_f32p, c_f32p, c_void]),
    'sunerf_mlp_points_fwd': (ctypes.c_int, [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_int64, c_f32p,
                                              c_void, ctypes.c_int, c_void, ctypes.c_size_t, c_void]),
    'sunerf_sample_pdf': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          c_f32p, c_void]),
    'sunerf_observer_rays': (ctypes.c_int, [c_void, c_void, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.POINTER(ctypes.c_float), ctypes.c_float, c_f32p, c_f32p, c_f32p, c_void]),
    'sunerf_simple_star_field': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f32p,
                                                 c_void]),
    'sunerf_train_workspace_bytes': (ctypes.c_size_t, []),
    'sunerf_training_loss': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int64, c_f32p, ctypes.c_int64,
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.c_int,
                                             ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                             c_f32p, c_f32p, c_f32p, c_void, ctypes.c_size_t, c_void]),
    'sunerf_clip_adam_step': (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_int64, c_f32p, c_f32p, c_void, ctypes.c_size_t,
                                              c_void, c_void]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class SunerfHipError(RuntimeError):
    pass


def load():
    """Loads the library once; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SunerfHipError(
                f'{LIB_PATH} not found: build it with 2024-hl-spi3s-sunerf_amd/csrc/build.sh '
                '(or __graft_entry__.build()); there is no CPU fallback for the render path')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.sunerf_abi_version() != 9:
            raise SunerfHipError('libsunerf_hip.so ABI version mismatch')
        _lib = lib
    return _lib


_ERRORS = {-1: 'bad argument (null pointer or non-positive size)',
           -2: 'unsupported configuration (d_filter / n_layers / sample count outside the compiled set)',
           -3: 'workspace too small'}


def check(status, what):
    if status == 0:
        return
    if status in (-1, -2, -3):
        raise ValueError(f'{what}: {_ERRORS[status]}')
    raise SunerfHipError(f'{what}: HIP error {status}')


def call(device, name, *args):
    """Runs C-ABI entry point ``name`` with ``device`` current (the library sizes its grids from hipGetDevice() and a
    kernel can only be launched into a stream of the current device: a module on cuda:1 while cuda:0 is current would
    otherwise fail or use the wrong CU count) and raises on a non-zero status."""
    fn = getattr(load(), name)
    with torch.cuda.device(device):
        status = fn(*args)
    check(status, name)
