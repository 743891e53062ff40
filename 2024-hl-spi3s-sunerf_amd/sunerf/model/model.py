"""Mirror of the reference's ``sunerf/model/model.py``: parameter containers with the reference's module tree
(state-dict keys ``in_layer.0.freq_bands``, ``in_layer.1.{weight,bias}``, ``layers.{i}.{weight,bias}``,
``out_layer.{weight,bias}``) and default ``nn.Linear`` initialisation.  The arithmetic is done by the fused HIP
kernels on a packed fp16 hi/lo image of these parameters (``sunerf_hip.ops.PackedMLP``)."""
import threading
from typing import Tuple

import torch
from torch import nn

from sunerf_hip import ops

_PACK_LOCK = threading.RLock()      # serialises (re)building the packed-weights caches (NeRF.packed)


class Sine(nn.Module):
    """model.py:66-72.  The fused kernels implement w0 = 1 (the only value the reference ever constructs)."""

    def __init__(self, w0: float = 1.):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        """Stand-alone use of the activation (model.py:71-72); inside ``NeRF`` it is part of the fused kernels."""
        return torch.sin(x * self.w0)


class PositionalEncoding(nn.Module):
    """model.py:92-132: [x, sin(x f_k / 2), cos(x f_k / 2)], f_k = 2^k, k = 0..n_freqs-1 (fused into the kernel)."""

    def __init__(self, d_input: int, n_freqs: int, scale_factor: float = 2., log_space: bool = True):
        super().__init__()
        if d_input != 4 or n_freqs != 10 or scale_factor != 2. or not log_space:
            raise ValueError('the fused kernel implements PositionalEncoding(d_input=4, n_freqs=10, scale_factor=2, '
                             'log_space=True), the only configuration NeRF constructs (model.py:29)')
        self.d_input = d_input
        self.n_freqs = n_freqs
        self.log_space = log_space
        self.d_output = d_input * (1 + 2 * n_freqs)
        self.register_buffer('freq_bands', 2. ** torch.linspace(0., n_freqs - 1, n_freqs))
        self.scale_factor = scale_factor

    def forward(self, x) -> torch.Tensor:
        """Stand-alone use of the encoder (model.py:123-132): (M, d) -> (M, d (1 + 2 n_freqs)) = [x, sin block, cos block],
        each block frequency-major.  Inside ``NeRF`` the encoding is computed in the fused kernels; this is for callers that
        apply the module by itself (element-wise torch on the tensor's own device)."""
        phase = (x.unsqueeze(1) * self.freq_bands.view(1, -1, 1) / self.scale_factor).flatten(1)
        return torch.cat((x, phase.sin(), phase.cos()), dim=-1)


class TrainablePositionalEncoding(nn.Module):
    """model.py:75-89 (constructed nowhere in the reference; kept so that ``from sunerf.model.model import *`` code finds it):
    ``d_input`` x ``n_freqs`` learnable exponents f, output [sin(pi 2^f x), cos(pi 2^f x)] / (pi 2^f), (M, 2 n_freqs d_input).
    Not connected to the fused kernels."""

    def __init__(self, d_input, n_freqs=20):
        super().__init__()
        exponents = torch.linspace(-3, 9, n_freqs, dtype=torch.float32).view(1, n_freqs, 1).repeat(1, 1, d_input)
        self.frequencies = nn.Parameter(exponents, requires_grad=True)
        self.d_output = n_freqs * 2 * d_input

    def forward(self, x):
        scale = torch.pi * torch.pow(2., self.frequencies)          # (1, n_freqs, d)
        phase = x.unsqueeze(1) * scale
        return torch.cat((phase.sin() / scale, phase.cos() / scale), dim=-1).flatten(1)


class NeRF(nn.Module):
    """model.py:7-57."""
    # class-level defaults: a NeRF unpickled from a reference-written .snf (sunerf.py:62-74) carries neither attribute
    _packed = None
    _packed_key = None

    def _replicate_for_data_parallel(self):
        # a model wrapped by itself (nn.DataParallel(rendering.fine_model), evaluation/loader.py:33-36 of the reference): see
        # sunerf.rendering.base_tracing.refuse_data_parallel
        from sunerf.rendering.base_tracing import refuse_data_parallel
        refuse_data_parallel(type(self).__name__)

    def __init__(self, d_input: int = 4, d_output: int = 2, n_layers: int = 8, d_filter: int = 512,
                 skip: Tuple[int] = (), encoding='positional'):
        super().__init__()
        if not 1 <= d_filter <= ops.SUPPORTED_D_FILTER[-1]:
            raise ValueError(f'd_filter={d_filter}: the fused kernels cover widths 1..{ops.SUPPORTED_D_FILTER[-1]} (compiled for '
                             f'{ops.SUPPORTED_D_FILTER}; other widths run zero-padded to the next of these, which is exact)')
        if d_input != 4:
            raise ValueError('the fused kernels take (x, y, z, t) query points: d_input = 4 (emission.py:11, the only value used)')
        self.d_input = d_input
        self.skip = skip
        self.act = Sine()
        if encoding == 'positional':      # model.py:28-33
            enc = PositionalEncoding(d_input=d_input, n_freqs=10)
            self.in_layer = nn.Sequential(enc, nn.Linear(enc.d_output, d_filter))
        else:                             # anything else: no encoding, the first layer reads the raw coordinates
            self.in_layer = nn.Linear(d_input, d_filter)
        self.layers = nn.ModuleList([nn.Linear(d_filter, d_filter) for _ in range(n_layers - 1)])
        self.out_layer = nn.Linear(d_filter, d_output)
        self._packed = None
        self._packed_key = None

    # -- parameter views in evaluation order -----------------------------------------------------------------
    def linears(self):
        first = self.in_layer[1] if isinstance(self.in_layer, nn.Sequential) else self.in_layer
        return [first] + list(self.layers) + [self.out_layer]

    def packed(self) -> 'ops.PackedMLP':
        """Packed image of the current parameters; re-packed when any parameter was modified in place
        (optimizer step, load_state_dict) or moved."""
        lin = self.linears()
        key = tuple((p.data_ptr(), p._version) for l in lin for p in (l.weight, l.bias))
        # evaluation/loader.py:226-229 submits ray batches of one frame to a ThreadPoolExecutor: the cache is (re)built by one
        # thread at a time
        with _PACK_LOCK:
            if self._packed is None or key != self._packed_key:
                ws, bs = [l.weight for l in lin], [l.bias for l in lin]
                if self._packed is None or self._packed.device != ws[0].device:
                    self._packed = ops.PackedMLP(ws, bs)
                else:
                    with self._packed._lock:
                        self._packed.repack(ws, bs)
                self._packed_key = key
            return self._packed

    def __getstate__(self):  # the packed image is a cache, not state (save_state pickles the module, sunerf.py:62-74)
        st = self.__dict__.copy()
        st.pop('_packed', None)
        st.pop('_packed_key', None)
        return st

    def forward(self, x: torch.Tensor):
        """(M, 4) query points -> {'inferences': (M, d_output)} (model.py:44-57)."""
        from sunerf.rendering.functional import mlp_points
        return {'inferences': mlp_points(self, x)}


class EmissionModel(NeRF):
    """model.py:60-63: ``NeRF`` with ``d_input = 4``, ``d_output = 2`` fixed."""

    def __init__(self, **kwargs):
        super().__init__(d_input=4, d_output=2, **kwargs)


class NeRF_DT(NeRF):
    """model.py:136-187: the same MLP read as (log density, log temperature) with base offsets, plus the 7 per-channel
    absorption scalars and the volumetric constant (same parameter names: ``log_absortpion.<wl>``, ``volumetric_constant``).
    The offsets are applied inside the DT integral kernel (``sunerf_dt_integral_fwd``)."""

    def __init__(self, d_input: int = 4, d_output: int = 2, n_layers: int = 8, d_filter: int = 512,
                 skip: Tuple[int] = (), encoding='positional', base_log_temperature: float = 5.0,
                 base_log_density: float = 10.0):
        super().__init__(d_input=d_input, d_output=d_output, n_layers=n_layers, d_filter=d_filter, skip=skip,
                         encoding=encoding)
        self.base_log_temperature = base_log_temperature
        self.base_log_density = base_log_density
        self.log_absortpion = nn.ParameterDict([[str(w), torch.tensor(1.0e-6, dtype=torch.float32)]
                                                for w in ops.AIA_WAVELENGTHS])
        self.volumetric_constant = nn.Parameter(torch.tensor(1.0, dtype=torch.float32, requires_grad=True))

    def log_abs_vector(self) -> torch.Tensor:
        return torch.stack([self.log_absortpion[str(w)] for w in ops.AIA_WAVELENGTHS])

    def forward(self, x: torch.Tensor):
        from sunerf.rendering.functional import mlp_points
        out = mlp_points(self, x)
        out = torch.stack([out[:, 0] + self.base_log_density, out[:, 1] + self.base_log_temperature], -1)
        return {'inferences': out, 'log_abs': self.log_absortpion, 'vol_c': self.volumetric_constant}
