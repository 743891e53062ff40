"""Tensor-level wrappers over the C ABI: argument validation, output allocation, current-stream plumbing.

PyTorch is used for device memory and streams only; every numerical op of the render path runs in the HIP
kernels of ``csrc/``.
"""
import ctypes
import os
import threading
from typing import Optional, Sequence, Tuple

import torch

from . import lib as _l

SAMPLER_STRATIFIED = 0
SAMPLER_SPHERICAL = 1
SUPPORTED_D_FILTER = (64, 128, 256, 512)
TRAINABLE_D_FILTER = (64, 128, 256, 512)
PRECISION_FAST, PRECISION_EXACT, PRECISION_HALF = 0, 1, 2     # include/sunerf_hip.h: SUNERF_PRECISION_*
PRECISION_AUTO = -1           # host-side policy (not a kernel mode): FAST while a probe shows it inside the gate, else EXACT
PRECISION_NAMES = {PRECISION_FAST: 'fast', PRECISION_EXACT: 'exact', PRECISION_HALF: 'half', PRECISION_AUTO: 'auto'}

# AUTO policy.  FAST (fp16 head + two fp8 correction products) behaves like arithmetic with ~58x the rounding noise of the
# fp32 reference, EXACT (three fp16 products) like ~7x (tests/tools/precision_scan.py): both are far inside
# the north-star gate (1e-4 relative) for freshly initialised and for trained networks, but the noise of ANY arithmetic
# -- the reference's included -- is amplified by the network's conditioning, and with all hidden weights x 4 FAST leaves
# the gate while EXACT stays inside.  So the mode is chosen by MEASUREMENT: every PROBE_EVERY-th parameter version (and the
# first) PROBE_RAYS rays spread evenly over the render call at hand are rendered in both modes and compared in gate units,
#     max_ray |fast - exact| / (1e-4 |exact| + 1e-6 max|exact|)      over image, height_map, absorption_map,
# FAST is kept while that stays below PROBE_LIMIT.  Calibration (tests/tools/probe_calibration.py, hidden weights x 1 ... x 4, two seeds):
# with 144 rays the probe tracks FAST's true worst gate units against the fp32 reference within 10 % (x 2: probe 0.37 / 0.38, true
# 0.32 / 0.41; x 3: 0.40 / 0.49, true 0.43 / 0.47; x 4: 1.19 / 0.92, true 1.26 / 0.95), so 0.5 keeps FAST below ~0.55 of the gate.
PROBE_EVERY = 64
PROBE_RAYS = 144
PROBE_LIMIT = 0.5
# The FIRST probe of an image is read at once (one 4-byte device -> host copy: nothing is known about the network yet); later
# re-probes are ASYNCHRONOUS: the measured units go to a pinned host word behind an event and the decision is taken by the first
# render call that finds the event complete -- the training step never waits for a probe.  Under a process group the units are
# MAX-all-reduced first, so every rank takes the same arithmetic at the same parameter version (equal step times, no rank-dependent
# forward).
PROBE_ASYNC = True
# Under a process group the decision of an asynchronous probe is taken at a DETERMINISTIC point -- the first render call at
# least PROBE_APPLY_AFTER parameter versions after the probe (its event has long completed by then; the call synchronises on it
# to be sure) -- not whenever event.query() first returns true, which differs from rank to rank.  Every rank takes part in the
# MAX all-reduce of the units whatever its own batch looks like (an empty batch contributes 0).
PROBE_APPLY_AFTER = 2
probe_group = None            # process group of the probe's all-reduce (None = the default group); ClipAdam(group=...) sets it


def _probe_world() -> int:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(probe_group)
    return 1


def default_precision(d_filter: int) -> int:
    """Forward arithmetic of newly packed models: ``SUNERF_FORWARD_PRECISION`` = ``auto`` (default: ``fast`` guarded by the
    probe above), ``fast`` (fp16 head product + two block-scaled fp8 correction products), ``exact`` (three fp16
    products per term) or ``half`` (opt-in: single fp16 operands -- the bf16-class arithmetic of BASELINE config 3; NOT
    within 1e-4 of the fp32 reference)."""
    import os
    mode = os.environ.get('SUNERF_FORWARD_PRECISION', 'auto').lower()
    names = {v: k for k, v in PRECISION_NAMES.items()}
    if mode not in names:
        raise ValueError(f"SUNERF_FORWARD_PRECISION must be one of {sorted(names)}, not {mode!r}")
    return names[mode]


_workspaces = {}                        # (device, stream) -> scratch of the d_filter = 512 render kernel
STASH_FP16, STASH_PHASE = 0, 1          # include/sunerf_hip.h: SUNERF_STASH_*


def training_stash_format(packed, n_rays: int, n_samples: int) -> int:
    """What the training forward leaves for the backward: 16-bit phases (half the bytes; what the layer-pipelined backward reads)
    whenever that backward is going to run -- d_filter 256 on a 256-CU device, SUNERF_BACKWARD not 'classic', no rank sharing the
    card --, fp16 sin + cos fragments for the two-kernel backward otherwise.  ``SUNERF_STASH=fp16`` forces the latter (and with
    it the two-kernel backward)."""
    if os.environ.get('SUNERF_STASH', '').strip().lower() == 'fp16' or n_rays <= 0:
        return STASH_FP16
    if backward_mode() != 'pipe':
        return STASH_FP16
    with torch.cuda.device(packed.device):
        ok = _l.load().sunerf_bwd_pipe_workspace_bytes(n_rays, n_samples, packed.d_filter, packed.n_linear) > 0
    if not ok or _shared_device(packed.device):
        return STASH_FP16
    return STASH_PHASE


def stash_format_of(stash, n_rays: int, n_samples: int, packed) -> int:
    """The format a stash tensor was written in, told by its size (the two formats differ by almost 2 x): the size of this batch
    exactly, or -- a stash of more rays used for its first ``n_rays`` (both formats are ray-major) -- a whole number of chunks of
    one format only."""
    lib = _l.load()
    need = {fmt: lib.sunerf_act_stash_bytes(n_rays, n_samples, packed.d_filter, packed.n_linear, fmt) for fmt in (STASH_PHASE, STASH_FP16)}
    for fmt, nbytes in need.items():
        if nbytes and stash.numel() == nbytes:
            return fmt
    chunk = {fmt: lib.sunerf_act_stash_bytes(1, 1, packed.d_filter, packed.n_linear, fmt) // 2 for fmt in need}     # (1 chunk + 1 spare)
    fits = [fmt for fmt, nbytes in need.items() if nbytes and stash.numel() > nbytes and stash.numel() % chunk[fmt] == 0]
    if len(fits) == 1:
        return fits[0]
    raise ValueError('the activation stash does not have the size of either format for this batch')


def _dev(t: torch.Tensor, name: str, shape=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if not t.is_cuda:
        raise _l.SunerfHipError(f'{name} is on {t.device}: the fused renderer has no CPU path '
                                '(move the module and its inputs to a ROCm device)')
    if t.dtype != torch.float32:
        raise TypeError(f'{name} must be float32, got {t.dtype}')
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f'{name} has shape {tuple(t.shape)}, expected {tuple(shape)}')
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class PackedMLP:
    """fp16 hi/lo A-fragment image of one NeRF MLP (see csrc/sunerf_common.h).  Re-pack after every
    parameter update (``repack``)."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], precision: Optional[int] = None):
        self.n_linear = len(weights)
        # Any width up to 512 and both input forms of NeRF (model.py:28-33) run on the compiled kernels by ZERO PADDING, which is
        # exact: a padded hidden unit has zero weights and bias, sin(0) = 0, and zero outgoing weights; a first layer without
        # positional encoding, Linear(4, d), is the 84-input layer with its weights in the four raw-coordinate columns (reference
        # columns 0..3 of the encoder output, model.py:127-132) and zeros under the 80 sin / cos features.
        self.d_model = int(weights[0].shape[0])
        self.d_in = int(weights[0].shape[1])
        if self.d_model < 1 or self.d_model > SUPPORTED_D_FILTER[-1]:
            raise ValueError(f'd_filter={self.d_model} is outside 1..{SUPPORTED_D_FILTER[-1]}')
        if self.d_in not in (4, 84):
            raise ValueError('the first layer takes the 84 positional-encoding features or the 4 raw coordinates')
        self.d_filter = next(d for d in SUPPORTED_D_FILTER if d >= self.d_model)
        self.padded = self.d_filter != self.d_model or self.d_in != 84
        precision = default_precision(self.d_filter) if precision is None else int(precision)
        self.auto = precision == PRECISION_AUTO
        self.precision = PRECISION_FAST if self.auto else precision          # the kernel mode of `buffer`
        self.probe_due = self.auto
        self.last_probe = None            # gate units measured by the last probe (AUTO only)
        self._versions_since_probe = 0
        self._pending_probe = None        # (pinned host word, event, sensitivity) of a probe whose result has not been read yet
        # evaluation/loader.py:226-229 calls the renderer from a ThreadPoolExecutor: (re)packing, probing and the buffer swap of
        # a mode change are serialised; a render call works on the (buffer, precision) pair it read under the lock
        self._lock = threading.RLock()
        self.d_out = int(weights[-1].shape[0])
        lib = _l.load()
        nbytes = lib.sunerf_packed_mlp_bytes(self.d_filter, self.n_linear)
        if nbytes == 0:
            raise ValueError(f'unsupported MLP shape: d_filter={self.d_filter}, n_linear={self.n_linear}')
        self.device = weights[0].device
        self.buffer = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.buffer_t = None        # transposed image for the backward pass, packed on demand
        self._t_valid = False
        self._pad_w = self._pad_b = None
        self.repack(weights, biases)

    def kernel_shapes(self):
        """[(weight shape, bias shape)] of the (padded) network the kernels see."""
        D = self.d_filter
        return [((self.d_out if i == self.n_linear - 1 else D, 84 if i == 0 else D), (self.d_out if i == self.n_linear - 1 else D,))
                for i in range(self.n_linear)]

    def _padded(self, weights, biases):
        """Copies the model's parameters into zero-initialised tensors of the kernel's shapes (device copies only)."""
        if self._pad_w is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            self._pad_w = [torch.zeros(ws, **f32) for ws, _ in self.kernel_shapes()]
            self._pad_b = [torch.zeros(bs, **f32) for _, bs in self.kernel_shapes()]
        for W, b, pw, pb in zip(weights, biases, self._pad_w, self._pad_b):
            pw[:W.shape[0], :W.shape[1]].copy_(W.detach())
            pb[:b.shape[0]].copy_(b.detach())
        return self._pad_w, self._pad_b

    def repack(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]):
        assert len(weights) == self.n_linear and len(biases) == self.n_linear
        if self.padded:
            for i, (w, b) in enumerate(zip(weights, biases)):
                d_in = self.d_in if i == 0 else self.d_model
                d_o = self.d_out if i == self.n_linear - 1 else self.d_model
                _dev(w.detach(), f'weight[{i}]', (d_o, d_in)); _dev(b.detach(), f'bias[{i}]', (d_o,))
            weights, biases = self._padded(weights, biases)
        ws, bs = [], []
        for i, (w, b) in enumerate(zip(weights, biases)):
            d_in = 84 if i == 0 else self.d_filter
            d_o = self.d_out if i == self.n_linear - 1 else self.d_filter
            ws.append(_dev(w.detach(), f'weight[{i}]', (d_o, d_in)))
            bs.append(_dev(b.detach(), f'bias[{i}]', (d_o,)))
        W = (ctypes.c_void_p * self.n_linear)(*[w.data_ptr() for w in ws])
        B = (ctypes.c_void_p * self.n_linear)(*[b.data_ptr() for b in bs])
        _l.call(self.device, 'sunerf_pack_mlp', W, B, self.n_linear, self.d_filter, self.d_out, self.precision,
                _ptr(self.buffer), _stream(self.device))
        self._keepalive = (ws, bs)   # until the pack kernel has run on the stream
        self._t_valid = False
        self._version = getattr(self, '_version', 0) + 1
        if self.auto:
            self._versions_since_probe += 1
            if self._versions_since_probe >= PROBE_EVERY:
                self.probe_due = True

    def _pack_into(self, buffer: torch.Tensor, precision: int):
        ws, bs = self._keepalive
        W = (ctypes.c_void_p * self.n_linear)(*[w.data_ptr() for w in ws])
        B = (ctypes.c_void_p * self.n_linear)(*[b.data_ptr() for b in bs])
        _l.call(self.device, 'sunerf_pack_mlp', W, B, self.n_linear, self.d_filter, self.d_out, precision, _ptr(buffer),
                _stream(self.device))

    def probe(self, rays_o, rays_d, times, z_vals, reg_radius: float, sensitivity: float = 1.0) -> float:
        """AUTO: renders PROBE_RAYS rays of the call in both arithmetics, keeps FAST if it is inside the gate with margin,
        switches this image to EXACT otherwise (and back when a later probe allows it).  One 4-byte device -> host read
        per PROBE_EVERY parameter versions.  Returns the measured gate units.  ``sensitivity``: how much more strongly than
        the emission image the caller's own integral reacts to an error of the raw output (the density-temperature image goes
        with exp(2 raw_0): 2) -- the measured units are multiplied by it."""
        total = rays_o.shape[0]
        n = min(PROBE_RAYS, total)
        self.probe_due = False
        self._versions_since_probe = 0
        world = _probe_world()
        if n == 0 and world <= 1:
            return 0.0
        units = torch.zeros(1, dtype=torch.float32, device=self.device)
        if n > 0:
            # PROBE_RAYS rays spread evenly over the call (the first rays of a frame are an off-disk corner of the image)
            sel = slice(0, (total // n) * n, total // n)
            rays_o, rays_d, z_vals = rays_o[sel].contiguous(), rays_d[sel].contiguous(), z_vals[sel].contiguous()
            times = times.reshape(-1)[sel].contiguous()
            if getattr(self, '_alt_buffer', None) is None:
                self._alt_buffer = torch.empty_like(self.buffer)
            other = PRECISION_EXACT if self.precision == PRECISION_FAST else PRECISION_FAST
            self._pack_into(self._alt_buffer, other)
            self._alt_version = self._version
            views = {self.precision: self.buffer, other: self._alt_buffer}
            outs = {}
            for mode, buf in views.items():
                shadow = object.__new__(PackedMLP)
                shadow.__dict__.update(self.__dict__)
                shadow.buffer, shadow.precision, shadow.auto, shadow._pending_probe = buf, mode, False, None
                outs[mode] = emission_render_fwd(shadow, rays_o, rays_d, times, z_vals, reg_radius, want_epilogues=True)
            for k in ('image', 'height_map', 'absorption_map'):
                f, e = outs[PRECISION_FAST][k].reshape(-1), outs[PRECISION_EXACT][k].reshape(-1)
                # absorption_map = sum(1 - a): the reference forms 1 - a in fp32, i.e. with 2^-24 absolute noise per sample.
                # The 1e-6 max|exact| term is NOT part of the parity gate (tests/conftest.py:gate_units is purely relative): it
                # keeps rays whose exact value is (next to) zero -- off-disk rays of a frame -- from deciding the arithmetic of
                # the whole image by 0 / 0; it can only make the probe more lenient on rays 1e-2 below the brightest one.
                floor = z_vals.shape[1] * 6e-8 if k == 'absorption_map' else 0.0
                units = torch.maximum(units, ((f - e).abs() / (1e-4 * e.abs() + 1e-6 * e.abs().max() + floor)).max())
            units = torch.nan_to_num(units, nan=float('inf'))      # NaN: non-finite outputs in either mode -> EXACT; the finite check reports them
        if world > 1:
            import torch.distributed as dist
            # every rank probes at the same parameter version and takes part whatever its own batch holds: one decision
            dist.all_reduce(units, op=dist.ReduceOp.MAX, group=probe_group)
        host = torch.empty(1, dtype=torch.float32, pin_memory=True)
        host.copy_(units, non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(self.device))
        first = self.last_probe is None
        self._pending_probe = (host, event, float(sensitivity), self._version, world > 1)
        self._apply_probe(block=first or not PROBE_ASYNC)
        return self.last_probe

    def _apply_probe(self, block: bool = False) -> None:
        """Takes the decision of a finished probe (``block``: waits for it)."""
        pending = self._pending_probe
        if pending is None:
            return
        host, event, sensitivity, version, collective = pending
        if block:
            event.synchronize()
        elif collective:
            # ranks must switch at the same parameter version: a fixed distance behind the probe, not "when the event is seen"
            if self._version < version + PROBE_APPLY_AFTER:
                return
            event.synchronize()
        elif not event.query():
            return
        self._pending_probe = None
        units = float(host[0]) * sensitivity
        self.last_probe = units
        want = PRECISION_FAST if units <= PROBE_LIMIT else PRECISION_EXACT
        if want != self.precision:
            # A render call of another thread may still hold (self.buffer, old precision) as the pair it read under the lock:
            # the live buffer is never re-packed in another arithmetic.  The image of the wanted mode goes into the alternate
            # buffer (the probe left it there if the parameters have not changed since) and the two are swapped.
            if getattr(self, '_alt_version', None) != self._version:
                self._alt_buffer = torch.empty_like(self.buffer)      # a fresh one: the old alternate may be some call's snapshot too
                self._pack_into(self._alt_buffer, want)
            self.buffer, self._alt_buffer = self._alt_buffer, self.buffer
            self.precision = want
            self._alt_version = None       # what is now the alternate holds the OTHER mode of this version at best: re-pack on use

    def wait_probe(self) -> Optional[float]:
        """Blocks until a probe in flight has been read and its decision taken; returns the last measured gate units."""
        with self._lock:
            self._apply_probe(block=True)
            return self.last_probe

    def transposed(self) -> torch.Tensor:
        """fp16 W^T image consumed by sunerf_mlp_dgrad (packed lazily, once per parameter version)."""
        lib = _l.load()
        if self.buffer_t is None:
            self.buffer_t = torch.empty(lib.sunerf_packed_mlp_t_bytes(self.d_filter, self.n_linear), dtype=torch.uint8,
                                        device=self.device)
        if not self._t_valid:
            ws = self._keepalive[0]
            W = (ctypes.c_void_p * self.n_linear)(*[w.data_ptr() for w in ws])
            _l.call(self.device, 'sunerf_pack_mlp_t', W, self.n_linear, self.d_filter, self.d_out,
                    _ptr(self.buffer_t), _stream(self.device))
            self._t_valid = True
        return self.buffer_t


def sample_z(kind: int, rays_o, rays_d, t_vals, distance: float, solar_R: float,
             t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _l.load()
    n = rays_o.shape[0]
    rays_o = _dev(rays_o, 'rays_o', (n, 3))
    rays_d = _dev(rays_d, 'rays_d', (n, 3))
    t_vals = _dev(t_vals.reshape(-1), 't_vals')
    s = t_vals.numel()
    if t_rand is not None:
        t_rand = _dev(t_rand, 't_rand', (n, s))
    z = torch.empty(n, s, dtype=torch.float32, device=rays_o.device)
    _l.call(rays_o.device, 'sunerf_sample_z', kind, _ptr(rays_o), _ptr(rays_d), _ptr(t_vals), _ptr(t_rand), n, s,
            float(distance), float(solar_R), _ptr(z), _stream(rays_o.device))
    return z


def emission_render_fwd(packed: PackedMLP, rays_o, rays_d, times, z_vals, reg_radius: float,
                        want_raw: bool = False, want_epilogues: bool = False, training: bool = False,
                        probe_sensitivity: float = 1.0):
    """One fused render pass.  Returns dict(image (N,1), weights (N,S), absorption (N,S)[, raw (N,S,2)]
    [, height_map (N,), absorption_map (N,), regularization (N,S)][, stash]).  ``training=True`` also writes the
    activation stash needed by :func:`emission_render_bwd` (and implies ``want_raw``)."""
    lib = _l.load()
    n, s = z_vals.shape
    dev = z_vals.device
    rays_o = _dev(rays_o, 'rays_o', (n, 3))
    rays_d = _dev(rays_d, 'rays_d', (n, 3))
    times = _dev(times.reshape(-1), 'times', (n,))
    z_vals = _dev(z_vals, 'z_vals', (n, s))
    if packed.device != dev:
        raise _l.SunerfHipError('packed weights and rays are on different devices')
    with packed._lock:
        packed._apply_probe()
        if packed.auto and packed.probe_due:
            packed.probe(rays_o, rays_d, times, z_vals, reg_radius, probe_sensitivity)
        weights_image, precision = packed.buffer, packed.precision        # this call's image, whatever other threads decide next
    f32 = dict(dtype=torch.float32, device=dev)
    out = {'image': torch.empty(n, 1, **f32), 'weights': torch.empty(n, s, **f32),
           'absorption': torch.empty(n, s, **f32)}
    want_raw = want_raw or training
    if training and packed.d_filter not in TRAINABLE_D_FILTER:
        raise NotImplementedError(f'training with d_filter={packed.d_filter} is not implemented (inference only); '
                                  f'trainable widths: {TRAINABLE_D_FILTER}')
    ws_bytes = lib.sunerf_render_workspace_bytes(packed.d_filter)
    ws = None
    if ws_bytes:
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _workspaces.get(key)
        if ws is None or ws.numel() < ws_bytes:
            ws = _workspaces[key] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    raw = torch.empty(n, s, 2, **f32) if want_raw else None
    stash, fmt = None, STASH_FP16
    if training:
        fmt = training_stash_format(packed, n, s)
        stash = torch.empty(lib.sunerf_act_stash_bytes(n, s, packed.d_filter, packed.n_linear, fmt), dtype=torch.uint8,
                            device=dev)
    hm = am = reg = None
    if want_epilogues:
        hm, am, reg = torch.empty(n, **f32), torch.empty(n, **f32), torch.empty(n, s, **f32)
    _l.call(dev, 'sunerf_emission_render_fwd', _ptr(weights_image), packed.d_filter, packed.n_linear,
            precision, _ptr(rays_o), _ptr(rays_d), _ptr(times), _ptr(z_vals), n, s, _ptr(out['image']),
            _ptr(out['weights']), _ptr(out['absorption']), _ptr(raw), _ptr(hm), _ptr(am), _ptr(reg),
            float(reg_radius), _ptr(stash), fmt, _ptr(ws), ws_bytes, _stream(dev))
    if want_raw:
        out['raw'] = raw
    if training:
        out['stash'] = stash
    if want_epilogues:
        out.update(height_map=hm, absorption_map=am, regularization=reg)
    return out


def mlp_points_fwd(packed: PackedMLP, points: torch.Tensor, training: bool = False):
    """NeRF.forward on free-standing points (M, 4) -> dict(raw (M, 2)[, stash, n_padded]).  The points are padded to whole
    32-point chunks; ``training=True`` also writes the activation stash :func:`mlp_backward` needs (with ``g_raw`` of shape
    (n_padded / 32, 32, 2))."""
    lib = _l.load()
    m = points.shape[0]
    dev = points.device
    points = _dev(points, 'points', (m, 4))
    if packed.device != dev:
        raise _l.SunerfHipError('packed weights and points are on different devices')
    with packed._lock:
        packed._apply_probe()
        if packed.auto and packed.probe_due and (m > 0 or _probe_world() > 1):
            # the measured choice of the arithmetic (AUTO) needs rays: PROBE_RAYS of the points as two-sample rays o = 0, d = xyz, z = 1
            # (a rank without points still takes part in the probe's all-reduce, with zero units)
            k = min(PROBE_RAYS, m)
            idx = torch.linspace(0, max(m - 1, 0), k, device=dev).long()
            sel = points[idx]
            packed.probe(torch.zeros(k, 3, device=dev), sel[:, :3].contiguous(), sel[:, 3].contiguous(), torch.ones(k, 2, device=dev), 0.0)
        weights_image, precision = packed.buffer, packed.precision
    m_pad = (m + 31) // 32 * 32
    if m_pad != m:
        points = torch.cat([points, points.new_zeros(m_pad - m, 4)])
    if training and packed.d_filter not in TRAINABLE_D_FILTER:
        raise NotImplementedError(f'training with d_filter={packed.d_filter} is not implemented (inference only)')
    ws_bytes = lib.sunerf_render_workspace_bytes(packed.d_filter)
    ws = None
    if ws_bytes:
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _workspaces.get(key)
        if ws is None or ws.numel() < ws_bytes:
            ws = _workspaces[key] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    raw = torch.empty(m_pad, 2, dtype=torch.float32, device=dev)
    stash, fmt = None, STASH_FP16
    if training:
        fmt = training_stash_format(packed, m_pad // 32, 32)
        stash = torch.empty(lib.sunerf_act_stash_bytes(m_pad // 32, 32, packed.d_filter, packed.n_linear, fmt), dtype=torch.uint8, device=dev)
    _l.call(dev, 'sunerf_mlp_points_fwd', _ptr(weights_image), packed.d_filter, packed.n_linear, precision, _ptr(points),
            m_pad, _ptr(raw), _ptr(stash), fmt, _ptr(ws), ws_bytes, _stream(dev))
    out = {'raw': raw[:m], 'n_padded': m_pad}
    if training:
        out['stash'] = stash
    return out


def hier_resample(z_vals, weights, u: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``u``: (S_f,) shared sample positions (perturb=False) or (N, S_f) per-ray (perturb=True)."""
    lib = _l.load()
    n, sc = z_vals.shape
    z_vals = _dev(z_vals, 'z_vals', (n, sc))
    weights = _dev(weights.detach(), 'weights', (n, sc))
    per_ray = int(u.dim() == 2)
    sf = u.shape[-1]
    order = None
    if per_ray:
        # perturb=True: random positions.  The inverse CDF is monotone, so SORTED positions give sorted new samples and the
        # kernel merges two ascending runs by rank; handed unsorted ones it falls back to one lane's insertion sort per ray
        # (2.6 ms instead of 10 us for 3072 rays x 64 + 128 samples).  The new samples are returned in the caller's order.
        u, order = torch.sort(u, dim=-1)
    u = _dev(u, 'u', (n, sf) if per_ray else (sf,))
    new_z = torch.empty(n, sf, dtype=torch.float32, device=z_vals.device)
    z_comb = torch.empty(n, sc + sf, dtype=torch.float32, device=z_vals.device)
    _l.call(z_vals.device, 'sunerf_hier_resample', _ptr(z_vals), _ptr(weights), _ptr(u), per_ray, n, sc, sf,
            _ptr(new_z), _ptr(z_comb), _stream(z_vals.device))
    if order is not None:
        new_z = torch.empty_like(new_z).scatter_(-1, order, new_z)
    return new_z, z_comb


def sample_pdf(bins, weights, u: torch.Tensor) -> torch.Tensor:
    """HierarchicalSampler.sample_pdf (sampling.py:128-169) on given bins (N, B) and weights (N, B-1) -> samples (N, S_f);
    ``u`` as in :func:`hier_resample`."""
    n, nb = bins.shape
    bins = _dev(bins, 'bins', (n, nb))
    weights = _dev(weights.detach(), 'weights', (n, nb - 1))
    per_ray = int(u.dim() == 2)
    sf = u.shape[-1]
    u = _dev(u, 'u', (n, sf) if per_ray else (sf,))
    samples = torch.empty(n, sf, dtype=torch.float32, device=bins.device)
    _l.call(bins.device, 'sunerf_sample_pdf', _ptr(bins), _ptr(weights), _ptr(u), per_ray, n, nb, sf, _ptr(samples),
            _stream(bins.device))
    return samples


def wgrad_split(n_linear: int, n_cus: int = 256, d_filter: int = 256) -> int:
    """Partial sums per layer in sunerf_mlp_wgrad: n_linear * split (* 4 workgroup blocks at d_filter = 512) workgroups
    must fit the chip in ONE wave (one workgroup per CU, all about equally long): 9 layers -> 28 (252 workgroups);
    288 would take two rounds."""
    blocks = 4 if d_filter > 256 else 1
    return max(1, n_cus // (n_linear * blocks))


def emission_render_bwd(packed: PackedMLP, rays_o, rays_d, z_vals, raw, stash, g_image, g_reg, g_reg_const: float,
                        reg_radius: float, grad_weights: Sequence[torch.Tensor], grad_biases: Sequence[torch.Tensor],
                        accumulate: bool = False, times=None):
    """Backward of one render pass: fills (or accumulates into) ``grad_weights[i]`` / ``grad_biases[i]`` (nn.Linear
    layouts) from the gradient w.r.t. the 'image' output (N,) or (N,1) and the 'regularization' output
    (``g_reg`` (N,S) tensor or None + the constant ``g_reg_const``).  ``times`` (N,): the forward's time coordinate -- with it
    the small-batch fp32 backward can recompute the activations (``mlp_backward(query=...)``); without it the fp16 kernels run."""
    lib = _l.load()
    n, s = z_vals.shape
    dev = z_vals.device
    rays_o = _dev(rays_o, 'rays_o', (n, 3))
    rays_d = _dev(rays_d, 'rays_d', (n, 3))
    z_vals = _dev(z_vals, 'z_vals', (n, s))
    raw = _dev(raw, 'raw', (n, s, 2))
    g_image = _dev(g_image.reshape(-1), 'g_image', (n,))
    if g_reg is not None:
        g_reg = _dev(g_reg, 'g_reg', (n, s))
    D, nl = packed.d_filter, packed.n_linear
    g_raw = torch.empty(n, s, 2, dtype=torch.float32, device=dev)
    absmax = torch.empty(1, dtype=torch.int32, device=dev)
    stream = _stream(dev)
    _l.call(dev, 'sunerf_emission_integral_bwd', _ptr(raw), _ptr(z_vals), _ptr(rays_o), _ptr(rays_d), _ptr(g_image),
            _ptr(g_reg), None, None, float(g_reg_const), float(reg_radius), n, s, _ptr(g_raw), _ptr(absmax), stream)
    mlp_backward(packed, g_raw, absmax, stash, grad_weights, grad_biases, accumulate,
                 query=None if times is None else ('rays', rays_o, rays_d, times, z_vals))
    return g_raw


def emission_integral_fwd(raw, z_vals, rays_d):
    """EmissionRadiativeTransfer.raw2outputs (emission.py:14-54) on a given raw tensor -> (image (N,1), weights, absorption)."""
    n, s = z_vals.shape
    dev = z_vals.device
    raw = _dev(raw, 'raw', (n, s, 2)); z_vals = _dev(z_vals, 'z_vals', (n, s)); rays_d = _dev(rays_d, 'rays_d', (n, 3))
    f32 = dict(dtype=torch.float32, device=dev)
    image, weights, absorption = torch.empty(n, 1, **f32), torch.empty(n, s, **f32), torch.empty(n, s, **f32)
    _l.call(dev, 'sunerf_emission_integral_fwd', _ptr(raw), _ptr(z_vals), _ptr(rays_d), n, s, _ptr(image), _ptr(weights),
            _ptr(absorption), _stream(dev))
    return image, weights, absorption


def emission_integral_bwd(raw, z_vals, rays_d, g_image=None, g_weights=None, g_absorption=None):
    """d / d raw of :func:`emission_integral_fwd` for gradients w.r.t. any of its three outputs -> g_raw (N,S,2)."""
    n, s = z_vals.shape
    dev = z_vals.device
    raw = _dev(raw, 'raw', (n, s, 2)); z_vals = _dev(z_vals, 'z_vals', (n, s)); rays_d = _dev(rays_d, 'rays_d', (n, 3))
    g_image = torch.zeros(n, dtype=torch.float32, device=dev) if g_image is None else _dev(g_image.reshape(-1), 'g_image', (n,))
    g_weights = None if g_weights is None else _dev(g_weights, 'g_weights', (n, s))
    g_absorption = None if g_absorption is None else _dev(g_absorption, 'g_absorption', (n, s))
    g_raw = torch.empty(n, s, 2, dtype=torch.float32, device=dev)
    absmax = torch.empty(1, dtype=torch.int32, device=dev)
    # (rays_o only enters through the regularization term, which is not an output of raw2outputs: g_reg = 0)
    _l.call(dev, 'sunerf_emission_integral_bwd', _ptr(raw), _ptr(z_vals), _ptr(rays_d), _ptr(rays_d), _ptr(g_image), None,
            _ptr(g_weights), _ptr(g_absorption), 0.0, 0.0, n, s, _ptr(g_raw), _ptr(absmax), _stream(dev))
    return g_raw


# ---- which backward runs -------------------------------------------------------------------------
# 'pipe' (default): the layer-pipelined kernel of csrc/bwd_pipe.hip where it applies (d_filter 256, n_linear >= 3, a 256-CU
# device), the two-kernel dgrad + wgrad elsewhere; 'classic': always the two kernels.  The pipelined launch needs all of its
# 256 workgroups resident at once: ranks that SHARE one GPU (the CPU-rehearsal tests) must use 'classic'.
_backward_forced = None
pipe_timing = False                       # True: every pipelined launch is bracketed by library-owned HIP events (flags bit 7)
_pipe_ws = {}                             # (device, stream) -> workspace
_pipe_checked = {}                        # workspaces whose sticky status word has not been looked at yet
PIPE_WS_STICKY, PIPE_WS_DEBUG = 0, 256    # include/sunerf_hip.h: fixed offsets of the sticky status block / the debug counters


def backward_mode() -> str:
    if _backward_forced is not None:
        return _backward_forced
    mode = os.environ.get('SUNERF_BACKWARD', 'pipe').lower()
    if mode not in ('pipe', 'classic'):
        raise ValueError(f"SUNERF_BACKWARD must be 'pipe' or 'classic', not {mode!r}")
    return mode


def _shared_device(dev) -> bool:
    """Ranks of one process group that drive the SAME GPU cannot all keep a 256-workgroup persistent launch resident: the
    process then uses the two-kernel backward (decided once, collectively, when the optimiser / gradient bucket is built:
    sunerf_hip.dist.ranks_share_a_device; here only the cached answer is read -- no collective inside a backward)."""
    global _backward_forced
    from . import dist as _dist
    if _dist.shared_device_known(dev):
        import warnings
        warnings.warn('several ranks of the process group share one GPU: the layer-pipelined backward needs the whole device, '
                      'this process uses the two-kernel backward (SUNERF_BACKWARD=classic)', RuntimeWarning)
        _backward_forced = 'classic'
        return True
    return False


def _env_on(name: str) -> bool:
    return os.environ.get(name, '0').lower() not in ('', '0', 'false', 'no', 'off')


def pipe_w_mode() -> str:
    """Precision of W^T in the pipelined backward's data gradient: 'auto' (default: a single fp16 W^T while a measured probe
    allows it, fp16 head + remainder otherwise -- see _pipe_w_probe), 'hi' (SUNERF_PIPE_HI_ONLY=1), 'hilo' (=0)."""
    v = os.environ.get('SUNERF_PIPE_HI_ONLY', 'auto').strip().lower()
    if v in ('', 'auto'):
        return 'auto'
    return 'hilo' if v in ('0', 'false', 'no', 'off') else 'hi'


def _pipe_flags() -> int:
    return (1 if pipe_w_mode() == 'hi' else 0) | (2 if _env_on('SUNERF_PIPE_DEBUG') else 0) | (0x80 if pipe_timing else 0)


# ---- single or split W^T in the pipelined backward: chosen by measurement, like the forward arithmetic -----------------------
# The data-gradient waves multiply dZ by W^T as fp16 head + fp16 remainder (32 matrix instructions per chunk) or by the head
# alone (16: the kernel -5.7 %, the training step -3.3 %, tools/experiments/r4_pipe_ab.sh).  The head alone is a SYSTEMATIC
# perturbation of the weights (2^-12 per element, the same for every sample), so its effect on the weight gradients does not
# average over the batch -- and for the same reason it can be measured on a few rays: the relative difference between the two
# arithmetics on the first 64 rays of a batch predicts the difference on 8192 rays within 2 %
# (tools/experiments/r4_hi_only_accuracy.py, profiles/r4_ab/r4_hi_only_accuracy.log: probe 5.2e-4 at default initialisation, 5.0e-4 with
# hidden weights x 2, 7.5e-4 at x 3, 8.5e-4 at x 4; against fp32 gradients: head + remainder 2.3 ... 2.8e-4 / 4.0e-4 / 6.6e-4, head
# alone 5.3 ... 6.0e-4 / 9.2e-4 / 8.4e-4).  Every PROBE_EVERY-th parameter version (and at the first pipelined backward of a model)
# both arithmetics run on those rays; the head alone is used while the worst weight tensor differs by at most PIPE_W_LIMIT, which
# keeps the gradients within ~0.6 of SURVEY 8d's 1e-3.  No collective: under data parallelism every rank decides for itself (the
# all-reduced gradient, and with it every replica, is the same on all ranks whichever arithmetic produced a rank's share).
PIPE_W_PROBE_RAYS = 64
PIPE_W_LIMIT = 6e-4


def _pipe_w_probe(packed, call_prefix):
    """Runs the pipelined backward on the first PIPE_W_PROBE_RAYS rays in both arithmetics into scratch gradients and leaves the
    worst relative difference of a weight tensor in a pinned host word behind an event (read at once for the first probe)."""
    dev = packed.device
    if getattr(packed, '_pipe_probe_bufs', None) is None:
        f32 = dict(dtype=torch.float32, device=dev)
        packed._pipe_probe_bufs = [([torch.empty(ws, **f32) for ws, _ in packed.kernel_shapes()],
                                    [torch.empty(bs, **f32) for _, bs in packed.kernel_shapes()]) for _ in range(2)]
    nl = packed.n_linear
    for hi_only, (gW, gb) in zip((0, 1), packed._pipe_probe_bufs):
        GW = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gW])
        GB = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gb])
        call_prefix(GW, GB, hi_only)
    (a, _), (b, _) = packed._pipe_probe_bufs
    diff = torch._foreach_norm(torch._foreach_sub(b, a))
    base = torch._foreach_norm(a)
    units = torch.nan_to_num((torch.stack(diff) / torch.stack(base)).max().reshape(1), nan=float('inf'))
    host = torch.empty(1, dtype=torch.float32, pin_memory=True)
    host.copy_(units, non_blocking=True)
    event = torch.cuda.Event()
    event.record(torch.cuda.current_stream(dev))
    first = getattr(packed, 'pipe_w_probe', None) is None
    packed._pipe_w_pending = (host, event)
    packed._pipe_probe_version = packed._version
    _pipe_w_apply(packed, block=first or not PROBE_ASYNC)


def _pipe_w_apply(packed, block: bool = False):
    pending = getattr(packed, '_pipe_w_pending', None)
    if pending is None:
        return
    host, event = pending
    if block:
        event.synchronize()
    elif not event.query():
        return
    packed._pipe_w_pending = None
    packed.pipe_w_probe = float(host[0])
    packed.pipe_hi_only = packed.pipe_w_probe <= PIPE_W_LIMIT


def pipe_kernel_time():
    """(sum of the kernel durations in ms, number of launches) of the pipelined-backward launches issued while
    ``ops.pipe_timing`` was set, measured by HIP events inside the C ABI on the launch stream; waits for them and forgets them."""
    ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
    _l.check(_l.load().sunerf_bwd_pipe_kernel_time(ctypes.byref(ms), ctypes.byref(n)), 'sunerf_bwd_pipe_kernel_time')
    return ms.value, n.value


def pipe_debug(dev=None):
    """SUNERF_PIPE_DEBUG=1: per-workgroup counters of the last pipelined launch, (8, 256, 8) int32: [0] loop ticks (100 MHz),
    fallback spins and their ticks on the input link, the same on the output link, chunks, layer, pipeline; [1] shader clocks / 16 of
    data wave 1 per phase of its loop (wait, barrier, k-steps, epilogue + decoder; of the latter: products + conversion, .. + stores); [2], [3] the same of weight waves 4 and 5 (top, operand
    reads + gate, matrix phase, counted wait, barrier); [4], [5] viewed as int64 (32, 8, 8): the stamps of iteration n / 2 of every
    wave of the workgroups of pipeline 0 (tools/pipe_check.py prints them as a timeline)."""
    for (d, _), ws in _pipe_ws.items():
        if dev is None or d == dev:
            return ws[PIPE_WS_DEBUG:PIPE_WS_DEBUG + 256 * 64 * 4].view(torch.int32).reshape(8, 256, 8).cpu()
    return None


def _pipe_workspace(dev, nbytes: int) -> torch.Tensor:
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _pipe_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None and key in _pipe_checked:
            pipe_status()          # the outgoing workspace's sticky word is looked at before it is dropped
        ws = _pipe_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws[PIPE_WS_STICKY:PIPE_WS_DEBUG].zero_()      # the sticky status block is the caller's: zeroed once, read + cleared by pipe_status
    _pipe_checked[key] = ws
    return ws


def pipe_status(raise_on_failure: Optional[bool] = None) -> int:
    """Worst status of ALL pipelined backward launches since the last call (one 4-byte read per workspace; call it where the
    step synchronises anyway).  The word is sticky: every launch's reduce kernel raises it to its own status, nothing but this
    function clears it -- a give-up of the fine model's launch is still there after the coarse model's launch on the same
    workspace.  Non-zero: a launch gave up (csrc/bwd_pipe.hip) -- its gradients were NaN, so the optimiser skipped that step.
    The process then switches to the two-kernel backward and says so: with a warning by default, with an exception when
    ``SUNERF_BACKWARD=pipe`` was asked for explicitly (or ``raise_on_failure=True``)."""
    global _backward_forced
    worst = 0
    for key, ws in list(_pipe_checked.items()):
        word = ws[PIPE_WS_STICKY:PIPE_WS_STICKY + 4].view(torch.int32)
        status = int(word.item())
        if status:
            word.zero_()
        worst = max(worst, status)
        del _pipe_checked[key]
    if worst:
        _backward_forced = 'classic'
        msg = (f'the pipelined backward gave up (status {worst}: 1 = workgroups not co-resident, 2 = a workgroup class was not '
               'placed on one XCD, 3 = a hand-off timed out); the step was skipped (NaN gradients) and this process now uses the '
               'two-kernel backward (SUNERF_BACKWARD=classic selects it from the start)')
        if raise_on_failure is None:
            raise_on_failure = os.environ.get('SUNERF_BACKWARD', '').lower() == 'pipe'
        if raise_on_failure:
            raise _l.SunerfHipError(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning)
    return worst


# ---- small batches: the reference's arithmetic (csrc/bwd_exact.hip) ---------------------------------------------------------
# The fp16 backward kernels carry ~2^-12 of relative rounding error per term of a gradient sum (dZ, cos, H are single fp16
# operands).  A training batch averages that away (every tensor within 1e-3 of the fp32 oracle from ~1e4 samples on); a batch of a
# few hundred samples whose bias sums cancel to a few per cent of their terms does not (tests/tools/bias_conditioning.py).  Up to
# EXACT_BACKWARD_SAMPLES samples per call -- where the fp16 kernels are launch-latency-bound anyway -- the backward therefore
# recomputes the activations and runs the chain in fp32.  SUNERF_EXACT_BACKWARD_SAMPLES overrides the limit (0: never); a
# backward kernel asked for BY NAME (SUNERF_BACKWARD, tests forcing a mode) is always honoured.
EXACT_BACKWARD_SAMPLES = 4096


def exact_backward_limit() -> int:
    v = os.environ.get('SUNERF_EXACT_BACKWARD_SAMPLES', '').strip()
    return EXACT_BACKWARD_SAMPLES if v == '' else max(0, int(v))


def _use_exact_backward(n_samples: int, query) -> bool:
    if query is None or n_samples <= 0:
        return False
    if _backward_forced is not None or os.environ.get('SUNERF_BACKWARD', '').strip():
        return False
    return n_samples <= exact_backward_limit()


_exact_ws = {}          # (device, stream) -> workspace of the fp32 backward


def _mlp_backward_exact(packed: PackedMLP, g_raw, query, grad_weights, grad_biases, accumulate: bool):
    lib = _l.load()
    dev = g_raw.device
    n, s = g_raw.shape[0], g_raw.shape[1]
    ws_, bs_ = packed._keepalive          # fp32 parameters of the kernel shapes (the padded copies for padded models)
    nl = packed.n_linear
    nbytes = lib.sunerf_mlp_backward_exact_workspace_bytes(n * s, packed.d_filter, nl)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _exact_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _exact_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    W = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in ws_])
    B = (ctypes.c_void_p * nl)(*[b.data_ptr() for b in bs_])
    GW = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in grad_weights])
    GB = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in grad_biases])
    if query[0] == 'rays':
        _, o, d, t, z = query
        o, d = _dev(o, 'rays_o', (n, 3)), _dev(d, 'rays_d', (n, 3))
        t, z = _dev(t.reshape(-1), 'times', (n,)), _dev(z, 'z_vals', (n, s))
        args = (_ptr(o), _ptr(d), _ptr(t), _ptr(z), None)
    else:
        pts = _dev(query[1], 'points', (n * s, 4))
        args = (None, None, None, None, _ptr(pts))
    g = g_raw if g_raw.shape[-1] == packed.d_out else g_raw[..., :packed.d_out]       # (N, S, d_out), densely packed
    g = _dev(g, 'g_raw')
    _l.call(dev, 'sunerf_mlp_backward_exact', W, B, nl, packed.d_filter, packed.d_out, *args, n, s, _ptr(g),
            _ptr(ws), nbytes, GW, GB, int(accumulate), _stream(dev))


def mlp_backward(packed: PackedMLP, g_raw, absmax, stash, grad_weights: Sequence[torch.Tensor],
                 grad_biases: Sequence[torch.Tensor], accumulate: bool = False, query=None):
    """dgrad + wgrad of the sine MLP from the gradient w.r.t. its raw output (N,S,2): fills / accumulates the nn.Linear
    gradients.  ``absmax``: 4-byte device scalar with the bit pattern of max |g_raw| (written by the integral backward).
    ``query``: what the forward was evaluated on -- ``('rays', rays_o, rays_d, times, z_vals)`` or ``('points', points (N*S, 4))``;
    given it, batches of at most ``exact_backward_limit()`` samples take the fp32 backward (csrc/bwd_exact.hip)."""
    lib = _l.load()
    n, s = g_raw.shape[0], g_raw.shape[1]
    dev = g_raw.device
    D, nl = packed.d_filter, packed.n_linear
    stream = _stream(dev)
    exact = _use_exact_backward(n * s, query)
    pipe_bytes = 0
    if n > 0 and not exact:
        # the forward chose the stash format for the backward it expected (training_stash_format): phases are read by the
        # pipelined kernel only, fp16 sin + cos fragments by the two kernels only
        fmt = stash_format_of(stash, n, s, packed)
        if fmt == STASH_PHASE:
            if backward_mode() != 'pipe':
                raise _l.SunerfHipError('this activation stash holds 16-bit phases (written for the layer-pipelined backward) but the '
                                        'two-kernel backward was selected after the forward ran: choose SUNERF_BACKWARD before the forward, '
                                        'or SUNERF_STASH=fp16')
            with torch.cuda.device(dev):
                pipe_bytes = lib.sunerf_bwd_pipe_workspace_bytes(n, s, D, nl)
            if not pipe_bytes:
                raise _l.SunerfHipError('phase stash but no pipelined backward for this shape / device')
    if not pipe_bytes and not exact:
        dz = torch.empty(lib.sunerf_dz_stash_bytes(n, s, D, nl), dtype=torch.uint8, device=dev)
        _l.call(dev, 'sunerf_mlp_dgrad', _ptr(packed.transposed()), D, nl, _ptr(g_raw), _ptr(absmax), _ptr(stash),
                _ptr(dz), n, s, stream)
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        cap = int(os.environ.get('SUNERF_GRID_CAP_WGRAD', 0))        # experiment knob, see csrc/sunerf_common.h
        split = wgrad_split(nl, cap if 0 < cap < cus else cus, D)
        ws = torch.empty(lib.sunerf_wgrad_workspace_bytes(packed.d_filter, nl, split), dtype=torch.uint8, device=dev)
    out_w, out_b = list(grad_weights), list(grad_biases)
    if packed.padded:
        # zero-padded model (PackedMLP.__init__): the kernels produce gradients of the padded shapes; the model's are their
        # leading blocks (the padding's own gradients are discarded: those weights are not parameters)
        for i, (gw, gb) in enumerate(zip(out_w, out_b)):
            d_in = packed.d_in if i == 0 else packed.d_model
            d_o = packed.d_out if i == nl - 1 else packed.d_model
            if gw.shape != (d_o, d_in) or gb.shape != (d_o,) or gw.dtype != torch.float32:
                raise ValueError(f'grad buffer {i} has the wrong shape / layout')
        if getattr(packed, '_pad_gw', None) is None:
            f32 = dict(dtype=torch.float32, device=dev)
            packed._pad_gw = [torch.empty(ws, **f32) for ws, _ in packed.kernel_shapes()]
            packed._pad_gb = [torch.empty(bs, **f32) for _, bs in packed.kernel_shapes()]
        grad_weights, grad_biases, kernel_accumulate = packed._pad_gw, packed._pad_gb, False
    else:
        kernel_accumulate = accumulate
    for i, (gw, gb) in enumerate(zip(grad_weights, grad_biases)):
        d_in = 84 if i == 0 else D
        d_o = packed.d_out if i == nl - 1 else D
        if gw.shape != (d_o, d_in) or gb.shape != (d_o,) or not gw.is_contiguous() or gw.dtype != torch.float32:
            raise ValueError(f'grad buffer {i} has the wrong shape / layout')
    GW = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in grad_weights])
    GB = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in grad_biases])
    if exact:
        _mlp_backward_exact(packed, g_raw, query, grad_weights, grad_biases, kernel_accumulate)
    elif pipe_bytes:
        ws = _pipe_workspace(dev, pipe_bytes)
        flags = _pipe_flags()
        if pipe_w_mode() == 'auto':
            with packed._lock:
                due = (getattr(packed, '_pipe_probe_version', None) is None
                       or packed._version - packed._pipe_probe_version >= PROBE_EVERY)
                if due and n >= PIPE_W_PROBE_RAYS and getattr(packed, '_pipe_w_pending', None) is None:
                    k = PIPE_W_PROBE_RAYS

                    def call_prefix(gw, gb, hi_only):      # the first k rays: a prefix of g_raw and of the (ray-major) stash
                        _l.call(dev, 'sunerf_mlp_backward_pipe', D, nl, packed.d_out, _ptr(packed.transposed()), _ptr(stash),
                                _ptr(g_raw), _ptr(absmax), k, s, _ptr(ws), pipe_bytes, gw, gb, 0, (flags & ~0x81) | hi_only, stream)
                    _pipe_w_probe(packed, call_prefix)
                _pipe_w_apply(packed)
                if getattr(packed, 'pipe_hi_only', False):
                    flags |= 1
        pargs = (D, nl, packed.d_out, _ptr(packed.transposed()), _ptr(stash), _ptr(g_raw), _ptr(absmax), n, s, _ptr(ws),
                 pipe_bytes, GW, GB, int(kernel_accumulate))
        _l.call(dev, 'sunerf_mlp_backward_pipe', *pargs, flags, stream)
    else:
        _l.call(dev, 'sunerf_mlp_wgrad', D, nl, packed.d_out, _ptr(packed.transposed()), _ptr(stash), _ptr(dz), _ptr(g_raw), _ptr(absmax), n, s, _ptr(ws),
                split, GW, GB, int(kernel_accumulate), stream)
    if packed.padded:
        for gw, gb, pw, pb in zip(out_w, out_b, grad_weights, grad_biases):
            if accumulate:
                gw.add_(pw[:gw.shape[0], :gw.shape[1]])
                gb.add_(pb[:gb.shape[0]])
            else:
                gw.copy_(pw[:gw.shape[0], :gw.shape[1]])
                gb.copy_(pb[:gb.shape[0]])


AIA_WAVELENGTHS = (94, 131, 171, 193, 211, 304, 335)


def dt_integral_fwd(raw, z_vals, rays_o, rays_d, wavelengths, table_logt, table_resp, log_abs, vol_c, base_log_density,
                    base_log_temperature, pixel_intensity_factor, reg_radius, want_epilogues=False):
    """DT radiative-transfer integral on the raw MLP output (density_temperature.py:192-274)."""
    lib = _l.load()
    n, s = z_vals.shape
    dev = z_vals.device
    w = wavelengths.shape[1]
    raw = _dev(raw, 'raw', (n, s, 2)); z_vals = _dev(z_vals, 'z_vals', (n, s))
    rays_o = _dev(rays_o, 'rays_o', (n, 3)); rays_d = _dev(rays_d, 'rays_d', (n, 3))
    wavelengths = _dev(wavelengths.to(torch.float32), 'wavelengths', (n, w))
    table_logt = _dev(table_logt, 'table_logt', (7, 101)); table_resp = _dev(table_resp, 'table_resp', (7, 101))
    log_abs = _dev(log_abs.detach(), 'log_abs', (7,)); vol_c = _dev(vol_c.detach().reshape(1), 'vol_c', (1,))
    f32 = dict(dtype=torch.float32, device=dev)
    out = {'image': torch.empty(n, w, **f32), 'weights': torch.empty(n, s, **f32), 'reg_q': torch.empty(n, s, **f32)}
    hm = am = reg = None
    if want_epilogues:
        hm, am, reg = torch.empty(n, **f32), torch.empty(n, **f32), torch.empty(n, s, **f32)
    _l.call(dev, 'sunerf_dt_integral_fwd', _ptr(raw), _ptr(z_vals), _ptr(rays_o), _ptr(rays_d), _ptr(wavelengths), w,
            _ptr(table_logt), _ptr(table_resp), _ptr(log_abs), _ptr(vol_c), float(base_log_density),
            float(base_log_temperature), float(pixel_intensity_factor), float(reg_radius), n, s, _ptr(out['image']),
            _ptr(out['weights']), _ptr(out['reg_q']), _ptr(hm), _ptr(am), _ptr(reg), _stream(dev))
    if want_epilogues:
        out.update(height_map=hm, absorption_map=am, regularization=reg)
    return out


def simple_star_field(rays_o, rays_d, z_vals, rho_0: float, h0: float, T0: float, Rs: float, t_photosphere: float):
    """SimpleStar.forward (stellar_model.py:53-102) at the sample points of every ray -> raw (N, S, 2) = (ln rho, log10 T)."""
    lib = _l.load()
    n, s = z_vals.shape
    rays_o = _dev(rays_o, 'rays_o', (n, 3)); rays_d = _dev(rays_d, 'rays_d', (n, 3)); z_vals = _dev(z_vals, 'z_vals', (n, s))
    raw = torch.empty(n, s, 2, dtype=torch.float32, device=z_vals.device)
    _l.call(z_vals.device, 'sunerf_simple_star_field', _ptr(rays_o), _ptr(rays_d), _ptr(z_vals), n, s, float(rho_0),
            float(h0), float(T0), float(Rs), float(t_photosphere), _ptr(raw), _stream(z_vals.device))
    return raw


def dt_integral_bwd(raw, z_vals, rays_o, rays_d, wavelengths, table_logt, table_resp, log_abs, vol_c, base_log_density,
                    base_log_temperature, pixel_intensity_factor, reg_radius, g_image, g_reg):
    """-> (g_raw (N,S,2), g_log_abs (7,), g_vol_c (1,), absmax).  The two scalar-head gradients are adjacent views of one buffer
    (``g_log_abs.storage`` holds [7 channels, vol_c, absmax]): one clear in the entry point, one add into a flat gradient bucket."""
    lib = _l.load()
    n, s = z_vals.shape
    dev = z_vals.device
    w = wavelengths.shape[1]
    raw = _dev(raw, 'raw', (n, s, 2)); z_vals = _dev(z_vals, 'z_vals', (n, s))
    rays_o = _dev(rays_o, 'rays_o', (n, 3)); rays_d = _dev(rays_d, 'rays_d', (n, 3))
    wavelengths = _dev(wavelengths.to(torch.float32), 'wavelengths', (n, w))
    log_abs = _dev(log_abs.detach(), 'log_abs', (7,)); vol_c = _dev(vol_c.detach().reshape(1), 'vol_c', (1,))
    g_image = _dev(g_image, 'g_image', (n, w))
    if g_reg is not None:
        g_reg = _dev(g_reg, 'g_reg', (n, s))
    f32 = dict(dtype=torch.float32, device=dev)
    g_raw = torch.empty(n, s, 2, **f32)
    small = torch.empty(9, **f32)
    g_la, g_vc, absmax = small[:7], small[7:8], small[8:9].view(torch.int32)
    _l.call(dev, 'sunerf_dt_integral_bwd', _ptr(raw), _ptr(z_vals), _ptr(rays_o), _ptr(rays_d), _ptr(wavelengths), w,
            _ptr(table_logt), _ptr(table_resp), _ptr(log_abs), _ptr(vol_c), float(base_log_density),
            float(base_log_temperature), float(pixel_intensity_factor), float(reg_radius), n, s, _ptr(g_image),
            _ptr(g_reg), _ptr(g_raw), _ptr(g_la), _ptr(g_vc), _ptr(absmax), _stream(dev))
    return g_raw, g_la, g_vc, absmax
