"""Output side of the render path: fused training loss and clip + Adam step, no host synchronisation.

``training_loss``   the loss section of the reference's ``training_step`` (sunerf/model/sunerf.py:105-125, :185-200) as ONE
                    kernel behind an autograd node: finite check, asinh scaling, 2 x MSE, regularization mean, PSNR.
``ClipAdam``        ``clip_grad_norm_(params, max_norm)`` + ``torch.optim.Adam.step()`` (+ the 1 / world averaging of the
                    all-reduced gradients) as two kernels over ONE flat parameter / gradient / moment buffer.
"""
import ctypes
import os
import weakref
from typing import Dict, Iterable, Optional, Sequence

import torch
import torch.distributed as dist
from torch.utils.weak import WeakIdKeyDictionary

from . import lib as _l
from .ops import _dev, _ptr, _stream

STAT_KEYS = ('loss', 'coarse', 'fine', 'regularization', 'psnr', 'non_finite')
_ws = {}      # (device, stream) -> zero-initialised reduction workspace

# Which flat bucket a parameter's gradient lives in: parameter -> (weak reference to the owning ClipAdam, offset, numel).  Kept
# HERE and not as an attribute of the nn.Parameter: torch pickles Parameter attributes, so a tag on the parameter would put the
# optimiser (its class name and both moment buffers) into every .snf written after configure_optimizers() -- a file the
# reference environment could then no longer unpickle -- and into every copy.deepcopy of the module.
_BUCKETS = WeakIdKeyDictionary()      # keyed by identity: Tensor.__eq__ is element-wise


def bucket_of(param):
    """(owner ClipAdam, offset, numel) of a parameter whose gradient is a view of a live optimiser's flat bucket, else None."""
    tag = _BUCKETS.get(param)
    if tag is None:
        return None
    owner = tag[0]()
    if owner is None:
        return None
    return owner, tag[1], tag[2]


def env_flag(name: str) -> bool:
    """An environment switch: unset, '', '0', 'false', 'no', 'off' are OFF."""
    return os.environ.get(name, '').strip().lower() not in ('', '0', 'false', 'no', 'off')


def remaining_slices(done, total):
    """[lo, hi) pieces of ``[0, total)`` not covered by the (possibly unordered) intervals ``done`` -- what ``ClipAdam.step`` still
    has to all-reduce after ``segment_ready`` sent some slices of the bucket early."""
    out, pos = [], 0
    for lo, hi in sorted(done) + [(total, total)]:
        if lo > pos:
            out.append((pos, min(lo, total)))
        pos = max(pos, hi)
    return out


def _workspace(dev) -> torch.Tensor:
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _ws.get(key)
    if ws is None:
        ws = _ws[key] = torch.zeros(_l.load().sunerf_train_workspace_bytes(), dtype=torch.uint8, device=dev)
    return ws


class _TrainingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coarse, fine, target, reg, cfg):
        lib = _l.load()
        dev = coarse.device
        coarse_c, fine_c = _dev(coarse, 'coarse_image'), _dev(fine, 'fine_image')
        target_c = _dev(target, 'target_image', coarse.shape)
        if fine.shape != coarse.shape:
            raise ValueError('coarse_image and fine_image differ in shape')
        n = coarse_c.numel()
        reg_c = _dev(reg, 'regularization') if reg is not None else None
        n_reg = reg_c.numel() if reg_c is not None else 0
        extras = [_dev(t.detach(), 'finite-check tensor') for t in cfg['finite_check']]
        ptrs = (ctypes.c_void_p * max(1, len(extras)))(*[t.data_ptr() for t in extras])
        sizes = (ctypes.c_int64 * max(1, len(extras)))(*[t.numel() for t in extras])
        g_coarse, g_fine = torch.empty_like(coarse_c), torch.empty_like(fine_c)
        stats = torch.empty(8, dtype=torch.float32, device=dev)
        ws = _workspace(dev)
        scaling = cfg['scaling']
        _l.call(dev, 'sunerf_training_loss', _ptr(coarse_c), _ptr(fine_c), _ptr(target_c), n, _ptr(reg_c), n_reg,
                ptrs, sizes, len(extras), 1 if scaling else 0, scaling[0] if scaling else 1.0, scaling[1] if scaling
                else 1.0, cfg['lambda_image'], cfg['lambda_regularization'], _ptr(g_coarse), _ptr(g_fine),
                _ptr(stats), _ptr(ws), ws.numel(), _stream(dev))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(g_coarse, g_fine)
        ctx.reg_shape = None if reg is None else tuple(reg.shape)
        ctx.reg_grad = cfg['lambda_regularization'] / n_reg if n_reg else 0.0
        ctx.mark_non_differentiable(stats)
        return stats[0], stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        if g_loss is None:
            return None, None, None, None, None
        g_coarse, g_fine = ctx.saved_tensors
        g_reg = None
        if ctx.reg_shape is not None and ctx.needs_input_grad[3]:
            g_reg = (g_loss * ctx.reg_grad).expand(ctx.reg_shape)
        return g_coarse * g_loss, g_fine * g_loss, None, g_reg, None


def training_loss(coarse_image: torch.Tensor, fine_image: torch.Tensor, target_image: torch.Tensor,
                  regularization: Optional[torch.Tensor], lambda_image: float = 1.0, lambda_regularization: float = 1.0,
                  asinh_scaling: Optional[Sequence[float]] = None, finite_check: Iterable[torch.Tensor] = ()):
    """``lambda_image * (MSE(s(coarse), s(target)) + MSE(s(fine), s(target))) + lambda_regularization * mean(reg)``
    with ``s`` = ``ImageAsinhScaling(vmax, a)`` when ``asinh_scaling = (vmax, a)`` is given, identity otherwise.

    Returns ``(loss, stats)``: ``loss`` is a differentiable 0-dim tensor; ``stats`` an 8-float device tensor
    (see ``STAT_KEYS``; ``stats[5]`` counts NaN / Inf values among the images, the regularization and ``finite_check``)."""
    cfg = {'lambda_image': float(lambda_image), 'lambda_regularization': float(lambda_regularization),
           'scaling': None if asinh_scaling is None else (float(asinh_scaling[0]), float(asinh_scaling[1])),
           'finite_check': list(finite_check)}
    if len(cfg['finite_check']) > 8:
        raise ValueError('at most 8 extra tensors can take part in the finite check')
    return _TrainingLoss.apply(coarse_image, fine_image, target_image, regularization, cfg)


def stats_dict(stats: torch.Tensor) -> Dict[str, torch.Tensor]:
    return {k: stats[i] for i, k in enumerate(STAT_KEYS)}


class ClipAdam(torch.optim.Optimizer):
    """Adam (torch.optim.Adam semantics: no weight decay, no amsgrad) with the gradient clip fused in, on flat buffers.

    On construction all parameters are moved into one flat fp32 buffer (``p.data`` become views of it), ``p.grad`` into a
    second one (so backward kernels write straight into the buffer that is all-reduced), and the two moment buffers are
    flat as well.  ``step()`` = optional sum all-reduce over ``group`` -> ||g / world|| -> clip -> Adam, two kernels, and
    no ``.item()``: the learning rate is a host scalar, the norm, the step counter and the skip decision stay on the device.

    Rank lock-step (the reference asserts on NaN / Inf, sunerf.py:105-107): the gradient bucket has ONE extra
    element at its tail that carries this rank's non-finite-output count of the step (``skip_if_positive``).  The same
    all-reduce that sums the gradients sums it, so every rank sees the global count and takes the same decision; a
    non-finite gradient norm (computed after the reduce) skips as well.  A skipped step leaves parameters, moments AND the
    bias-correction step counter untouched.

    ``param_groups`` carries ``lr`` / ``betas`` / ``eps`` like torch's Adam, so ``ExponentialLR`` (sunerf.py:32) works
    unchanged.  ``max_norm=None`` leaves clipping to the caller (e.g. Lightning's ``gradient_clip_val``)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm: Optional[float] = None, group=None,
                 overlap: bool = False):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise ValueError('ClipAdam keeps one flat buffer: a single parameter group')
        self.max_norm = max_norm
        self.group = group
        if group is not None:
            from . import ops as _ops
            _ops.probe_group = group        # the AUTO probe's MAX all-reduce runs over the same ranks as the gradient exchange
        self.reduce_single_rank = False     # tests: issue the collective even in a one-rank group (exercises RCCL on one GPU)
        # overlap: the backward of the fine model finishes before the coarse model's starts (the two graphs
        # are independent, sampling.py:120), so its slice of the bucket can be all-reduced -- asynchronously, on the
        # collective's own stream -- while the coarse backward kernels still run.  Needs exactly ONE backward per model and
        # step (no gradient accumulation over micro-batches): opt-in.
        self.overlap = overlap
        self._early = []                    # [(lo, hi, work)] slices already handed to the collective this step
        self._params = [p for p in self.param_groups[0]['params'] if p.requires_grad]
        if not self._params:
            raise ValueError('no trainable parameters')
        dev = self._params[0].device
        if dev.type != 'cuda' or any(p.device != dev or p.dtype != torch.float32 for p in self._params):
            raise _l.SunerfHipError('ClipAdam needs float32 parameters on one ROCm device')
        from . import dist as _sdist
        _sdist.ranks_share_a_device(dev, group)      # (collective, once per device: ranks sharing a card take the two-kernel backward)
        n = sum(p.numel() for p in self._params)
        self.n_params = n
        self.flat_params = torch.empty(n, dtype=torch.float32, device=dev)
        self.bucket = torch.zeros(n + 1, dtype=torch.float32, device=dev)     # gradients + the non-finite count
        self.flat_grads = self.bucket[:n]
        self.nonfinite = self.bucket[n:]
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(4, dtype=torch.float32, device=dev)      # total gradient norm, clip coefficient, skipped, 0
        self.applied_steps = torch.zeros(1, dtype=torch.int64, device=dev)
        self._grad_views = []
        off = 0
        for p in self._params:
            k = p.numel()
            view = self.flat_params[off:off + k].view_as(p)
            view.copy_(p.data)
            p.data = view
            gv = self.flat_grads[off:off + k].view_as(p)
            p.grad = gv
            self._grad_views.append(gv)
            self.state[p] = {'step': torch.tensor(0.), 'exp_avg': self.exp_avg[off:off + k].view_as(p),
                             'exp_avg_sq': self.exp_avg_sq[off:off + k].view_as(p)}
            _BUCKETS[p] = (weakref.ref(self), off, k)      # lets the backward node announce a finished slice (segment_ready)
            off += k

    @property
    def step_count(self) -> int:
        """Number of updates actually applied (one device -> host read; not used inside the step)."""
        return int(self.applied_steps.item())

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes the flat gradient bucket and re-attaches the views (``set_to_none`` is ignored: the backward kernels
        and the all-reduce work in place on the bucket)."""
        self.bucket.zero_()
        self._early = []
        for p, gv in zip(self._params, self._grad_views):
            p.grad = gv

    def _world(self) -> int:
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(self.group)
            if world > 1 or self.reduce_single_rank:
                return world
        return 0

    def segment_ready(self, params) -> None:
        """Called by a backward node whose kernels have accumulated the FINAL gradients of ``params`` into the bucket: starts
        the all-reduce of that slice right away (``overlap=True`` and a process group; no-op otherwise).  The collective is
        ordered behind the gradient kernels on the current stream and runs beside whatever is launched next."""
        if not self.overlap or not self._world() or env_flag('SUNERF_NO_OVERLAP'):      # (the env switch: rehearsals / A-B runs)
            return
        tags = [bucket_of(p) for p in params]
        if any(t is None or t[0] is not self for t in tags):
            return        # not (all) this optimiser's parameters: step() reduces them
        for p, (_, off, k) in zip(params, tags):
            # the kernels must have written into the bucket itself: a .grad that autograd (or the caller) replaced is copied
            # into the bucket by step() -- AFTER an early all-reduce of this slice it would overwrite the reduced values
            if p.grad is None or p.grad.data_ptr() != self.flat_grads[off:off + k].data_ptr():
                return
        spans = sorted(t[1:] for t in tags)
        lo, hi = spans[0][0], spans[-1][0] + spans[-1][1]
        if sum(k for _, k in spans) != hi - lo:
            return        # not one contiguous slice of the bucket: leave it to step()
        if any(a < hi and lo < b for a, b, _ in self._early):
            # a SECOND backward contribution to a slice that has already been all-reduced (gradient accumulation over
            # micro-batches, a model used twice in one step): step() only reduces what was not sent early, so the replicas
            # would silently diverge.  overlap=True promises exactly one backward per model and step.
            raise RuntimeError('ClipAdam(overlap=True): a slice of the gradient bucket received a second backward contribution '
                               'after its all-reduce had started; use overlap=False with gradient accumulation / shared models')
        self._early.append((lo, hi, dist.all_reduce(self.bucket[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def _collect(self):
        for p, gv in zip(self._params, self._grad_views):   # autograd may have put a fresh tensor into .grad
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
            p.grad = gv

    @torch.no_grad()
    def step(self, closure=None, skip_if_positive: Optional[torch.Tensor] = None):
        loss = None
        if closure is not None:                 # torch.optim convention (Lightning's automatic optimisation passes the
            with torch.enable_grad():           # forward + backward of the step as a closure)
                loss = closure()
        self._collect()
        if skip_if_positive is not None:
            self.nonfinite.copy_(skip_if_positive.reshape(1))
        else:
            self.nonfinite.zero_()
        world = self._world()
        if world:
            # what segment_ready has not already sent: the remaining slices of the bucket, the last one with the count
            for lo, hi in remaining_slices([(lo, hi) for lo, hi, _ in self._early], self.n_params + 1):
                dist.all_reduce(self.bucket[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            for _, _, work in self._early:
                work.wait()                 # orders the current stream behind the early collectives
            self._early = []
        world = max(world, 1)
        g = self.param_groups[0]
        dev = self.flat_params.device
        ws = _workspace(dev)
        max_norm = float(self.max_norm) if self.max_norm else 0.0
        _l.call(dev, 'sunerf_clip_adam_step', _ptr(self.flat_params), _ptr(self.flat_grads), _ptr(self.exp_avg),
                _ptr(self.exp_avg_sq), self.n_params, float(g['lr']), float(g['betas'][0]), float(g['betas'][1]),
                float(g['eps']), max_norm, 1.0 / world, 0, _ptr(self.nonfinite), _ptr(self.norm), _ptr(ws), ws.numel(),
                _ptr(self.applied_steps), _stream(dev))
        # the kernels wrote through raw pointers: tell autograd / the packed-weights cache that the values changed
        torch.autograd.graph.increment_version(self._params)
        return loss

    def skipped_last_step(self) -> bool:
        """True when the last ``step()`` was skipped on all ranks (one 4-byte read)."""
        return bool(self.norm[2].item() > 0)

    def state_dict(self):
        steps = float(self.step_count)
        for p in self._params:
            self.state[p]['step'] = torch.tensor(steps)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Accepts a ``torch.optim.Adam`` state dict (reference checkpoints): moments are copied into the flat buffers."""
        groups = state_dict['param_groups']
        ids = groups[0]['params']
        for k in ('lr', 'betas', 'eps'):
            if k in groups[0]:
                self.param_groups[0][k] = groups[0][k]
        all_params = self.param_groups[0]['params']
        steps = []
        for pid, p in zip(ids, all_params):
            st = state_dict['state'].get(pid)
            if st is None or p not in self.state:
                continue
            self.state[p]['exp_avg'].copy_(st['exp_avg'])
            self.state[p]['exp_avg_sq'].copy_(st['exp_avg_sq'])
            self.state[p]['step'] = torch.as_tensor(float(st['step']))
            steps.append(int(float(st['step'])))
        if steps:
            if len(set(steps)) != 1:
                raise ValueError('parameters with different step counts cannot share the fused bias correction')
            self.applied_steps.fill_(steps[0])
