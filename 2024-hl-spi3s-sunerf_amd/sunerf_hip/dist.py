"""Multi-GPU data parallelism for the renderer: one process per GPU, rays sharded, one all-reduce per step.

The reference's only parallel strategy is single-process ``nn.DataParallel`` (run_emission.py:64-69): scatter the
batch over GPUs, replicate the module, reduce gradients onto GPU 0.  The MI355X-native equivalent:
every rank owns a contiguous block of rays, parameters and optimiser state are replicated, and the ONLY exchange is a
sum all-reduce (RCCL over xGMI; gloo in the CPU tests) of one flat fp32 gradient bucket per step -- 3.86 MB at
d_filter = 256, far below the per-link bandwidth-delay product, so it is a latency, not a bandwidth, item.
"""
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) block of ``n_items`` for ``rank`` (first ``n_items % world`` ranks get one more)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class GradBucket:
    """Flat fp32 view over the gradients of ``params`` so that a step needs exactly one collective.

    ``p.grad`` of every parameter is re-pointed into the bucket (like DDP's gradient-as-bucket-view), so backward
    kernels that accumulate into ``p.grad`` write straight into the buffer that is all-reduced."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        if dev.type == 'cuda':
            ranks_share_a_device(dev)          # (collective, once: decides pipelined vs two-kernel backward for shared cards)
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):   # optimizers / autograd may have replaced .grad
            p.grad = v

    def gather_grads(self):
        """Copies gradients that autograd allocated elsewhere back into the bucket (no-op for bucket views)."""
        for p, v in zip(self.params, self.views):
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v

    def all_reduce_mean(self, group=None):
        """Sum over ranks, then 1/world: the loss of every rank is a mean over its own equally sized ray block, so the
        global-batch gradient of the reference's ``dp`` semantics (mean over B*N_GPUS rays) is the rank average."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.mul_(1.0 / dist.get_world_size(group))
        return self.flat

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """``torch.nn.utils.clip_grad_norm_`` on the (already reduced) bucket: global L2 norm, scale if above."""
        total = torch.linalg.vector_norm(self.flat)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.flat.mul_(coef)
        return total


def device_identity(dev) -> tuple:
    """(host, what identifies the physical GPU behind ``dev`` in this process) -- compared across ranks by
    :func:`ranks_share_a_device`."""
    import socket
    props = torch.cuda.get_device_properties(dev)
    ident = getattr(props, 'uuid', None)
    if ident is None or not str(ident).strip('0-'):      # (some ROCm builds report an all-zero uuid)
        ident = tuple(getattr(props, k, None) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id'))
    return socket.gethostname(), str(ident)


def any_shared(identities) -> bool:
    """True when two ranks of ``identities`` (one :func:`device_identity` per rank) name the same GPU."""
    identities = list(identities)
    return len(set(identities)) < len(identities)


_shared_cache = {}


def ranks_share_a_device(dev, group=None) -> bool:
    """Collective (every rank of ``group`` must call it at the same point): do two ranks drive the same physical GPU?  The
    layer-pipelined backward needs all 256 of its workgroups resident at once, each filling a whole CU; ranks that share a card
    (CPU-style rehearsals of the multi-rank path on one GPU) would starve each other's launches into their time-outs, so they
    take the two-kernel backward.  One all_gather_object per device, cached; called where every rank passes exactly once --
    the construction of the optimiser / gradient bucket (``ClipAdam``, ``GradBucket``)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return False
    key = str(torch.device(dev))
    if key not in _shared_cache:
        mine = device_identity(dev)
        everyone = [None] * dist.get_world_size(group)
        dist.all_gather_object(everyone, mine, group=group)
        _shared_cache[key] = any_shared(everyone)
    return _shared_cache[key]


def shared_device_known(dev) -> bool:
    """The cached answer of :func:`ranks_share_a_device` for ``dev`` (False when nobody has asked): what the backward consults --
    it must not start a collective of its own, since ranks may take different backward paths for batches of different sizes."""
    return _shared_cache.get(str(torch.device(dev)), False)

