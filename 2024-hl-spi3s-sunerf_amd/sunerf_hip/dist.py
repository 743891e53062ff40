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
    :func:`ranks_share_a_device`.  The device's uuid when the runtime reports a real one, else its PCI address, else the entry of
    the visibility list (``HIP_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES``) the index stands for -- never a value that would make
    DIFFERENT cards of one host look alike (that would silently put every rank on the slower two-kernel backward)."""
    import os
    import socket
    dev = torch.device(dev)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    props = torch.cuda.get_device_properties(index)
    ident = getattr(props, 'uuid', None)
    if ident is None or not str(ident).strip('0-'):      # (some ROCm builds report an all-zero uuid)
        pci = tuple(getattr(props, k, None) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id'))
        if all(v is not None for v in pci):
            ident = 'pci:%s' % (pci,)
        else:
            ident = 'visible:%d' % index
            for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
                listed = [x.strip() for x in os.environ.get(var, '').split(',') if x.strip()]
                if index < len(listed):
                    ident = 'visible:%s' % listed[index]
                    break
    return socket.gethostname(), str(ident)


def identity_words(identity) -> list:
    """Two int64 words of a hash of ``identity``: what the ranks exchange (a fixed-size tensor all-gather -- the plainest collective
    every backend has -- instead of pickled objects)."""
    import hashlib
    digest = hashlib.sha256('|'.join(identity).encode()).digest()
    return [int.from_bytes(digest[:8], 'little', signed=True), int.from_bytes(digest[8:16], 'little', signed=True)]


def any_shared(identities) -> bool:
    """True when two ranks of ``identities`` (one :func:`device_identity`, or its hash words, per rank) name the same GPU."""
    identities = [tuple(i) for i in identities]
    return len(set(identities)) < len(identities)


_shared_cache = {}


def ranks_share_a_device(dev, group=None) -> bool:
    """Collective (every rank of ``group`` must call it at the same point): do two ranks drive the same physical GPU?  The
    layer-pipelined backward needs all 256 of its workgroups resident at once, each filling a whole CU; ranks that share a card
    (CPU-style rehearsals of the multi-rank path on one GPU) would starve each other's launches into their time-outs, so they
    take the two-kernel backward.  One all-gather of two int64 words per device (on the device under RCCL, on the host under
    gloo), cached; called where every rank passes exactly once -- the construction of the optimiser / gradient bucket
    (``ClipAdam``, ``GradBucket``)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return False
    key = str(torch.device(dev))
    if key not in _shared_cache:
        where = torch.device(dev) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
        mine = torch.tensor(identity_words(device_identity(dev)), dtype=torch.int64, device=where)
        everyone = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(everyone, mine, group=group)
        _shared_cache[key] = any_shared(t.tolist() for t in everyone)
    return _shared_cache[key]


def shared_device_known(dev) -> bool:
    """The cached answer of :func:`ranks_share_a_device` for ``dev`` (False when nobody has asked): what the backward consults --
    it must not start a collective of its own, since ranks may take different backward paths for batches of different sizes."""
    return _shared_cache.get(str(torch.device(dev)), False)

