"""Batch feed of the render path: the reference's on-disk training set as a device-resident ray pool.

The reference writes pre-shuffled arrays to ``working_dir`` -- ``rays_batches.npy (P,2,3)``, ``times_batches.npy (P,1)``,
``images_batches.npy (P,1 | W)`` and, for the multi-channel loader, ``wavelengths`` ``(P,W)``
(``sunerf/data/loader/single_channel.py:44-80``, ``multi_thermal_loader.py:59-94``) -- and feeds them through
``MmapDataset`` + ``DataLoader(batch_size=None, shuffle=True, num_workers=os.cpu_count())``
(``sunerf/data/dataset.py:7-30``, ``base_loader.py:41-56``): every step copies one batch from the page cache through a
worker process and pinned memory to the GPU.  At 10^8 ray-samples/s a GPU consumes ~10^6 rays/s (32 B each); the whole
training set of a run (10^7 - 10^9 rays) is 0.3 - 30 GB and fits the 288 GB of HBM many times over.  So: upload once,
shard by rank, and hand out batches as zero-copy views.

Semantics kept: a batch is a contiguous block of ``batch_size`` rays of the (pre-shuffled) file; an epoch visits every
batch once, in a fresh random order (what ``shuffle=True`` over a ``batch_size=None`` dataset does); the last batch may be
short.  Multi-GPU: rank r owns the contiguous shard ``shard_range(P, r, world)`` of the file and every rank
draws ``batch_size`` rays per step from its own shard, so the global batch is ``batch_size * world`` rays as under the
reference's ``dp`` (single_channel.py:67-68).
"""
from typing import Dict, Iterator, Mapping, Optional

import numpy as np
import torch

from .dist import shard_range

_UPLOAD_CHUNK = 1 << 26      # bytes per staged host -> device copy


def _upload(array: np.ndarray, begin: int, end: int, device) -> torch.Tensor:
    """Rows [begin, end) of a (possibly memory-mapped) array -> one contiguous float32 device tensor, streamed in chunks
    so that the host never holds more than one chunk of the file."""
    shape = (end - begin,) + tuple(array.shape[1:])
    out = torch.empty(shape, dtype=torch.float32, device=device)
    row_bytes = max(1, int(np.prod(shape[1:], dtype=np.int64)) * 4)
    rows = max(1, _UPLOAD_CHUNK // row_bytes)
    pin = torch.device(device).type == 'cuda'
    for b in range(begin, end, rows):
        e = min(end, b + rows)
        chunk = torch.from_numpy(np.array(array[b:e], dtype=np.float32))     # copy: memmaps are read-only
        if pin:
            chunk = chunk.pin_memory()
        out[b - begin:e - begin].copy_(chunk, non_blocking=pin)
    if pin:
        torch.cuda.current_stream(device).synchronize()     # the pinned chunks may be released now
    return out


class RayPool:
    """Device-resident shard of a pre-shuffled ray set; iterating yields ``{key: view}`` batches for one epoch.

    ``arrays``: mapping name -> array of shape ``(P, ...)`` (numpy arrays or ``np.load(..., mmap_mode='r')`` memmaps), all
    with the same ``P``.  The reference's names are ``rays`` ``(P,2,3)``, ``time`` ``(P,1)``, ``target_image`` ``(P,C)``
    and ``wavelength`` ``(P,W)``."""

    def __init__(self, arrays: Mapping[str, np.ndarray], batch_size: int = 2 ** 13, rank: int = 0, world: int = 1,
                 device='cuda', shuffle: bool = True, seed: int = 0, drop_last: bool = False):
        sizes = {k: v.shape[0] for k, v in arrays.items()}
        if len(set(sizes.values())) != 1:
            raise ValueError(f'arrays differ in their number of rays: {sizes}')
        self.total_rays = next(iter(sizes.values()))
        self.begin, self.end = shard_range(self.total_rays, rank, world)
        self.batch_size, self.shuffle, self.seed, self.drop_last = int(batch_size), shuffle, seed, drop_last
        self.rank, self.world = rank, world
        self.data: Dict[str, torch.Tensor] = {k: _upload(v, self.begin, self.end, device) for k, v in arrays.items()}
        self.epoch = 0

    @classmethod
    def from_files(cls, paths: Mapping[str, str], **kwargs) -> 'RayPool':
        """``paths`` as handed to the reference's ``MmapDataset`` (dataset.py:9-15): name -> ``.npy`` file."""
        return cls({k: np.load(p, mmap_mode='r') for k, p in paths.items()}, **kwargs)

    @classmethod
    def from_dataset(cls, dataset, **kwargs) -> 'RayPool':
        """From a reference ``MmapDataset`` (``batches_file_paths``) or ``ArrayDataset`` (``array_dict``); the dataset's
        own ``batch_size`` is the default."""
        kwargs.setdefault('batch_size', getattr(dataset, 'batch_size', 2 ** 13))
        if hasattr(dataset, 'batches_file_paths'):
            return cls.from_files(dataset.batches_file_paths, **kwargs)
        return cls(dataset.array_dict, **kwargs)

    @property
    def n_rays(self) -> int:
        return self.end - self.begin

    def __len__(self) -> int:
        """Batches per epoch (dataset.py:17-20: ceil(P / batch_size), over this rank's shard)."""
        if self.drop_last:
            return self.n_rays // self.batch_size
        return -(-self.n_rays // self.batch_size)

    def batch(self, index: int) -> Dict[str, torch.Tensor]:
        """Batch ``index`` of the shard: views, no copy (dataset.py:22-26 copies; consumers here only read)."""
        b = index * self.batch_size
        return {k: v[b:b + self.batch_size] for k, v in self.data.items()}

    def order(self, epoch: Optional[int] = None) -> np.ndarray:
        n = len(self)
        if not self.shuffle:
            return np.arange(n)
        epoch = self.epoch if epoch is None else epoch
        return np.random.default_rng([self.seed, self.rank, epoch]).permutation(n)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        order = self.order()
        self.epoch += 1
        for i in order:
            yield self.batch(int(i))


def training_batches(pool: RayPool, steps: int) -> Iterator[Dict[str, Dict[str, torch.Tensor]]]:
    """``steps`` batches in the structure the Lightning modules' ``training_step`` reads (``batch['tracing'][...]``,
    sunerf.py:99; base_loader.py:41-56 wraps the loaders in a dict keyed by dataset name), cycling over epochs."""
    done = 0
    while done < steps:
        for b in pool:
            if done >= steps:
                return
            yield {'tracing': b}
            done += 1
