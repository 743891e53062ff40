"""Batch feed: the reference's on-disk ray files as a sharded, resident ray pool.  Host logic only."""
import numpy as np
import torch

from sunerf_hip.feed import RayPool, training_batches


def _files(tmp_path, p=1000, channels=1):
    rng = np.random.default_rng(0)
    arrays = {'rays': rng.normal(size=(p, 2, 3)).astype(np.float32), 'time': rng.random((p, 1), dtype=np.float32),
              'target_image': rng.random((p, channels), dtype=np.float32)}
    names = {'rays': 'rays_batches.npy', 'time': 'times_batches.npy', 'target_image': 'images_batches.npy'}
    paths = {}
    for k, v in arrays.items():
        paths[k] = str(tmp_path / names[k])
        np.save(paths[k], v)
    return arrays, paths


def test_pool_batches_cover_the_file_once_per_epoch(tmp_path):
    arrays, paths = _files(tmp_path, p=1000)
    pool = RayPool.from_files(paths, batch_size=128, device='cpu', seed=3)
    assert len(pool) == 8 and pool.n_rays == 1000
    seen = []
    for b in pool:
        assert set(b) == {'rays', 'time', 'target_image'} and b['rays'].shape[1:] == (2, 3)
        # a batch is a contiguous block of the file, like MmapDataset.__getitem__ (dataset.py:22-26)
        start = int(np.flatnonzero((arrays['time'][:, 0] == b['time'][0, 0].item()))[0])
        assert start % 128 == 0
        n = b['time'].shape[0]
        assert np.array_equal(b['rays'].numpy(), arrays['rays'][start:start + n])
        assert np.array_equal(b['target_image'].numpy(), arrays['target_image'][start:start + n])
        seen.append(start // 128)
    assert sorted(seen) == list(range(8)) and seen != list(range(8))          # every batch once, shuffled order
    assert [b['time'].shape[0] for b in [pool.batch(7)]] == [1000 - 7 * 128]   # short last batch
    second = [int(np.flatnonzero(arrays['time'][:, 0] == b['time'][0, 0].item())[0]) // 128 for b in pool]
    assert sorted(second) == list(range(8)) and second != seen                # a new order every epoch
    again = RayPool.from_files(paths, batch_size=128, device='cpu', seed=3)
    assert list(again.order(0)) == seen                                      # reproducible from (seed, rank, epoch)


def test_pool_shards_are_disjoint_and_complete(tmp_path):
    arrays, paths = _files(tmp_path, p=1003, channels=7)
    pools = [RayPool.from_files(paths, batch_size=100, rank=r, world=4, device='cpu', shuffle=False) for r in range(4)]
    assert [p.n_rays for p in pools] == [251, 251, 251, 250]
    cat = np.concatenate([np.concatenate([b['target_image'].numpy() for b in p]) for p in pools])
    assert np.array_equal(cat, arrays['target_image'])
    drop = RayPool.from_files(paths, batch_size=100, rank=3, world=4, device='cpu', drop_last=True)
    assert len(drop) == 2 and all(b['time'].shape[0] == 100 for b in drop)


def test_training_batches_cycle_epochs_and_match_module_structure(tmp_path):
    arrays, _ = _files(tmp_path, p=300)

    class _ArrayDataset:          # the reference's ArrayDataset surface (dataset.py:33-52)
        array_dict = arrays
        batch_size = 64
    pool = RayPool.from_dataset(_ArrayDataset(), device='cpu')
    assert pool.batch_size == 64 and len(pool) == 5
    got = list(training_batches(pool, 12))
    assert len(got) == 12 and pool.epoch == 3
    assert all(set(b) == {'tracing'} and b['tracing']['rays'].dtype == torch.float32 for b in got)
    # views into the resident shard, not copies
    assert got[0]['tracing']['rays'].untyped_storage().data_ptr() == pool.data['rays'].untyped_storage().data_ptr()
