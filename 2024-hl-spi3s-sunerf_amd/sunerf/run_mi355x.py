"""Multi-GPU entry for the reference's UNCHANGED run scripts on an MI355X node: one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m sunerf.run_mi355x /path/to/run_emission.py --config config.yaml

``run_emission.py:64-69`` / ``run_density_temperature.py:80-85`` choose ``devices = torch.cuda.device_count()`` and
``strategy='dp'`` (``nn.DataParallel``: one process, module replicas per step) whenever more than one GPU is visible.  The fused
renderer refuses such replicas (``sunerf.rendering.base_tracing.refuse_data_parallel``); its data parallelism is one process
per GPU with ONE RCCL all-reduce of the flat gradient bucket per step (``sunerf_hip.train.ClipAdam``).  This
wrapper makes the script take its own single-GPU branch in every rank, without editing it:

1. before anything touches the GPU, the rank is restricted to ITS device (``HIP_VISIBLE_DEVICES`` = entry ``LOCAL_RANK`` of
   the devices visible to the launcher), so ``torch.cuda.device_count()`` is 1 and the script builds ``Trainer(devices=1,
   strategy=None)``;
2. the default process group is initialised (``nccl`` = RCCL on ROCm; ``SUNERF_DIST_BACKEND=gloo`` for CPU rehearsals):
   ``ClipAdam.step`` -- the optimiser ``configure_optimizers`` returns -- then sums the gradient bucket (and the non-finite
   count at its tail) over the ranks before clipping, every rank applies the same update;
3. the random generators are seeded per rank (``SUNERF_SEED``, default 0, + rank), so the ranks draw DIFFERENT ray batches from
   the data module's shuffled loader: N times the rays per optimiser step, the reference's ``dp`` semantics;
4. the script runs as ``__main__`` with the remaining arguments.

Rank 0 is the only rank that should log and write checkpoints; Lightning's ``rank_zero_only`` reads ``RANK`` from the
environment ``torch.distributed.run`` sets.  (Lightning is not installed in the build image: the wrapper is exercised there with a
stand-in script over gloo, tests/test_run_wrapper.py; the training semantics it relies on are those of ``fit_steps`` /
``ClipAdam``, tests/test_dist_gloo.py and tests/test_gpu_dist.py.)
"""
import os
import runpy
import sys


def visible_device_for(local_rank: int, environ=None):
    """(variable, value) that restricts this rank to its GPU: entry ``local_rank`` of the launcher's own visibility list when one
    is set (``HIP_VISIBLE_DEVICES``, else ``CUDA_VISIBLE_DEVICES``), else the index itself."""
    environ = os.environ if environ is None else environ
    for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        listed = [x.strip() for x in environ.get(var, '').split(',') if x.strip()]
        if listed:
            if local_rank >= len(listed):
                raise RuntimeError(f'LOCAL_RANK={local_rank} but {var}={environ[var]!r} lists only {len(listed)} device(s): '
                                   'start at most one rank per visible GPU')
            return var, listed[local_rank]
    return 'HIP_VISIBLE_DEVICES', str(local_rank)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 2
    script, rest = argv[0], argv[1:]
    if not os.path.isfile(script):
        raise SystemExit(f'sunerf.run_mi355x: {script!r} is not a file (pass the path of run_emission.py / '
                         'run_density_temperature.py, then its own arguments)')
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    backend = os.environ.get('SUNERF_DIST_BACKEND', 'nccl')
    if backend == 'nccl':
        # (the gloo rehearsal leaves the visibility alone: its ranks may share one card, or have none)
        var, value = visible_device_for(local_rank)
        os.environ[var] = value
        if var != 'HIP_VISIBLE_DEVICES':
            os.environ.pop('HIP_VISIBLE_DEVICES', None)
    import torch          # after the visibility is fixed: nothing has initialised the GPU yet
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.cuda.set_device(0)
            dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
        else:
            dist.init_process_group(backend)
    torch.manual_seed(int(os.environ.get('SUNERF_SEED', 0)) + rank)
    try:
        import numpy as np
        np.random.seed((int(os.environ.get('SUNERF_SEED', 0)) + rank) % (2 ** 32))
    except ImportError:      # pragma: no cover
        pass
    sys.argv = [script] + rest
    try:
        runpy.run_path(script, run_name='__main__')
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
