"""Do concurrent asynchronous gloo all-reduces on CUDA tensors stall when several ranks share one GPU?  (The N-rank gloo REHEARSAL of
bench.py on a one-GPU box showed 6.3 s per step at 4 ranks in the lines that use ClipAdam(overlap=True), none at 2 or 3 ranks and none
with CPU tensors.)   python -m torch.distributed.run --nproc-per-node N tools/probes/gloo_cuda_concurrent.py"""
import os, time, torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
n = 483331
bucket = torch.zeros(2 * n + 1, device='cuda')
a = torch.randn(4096, 4096, device='cuda')


def busy():            # ~ several ms of GPU work queued behind the asynchronous collectives, like the coarse model's backward
    x = a
    for _ in range(6):
        x = x @ a * 1e-3
    return x


for mode in ('sequential', 'concurrent', 'concurrent+kernels'):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode.startswith('concurrent'):
            w1 = dist.all_reduce(bucket[n:2 * n], async_op=True)
            if mode.endswith('kernels'):
                busy()
            w2 = dist.all_reduce(bucket[0:n], async_op=True)
            if mode.endswith('kernels'):
                busy()
            dist.all_reduce(bucket[2 * n:2 * n + 1])
            w1.wait(); w2.wait()
        else:
            dist.all_reduce(bucket[n:2 * n]); dist.all_reduce(bucket[0:n]); dist.all_reduce(bucket[2 * n:2 * n + 1])
        torch.cuda.synchronize()
        if rank == 0:
            print(f'world {world} {mode} iter {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True)
dist.destroy_process_group()
