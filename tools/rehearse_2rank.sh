export SUNERF_DIST_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8192 --res 256 > gpurun_out/rehearse2.log 2>&1
tail -2 gpurun_out/rehearse2.log | cut -c1-900
