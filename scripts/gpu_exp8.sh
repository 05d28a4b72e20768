#!/bin/bash
# multi-rank control flow of bench.py on a 1-GPU box (two ranks share cuda:0, gloo collectives)
export CSR5_BENCH_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 200 --warmup 20 2>&1 | tail -3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --steps 20 --warmup 2 --workload rmat20 2>&1 | tail -2
