#!/bin/bash
# multi-rank control flow of bench.py on a 1-GPU box (two ranks share cuda:0, gloo collectives)
export CSR5_BENCH_SHARE_GPU=1
run() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 "${@:2}" 2>&1 | grep '"metric"' | python scripts/benchline.py; }
run 29555 --steps 200 --warmup 20
run 29556 --steps 20 --warmup 2 --workload rmat20
run 29557 --steps 20 --warmup 2 --workload rmat20 --scaling strong
run 29558 --steps 100 --warmup 5 --workload webbase --scaling strong
