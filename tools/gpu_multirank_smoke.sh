#!/bin/bash
# N>1 control flow of bench.py on a 1-GPU box: ranks share cuda:0, gloo instead of RCCL.
export TMPDIR=/tmp CPH_BENCH_SHARE_GPU=1
for N in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N \
    bench.py --gpus $N --steps 2 --warmup 1 --rows 20000000 --customers 1000000 --products 10000 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-900
done
