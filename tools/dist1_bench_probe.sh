#!/bin/bash
# world_size=1 rehearsal of the multi-process bench path (NCCL init, sharded-Split code path is skipped at
# world 1, graph capture, JSON line) -- the 8-GPU run itself is the driver's.
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 \
    bench.py --gpus 1 --steps 20 --warmup 3 --no-extras
