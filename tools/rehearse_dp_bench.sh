#!/bin/bash
# Rehearsal of `bench.py --gpus 2` on a ONE-GPU box: two ranks share device 0, gloo carries the buckets.
# (The driver's real run uses torch.distributed.run with one rank per GPU over RCCL.)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 WORLD_SIZE=2 ASR_DIST_BACKEND=gloo LOCAL_RANK=0
RANK=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/rank1.log 2>&1 &
P1=$!
RANK=0 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/tmp/rank0.err | tail -1 | cut -c1-400
wait $P1; echo "rank1 exit $?"; tail -2 /tmp/rank1.log | cut -c1-200
