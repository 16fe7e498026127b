#!/bin/bash
# two bench ranks on the one GPU of a test box: RCCL refuses two ranks on one device, so this exercises the agreed
# fall-back to the gloo all-gather (and the bring-up deadline thread) of bench.py
export ICICLE_SNARK_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --constraints 100000
