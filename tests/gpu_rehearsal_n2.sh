#!/bin/bash
# One-GPU rehearsal of the N > 1 bench path: two ranks share cuda:0, halo messages go through gloo (host staging).
# The driver's real runs use one rank per GPU over RCCL; this only proves the code path and gives a kernel table.
export NUDGE_BENCH_BACKEND=gloo NUDGE_BENCH_ONE_DEVICE=1
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --profile-kernels "$@"
