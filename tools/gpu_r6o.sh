#!/bin/bash
# round 6, call o: quiet refreshes (a partitioned world at rest skips the refresh at an epoch boundary): partition / c5 / call-site tests, then the one-rank partition path through RCCL
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_partition.py tests/test_gpu_c5.py tests/test_gpu_callsite.py -x -q -m gpu 2>&1 | tail -n 5
for r in 1 2; do NUDGE_BENCH_LOOPBACK=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 100 --warmup 5 --tiles 248 --no-cpu 2> gpurun_out/r06o_loop.err | grep "^{" > gpurun_out/r06o_loop.json; python -c "
import json; d=json.load(open('gpurun_out/r06o_loop.json')); c=d['config']; print('one rank partition path', round(d['value'],1), round(d['ms_per_step'],4), c.get('sharding','')[:160]); print({k: c[k] for k in c if 'refresh' in k or 'halo' in k})"; done
