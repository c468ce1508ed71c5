#!/bin/bash
# round 6, call p: halo split (the next sub-step's exchange beside the interior's solver): partition / call-site / c5 tests, then the one-rank partition path with and without
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_partition.py tests/test_gpu_callsite.py tests/test_gpu_c5.py -x -q -m gpu -s 2>&1 | grep -E "nh_partition_step|passed|failed|Error|error" | tail -n 8
for e in NH_HALO_OVERLAP ""; do for r in 1 2; do env ${e:+$e=1} NUDGE_BENCH_LOOPBACK=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 100 --warmup 5 --tiles 248 --no-cpu 2> gpurun_out/r06p_loop.err | grep "^{" > gpurun_out/r06p_loop.json; python -c "
import json; d=json.load(open('gpurun_out/r06p_loop.json')); print('one rank partition path ${e:-between sub-steps}', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a gpurun_out/r06p_ab.log; done; done
