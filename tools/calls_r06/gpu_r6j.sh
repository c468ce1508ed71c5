#!/bin/bash
# round 6, call j: the still solver reports the step's counters itself (no runtime copy per step): tests + same-box A/B of the default line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_still.py tests/test_gpu_partition.py tests/test_gpu_callsite.py -x -q -m gpu 2>&1 | tail -n 4
tools/gpu_ab_two.sh base selfrep | tee gpurun_out/r06j_ab.log
for l in base selfrep; do NUDGE_HIP_LIBRARY=$PWD/nudge_amd/_ab/lib$l.so python bench.py --scene mixed --tiles 248 --no-cpu --no-full-window 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3 $l', round(d['value'],1), round(d['ms_per_step'],5))" | tee -a gpurun_out/r06j_ab.log; done
