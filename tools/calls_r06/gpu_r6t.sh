#!/bin/bash
# round 6, call t: the last verdict of an nh_step call polled (early counters): tests, calls of 1 .. 100 sub-steps with and without, the driver line in both orders
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_still.py tests/test_gpu_callsite.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 2
timeout 600 python tests/gpu_call_rates.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06t_call_rates.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); fw=d.get('full_window') or {}
print('$1', round(d['value'],1), 'steps/s', round(d['ms_per_step'],5), 'ms  min', round(d['config']['block_ms_per_step']['min'],5), ' full window', fw.get('steps_per_s') and round(fw['steps_per_s'],1))"; }
for r in 1 2; do for e in NH_NO_EARLY_COUNTS ""; do
  env ${e:+$e=1} python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | line "c2 driver line ${e:-early counters}"
done; done | tee gpurun_out/r06t_driver_ab.log
