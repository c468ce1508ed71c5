#!/bin/bash
# round 6, call r: early counters (a full step's counters read by the host while its solver runs): the whole GPU suite, then A/B lines with option no_early_counts
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06r_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 4 gpurun_out/r06r_gpu_tests.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); fw=d.get('full_window') or {}
print('$1', round(d['value'],1), 'steps/s', round(d['ms_per_step'],5), 'ms  min', round(d['config']['block_ms_per_step']['min'],5), ' full window', fw.get('steps_per_s') and round(fw['steps_per_s'],1))"; }
{
for r in 1 2; do for e in "" NH_NO_EARLY_COUNTS; do
  env ${e:+$e=1} NH_NO_STILL=1 python bench.py --no-cpu --no-full-window 2>/dev/null | line "c2 full steps (NH_NO_STILL=1) ${e:-early counters}"
done; done
for r in 1 2; do for e in "" NH_NO_EARLY_COUNTS; do
  env ${e:+$e=1} python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | line "c2 driver line ${e:-early counters}"
done; done
for e in "" NH_NO_EARLY_COUNTS; do
  env ${e:+$e=1} NH_NO_STILL=1 python bench.py --no-cpu --no-full-window --scene mixed --tiles 248 2>/dev/null | line "c3 full steps ${e:-early counters}"
done
for e in "" NH_NO_EARLY_COUNTS; do
  env ${e:+$e=1} python bench.py --no-cpu --scene pit --steps 10 2>/dev/null | line "c4 pit ${e:-early counters}"
done
} | tee gpurun_out/r06r_early_counters_ab.log
