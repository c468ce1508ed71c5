#!/bin/bash
# round 6, call s: early counters with bench.py's one-kernel timing on (call r excluded every timed context: no effect measured); the tests that go through full steps, A/B lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py tests/test_gpu_callsite.py -x -q -m gpu 2>&1 | tail -n 2
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); fw=d.get('full_window') or {}
print('$1', round(d['value'],1), 'steps/s', round(d['ms_per_step'],5), 'ms  min', round(d['config']['block_ms_per_step']['min'],5), ' full window', fw.get('steps_per_s') and round(fw['steps_per_s'],1))"; }
{
for r in 1 2; do for e in "" NH_NO_EARLY_COUNTS; do
  env ${e:+$e=1} NH_NO_STILL=1 python bench.py --no-cpu --no-full-window 2>/dev/null | line "c2 full steps (NH_NO_STILL=1) ${e:-early counters}"
done; done
for e in "" NH_NO_EARLY_COUNTS; do
  env ${e:+$e=1} python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | line "c2 driver line ${e:-early counters}"
done
for e in "" NH_NO_EARLY_COUNTS; do
  env ${e:+$e=1} NH_NO_STILL=1 python bench.py --no-cpu --no-full-window --scene mixed --tiles 248 2>/dev/null | line "c3 full steps ${e:-early counters}"
done
for e in "" NH_NO_EARLY_COUNTS; do
  env ${e:+$e=1} python bench.py --no-cpu --scene pit --steps 10 2>/dev/null | line "c4 pit ${e:-early counters}"
done
NH_NO_STILL=1 bash tools/gpu_timeline.sh 2>&1 | tail -n 32
} | tee gpurun_out/r06s_early_counters_ab.log
