#!/bin/bash
# round 6, call v: sleeper lists (the sleepers form's narrowphase and solver over the groups with something to do): the tests with sleepers in them, the rates table with and without
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu -k "sleep or asleep or early" 2>&1 | tail -n 12
for e in "" NH_NO_SLEEPER_LISTS; do echo "--- ${e:-sleeper lists}"; for f in 0.1 0.5 0.9; do env ${e:+$e=1} timeout 300 python tests/gpu_sleepers_rates.py --table --only $f 2>&1 | grep -E "awake|%"; done; done | tee gpurun_out/r06v_sleeper_lists.log
