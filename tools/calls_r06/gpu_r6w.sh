#!/bin/bash
# round 6, call w: sleepers ahead (pair ahead while the sleeping set stands still): the tests with sleepers in them, the rates table with and without
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu -s -k "sleep or asleep" 2>&1 | grep -E "sleeper skip|passed|failed|Error|assert" | tail -n 12
for e in "" NH_NO_SLEEPER_AHEAD; do echo "--- ${e:-sleepers ahead}"; for f in 0.1 0.5 0.9; do env ${e:+$e=1} timeout 300 python tests/gpu_sleepers_rates.py --only $f 2>&1 | grep -E "%"; done; done | tee gpurun_out/r06w_sleepers_ahead.log
