#!/bin/bash
# Dev script (GPU): A/B of library builds on config 4 (the 4M-sphere pit), alternating in ONE gpurun call (box-to-box differences are 2-3 %).
# usage: tests/gpu_ab_pit.sh lib1.so lib2.so ...   (paths relative to the repo root; each run is its own process)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for lib in "$@"; do
    NUDGE_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --scene pit --no-cpu --steps 10 --min-time 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', round(d['value'], 2), 'steps/s', round(d['ms_per_step'], 3), 'ms', d.get('launches_per_step'), 'launches')"
  done
done
