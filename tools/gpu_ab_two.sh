#!/bin/bash
# tools/gpu_ab_two.sh <lib> <lib> ...   (on the GPU box) -- the default bench line of each A/B library of nudge_amd/_ab, twice, alternating (one box: comparable clocks)
for r in 1 2; do for l in "$@"; do NUDGE_HIP_LIBRARY=$PWD/nudge_amd/_ab/lib$l.so python bench.py --no-cpu --no-full-window 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l', round(d['value'],1), round(d['ms_per_step'],5), 'min', round(d['config']['block_ms_per_step']['min'],5), 'lw', round(d['landed_window']['ms_per_step'],5))"; done; done
