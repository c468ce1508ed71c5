#!/bin/bash
# tools/gpu_ab_scene.sh "<bench arguments>" <lib> <lib> ...   (on the GPU box) -- the bench line of each A/B library of nudge_amd/_ab for one scene, twice, alternating
A=$1; shift
for r in 1 2; do for l in "$@"; do NUDGE_HIP_LIBRARY=$PWD/nudge_amd/_ab/lib$l.so python bench.py --no-cpu --no-full-window $A 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l', round(d['value'],1), round(d['ms_per_step'],5), 'min', round(d['config']['block_ms_per_step']['min'],5))"; done; done
