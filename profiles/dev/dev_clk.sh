#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for MT in 0.5 3 8; do
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time $MT 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('min-time', $MT, round(d['value'],1), d['config']['block_ms_per_step'], 'solver avg ms', round(d['roofline']['avg_launch_ms'],4), 'blocks', d['config']['timed_blocks'])"
done
rocm-smi --showclocks 2>/dev/null | head -20
