#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for V in 0 1 0 1; do
[ $V = 1 ] && export NH_MEASURE_SKIP_CHAINS=1 || unset NH_MEASURE_SKIP_CHAINS
python $R/bench.py --gpus 1 --steps 20 --warmup 12 --no-cpu --no-full-window --min-time 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('skip', $V, round(d['value'],1), round(d['ms_per_step'],4), d['launches_per_step'], d['config']['contacts_per_step'])"
done
