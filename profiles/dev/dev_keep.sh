#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for V in 0 1 0 1; do
[ $V = 1 ] && export NH_BLK_NO_KEEP=1 || unset NH_BLK_NO_KEEP
python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('no_keep', $V, round(d['value'],2), round(d['ms_per_step'],3), d['launches_per_step'], round(d['roofline']['avg_launch_ms'],4))"
done
