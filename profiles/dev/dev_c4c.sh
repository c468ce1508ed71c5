#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do
timeout 200 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 0.5 2>&1 | tail -2 | cut -c1-200
done
echo "--- no blocks"
NH_NO_BLOCKS=1 timeout 200 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 0.5 2>&1 | tail -2 | cut -c1-200
