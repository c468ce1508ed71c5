#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
NH_BLK_PROFILE=1 timeout 300 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 0.3 > $O/bench_c4_prof.json 2> $O/bench_c4_prof.log
grep blk_profile $O/bench_c4_prof.log | tail -8
