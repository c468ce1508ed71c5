#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/n2; mkdir -p $O
cd $R
timeout 900 tests/gpu_rehearsal_n2.sh > $O/rehearsal.out 2> $O/rehearsal.err
grep '^{' $O/rehearsal.out | tail -1 | cut -c1-1500
grep -v "^\[" $O/rehearsal.err | head -45
timeout 600 python -m pytest tests/test_gpu_partition.py tests/test_gpu_callsite.py -x -q -m gpu 2>&1 | tail -5
