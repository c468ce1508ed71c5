#!/bin/bash
# the whole -m gpu suite, as the driver runs it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/suite; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
tail -30 $O/gpu_tests.log
