#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c4; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_blocks.py -x -q -m gpu -k "pit or ball or sleeping or golden or pile or blocks or garbage" 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 1 --profile-kernels > $O/bench_c4.json 2> $O/bench_c4_kernels.log
cut -c1-220 $O/bench_c4.json; head -34 $O/bench_c4_kernels.log
