#!/bin/bash
# everything: the -m gpu suite as the driver runs it, then the c4 and c2 lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/full; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
tail -6 $O/gpu_tests.log
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 1 --profile-kernels > $O/bench_c4.json 2> $O/bench_c4_kernels.log
cut -c1-220 $O/bench_c4.json; head -12 $O/bench_c4_kernels.log
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --profile-kernels > $O/bench_c2.json 2> $O/bench_c2_kernels.log
cut -c1-220 $O/bench_c2.json; head -8 $O/bench_c2_kernels.log
