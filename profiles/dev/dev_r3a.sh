#!/bin/bash
# round 3, first GPU call: the blocked solver's tests, the advisor tests, the holdable-union test, c4 and c2 bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_blocks.py -x -q -m gpu -s > $O/blocks.log 2>&1; echo "blocks rc=$?" >> $O/blocks.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "restore_sleepers or garbage_colour or ball_pit or determinism" -s > $O/advice.log 2>&1; echo "advice rc=$?" >> $O/advice.log
timeout 600 python -m pytest tests/test_gpu_atsize.py -x -q -m gpu -k "union_the_reference_can_hold" -s > $O/union.log 2>&1; echo "union rc=$?" >> $O/union.log
cd /tmp; export TMPDIR=/tmp
timeout 500 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 1 --profile-kernels > $O/bench_c4.json 2> $O/bench_c4_kernels.log; echo "c4 rc=$?" >> $O/bench_c4_kernels.log
NH_NO_BLOCKS=1 timeout 500 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 1 > $O/bench_c4_noblocks.json 2> $O/bench_c4_noblocks.err
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --profile-kernels > $O/bench_c2.json 2> $O/bench_c2_kernels.log; echo "c2 rc=$?" >> $O/bench_c2_kernels.log
tail -3 $O/blocks.log $O/advice.log $O/union.log; cat $O/bench_c4.json | cut -c1-600; cat $O/bench_c4_noblocks.json | cut -c1-300; cat $O/bench_c2.json | cut -c1-400
