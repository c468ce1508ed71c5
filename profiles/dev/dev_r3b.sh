#!/bin/bash
# c4 tuning: blocked solver with the two-deep pipeline, block size sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_blocks.py -x -q -m gpu > $O/blocks.log 2>&1; echo "blocks rc=$?" >> $O/blocks.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "restore_sleepers or garbage_colour" > $O/advice.log 2>&1; echo "advice rc=$?" >> $O/advice.log
cd /tmp; export TMPDIR=/tmp
for T in 900 600 1300; do
NH_BLK_TARGET=$T timeout 300 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 1 --profile-kernels > $O/bench_c4_t$T.json 2> $O/bench_c4_t${T}_kernels.log; echo "c4 rc=$?" >> $O/bench_c4_t${T}_kernels.log
done
tail -3 $O/blocks.log; tail -3 $O/advice.log; for T in 900 600 1300; do cut -c1-330 $O/bench_c4_t$T.json; grep "blk_\|apply_level" $O/bench_c4_t${T}_kernels.log; done
