#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/c4stat; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_atsize.py -x -q -m gpu -s -k "four_million or column_of_160" > $O/blocks.log 2>&1
grep "^\[pit\|^\[column" $O/blocks.log
echo "--- per-colour launches"
NH_NO_BLOCKS=1 timeout 600 python -m pytest tests/test_gpu_atsize.py -x -q -m gpu -s -k "four_million" > $O/colours.log 2>&1
grep "^\[pit" $O/colours.log
tail -2 $O/blocks.log $O/colours.log
