#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/n2; mkdir -p $O
cd $R
for V in default nocoop nofused; do
  export NH_NO_COOPERATIVE= NUDGE_BENCH_NO_FUSED=
  unset NH_NO_COOPERATIVE NUDGE_BENCH_NO_FUSED
  [ $V = nocoop ] && export NH_NO_COOPERATIVE=1
  [ $V = nofused ] && export NUDGE_BENCH_NO_FUSED=1
  timeout 600 tests/gpu_rehearsal_n2.sh > $O/rehearsal_$V.out 2> $O/rehearsal_$V.err
  echo "== $V"; grep '^{' $O/rehearsal_$V.out | tail -1 | cut -c1-260
done
