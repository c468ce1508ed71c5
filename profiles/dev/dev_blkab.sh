#!/bin/bash
# A/B of block geometry variants on the c4 bench: dev_blkab.sh "lib target" ...   (lib = base or a name under nudge_amd/_ab/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/blkab; mkdir -p $O
cd /tmp
for spec in "$@"; do
  set -- $spec; LIB=$1; T=$2
  L=$R/nudge_amd/libnudge_hip.so; [ "$LIB" != base ] && L=$R/nudge_amd/_ab/lib_$LIB.so
  NUDGE_HIP_LIBRARY=$L NH_BLK_TARGET=$T timeout 300 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 1 --profile-kernels > $O/${LIB}_$T.json 2> $O/${LIB}_$T.err
  echo "$LIB target $T: $(python -c "import json;d=json.loads(open('$O/${LIB}_$T.json').read().strip().split(chr(10))[-1]);print(round(d['value'],2),'steps/s',round(d['ms_per_step'],3),'ms')") | $(grep -E 'blk_sweep|blk_prepare|blk_contact' $O/${LIB}_$T.err | awk '{printf "%s %s  ", $1, $2}')"
done
