#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags] [git rev of the sources, default: working tree]  ->  nudge_amd/_ab/lib<name>.so   (A/B builds for tests/gpu_ab_libs.py)
set -e
name=$1; extra=${2:-}; rev=${3:-}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/nhvar.XXXX)
mkdir -p $tmp/nudge_amd/csrc $tmp/include $root/nudge_amd/_ab
if [ -n "$rev" ]; then
  for f in nh_context.hip nh_util.hip nh_collide.hip nh_solve.hip nh_partition.hip nh_fused.hip nh_internal.h nh_math.h nh_narrowphase.h nh_solver.h nh_blocks.h; do git -C $root show $rev:nudge_amd/csrc/$f > $tmp/nudge_amd/csrc/$f 2>/dev/null || rm -f $tmp/nudge_amd/csrc/$f; done
  git -C $root show $rev:include/nudge_hip.h > $tmp/include/nudge_hip.h
else
  cp $root/nudge_amd/csrc/*.hip $root/nudge_amd/csrc/*.h $tmp/nudge_amd/csrc/; cp $root/include/nudge_hip.h $tmp/include/
fi
cd $tmp/nudge_amd/csrc
if [ -n "$NH_VARIANT_PATCH" ]; then python $root/$NH_VARIANT_PATCH nh_solve.hip nh_solve.patched && mv nh_solve.patched nh_solve.hip; fi
for f in nh_context nh_util nh_collide nh_solve nh_partition nh_fused; do
  [ -f $f.hip ] || continue
  /opt/rocm/bin/hipcc $extra -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -w -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o $root/nudge_amd/_ab/lib$name.so
rm -rf $tmp
ls -la $root/nudge_amd/_ab/lib$name.so
