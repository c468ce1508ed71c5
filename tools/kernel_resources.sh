#!/bin/bash
# Register / LDS / spill figures of the kernels of one translation unit, from the compiler's own remarks (no GPU needed).
#   tools/kernel_resources.sh nh_solve.hip [name-filter]
cd "$(dirname "$0")/../nudge_amd/csrc" || exit 1
src=${1:-nh_solve.hip}; filt=${2:-.}
/opt/rocm/bin/hipcc $EXTRA -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC --cuda-device-only -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | awk -v f="$filt" '
  /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); keep = (name ~ f) }
  keep && /(VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs:|LDS Size)/ { line=$0; sub(/.*remark: [^ ]* /,"",line); sub(/.*\]  */,"",line); out[name]=out[name] " | " line }
  END { for (n in out) print n out[n] }' | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | sort
