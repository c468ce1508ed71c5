#!/bin/bash
# tools/gpu_ab_c4.sh <lib> ...   (on the GPU box): config 4 (4 M-sphere pit, steps [506, 516)) per A/B library of nudge_amd/_ab: steps/s and the kernels named in $KERNELS
K=${KERNELS:-"find_pairs blk_contact blk_bounds contact_class gather_contacts blk_prepare narrowphase_sph"}
for l in "$@"; do
  NUDGE_HIP_LIBRARY=$PWD/nudge_amd/_ab/lib$l.so timeout 400 python bench.py --scene pit --no-cpu --steps 10 --profile-kernels > gpurun_out/ab_c4_$l.json 2> gpurun_out/ab_c4_$l.log
  python - "$l" "$K" <<'PY'
import json, sys, re
l, ks = sys.argv[1], sys.argv[2].split()
d = json.load(open(f"gpurun_out/ab_c4_{l}.json"))
t = {}
for line in open(f"gpurun_out/ab_c4_{l}.log"):
    m = re.match(r"\s+(\S+)\s+([0-9.]+) ms/step", line)
    if m: t[m.group(1)] = float(m.group(2))
print(l, round(d["value"], 2), "steps/s", round(d["ms_per_step"], 3), "ms |", "  ".join(f"{k} {t.get(k, 0):.3f}" for k in ks), "| sum of table", round(sum(t.values()), 3))
PY
done
