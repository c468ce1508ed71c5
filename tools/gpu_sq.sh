#!/bin/bash
# tools/gpu_sq.sh <bench arguments...>   (on the GPU box): SQ wave-state counters of every kernel of `bench.py <arguments>` (profiles/export_sq.py), printed
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/sqtmp
rm -rf $D
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $D -o sq -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 0 "$@" > $R/gpurun_out/sq_bench.json 2> $R/gpurun_out/sq.log
F=$(find $D -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/profiles/export_sq.py $(dirname $F) $R/gpurun_out/sq_states.json 20 | head -12
python - <<PY
import json
d = json.load(open("$R/gpurun_out/sq_states.json"))
for k, v in d.get("kernels", {}).items():
    if "solve_one_body" in k:
        c = v["counters_per_launch"]
        print(k[:70], "waves", round(c.get("SQ_WAVES", 0)), "valu/wave", round(v.get("valu_instructions_per_wave", 0)), "wave quad-cycles per wave", round(c.get("SQ_WAVE_CYCLES", 0) / max(c.get("SQ_WAVES", 1), 1)))
PY
rm -rf $D
