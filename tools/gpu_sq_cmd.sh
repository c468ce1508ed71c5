#!/bin/bash
# tools/gpu_sq_cmd.sh <command...>   (on the GPU box): SQ wave-state counters and kernel durations of the LAST 20 dispatches of every kernel of any command
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/sqtmp
rm -rf $D
( cd $R; rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $D -o sq -- "$@" > $R/gpurun_out/sq_cmd.out 2> $R/gpurun_out/sq_cmd.log )
F=$(find $D -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/profiles/export_sq.py $(dirname $F) $R/gpurun_out/sq_states.json 20 | head -10
python - "$D" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1][-20:]))[:8]:
    t = v[-20:]
    print(f"{k:60s} last {len(t)} launches: mean {sum(t) / len(t) / 1e3:8.1f} us")
PY
rm -rf $D
