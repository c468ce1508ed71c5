#!/bin/bash
# tools/gpu_trace_cmd.sh <command...>   (on the GPU box): rocprofv3 --kernel-trace of any command; mean duration of the last 200 launches of the top kernels, launches per solver launch
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; D=$R/gpurun_out/trc
rm -rf $D
( cd $R; rocprofv3 --kernel-trace --output-format csv -d $D -o t -- "$@" > $R/gpurun_out/trc_cmd.out 2> $R/gpurun_out/trc_cmd.log ); grep "^{" $R/gpurun_out/trc_cmd.out | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('line:', round(d['value'],1), round(d['ms_per_step'],4))"
python - "$D" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:64]
per = collections.defaultdict(list)
for r in rows[-12000:]:
    per[name(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k:64s} x{len(v):5d}  mean {sum(v) / len(v) / 1e3:8.1f} us")
# one steady step around the end: from one k_pair_begin to the next
pb = [i for i, r in enumerate(rows) if name(r).startswith("k_pair_begin")]
if len(pb) > 20:
    a, b = pb[-12], pb[-11]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b + 1]:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {name(r):64s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  queue {r.get('Queue_Id', '?')}")
PY
rm -rf $D
