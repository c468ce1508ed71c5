#!/bin/bash
# tools/gpu_timeline.sh [env assignments / bench arguments...]  (on the GPU box): the timeline of ONE steady step of `bench.py` -- every launch with its duration and the idle gap
# before it -- from a rocprofv3 kernel trace.  Usage: NH_NO_STILL=1 tools/gpu_timeline.sh            (the full step of config 2)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 0 "$@" > $R/gpurun_out/tl_bench.json 2> $R/gpurun_out/tl.log
F=$(find $R/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
# a step starts at k_collide_begin; take the step 12 before the last
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_collide_begin")]
if len(starts) < 16: print("too few steps", len(starts)); sys.exit(0)
a, b = starts[-12], starts[-11]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None; busy = 0; gaps = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    print(f"{(s - t0) / 1e3:9.2f} us  {name(r):60s} {(e - s) / 1e3:8.2f} us   gap {gap / 1e3:7.2f}")
    busy += e - s; gaps += max(gap, 0); prev_end = max(e, prev_end or 0)
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches, busy {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us")
# average over the last 100 steps
tot = collections.Counter(); n = 0
for i in range(len(starts) - 101, len(starts) - 1):
    n += 1
tspan = int(rows[starts[-1]]["Start_Timestamp"]) - int(rows[starts[-101]]["Start_Timestamp"])
print(f"mean of the last 100 steps: {tspan / 100e3:.1f} us per step, {(starts[-1] - starts[-101]) / 100:.1f} launches per step")
PY
rm -rf $R/gpurun_out/tl
