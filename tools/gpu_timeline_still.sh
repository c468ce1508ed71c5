#!/bin/bash
# tools/gpu_timeline_still.sh <command...>  (on the GPU box): one steady PAIR step (between two k_pair_begin launches, the 20th from the end) of any command, launch by launch --
# gpu_timeline_cmd.sh delimits steps by k_collide_begin, which a still step does not launch
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl
( cd $R; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o t -- "$@" > $R/gpurun_out/tl_cmd.out 2> $R/gpurun_out/tl_cmd.log ); grep -v amdgpu $R/gpurun_out/tl_cmd.out | tail -3
F=$(find $R/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:64]
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_pair_begin")]
a, b = starts[-20], starts[-19]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None; busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    print(f"{(s - t0) / 1e3:9.2f} us  {name(r):64s} {(e - s) / 1e3:8.2f} us   gap {gap / 1e3:7.2f}   workgroups {int(r.get('Grid_Size', 0)) // max(1, int(r.get('Workgroup_Size', 1)))}")
    busy += e - s; prev_end = max(e, prev_end or 0)
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches, busy {busy / 1e3:.1f} us")
tspan = int(rows[starts[-10]]["Start_Timestamp"]) - int(rows[starts[-110]]["Start_Timestamp"])
print(f"mean of 100 steps: {tspan / 100e3:.1f} us per step, {(starts[-10] - starts[-110]) / 100:.1f} launches per step (under rocprofv3)")
PY
rm -rf $R/gpurun_out/tl
