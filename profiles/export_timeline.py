"""One step of a rocprofv3 --kernel-trace (csv) as a timeline: every launch with its start offset, duration and the idle gap before it.
The step is the last complete one: from the last-but-one k_collide_begin to the last one.
usage: python profiles/export_timeline.py <dir with *_kernel_trace.csv> <out.log>"""
import csv, glob, sys

d, out = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("k_collide_begin")]
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0 = step[0][0]
busy = sum(e - s for s, e, _ in step)
span = rows[b][0] - t0
with open(out, "w") as o:
    o.write(f"# one step: {len(step)} launches, span {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {100.0 * (span - busy) / span:.1f} %\n")
    o.write("#   start_us   dur_us   gap_before_us  kernel\n")
    prev_end = t0
    for s, e, n in step:
        o.write(f"{(s - t0) / 1e3:11.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:10.1f}      {n}\n")
        prev_end = max(prev_end, e)
    o.write(f"{span / 1e3:11.1f}      (next k_collide_begin; gap {(rows[b][0] - prev_end) / 1e3:.1f} us)\n")
print(open(out).readline().strip())
