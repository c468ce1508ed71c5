"""Turns a rocprofv3 results .db (rocpd sqlite, `rocprofv3 --kernel-trace --stats`) into a small per-kernel summary.

usage: python profiles/export_rocprof.py <results.db> <out.tsv> "note" [timed_steps total_steps]

Columns `calls total_us avg_us percent` are rocprofv3's own `top_kernels` view (whole process: warm-up steps,
during which the boxes are still falling and most kernels have nothing to do, included).  When `timed_steps
total_steps` are given, `timed_avg_us` is the mean duration of each kernel's LAST calls*timed/total dispatches,
i.e. of the dispatches issued inside bench.py's timed region -- the figure bench.py's `roofline.avg_launch_ms`
(HIP events) has to agree with."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
timed = None
if len(sys.argv) > 5:
    timed = (int(sys.argv[4]), int(sys.argv[5]))
    per = {}
    for name, dur in db.execute("select name, duration from kernels order by start"):
        per.setdefault(name, []).append(dur)
with open(sys.argv[2], "w") as f:
    f.write("# " + (sys.argv[3] if len(sys.argv) > 3 else "") + "\n")
    f.write("# rocprofv3 --kernel-trace --stats ; view top_kernels ; durations in microseconds\n")
    if timed:
        f.write(f"# timed_avg_us = mean over each kernel's last calls*{timed[0]}/{timed[1]} dispatches (the timed region of bench.py)\n")
    f.write("kernel\tcalls\ttotal_us\tavg_us\tpercent" + ("\ttimed_calls\ttimed_avg_us" if timed else "") + "\n")
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        line = f"{short}\t{calls}\t{tot:.1f}\t{avg:.2f}\t{pct:.2f}"
        if timed:
            d = per.get(name, [])
            n = len(d) * timed[0] // timed[1]
            line += f"\t{n}\t{(sum(d[-n:]) / n / 1e3) if n else 0.0:.2f}"
        f.write(line + "\n")
print("wrote", sys.argv[2], len(rows), "kernels")
