"""Turns a rocprofv3 results .db (rocpd sqlite, `rocprofv3 --kernel-trace --stats`) into a small per-kernel summary.
usage: python profiles/export_rocprof.py gpurun_out/prof/r01_results.db profiles/r01_rocprofv3_kernel_stats.tsv "note" """
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w") as f:
    f.write("# " + (sys.argv[3] if len(sys.argv) > 3 else "") + "\n")
    f.write("# rocprofv3 --kernel-trace --stats ; view top_kernels ; durations in microseconds\n")
    f.write("kernel\tcalls\ttotal_us\tavg_us\tpercent\n")
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        f.write(f"{short}\t{calls}\t{tot:.1f}\t{avg:.2f}\t{pct:.2f}\n")
print("wrote", sys.argv[2], len(rows), "kernels")
