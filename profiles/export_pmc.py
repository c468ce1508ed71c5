"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace, csv).
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter values are KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, i.e. reads are reported at HALF their size:
the corrected read bytes are 2 x FETCH_SIZE (upper bound: narrow / gather accesses are not halved, so the truth lies between).
usage: python profiles/export_pmc.py gpurun_out/pmc profiles/r01_pmc_traffic.json [last_n_dispatches_per_kernel] [bench_line.json]

`launches_per_step` is counted over the LAST `last_n` STEPS of the profiled command -- the steady steps of bench.py's timed region / landed window -- where a step
is delimited by the dispatches of the one-body solver (every step, still or full, launches k_solve_one_body exactly once; a run without that kernel falls back
to k_xform).  Rounds 1-5 divided every kernel's dispatches of the whole run by the number of k_xform dispatches, which stopped being the number of steps when
xform ahead removed k_xform from 95 % of the steps: the r05 export said 4.03 solver launches per step.  When the bench line of
the profiled run is given, its step count is recorded next to the dispatch-derived one as a cross-check."""
import csv, json, sys, collections

d, out = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 4
line = sys.argv[4] if len(sys.argv) > 4 else None


def short(name):
    return name.split("(")[0].replace("void ", "")


def read_pass(counter):
    """[(dispatch id, kernel, bytes)] of one pass, in dispatch order."""
    rows = []
    for row in csv.DictReader(open(f"{d}/{counter}_counter_collection.csv")):
        if row["Counter_Name"] != counter:
            continue
        rows.append((int(row["Dispatch_Id"]), short(row["Kernel_Name"]), float(row["Counter_Value"]) * 1024.0))
    rows.sort()
    return rows


def window(rows, n_steps):
    """Rows of the last n_steps steps and the number of steps they really span.  A step ends with its k_solve_one_body dispatch."""
    marks = [i for i, (_, name, _) in enumerate(rows) if name.startswith("k_solve_one_body")]
    if len(marks) < 2:
        marks = [i for i, (_, name, _) in enumerate(rows) if name.startswith("k_xform")]
        if len(marks) < 2:
            return [], 0, 0          # (a filtered run, e.g. config 4's --kernel-include-regex: per-launch means of the last dispatches only)
    n = min(n_steps, len(marks) - 1)
    return rows[marks[-(n + 1)] + 1: marks[-1] + 1], n, len(marks)


def steps_of(line_file):
    """Steps the profiled bench.py command ran, from the line it printed (None when there is none)."""
    try:
        txt = [l for l in open(line_file).read().splitlines() if l.startswith("{")][-1]
        dd = json.loads(txt)
        c = dd["config"]
        restored = "from a device checkpoint" in c.get("timed_window", "")
        n = c["landed_after_steps"] + dd["warmup"] + 3 + (c["timed_blocks"] if restored else 0) + c["timed_steps"]
        if dd.get("landed_window"):
            n += 201
        if dd.get("full_window"):
            n += 2000
        return n
    except Exception:
        return None


passes = {c: read_pass(c) for c in ("FETCH_SIZE", "WRITE_SIZE")}
wins = {c: window(passes[c], last) for c in passes}
per_launch, per_step = {}, {}
for c, (rows, n, _total) in wins.items():
    acc = collections.defaultdict(list)
    for _, name, v in rows:
        acc[name].append(v)
    per_launch[c] = {k: sum(v) / len(v) for k, v in acc.items()}
    per_step[c] = {k: len(v) / float(max(n, 1)) for k, v in acc.items()}
# kernels that did not run in the steady window (landing only): mean of their last dispatches of the whole run, zero launches per steady step
for c, rows in passes.items():
    acc = collections.defaultdict(list)
    for _, name, v in rows:
        acc[name].append(v)
    for k, v in acc.items():
        if k not in per_launch[c]:
            per_launch[c][k] = sum(v[-last:]) / len(v[-last:])
            per_step[c][k] = 0.0 if wins[c][1] else 1.0

res = {}
for name in per_launch["FETCH_SIZE"]:
    if not name.startswith(("k_", "rs_", "sc_", "os_", "bk_")):
        continue
    fr = per_launch["FETCH_SIZE"][name]
    wr = per_launch["WRITE_SIZE"].get(name, 0.0)
    res[name] = {"fetch_bytes_raw": fr, "write_bytes": wr, "hbm_bytes_raw": fr + wr, "hbm_bytes_corrected": 2 * fr + wr,
                 "launches_per_step": per_step["FETCH_SIZE"][name]}
step_raw = sum(v["hbm_bytes_raw"] * v["launches_per_step"] for v in res.values())
step_cor = sum(v["hbm_bytes_corrected"] * v["launches_per_step"] for v in res.values())
json.dump({"note": "bytes per launch = mean over the dispatches of the last steps of the profiled command (steady steps of the timed region / landed window); corrected = "
                   "2*FETCH + WRITE (gfx950 half-count of wide reads; an upper bound where the kernel gathers), raw = FETCH + WRITE (lower bound); launches_per_step = "
                   "dispatches in that window / its steps (a step ends with its k_solve_one_body dispatch); kernels that only run while the scene lands: 0 per steady step",
           "steps_in_window": wins["FETCH_SIZE"][1], "steps_of_the_run_by_solver_dispatches": wins["FETCH_SIZE"][2], "steps_of_the_run_by_the_bench_line": steps_of(line) if line else None,
           "steady_step_bytes_raw": step_raw, "steady_step_bytes_corrected": step_cor,
           "kernels": res}, open(out, "w"), indent=1)
print(f"steady step (last {wins['FETCH_SIZE'][1]} steps): raw {step_raw/1e6:.1f} MB  corrected {step_cor/1e6:.1f} MB   [run: {wins['FETCH_SIZE'][2]} steps by solver dispatches, "
      f"{steps_of(line) if line else '?'} by the bench line]")
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"] * max(kv[1]["launches_per_step"], 1e-3))[:16]:
    print(f"{k:52s} x{v['launches_per_step']:5.2f}/step  fetch {v['fetch_bytes_raw']/1e6:8.1f} MB  write {v['write_bytes']/1e6:8.1f} MB  corrected {v['hbm_bytes_corrected']/1e6:8.1f} MB")
