"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace, csv).
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter values are KiB;
on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, i.e. reads are reported at HALF their size:
the corrected read bytes are 2 x FETCH_SIZE (upper bound: narrow / gather accesses are not halved, so the truth lies between).
usage: python profiles/export_pmc.py gpurun_out/pmc profiles/r01_pmc_traffic.json [last_n_dispatches_per_kernel]"""
import csv, json, sys, collections

d, out = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 4
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f"{d}/{c}_counter_collection.csv")):
        if row["Counter_Name"] != c:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        per[name].append(float(row["Counter_Value"]))
    vals[c] = per
# launches per step over the whole profiled run: k_xform (either instantiation) runs exactly once per step
steps = sum(len(v) for n, v in vals["FETCH_SIZE"].items() if n.startswith("k_xform")) or 1
res = {}
for name in vals["FETCH_SIZE"]:
    f = vals["FETCH_SIZE"][name][-last:]
    w = vals["WRITE_SIZE"].get(name, [0.0])[-last:]
    if not name.startswith(("k_", "rs_", "sc_", "os_", "bk_")):
        continue
    fr = sum(f) / len(f) * 1024.0
    wr = sum(w) / len(w) * 1024.0
    res[name] = {"fetch_bytes_raw": fr, "write_bytes": wr, "hbm_bytes_raw": fr + wr, "hbm_bytes_corrected": 2 * fr + wr, "dispatches_averaged": len(f),
                 "launches_per_step": len(vals["FETCH_SIZE"][name]) / steps}
json.dump({"note": "bytes per launch, mean of the last dispatches (landed, awake steps of the 1M-box scene); corrected = 2*FETCH + WRITE (gfx950 half-count of wide reads); "
                   "launches_per_step = this kernel's dispatches / steps over the whole profiled run (landing included)",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"])[:14]:
    print(f"{k:28s} fetch {v['fetch_bytes_raw']/1e6:8.1f} MB  write {v['write_bytes']/1e6:8.1f} MB  corrected {v['hbm_bytes_corrected']/1e6:8.1f} MB")
