"""Per-kernel SQ wave-state accounting from rocprofv3 --pmc passes (--kernel-trace, csv): where the wave cycles of each kernel go.
SQ_WAVE_CYCLES ~= SQ_ACTIVE_INST_ANY + SQ_WAIT_ANY (parked at s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall), all in quad-cycles
(MI355X_MICROARCH.md).  usage: python profiles/export_sq.py <dir with *_counter_collection.csv> out.json [last_n]"""
import csv, glob, json, sys, collections

d, out = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 20
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))     # kernel -> dispatch -> counter -> value
for f in glob.glob(f"{d}/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith(("k_", "rs_", "sc_", "os_", "bk_")):
            continue
        per[name][int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
res = {}
for name, disp in per.items():
    ids = sorted(disp)[-last:]
    mean = {c: sum(disp[k][c] for k in ids) / len(ids) for c in disp[ids[0]]}
    r = {"dispatches_averaged": len(ids), "counters_per_launch": mean}
    wc = mean.get("SQ_WAVE_CYCLES")
    if wc:
        for c, key in (("SQ_ACTIVE_INST_ANY", "active"), ("SQ_WAIT_ANY", "parked_at_waitcnt"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_ACTIVE_INST_VALU", "valu")):
            if c in mean:
                r[key + "_share_of_wave_cycles"] = mean[c] / wc
    if mean.get("SQ_WAVES") and "SQ_INSTS_VALU" in mean:
        r["valu_instructions_per_wave"] = mean["SQ_INSTS_VALU"] / mean["SQ_WAVES"]
    res[name] = r
json.dump({"note": "mean per launch over the last dispatches of each kernel (landed, awake steps); quad-cycles", "kernels": res}, open(out, "w"), indent=1)
rows = sorted(res.items(), key=lambda kv: -kv[1]["counters_per_launch"].get("SQ_WAVE_CYCLES", 0))[:24]
for k, v in rows:
    print(f"{k[:44]:44s} wave_cycles {v['counters_per_launch'].get('SQ_WAVE_CYCLES', 0)/1e6:8.2f} M  active {v.get('active_share_of_wave_cycles', 0):5.2f}  parked {v.get('parked_at_waitcnt_share_of_wave_cycles', 0):5.2f}  issue-stall {v.get('issue_stall_share_of_wave_cycles', 0):5.2f}  valu {v.get('valu_share_of_wave_cycles', 0):5.2f}")
