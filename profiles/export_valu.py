"""VALU issue accounting of the fused solver from one rocprofv3 --pmc pass (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAVES GRBM_GUI_ACTIVE; --kernel-trace, csv).  SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE
counts shader-clock cycles of the whole dispatch.  valu_busy = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE * SIMDs), the share of all SIMD
issue cycles of the launch spent issuing VALU instructions.
usage: python profiles/export_valu.py <dir with *_counter_collection.csv> out.json [kernel prefix] [last_n]"""
import csv, glob, json, sys, collections

d, out = sys.argv[1], sys.argv[2]
prefix = sys.argv[3] if len(sys.argv) > 3 else "k_solve_one_body<4, true, true>"
last = int(sys.argv[4]) if len(sys.argv) > 4 else 20
SIMDS = 256 * 4
per = collections.defaultdict(lambda: collections.defaultdict(float))     # dispatch id -> counter -> value
order = []
for f in glob.glob(f"{d}/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith(prefix):
            continue
        k = int(row["Dispatch_Id"])
        if k not in per:
            order.append(k)
        per[k][row["Counter_Name"]] += float(row["Counter_Value"])
order = sorted(order)[-last:]
mean = {c: sum(per[k][c] for k in order) / len(order) for c in per[order[0]]}
res = {"kernel": prefix, "dispatches_averaged": len(order), "counters_per_launch": mean}
if "SQ_ACTIVE_INST_VALU" in mean and "GRBM_GUI_ACTIVE" in mean and mean["GRBM_GUI_ACTIVE"]:
    res["valu_busy"] = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / (mean["GRBM_GUI_ACTIVE"] * SIMDS)
if "SQ_INSTS_VALU" in mean and "SQ_WAVES" in mean and mean["SQ_WAVES"]:
    res["valu_instructions_per_wave"] = mean["SQ_INSTS_VALU"] / mean["SQ_WAVES"]
if "SQ_WAVE_CYCLES" in mean and "SQ_ACTIVE_INST_VALU" in mean and mean["SQ_WAVE_CYCLES"]:
    res["valu_share_of_wave_cycles"] = mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_WAVE_CYCLES"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
