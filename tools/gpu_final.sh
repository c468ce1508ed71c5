#!/bin/bash
# round 6: the whole GPU suite + smoke on the final build, then everything profiles/ holds (profiles/collect.sh) + movers / sleepers tables
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 4 gpurun_out/r06_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
bash profiles/collect.sh r06 > gpurun_out/r06_collect.log 2>&1
timeout 600 python tests/gpu_movers_rates.py --quick > gpurun_out/r06/export/r06_movers_local_speculation.log 2>&1
timeout 600 python tests/gpu_sleepers_rates.py > gpurun_out/r06/export/r06_sleepers_rates.log 2>&1
timeout 600 python tests/gpu_dropin_rates.py > gpurun_out/r06/export/r06_dropin_rates.log 2>&1
ls gpurun_out/r06/export | wc -l
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/export/r06_bench_*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f.split("/")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), r.get("kernel"), r.get("frac") and round(r["frac"], 3), "traffic", r.get("traffic") and round(r["traffic"] / 1e6), r.get("pricing_check"), "pmc_frac", d.get("step_traffic", {}).get("pmc_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
