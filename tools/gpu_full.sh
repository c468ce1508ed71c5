#!/bin/bash
# the whole GPU suite + smoke + the driver's bench invocation
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 6 gpurun_out/r05_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver.json 2> gpurun_out/r05_bench_driver.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_driver.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("limited_by"), "valu", (d["roofline"].get("valu") or {}).get("valu_frac"))
print("landed", d.get("landed_window", {}).get("steps_per_s"), "full", d.get("full_window", {}).get("steps_per_s"), "cpu", d.get("cpu_baseline", {}).get("value"))
print(d.get("kernel_ms_per_step"))
PY
