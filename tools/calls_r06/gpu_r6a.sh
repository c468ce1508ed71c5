#!/bin/bash
# round 6, call a: whole GPU suite on the advisor fixes; the undo test against the library as it was BEFORE the fix (must fail); driver line
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06a_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 8 gpurun_out/r06a_gpu_tests.log
NUDGE_HIP_LIBRARY=$PWD/nudge_amd/_ab/libprefix.so timeout 300 python -m pytest tests/test_gpu_still.py -x -q -m gpu -k "undone" -s > gpurun_out/r06a_undo_prefix.log 2>&1; echo "undo test on the pre-fix library rc=$? (expected: 1)"; tail -n 12 gpurun_out/r06a_undo_prefix.log
timeout 300 python -m pytest tests/test_gpu_still.py -x -q -m gpu -k "undone" -s 2>&1 | tail -n 4
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a_bench_driver.json 2> gpurun_out/r06a_bench_driver.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06a_bench_driver.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], "step_traffic", d["step_traffic"])
print("landed", d.get("landed_window", {}).get("steps_per_s"), "full", d.get("full_window", {}).get("steps_per_s"), "cpu", d.get("cpu_baseline", {}).get("value"))
print(d.get("kernel_ms_per_step"))
PY
