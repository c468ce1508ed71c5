#!/bin/bash
# round 6, call g: whole GPU suite on pair ahead (default on) + smoke + driver line, A/B against no_pair_ahead
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06g_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 6 gpurun_out/r06g_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
for V in pair nopair; do
  if [ $V = nopair ]; then export NH_NO_PAIR_AHEAD=1; else unset NH_NO_PAIR_AHEAD; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/r06g_bench_$V.json 2> gpurun_out/r06g_bench_$V.err
  python - $V <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r06g_bench_{sys.argv[1]}.json"))
    print(sys.argv[1], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "avg_launch_ms", round(d["roofline"]["avg_launch_ms"], 4),
          "landed", round(d.get("landed_window", {}).get("steps_per_s", 0), 1), "full", round(d.get("full_window", {}).get("steps_per_s", 0), 1), "pair", d["config"].get("pair_ahead", {}).get("share_of_timed_steps"), "still", d.get("still_steps", {}).get("confirmed"), d.get("still_steps", {}).get("replayed"))
except Exception as e:
    print(sys.argv[1], "no line:", e); print(open(f"gpurun_out/r06g_bench_{sys.argv[1]}.err").read()[-1500:])
PY
done
