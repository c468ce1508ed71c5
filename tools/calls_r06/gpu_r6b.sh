#!/bin/bash
# round 6, call b: the one-kernel still step -- its tests first, then the driver line with and without it, and a kernel trace
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu -s -k "fused or undone or touch or legacy" > gpurun_out/r06b_fused_tests.log 2>&1; echo "fused tests rc=$?"; grep -v "^$" gpurun_out/r06b_fused_tests.log | tail -n 25
for V in fused twokernel; do
  if [ $V = twokernel ]; then export NH_NO_FUSED_STILL=1; else unset NH_NO_FUSED_STILL; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window > gpurun_out/r06b_bench_$V.json 2> gpurun_out/r06b_bench_$V.err
  python - $V <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r06b_bench_{sys.argv[1]}.json"))
    print(sys.argv[1], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "avg_launch_ms", round(d["roofline"]["avg_launch_ms"], 4),
          "landed", round(d.get("landed_window", {}).get("steps_per_s", 0), 1), "fused", d["config"].get("fused_still", {}).get("share_of_timed_steps"), "still", d.get("still_steps"))
except Exception as e:
    print(sys.argv[1], "no line:", e); print(open(f"gpurun_out/r06b_bench_{sys.argv[1]}.err").read()[-1500:])
PY
done
unset NH_NO_FUSED_STILL
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06b_prof -o stats -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r06b_prof.log
cd $GRAFT_REPO_ROOT
F=$(find gpurun_out/r06b_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -8 $F | cut -c1-200
rm -rf gpurun_out/r06b_prof
