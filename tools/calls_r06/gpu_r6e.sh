#!/bin/bash
# round 6, call c: the one-kernel still step with segment sums -- tests, A/B lines, SQ counters of the fused kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py -q -m gpu -s -k "fused or undone or touch" > gpurun_out/r06e_fused_tests.log 2>&1; echo "fused tests rc=$?"; grep -v "^$" gpurun_out/r06e_fused_tests.log | tail -n 40
for V in fused twokernel; do
  if [ $V = twokernel ]; then export NH_NO_FUSED_STILL=1; else unset NH_NO_FUSED_STILL; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window > gpurun_out/r06e_bench_$V.json 2> gpurun_out/r06e_bench_$V.err
  python - $V <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r06e_bench_{sys.argv[1]}.json"))
    print(sys.argv[1], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "avg_launch_ms", round(d["roofline"]["avg_launch_ms"], 4),
          "landed", round(d.get("landed_window", {}).get("steps_per_s", 0), 1), "fused", d["config"].get("fused_still", {}).get("share_of_timed_steps"), "still", d.get("still_steps", {}).get("confirmed"), d.get("still_steps", {}).get("replayed"))
except Exception as e:
    print(sys.argv[1], "no line:", e); print(open(f"gpurun_out/r06e_bench_{sys.argv[1]}.err").read()[-1500:])
PY
done
unset NH_NO_FUSED_STILL
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --kernel-include-regex "k_still_fused|k_solve_one_body|k_narrowphase" --kernel-trace --output-format csv -d $R/gpurun_out/r06e_sq -o sq -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 0 > /dev/null 2> $R/gpurun_out/r06e_sq.log
F=$(find $R/gpurun_out/r06e_sq -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/profiles/export_sq.py $(dirname $F) $R/gpurun_out/r06e_sq_wave_states.json 20 | head -8
rm -rf $R/gpurun_out/r06e_sq
