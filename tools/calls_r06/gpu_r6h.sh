#!/bin/bash
# round 6, call h: nh_partition_step -- partition tests, then the one-rank partition path over RCCL with and without it
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_partition.py -x -q -m gpu -s > gpurun_out/r06h_partition_tests.log 2>&1; echo "partition tests rc=$?"; grep -E "^\[|passed|failed|Error|assert" gpurun_out/r06h_partition_tests.log | tail -n 12
for V in multi single; do
  if [ $V = single ]; then export NUDGE_BENCH_MULTI_STEP=0; else unset NUDGE_BENCH_MULTI_STEP; fi
  NUDGE_BENCH_LOOPBACK=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 100 --warmup 5 --tiles 248 --no-cpu 2> gpurun_out/r06h_loop_$V.err | grep "^{" > gpurun_out/r06h_loop_$V.json
  python - $V <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r06h_loop_{sys.argv[1]}.json"))
    print(sys.argv[1], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), d["halo"]["per_step_transport"][:60], "loop records/step", d["halo"]["loopback_records_per_step"], "still", d.get("still_steps"), "pair", d["config"].get("pair_ahead", {}).get("share_of_timed_steps"), "ahead", d["config"]["xform_ahead"]["share_of_timed_steps"], d.get("per_rank"))
except Exception as e:
    print(sys.argv[1], "no line:", e); print(open(f"gpurun_out/r06h_loop_{sys.argv[1]}.err").read()[-2500:])
PY
done
unset NUDGE_BENCH_MULTI_STEP
timeout 300 python bench.py --gpus 1 --tiles 248 --steps 20 --warmup 5 --no-cpu --no-full-window > gpurun_out/r06h_248_plain.json 2> gpurun_out/r06h_248_plain.err; python -c "
import json; d=json.load(open('gpurun_out/r06h_248_plain.json')); print('plain 248 tiles', round(d['value'],1), round(d['ms_per_step'],4))"
