#!/bin/bash
# occupancy of the solver builds from PMC: average waves per SIMD = SQ_WAVE_CYCLES (quad-cycles) x 4 / (GRBM_GUI_ACTIVE x 1024)
mkdir -p gpurun_out/r5d; R=$PWD; cd /tmp; export TMPDIR=/tmp
for lib in base new3b; do
  NUDGE_HIP_LIBRARY=$R/nudge_amd/_ab/lib$lib.so rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-include-regex "k_solve_one_body" --kernel-trace --output-format csv -d $R/gpurun_out/r5d/$lib -o p -- python $R/tests/gpu_ab_libs.py --one > $R/gpurun_out/r5d/$lib.json 2> $R/gpurun_out/r5d/$lib.err
  F=$(find $R/gpurun_out/r5d/$lib -name "*counter_collection.csv" | head -1)
  echo "== $lib"; python - "$F" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); name = {}
for row in csv.DictReader(open(sys.argv[1])):
    per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"]); name[int(row["Dispatch_Id"])] = row["Kernel_Name"][:60]
ids = sorted(per)[-20:]
for c in ("SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"):
    print(c, sum(per[i][c] for i in ids) / len(ids))
m = {c: sum(per[i][c] for i in ids) / len(ids) for c in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")}
print(name[ids[-1]], "avg waves per SIMD:", m["SQ_WAVE_CYCLES"] * 4 / (m["GRBM_GUI_ACTIVE"] * 1024))
PY
done
cd $R
for nw in 2 4; do NH_SOLVER_WAVES=$nw NUDGE_HIP_LIBRARY=$R/nudge_amd/_ab/libnew3b.so python tests/gpu_ab_libs.py --one; done
# the N = 2 rehearsal of the new timed region (two ranks on one GPU, gloo) and the one-rank RCCL loop-back that carries records
timeout 600 tests/gpu_rehearsal_n2.sh --no-full-window 2> gpurun_out/r5d/n2.err | grep '^{' | tail -1 > gpurun_out/r5d/n2.json; python -c "
import json; d=json.load(open('gpurun_out/r5d/n2.json')); print('N2 value', d['value'], d['config']['timed_blocks'], d['config']['timed_seconds'], d.get('per_rank'))"
NUDGE_BENCH_LOOPBACK=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 20 --warmup 5 --tiles 248 --no-cpu --no-full-window 2> gpurun_out/r5d/loop.err | grep '^{' | tail -1 > gpurun_out/r5d/loop.json; python -c "
import json; d=json.load(open('gpurun_out/r5d/loop.json')); print('loop value', d['value'], d['config']['timed_blocks'], d.get('halo'), d.get('per_rank'))"
tail -n 3 gpurun_out/r5d/n2.err; tail -n 3 gpurun_out/r5d/loop.err
