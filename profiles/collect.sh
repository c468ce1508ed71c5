#!/bin/bash
# profiles/collect.sh <tag> -- run ON THE GPU BOX (gpurun): collects what profiles/ holds for one build.
#   bench lines (driver invocation, default, c3, c4), per-kernel tables, rocprofv3 --kernel-trace --stats of the driver invocation,
#   two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950), one SQ wave-state pass and the one-GPU rehearsal of the N = 2 path.
# Raw output goes to gpurun_out/<tag>/ (scratch); the summaries are exported into gpurun_out/<tag>/export/ ready to be copied to profiles/.
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O/export
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/export/${TAG}_bench_driver_invocation.json 2> $O/bench_driver.err
python $R/bench.py --profile-kernels > $O/export/${TAG}_bench_default.json 2> $O/export/${TAG}_bench_kernel_table.log
python $R/bench.py --scene mixed --tiles 248 --profile-kernels > $O/export/${TAG}_bench_c3_mixed_2M.json 2> $O/export/${TAG}_bench_c3_kernel_table.log
timeout 400 python $R/bench.py --scene pit --no-cpu --steps 10 --profile-kernels > $O/export/${TAG}_bench_c4_pit_4M_settled.json 2> $O/export/${TAG}_bench_c4_kernel_table.log
# config 5 as ONE world on one GPU (the denominator of ">= 6 x single-GPU throughput at 8 GPUs") and one rank's load (248 tiles) alone on the GPU
python $R/bench.py --gpus 1 --tiles 1984 --steps 20 --warmup 5 --profile-kernels > $O/export/${TAG}_bench_c5_single_gpu_16M.json 2> $O/export/${TAG}_bench_c5_single_gpu_16M_kernel_table.log
python $R/bench.py --gpus 1 --tiles 248 --steps 20 --warmup 5 --no-cpu --no-full-window > $O/export/${TAG}_bench_248_tiles_single_gpu.json 2> $O/bench_248.err
# the partitioned code path with ONE rank under torch.distributed.run, backend nccl: the library-driven RCCL transport checked by a loop-back, per-step host cost of the path
NUDGE_BENCH_LOOPBACK=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 $R/bench.py --gpus 1 --steps 100 --warmup 5 --tiles 248 --no-cpu 2> $O/loop.err | grep "^{" > $O/export/${TAG}_bench_one_rank_partition_path_rccl.json
NH_NO_BLOCKS=1 timeout 400 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 1 > $O/export/${TAG}_bench_c4_per_colour_launches.json 2> $O/c4_noblocks.err
# rocprofv3: kernel trace + stats of the driver's invocation
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window > $O/prof_bench.json 2> $O/prof.log
DB=$(find $O/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python - "$O/prof_bench.json" "$DB" "$O/export/${TAG}_rocprofv3_kernel_stats.tsv" <<'PY'
import json, subprocess, sys, os
d = json.load(open(sys.argv[1]))
c = d["config"]
total = c["landed_after_steps"] + d["warmup"] + 3 + c["timed_steps"] + (c["timed_blocks"] - 1) * 0
note = (f"python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu under rocprofv3 --kernel-trace --stats: {d['value']:.1f} steps/s, "
        f"{c['timed_steps']} timed steps of {total} ({c['timed_window']})")
root = os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[3])))
subprocess.check_call([sys.executable, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "profiles", "export_rocprof.py"), sys.argv[2], sys.argv[3], note, str(c["timed_steps"]), str(total)])
PY
fi
# PMC passes (one block of the timed region is enough; counters serialise the kernels)
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/pmc -o $CNT -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 0 > $O/pmc_$CNT.json 2> $O/pmc_$CNT.log
done
F=$(find $O/pmc -name "FETCH_SIZE_counter_collection.csv" | head -1)
if [ -n "$F" ]; then
  D=$(dirname $F); W=$(find $O/pmc -name "WRITE_SIZE_counter_collection.csv" | head -1); [ "$(dirname $W)" != "$D" ] && cp $W $D/
  python $R/profiles/export_pmc.py $D $O/export/${TAG}_pmc_traffic.json 20 > $O/export/${TAG}_pmc_traffic.log
fi
# the same two PMC passes for config 3 (2 M mixed) and for the 16 M-box world of config 5 on one GPU: their bench lines read these exports for roofline.traffic
for CFG in "c3 --scene mixed --tiles 248" "c5 --tiles 1984"; do
  set -- $CFG; NAME=$1; shift
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/pmc_$NAME -o $CNT -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 0 "$@" > $O/pmc_${NAME}_$CNT.json 2> $O/pmc_${NAME}_$CNT.log
  done
  F=$(find $O/pmc_$NAME -name "FETCH_SIZE_counter_collection.csv" | head -1)
  if [ -n "$F" ]; then
    D=$(dirname $F); W=$(find $O/pmc_$NAME -name "WRITE_SIZE_counter_collection.csv" | head -1); [ "$(dirname $W)" != "$D" ] && cp $W $D/
    python $R/profiles/export_pmc.py $D $O/export/${TAG}_pmc_traffic_$NAME.json 20 > $O/export/${TAG}_pmc_traffic_$NAME.log
  fi
  rm -rf $O/pmc_$NAME
done
# SQ wave-state counters of every kernel (where the wave cycles go: issuing, parked at s_waitcnt, stalled at issue)
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/sq -o sq -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 0 > $O/sq_bench.json 2> $O/sq.log
F=$(find $O/sq -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/profiles/export_sq.py $(dirname $F) $O/export/${TAG}_sq_wave_states.json > $O/export/${TAG}_sq_wave_states.log 2>&1
rm -rf $O/sq $O/pmc $O/prof
# config 4: the blocked solver's kernels -- SQ wave states, FETCH / WRITE per launch, and the per-workgroup wall-clock profile
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --kernel-include-regex "k_blk_|k_rows_general|k_level_exec" --kernel-trace --output-format csv -d $O/sq4 -o c -- python $R/bench.py --scene pit --no-cpu --steps 3 --min-time 0 > $O/sq4_bench.json 2> $O/sq4.log
F=$(find $O/sq4 -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/profiles/export_sq.py $(dirname $F) $O/export/${TAG}_sq_wave_states_c4.json 64 > $O/export/${TAG}_sq_wave_states_c4.log 2>&1
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --kernel-include-regex "k_blk_|k_rows_general|k_level_exec" --kernel-trace --output-format csv -d $O/pmc4 -o $CNT -- python $R/bench.py --scene pit --no-cpu --steps 3 --min-time 0 > $O/pmc4_$CNT.json 2> $O/pmc4_$CNT.log
done
F=$(find $O/pmc4 -name "FETCH_SIZE_counter_collection.csv" | head -1)
if [ -n "$F" ]; then
  D=$(dirname $F); W=$(find $O/pmc4 -name "WRITE_SIZE_counter_collection.csv" | head -1); [ "$(dirname $W)" != "$D" ] && cp $W $D/
  python $R/profiles/export_pmc.py $D $O/export/${TAG}_pmc_traffic_c4.json 64 > $O/export/${TAG}_pmc_traffic_c4.log
fi
NH_BLK_PROFILE=1 timeout 300 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 0.3 > $O/c4_prof.json 2> $O/c4_prof.log
grep blk_profile $O/c4_prof.log | tail -8 > $O/export/${TAG}_blocked_sweep_workgroup_profile.log
rm -rf $O/sq4 $O/pmc4
# N = 2 rehearsal on one GPU
cd $R; timeout 600 tests/gpu_rehearsal_n2.sh > $O/rehearsal.out 2> $O/export/${TAG}_bench_n2_rehearsal_kernel_table.log
grep '^{' $O/rehearsal.out | tail -1 > $O/export/${TAG}_bench_n2_rehearsal_one_gpu.json
ls -la $O/export
