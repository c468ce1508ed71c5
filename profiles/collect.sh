#!/bin/bash
# profiles/collect.sh <tag> -- run ON THE GPU BOX (gpurun): collects what profiles/ holds for one build.
#   FIRST the counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes: they do not fit one on gfx950; SQ wave states) for c2, c3, c5-on-one-GPU and c4,
#   exported AND copied into $R/profiles/ on the box, THEN the bench lines -- so that every committed line reads the PMC export of ITS OWN build and carries
#   roofline.traffic + pricing_check (round 5 wrote the c3 / c5 lines before their exports existed).
#   Then rocprofv3 --kernel-trace --stats of the driver invocation, the one-rank partition path over RCCL, and the one-GPU rehearsal of the N = 2 path.
# Raw output goes to gpurun_out/<tag>/ (scratch); the summaries are exported into gpurun_out/<tag>/export/ ready to be copied to profiles/.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O/export
cd /tmp; export TMPDIR=/tmp
SQ_COUNTERS="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU"

pmc_export() {      # pmc_export <suffix ("" | _c3 | _c5 | _c4)> <last_n> <bench arguments...>: two PMC passes of `bench.py <arguments>` -> ${TAG}_pmc_traffic<suffix>.json/.log
  local SUF=$1 LAST=$2; shift 2
  local D=$O/pmc$SUF
  for CNT in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $D -o $CNT -- python $R/bench.py "$@" > $O/pmc${SUF}_$CNT.json 2> $O/pmc${SUF}_$CNT.log
  done
  local F=$(find $D -name "FETCH_SIZE_counter_collection.csv" | head -1)
  if [ -n "$F" ]; then
    local DD=$(dirname $F) W=$(find $D -name "WRITE_SIZE_counter_collection.csv" | head -1)
    [ "$(dirname $W)" != "$DD" ] && cp $W $DD/
    # (the step count of the profiled run comes from the line that run printed, not from counting k_xform dispatches)
    python $R/profiles/export_pmc.py $DD $O/export/${TAG}_pmc_traffic$SUF.json $LAST $O/pmc${SUF}_FETCH_SIZE.json > $O/export/${TAG}_pmc_traffic$SUF.log
    cp $O/export/${TAG}_pmc_traffic$SUF.json $R/profiles/
  fi
  rm -rf $D
}

sq_export() {       # sq_export <suffix> <last_n> <bench arguments...>
  local SUF=$1 LAST=$2; shift 2
  local D=$O/sq$SUF
  rocprofv3 --pmc $SQ_COUNTERS --kernel-trace --output-format csv -d $D -o sq -- python $R/bench.py "$@" > $O/sq${SUF}_bench.json 2> $O/sq$SUF.log
  local F=$(find $D -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then
    python $R/profiles/export_sq.py $(dirname $F) $O/export/${TAG}_sq_wave_states$SUF.json $LAST > $O/export/${TAG}_sq_wave_states$SUF.log 2>&1
    cp $O/export/${TAG}_sq_wave_states$SUF.json $R/profiles/
  fi
  rm -rf $D
}

# ---- 1. counters first (one block of the timed region is enough; counters serialise the kernels) ----
COMMON="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 0"
pmc_export ""    20 $COMMON
pmc_export "_c3" 20 $COMMON --scene mixed --tiles 248
pmc_export "_c5" 20 $COMMON --tiles 1984
sq_export  ""    20 $COMMON
sq_export  "_c3" 20 $COMMON --scene mixed --tiles 248
# config 4: the general path's kernels only
PIT="--scene pit --no-cpu --steps 3 --min-time 0"
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --kernel-include-regex "k_blk_|k_rows_general|k_level_exec|k_gen_" --kernel-trace --output-format csv -d $O/pmc4 -o $CNT -- python $R/bench.py $PIT > $O/pmc4_$CNT.json 2> $O/pmc4_$CNT.log
done
F=$(find $O/pmc4 -name "FETCH_SIZE_counter_collection.csv" | head -1)
if [ -n "$F" ]; then
  D=$(dirname $F); W=$(find $O/pmc4 -name "WRITE_SIZE_counter_collection.csv" | head -1); [ "$(dirname $W)" != "$D" ] && cp $W $D/
  python $R/profiles/export_pmc.py $D $O/export/${TAG}_pmc_traffic_c4.json 64 $O/pmc4_FETCH_SIZE.json > $O/export/${TAG}_pmc_traffic_c4.log
  cp $O/export/${TAG}_pmc_traffic_c4.json $R/profiles/
fi
rocprofv3 --pmc $SQ_COUNTERS --kernel-include-regex "k_blk_|k_rows_general|k_level_exec|k_gen_" --kernel-trace --output-format csv -d $O/sq4 -o c -- python $R/bench.py $PIT > $O/sq4_bench.json 2> $O/sq4.log
F=$(find $O/sq4 -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/profiles/export_sq.py $(dirname $F) $O/export/${TAG}_sq_wave_states_c4.json 64 > $O/export/${TAG}_sq_wave_states_c4.log 2>&1
rm -rf $O/sq4 $O/pmc4

# ---- 2. the bench lines (they read the exports just copied into profiles/) ----
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/export/${TAG}_bench_driver_invocation.json 2> $O/bench_driver.err
python $R/bench.py --profile-kernels > $O/export/${TAG}_bench_default.json 2> $O/export/${TAG}_bench_kernel_table.log
python $R/bench.py --scene mixed --tiles 248 --profile-kernels > $O/export/${TAG}_bench_c3_mixed_2M.json 2> $O/export/${TAG}_bench_c3_kernel_table.log
timeout 400 python $R/bench.py --scene pit --no-cpu --steps 10 --profile-kernels > $O/export/${TAG}_bench_c4_pit_4M_settled.json 2> $O/export/${TAG}_bench_c4_kernel_table.log
# config 5 as ONE world on one GPU (the denominator of ">= 6 x single-GPU throughput at 8 GPUs") and one rank's load (248 tiles) alone on the GPU
python $R/bench.py --gpus 1 --tiles 1984 --steps 20 --warmup 5 --profile-kernels > $O/export/${TAG}_bench_c5_single_gpu_16M.json 2> $O/export/${TAG}_bench_c5_single_gpu_16M_kernel_table.log
python $R/bench.py --gpus 1 --tiles 248 --steps 20 --warmup 5 --no-cpu --no-full-window > $O/export/${TAG}_bench_248_tiles_single_gpu.json 2> $O/bench_248.err
# the partitioned code path with ONE rank under torch.distributed.run, backend nccl: the library-driven RCCL transport checked by a loop-back, per-step host cost of the path
NUDGE_BENCH_LOOPBACK=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 $R/bench.py --gpus 1 --steps 100 --warmup 5 --tiles 248 --no-cpu 2> $O/loop.err | grep "^{" > $O/export/${TAG}_bench_one_rank_partition_path_rccl.json

# ---- 3. rocprofv3: kernel trace + stats of the driver's invocation ----
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window > $O/prof_bench.json 2> $O/prof.log
DB=$(find $O/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python - "$O/prof_bench.json" "$DB" "$O/export/${TAG}_rocprofv3_kernel_stats.tsv" <<'PY'
import json, subprocess, sys, os
d = json.load(open(sys.argv[1]))
c = d["config"]
total = c["landed_after_steps"] + d["warmup"] + 3 + c["timed_steps"] + c["timed_blocks"] + (201 if d.get("landed_window") else 0)
note = (f"python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu under rocprofv3 --kernel-trace --stats: {d['value']:.1f} steps/s, "
        f"{c['timed_steps']} timed steps of {total} ({c['timed_window']})")
subprocess.check_call([sys.executable, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "profiles", "export_rocprof.py"), sys.argv[2], sys.argv[3], note, str(c["timed_steps"]), str(total)])
PY
fi
rm -rf $O/prof
NH_BLK_PROFILE=1 timeout 300 python $R/bench.py --scene pit --no-cpu --steps 10 --min-time 0.3 > $O/c4_prof.json 2> $O/c4_prof.log
grep blk_profile $O/c4_prof.log | tail -8 > $O/export/${TAG}_blocked_sweep_workgroup_profile.log
# N = 2 rehearsal on one GPU
cd $R; timeout 600 tests/gpu_rehearsal_n2.sh > $O/rehearsal.out 2> $O/export/${TAG}_bench_n2_rehearsal_kernel_table.log
grep '^{' $O/rehearsal.out | tail -1 > $O/export/${TAG}_bench_n2_rehearsal_one_gpu.json
ls -la $O/export
