# dev script (GPU box): SQ wave-state counters for every kernel of a scene.   usage: dev_sq_all.sh <tag> <bench args...>
set -u
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/r2h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_$TAG -o c -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --min-time 0 "$@" > $O/pmc_$TAG.json 2> $O/pmc_$TAG.log
F=$(find $O/pmc_$TAG -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/profiles/export_sq.py $(dirname $F) $O/sq_$TAG.json > $O/sq_$TAG.txt 2>&1
rm -rf $O/pmc_$TAG
cat $O/sq_$TAG.txt
