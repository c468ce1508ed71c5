#!/bin/bash
# c4: where do the wave cycles of the blocked sweep go (SQ counters), and what does it move (FETCH/WRITE)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --kernel-include-regex "k_blk_" --kernel-trace --output-format csv -d $O/sq -o c -- python $R/bench.py --scene pit --no-cpu --steps 3 --min-time 0 > $O/sq_bench.json 2> $O/sq.log
F=$(find $O/sq -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python $R/profiles/export_sq.py $(dirname $F) $O/sq_c4_blk.json > $O/sq_c4_blk.txt 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "k_blk_sweep" --kernel-trace --output-format csv -d $O/lds -o c -- python $R/bench.py --scene pit --no-cpu --steps 3 --min-time 0 > $O/lds_bench.json 2> $O/lds.log
F=$(find $O/lds -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && python - "$F" > $O/lds_c4_blk.txt <<'PY'
import csv,sys,collections
per=collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    per[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in per.items():
    print(k[:60], {c: sum(x[-64:])/len(x[-64:]) for c,x in v.items()})
PY
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --kernel-include-regex "k_blk_|k_rows_general|k_level_exec" --kernel-trace --output-format csv -d $O/pmc -o $CNT -- python $R/bench.py --scene pit --no-cpu --steps 3 --min-time 0 > $O/pmc_$CNT.json 2> $O/pmc_$CNT.log
done
F=$(find $O/pmc -name "FETCH_SIZE_counter_collection.csv" | head -1)
if [ -n "$F" ]; then
  D=$(dirname $F); W=$(find $O/pmc -name "WRITE_SIZE_counter_collection.csv" | head -1); [ "$(dirname $W)" != "$D" ] && cp $W $D/
  python $R/profiles/export_pmc.py $D $O/pmc_c4_blk.json 64 > $O/pmc_c4_blk.txt
fi
rm -rf $O/sq $O/lds $O/pmc
cat $O/sq_c4_blk.txt $O/lds_c4_blk.txt $O/pmc_c4_blk.txt; tail -3 $O/sq.log
