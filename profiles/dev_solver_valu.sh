# dev script (run on the GPU box): exhaustive check of the short division forms, parity suite, bench, and SQ counter passes over the fused solver
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
cd $R
tests/fastmath/exhaustive > $O/exhaustive.txt 2>&1; echo "exhaustive rc $?" >> $O/exhaustive.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/parity.txt 2>&1; tail -3 $O/parity.txt
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --profile-kernels > $O/c2.json 2> $O/c2.err
for P in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT"; do
  T=$(echo $P | cut -d' ' -f1)
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$T -o c -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --min-time 0 > $O/pmc_$T.json 2> $O/pmc_$T.log
  F=$(find $O/pmc_$T -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python $R/profiles/export_valu.py $(dirname $F) $O/solver_$T.json > $O/solver_$T.txt 2>&1
  rm -rf $O/pmc_$T
done
cat $O/exhaustive.txt; head -6 $O/c2.err; cat $O/solver_*.txt; tail -5 $O/pmc_SQ_INSTS_VMEM_RD.log
