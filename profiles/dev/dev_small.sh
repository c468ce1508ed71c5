# dev script (GPU box): parity + callsite + partition tests, small-world profile, drop-in rates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_callsite.py tests/test_gpu_partition.py -m gpu -q -x > $O/tests.txt 2>&1; tail -8 $O/tests.txt
python tests/gpu_small_profile.py > $O/small_profile.txt 2>&1; tail -22 $O/small_profile.txt
timeout 900 python tests/gpu_dropin_rates.py > $O/dropin.txt 2>&1; grep "example scene" $O/dropin.txt
