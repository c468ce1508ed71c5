# dev script (GPU box): resident-solver tests + small-scene rates + drop-in rates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2l; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lds_resident or colours_seeded or exact_order_steps" > $O/resb.txt 2>&1; tail -12 $O/resb.txt
timeout 600 python tests/gpu_small_scenes.py > $O/small.txt 2>&1; cat $O/small.txt
timeout 900 python tests/gpu_dropin_rates.py > $O/dropin.txt 2>&1; tail -30 $O/dropin.txt
