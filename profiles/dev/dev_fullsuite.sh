# dev script (GPU box): the whole -m gpu suite + smoke
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -m gpu -q -x --durations=12 > $O/gpu_suite.txt 2>&1; tail -25 $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
