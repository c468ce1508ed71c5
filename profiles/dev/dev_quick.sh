# dev script (GPU box): parity file + c2 bench with kernel table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/parity.txt 2>&1; tail -15 $O/parity.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --profile-kernels > $O/c2.json 2> $O/c2.err; head -30 $O/c2.err
python -c "import json;d=json.loads(open('$O/c2.json').read().strip().split(chr(10))[-1]);print(d['value'],d['ms_per_step'],d['roofline'],d['launches_per_step'])"
