# dev script (GPU box): seeded-colour test, pit invariants, c4 bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "colours_seeded or ball_pit or lds_resident" > $O/colour.txt 2>&1; tail -12 $O/colour.txt
timeout 600 python bench.py --scene pit --no-cpu --steps 10 --profile-kernels > $O/c4.json 2> $O/c4.err; head -14 $O/c4.err
python -c "import json;d=json.loads(open('$O/c4.json').read().strip().split(chr(10))[-1]);print('c4',d['value'],d['ms_per_step'],d['launches_per_step'])"
