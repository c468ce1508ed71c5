# dev script (GPU box): c3 and c4 bench lines with kernel tables
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O; cd $R
python bench.py --scene mixed --tiles 248 --no-cpu --profile-kernels > $O/c3.json 2> $O/c3.err; head -32 $O/c3.err
python -c "import json;d=json.loads(open('$O/c3.json').read().strip().split(chr(10))[-1]);print('c3',d['value'],d['ms_per_step'],d['launches_per_step'])"
timeout 600 python bench.py --scene pit --no-cpu --steps 10 --profile-kernels > $O/c4.json 2> $O/c4.err; head -24 $O/c4.err
python -c "import json;d=json.loads(open('$O/c4.json').read().strip().split(chr(10))[-1]);print('c4',d['value'],d['ms_per_step'],d['launches_per_step'])"
