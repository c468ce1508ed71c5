# dev script (GPU box): A/B of an environment knob on the c2 bench.  usage: dev_ab.sh VAR v1 v2 ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O; VAR=$1; shift
for V in "$@"; do
  env $VAR=$V python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --profile-kernels > $O/ab_$V.json 2> $O/ab_$V.err
  echo "$VAR=$V: $(python -c "import json;d=json.loads(open('$O/ab_$V.json').read().strip().split(chr(10))[-1]);print(round(d['value'],1),'steps/s', round(d['roofline']['avg_launch_ms']*1e3,1),'us solver')")"
  head -3 $O/ab_$V.err | tail -2
done
