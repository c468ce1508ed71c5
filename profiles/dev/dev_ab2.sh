# dev script (GPU box): A/B of an environment knob on c2, c3 and c4.  usage: dev_ab2.sh VAR v1 v2 ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n; mkdir -p $O; VAR=$1; shift
for V in "$@"; do
  for CFG in "c2:--steps 20 --warmup 5" "c3:--scene mixed --tiles 248" "c4:--scene pit --steps 10"; do
    N=${CFG%%:*}; A=${CFG#*:}
    env $VAR=$V timeout 600 python $R/bench.py --no-cpu --profile-kernels $A > $O/${N}_$V.json 2> $O/${N}_$V.err
    echo "$VAR=$V $N: $(python -c "import json;d=json.loads(open('$O/${N}_$V.json').read().strip().split(chr(10))[-1]);print(round(d['value'],1),'steps/s')") $(grep find_pairs $O/${N}_$V.err)"
  done
done
