#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/movers; mkdir -p $O
cd /tmp
timeout 800 python $R/tests/gpu_movers_rates.py > $O/movers.log 2>&1; cat $O/movers.log
for rep in 1 2; do for V in default NH_NO_INCREMENTAL; do
  E=""; [ $V != default ] && E="$V=1"
  env $E python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window > $O/c2_${V}_$rep.json 2> /dev/null
  echo "c2 $V: $(python -c "import json;d=json.loads(open('$O/c2_${V}_$rep.json').read().strip().split(chr(10))[-1]);print(round(d['value'],1),'steps/s', round(d['ms_per_step'],4),'ms')")"
done; done
