#!/bin/bash
# launch-by-launch timeline of one settled c4 step (and one c2 step)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tl; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl4 /tmp/tl2
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl4 -- python $R/bench.py --scene pit --no-cpu --steps 3 --min-time 0 --no-full-window > $O/c4.json 2> $O/c4.err
python $R/profiles/export_timeline.py /tmp/tl4 $O/c4_step_timeline.log
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl2 -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --min-time 0 --no-full-window > $O/c2.json 2> $O/c2.err
python $R/profiles/export_timeline.py /tmp/tl2 $O/c2_step_timeline.log
