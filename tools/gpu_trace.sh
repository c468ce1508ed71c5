#!/bin/bash
# kernel trace of the driver line (or of `bench.py <arguments>`): per-kernel calls / average durations
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time 1 "$@" > $R/gpurun_out/trace_bench.json 2> $R/gpurun_out/trace.log
DB=$(find $R/gpurun_out/trace -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, calls, tot, avg, pct in list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))[:14]:
    print(f"{name.split('(')[0][:70]:70s} calls {calls:6d}  avg {avg/1e3:9.2f} us  {pct:5.1f} %")
PY
python -c "
import json; d=json.load(open('$R/gpurun_out/trace_bench.json')); print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],4))"
rm -rf $R/gpurun_out/trace
