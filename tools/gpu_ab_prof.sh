#!/bin/bash
# tools/gpu_ab_prof.sh <lib name> ...   (on the GPU box) -- rocprofv3 --kernel-trace --stats of the default bench run for each A/B library of nudge_amd/_ab, twice, alternating
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for l in "$@"; do
  rm -rf /tmp/prof_$l; NUDGE_HIP_LIBRARY=$GRAFT_REPO_ROOT/nudge_amd/_ab/lib$l.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$l -o out -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-full-window > /tmp/prof_$l.json 2> /tmp/prof_$l.err
  echo "== $l"; grep "^{" /tmp/prof_$l.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  steps/s', round(d['value'],1), 'ms', round(d['ms_per_step'],5), 'block min', round(d['config']['block_ms_per_step']['min'],5))"
  f=$(find /tmp/prof_$l -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_narrowphase<false, true","k_solve_one_body<4, true, true, 1, true","k_xform<true>","k_solve_still_pipe")):
        print("  ", n[:64].ljust(64), r["Calls"].rjust(6), round(float(r["AverageNs"])/1e3,2), "us")
PY
done
done
