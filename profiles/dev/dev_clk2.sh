#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/clk; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for MT in 0.5 3; do
rm -rf $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window --min-time $MT > $O/b_$MT.json 2> $O/p_$MT.log
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
echo "== min-time $MT: $(python3 -c "import json;d=json.load(open('$O/b_$MT.json'));print(d['value'], d['roofline']['avg_launch_ms'])")"
[ -n "$F" ] && head -4 $F | cut -c1-200
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python3 - "$DB" <<'PY'
import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]
ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
sol=[(a,b) for a,b,n in rows if 'k_solve_one_body' in n]
import statistics
d=[(b-a)/1000 for a,b in sol]
n=len(d)
print('solver dispatches',n,'first 100 avg',statistics.mean(d[:100]),'middle',statistics.mean(d[n//2-50:n//2+50]),'last 100',statistics.mean(d[-100:]))
# gap before solver
PY
done
