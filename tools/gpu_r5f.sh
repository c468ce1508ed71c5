#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu 2>&1 | tail -n 3
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && /tmp/valu_rate > gpurun_out/r05_valu_rate.log 2>&1; grep -n "v_mul\|v_add\|v_fmac\|v_cnd\|3 src\|mul/add" gpurun_out/r05_valu_rate.log
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5f_prof -o stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window > $R/gpurun_out/r5f_bench.json 2> $R/gpurun_out/r5f_prof.log
cd $R; python - <<'PY'
import json, glob, sqlite3
d = json.load(open("gpurun_out/r5f_bench.json")); print("bench under rocprof:", d["value"], d["ms_per_step"])
import subprocess, sys
db = glob.glob("gpurun_out/r5f_prof/**/*.db", recursive=True)
print(db)
PY
DB=$(find gpurun_out/r5f_prof -name "*.db" | head -1); python profiles/export_rocprof.py $DB gpurun_out/r5f_stats.tsv "r5f" 9000 9084 2>&1 | tail -n 2; head -n 8 gpurun_out/r5f_stats.tsv | cut -c1-160
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window | python -c "import json,sys; d=json.load(sys.stdin); print('plain bench', d['value'], d['ms_per_step'], d.get('landed_window',{}).get('steps_per_s'), d['kernel_ms_per_step'])"
