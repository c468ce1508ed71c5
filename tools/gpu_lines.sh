#!/bin/bash
# bench lines that read the PMC / SQ exports committed under profiles/ (run after profiles/collect.sh's exports have been copied there)
O=gpurun_out/r05_lines; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_driver_invocation.json 2> $O/driver.err
python bench.py --scene mixed --tiles 248 --profile-kernels > $O/r05_bench_c3_mixed_2M.json 2> $O/r05_bench_c3_kernel_table.log
python bench.py --gpus 1 --tiles 1984 --steps 20 --warmup 5 --profile-kernels > $O/r05_bench_c5_single_gpu_16M.json 2> $O/r05_bench_c5_single_gpu_16M_kernel_table.log
python - <<'PY'
import json
for n in ("driver_invocation", "c3_mixed_2M", "c5_single_gpu_16M"):
    d = json.load(open(f"gpurun_out/r05_lines/r05_bench_{n}.json")); r = d["roofline"]
    print(n, round(d["value"], 1), r["kernel"], round(r["frac"], 3), r.get("traffic"), r.get("traffic_source"), r.get("pricing_check"), (r.get("valu") or {}).get("valu_pipe_busy_by_instruction_prices"), (d.get("cpu_baseline") or {}).get("value"))
PY
