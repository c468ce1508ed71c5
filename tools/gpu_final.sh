#!/bin/bash
# round 5: everything profiles/ holds for the final build (run on the GPU box)
bash profiles/collect.sh r05 > gpurun_out/r05_collect.log 2>&1
timeout 600 python tests/gpu_movers_rates.py --quick > gpurun_out/r05/export/r05_movers_local_speculation.log 2>&1
timeout 600 python tests/gpu_sleepers_rates.py > gpurun_out/r05/export/r05_sleepers_rates.log 2>&1
tail -n 5 gpurun_out/r05_collect.log; ls gpurun_out/r05/export | wc -l
