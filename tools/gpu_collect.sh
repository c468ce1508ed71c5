#!/bin/bash
# round 6: everything profiles/ holds for the final build (run on the GPU box)
bash profiles/collect.sh r06 > gpurun_out/r06_collect.log 2>&1
timeout 600 python tests/gpu_movers_rates.py --quick > gpurun_out/r06/export/r06_movers_local_speculation.log 2>&1
timeout 600 python tests/gpu_sleepers_rates.py > gpurun_out/r06/export/r06_sleepers_rates.log 2>&1
tail -n 5 gpurun_out/r06_collect.log; ls gpurun_out/r06/export | wc -l
