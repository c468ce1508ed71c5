#!/bin/bash
# round 5, first GPU call: still-step tests (local speculation, observers) + movers table
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu -s > gpurun_out/r05a_still_tests.log 2>&1; echo "still tests rc=$?" 
tail -5 gpurun_out/r05a_still_tests.log
timeout 600 python tests/gpu_movers_rates.py --quick > gpurun_out/r05a_movers.log 2>&1; echo "movers rc=$?"
cat gpurun_out/r05a_movers.log
