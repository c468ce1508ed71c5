#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu -s > gpurun_out/r05e_still_tests.log 2>&1; echo "still tests rc=$?"; grep -n "^\[" gpurun_out/r05e_still_tests.log; tail -n 12 gpurun_out/r05e_still_tests.log
timeout 600 python tests/gpu_sleepers_rates.py > gpurun_out/r05e_sleepers.log 2>&1; echo "sleepers rc=$?"; cat gpurun_out/r05e_sleepers.log | tail -n 8
