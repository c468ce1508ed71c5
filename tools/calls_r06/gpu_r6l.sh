#!/bin/bash
# round 6, call l: the sleepers form skips what is asleep (k_xform<true>, still narrowphase, solver waves): tests, then the sleepers table
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_still.py tests/test_gpu_atsize.py -x -q -m gpu 2>&1 | tail -n 6
timeout 600 python tests/gpu_sleepers_rates.py --table 2>&1 | tee gpurun_out/r06l_sleepers.log | cut -c1-400
