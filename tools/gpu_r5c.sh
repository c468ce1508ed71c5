#!/bin/bash
mkdir -p gpurun_out
python tests/gpu_ab_libs.py nudge_amd/_ab/libbase.so nudge_amd/_ab/libnew2b.so nudge_amd/_ab/libnew3b.so nudge_amd/_ab/libbase.so nudge_amd/_ab/libnew3b.so 2>&1 | tee gpurun_out/r05c_ab.log
NUDGE_HIP_LIBRARY=$PWD/nudge_amd/_ab/libnew3b.so timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu 2>&1 | tail -3
