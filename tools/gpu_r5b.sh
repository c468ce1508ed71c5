#!/bin/bash
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && /tmp/valu_rate > gpurun_out/r05_valu_rate.log 2>&1
cat gpurun_out/r05_valu_rate.log
timeout 600 python -m pytest tests/test_gpu_still.py -x -q -m gpu -s -k "hops or counts_and" 2>&1 | tail -8
