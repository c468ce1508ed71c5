#!/bin/bash
# round 6, call u: calls of 1 .. 100 sub-steps with and without early counters, awake steps only
mkdir -p gpurun_out
timeout 600 python tests/gpu_call_rates.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06u_call_rates.log
