#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_partition.py tests/test_gpu_callsite.py -x -q -m gpu 2>&1 | tail -n 15
