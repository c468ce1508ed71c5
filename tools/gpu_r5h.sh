#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu 2>&1 | tail -n 3
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-full-window | python -c "import json,sys; d=json.load(sys.stdin); print('driver invocation', d['value'], d['ms_per_step'], d.get('landed_window',{}).get('steps_per_s'), d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['block_ms_per_step'])"
timeout 600 python tests/gpu_movers_rates.py --quick 2>&1 | grep -v amdgpu.ids | cut -c1-150 | head -9
