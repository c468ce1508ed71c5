#!/bin/bash
# round 6, call i: the count scan in pair steps of large worlds (c5 on one GPU), the looser re-insertion threshold (movers table), still tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_still.py -x -q -m gpu 2>&1 | tail -n 3
python bench.py --gpus 1 --tiles 1984 --steps 20 --warmup 5 --no-cpu --no-full-window > gpurun_out/r06i_c5.json 2> gpurun_out/r06i_c5.err; python -c "
import json; d=json.load(open('gpurun_out/r06i_c5.json')); print('c5 single', round(d['value'],1), round(d['ms_per_step'],4), d.get('still_steps'), d['config']['pair_ahead']['share_of_timed_steps'], round(d['landed_window']['ms_per_step'],4) if d.get('landed_window') else None)"
timeout 900 python tests/gpu_movers_rates.py --quick > gpurun_out/r06i_movers.log 2>&1; cat gpurun_out/r06i_movers.log | cut -c1-220
