#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export RB3GPU_LIB=$R/ropebwt3_amd/prof/profstep.so
timeout 300 python bench.py --only headline --steps 1 --warmup 0 --mtb ${K:-152} 2> gpurun_out/exp4.err > gpurun_out/exp4.json
grep "common step" gpurun_out/exp4.err | awk 'NR%15==1' | head -12
grep "common step" gpurun_out/exp4.err | tail -2
python -c "
import json; d=json.loads(open('gpurun_out/exp4.json').read().strip().splitlines()[-1]); print(d['roofline']['ms_per_launch'], d['config']['lf_steps_per_step'])"
