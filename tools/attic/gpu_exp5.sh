#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -4
unset RB3GPU_LIB
timeout 300 python bench.py --only headline --steps 2 --warmup 1 > gpurun_out/exp5.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/exp5.json").read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], d["phases_ms_per_step"], "kchain/launch", d["roofline"]["ms_per_launch"], "steps", d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"])
PY
export RB3GPU_LIB=$R/ropebwt3_amd/prof/profstep.so
timeout 300 python bench.py --only headline --steps 1 --warmup 0 2> gpurun_out/exp5.err > /dev/null
grep "common step" gpurun_out/exp5.err | awk 'NR%30==1' | head -6
grep "common step" gpurun_out/exp5.err | tail -1
