#!/bin/bash
# experiment: k_chain fast step; walker spacing
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -5
for ws in 384 256 192; do
	timeout 300 python bench.py --only headline --steps 2 --warmup 1 --walker-step $ws > gpurun_out/exp1_ws$ws.json 2>/dev/null
	python - $ws <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp1_ws%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("ws", sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], d["phases_ms_per_step"], "kchain/launch", d["roofline"]["ms_per_launch"], "steps", d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"])
PY
done
