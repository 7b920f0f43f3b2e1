#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for ws in 384 359 270 216 540; do
	timeout 300 python bench.py --only headline --steps 2 --warmup 1 --walker-step $ws > gpurun_out/exp16.json 2>/dev/null
	python - "$ws" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp16.json").read().strip().splitlines()[-1])
print("walker-step", sys.argv[1], "ms", d["ms_per_step"], "k_chain", d["phases_ms_per_step"]["k_chain"], "rank", d["phases_ms_per_step"]["rank"], "steps", d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"])
PY
done
