#!/bin/bash
# GPU box: A/B of tune switches on the headline leg:  bash tools/gpu_ab.sh "RB3GPU_X=1" "RB3GPU_X=2 RB3GPU_Y=3" ...   ("" = defaults)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for cfg in "$@"; do
	env $cfg timeout 600 python bench.py --only headline --steps ${STEPS:-3} --warmup 1 > gpurun_out/ab.json 2>/dev/null
	python - "$cfg" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
print("%-40s ms %.1f  k_chain %.1f rank %.1f rebuild %.1f lf %.1f  fallbacks %s md5ok %s" % (sys.argv[1] or "(defaults)", d["ms_per_step"], p["k_chain"], p["rank"], p["rebuild"], p["lf"], d["config"]["rank_phase_fallbacks"], d["config"]["fmd_identical_to_reference"]))
PY
done
