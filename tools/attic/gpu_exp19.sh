#!/bin/bash
# walker spacing against k_chain / rank phase on the headline build
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for ws in ${WS:-384 256 192 128 512}; do
	timeout 300 python bench.py --only headline --steps 2 --warmup 1 --walker-step $ws > gpurun_out/exp19.json 2>/dev/null
	python - "$ws" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp19.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
print("walker-step %s: ms %.1f k_chain %.1f rank %.1f rebuild %.1f lf %.1f host %.1f steps %d md5ok %s fb %s" % (sys.argv[1], d["ms_per_step"], p["k_chain"], p["rank"], p["rebuild"], p["lf"], p["host_and_sync_inside_merge_calls"], d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"]))
PY
done
