#!/bin/bash
# how often does a merge of the headline build fall back (rank phase redone without tentative records) against the walker spacing?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for ws in ${WS:-0 224 232 240 256}; do
	timeout 600 python bench.py --only headline --steps ${STEPS:-8} --warmup 1 --walker-step $ws > gpurun_out/exp26.json 2>/dev/null
	python - "$ws" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp26.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
print("walker-step %s: ms %.1f k_chain %.1f rank %.1f  fallbacks %s in %d builds, long settles %s  md5ok %s" % (sys.argv[1], d["ms_per_step"], p["k_chain"], p["rank"], d["config"]["rank_phase_fallbacks"], d["steps"], d["config"]["long_settles"], d["config"]["fmd_identical_to_reference"]))
PY
done
