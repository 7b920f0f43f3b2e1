#!/bin/bash
# k_chain's grid against the resident wave slots (88 VGPRs -> 5 waves per SIMD -> 1280 blocks of 256): block cap x walker spacing
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for cap in 2048 1280; do for ws in ${WS:-384 256 224 208 192 160}; do
	RB3GPU_BLKCAP=$cap timeout 300 python bench.py --only headline --steps 2 --warmup 1 --walker-step $ws > gpurun_out/exp20.json 2>/dev/null
	python - "$cap" "$ws" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp20.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
print("blkcap %s walker-step %s: ms %.1f k_chain %.1f rank %.1f rebuild %.1f host %.1f steps %d md5ok %s fb %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], p["k_chain"], p["rank"], p["rebuild"], p["host_and_sync_inside_merge_calls"], d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"]))
PY
done; done
