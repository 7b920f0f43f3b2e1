#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for bs in 256 128 64; do
	RB3GPU_CHAIN_BS=$bs timeout 300 python bench.py --only headline --steps 2 --warmup 1 > gpurun_out/exp15.json 2>/dev/null
	python - "$bs" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp15.json").read().strip().splitlines()[-1])
print("chain_bs", sys.argv[1], "ms", d["ms_per_step"], "k_chain", d["phases_ms_per_step"]["k_chain"], "rank", d["phases_ms_per_step"]["rank"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"])
PY
done
