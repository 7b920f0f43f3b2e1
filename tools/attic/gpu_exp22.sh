#!/bin/bash
# the age at which a walker may record tentatively (RB3_TENT_MIN_AGE: 32) against steps, k_chain and fallbacks; soak with each
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rep in 1 2; do for v in release age24 age16 age8; do
	if [ "$v" = release ]; then unset RB3GPU_LIB; else export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so; fi
	timeout 300 python bench.py --only headline --steps 2 --warmup 1 > gpurun_out/exp22.json 2>/dev/null
	python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp22.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
print("%-8s ms %.1f k_chain %.1f rank %.1f rebuild %.1f steps %d md5ok %s fb %s long %s" % (sys.argv[1], d["ms_per_step"], p["k_chain"], p["rank"], p["rebuild"], d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"], d["config"]["long_settles"]))
PY
done; done
for v in age16 age8; do RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so timeout 600 python tools/soak.py 120 9000 2>&1 | tail -1 | sed "s/^/$v soak: /"; done
