#!/bin/bash
# the age from which a walker records (RB3_TENT_MIN_AGE) with the walkers' pre-roll (RB3H_PREROLL) set to the same value
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rep in 1 2; do for a in 32 24 20 16 12; do
	if [ $a = 32 ]; then unset RB3GPU_LIB RB3HOST_LIB; else export RB3GPU_LIB=$R/ropebwt3_amd/prof/age$a.so RB3HOST_LIB=$R/ropebwt3_amd/prof/host_pre$a.so; fi
	timeout 300 python bench.py --only headline --steps 2 --warmup 1 > gpurun_out/exp23.json 2>/dev/null
	python - "$a" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/exp23.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
print("age/pre %-3s ms %.1f k_chain %.1f rank %.1f rebuild %.1f steps %d md5ok %s fb %s long %s" % (sys.argv[1], d["ms_per_step"], p["k_chain"], p["rank"], p["rebuild"], d["config"]["lf_steps_per_step"], d["config"]["fmd_identical_to_reference"], d["config"]["rank_phase_fallbacks"], d["config"]["long_settles"]))
PY
done; done
for a in 16 12; do RB3GPU_LIB=$R/ropebwt3_amd/prof/age$a.so RB3HOST_LIB=$R/ropebwt3_amd/prof/host_pre$a.so timeout 600 python tools/soak.py 120 13000 2>&1 | tail -1 | sed "s/^/age $a soak: /"; done
