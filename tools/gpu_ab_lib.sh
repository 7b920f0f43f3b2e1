#!/bin/bash
# GPU box: A/B of library variants (tools/build_variant.sh NAME ... -> ropebwt3_amd/prof/NAME.so) on the headline leg:
#   bash tools/gpu_ab_lib.sh name1 name2 ...     ("release" = the in-tree library);  REPS=2 repeats the whole list
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-1}); do
for v in "$@"; do
	if [ "$v" = release ]; then unset RB3GPU_LIB; else export RB3GPU_LIB=$R/ropebwt3_amd/prof/$v.so; fi
	timeout 600 python bench.py --only headline --steps ${STEPS:-3} --warmup 1 > gpurun_out/ab.json 2>gpurun_out/ab.err
	python - "$v" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); p = d["phases_ms_per_step"]
    print("%-12s ms %.1f  k_chain %.1f (%.4f ms/launch, frac %.3f) rank %.1f rebuild %.1f lf %.1f steps %s fallbacks %s md5ok %s" % (sys.argv[1], d["ms_per_step"], p["k_chain"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], p["rank"], p["rebuild"], p["lf"], d["config"].get("lf_steps_per_step"), d["config"]["rank_phase_fallbacks"], d["config"]["fmd_identical_to_reference"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open("gpurun_out/ab.err").read()[-2000:])
PY
done; done
